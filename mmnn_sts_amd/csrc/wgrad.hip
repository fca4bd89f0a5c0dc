// Host-side dispatch of the weight-gradient kernels (wgrad.hpp).
#include "wgrad.hpp"

#include <algorithm>

namespace mmnn {

static void wg3_tile(int W, int& TD, int& TH, int& TW) {
  if (W > 16) { TD = 1; TH = 2; TW = 32; }
  else if (W > 8) { TD = 1; TH = 4; TW = 16; }
  else if (W > 4) { TD = 2; TH = 4; TW = 8; }
  else { TD = 4; TH = 4; TW = 4; }
}

// channel-group width: 8 waves x 32 channels from 256 input channels on, else 4 waves (2-wave blocks spent more time staging the
// 128-row dOut operand than multiplying; with 4 waves the ones beyond Cin just help staging).  Below 32^3 voxels the wide group already
// pays from 97 channels on (r03, 16^3: all twelve layers of block 2 in one 8-wave launch 146 us against 159 us in two launches; at
// 32^3 the same choice costs 340 against 281 us).
static int wg1_wc(int Cin, long V) {
  static const int thr_env = [] { const char* e = getenv("MMNN_WG1_WC8_FROM"); int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
  const int thr = thr_env > 0 ? thr_env : (V >= 32768 ? 256 : 97);
  return Cin >= thr ? 8 : 4;
}

int wgrad_pick_splits(int taps, int N, int D, int H, int W, int M, int Cin, int batch) {
  if (batch < 1) batch = 1;      // layers that share one launch: the block budget below is that of the whole launch
  if (taps == 27) {
    int TD, TH, TW;
    wg3_tile(W, TD, TH, TW);
    const long ntiles = (long)N * cdiv(D, TD) * cdiv(H, TH) * cdiv(W, TW);
    const int cgroups = cdiv(Cin, 32);
    static const int cap = [] { const char* e = getenv("MMNN_WG3_SPLIT_CAP"); int v = e ? atoi(e) : 0; return v > 0 ? v : 64; }();
    long s = (cap > 64 ? 1024 : 512) / ((long)cgroups * batch);
    if (s > cap) s = cap;
    if (s > ntiles / 2) s = ntiles / 2;
    return s < 1 ? 1 : (int)s;
  }
  const long V = (long)D * H * W;
  const long nchunks = (long)N * cdiv(V, 64);
  const int wc = wg1_wc(Cin, V);
  // Each block owns a (128 x 32*wc) tile of the weight gradient and loops over its share of the 64-voxel chunks.  The kernel is
  // HBM-heavy at block 1 (every (layer, channel group) pair re-reads the 128-row dOut operand: ~0.9 GB per launch), so what
  // matters is balance: ~2048 equal blocks per launch = four rounds of two blocks per CU measured best (r02, after the loads left
  // the FLAT path: 317 / 281 / 266 / 295 us at 1024 / 1536 / 2048 / 3072; step 8.84 / 8.81 / 8.79 / 8.85 ms including the larger
  // reduction), bounded by the slab the reduction kernel then has to read (64 MiB per layer).
  static const int target = [] { const char* e = getenv("MMNN_WG1_BLOCKS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 2048; }();
  const long groups = (long)cdiv(Cin, 32 * wc) * cdiv(M, 128);
  // batch > 1: the number of (layer, channel group) pairs that share the launch -- every block of the launch then gets the same
  // number of chunks, whatever its layer's channel count
  long s = batch > 1 ? target / batch : target / groups;
  const long slab_cap = ((long)64 << 20) / ((long)M * Cin * 4);
  if (s > slab_cap) s = slab_cap;
  if (s > nchunks / 2) s = nchunks / 2;
  return s < 1 ? 1 : (int)s;
}

int wgrad1_channel_width(int Cin, long V) { return 32 * wg1_wc(Cin, V); }

// kernel translation units
int wgrad3_launch(const WgradArgs& a, int pro_x, hipStream_t s);
int wgrad3_launch_batched(const WgradArgs* host, const WgradArgs* dev, int count, uint64_t seed, hipStream_t stream);
int wgrad1_launch(const WgradArgs& a, int pro_x, int wc, hipStream_t s);
int wgrad1_launch_batched(const WgradArgs* host, const WgradArgs* dev, int count, uint64_t seed, int wc, hipStream_t stream);

int launch_wgrad_batched(const WgradArgs* host, const WgradArgs* dev, int count, uint64_t seed, int taps, int pro_x, hipStream_t stream) {
  MMNN_REQUIRE(host && dev && count >= 1 && count <= 65535, "wgrad batch: bad table (count %d)", count);
  MMNN_REQUIRE(pro_x == PRO_BNRELU, "wgrad batch: only the dense-layer kernels (BN+ReLU input prologue) are batched");
  MMNN_REQUIRE(taps == 1 || taps == 27, "wgrad: taps must be 1 or 27");
  const WgradArgs& f = host[0];
  for (int i = 0; i < count; ++i) {
    const WgradArgs& a = host[i];
    MMNN_REQUIRE(a.N == f.N && a.D == f.D && a.H == f.H && a.W == f.W && a.M == f.M, "wgrad batch: layer %d differs in extent", i);
    MMNN_REQUIRE(a.N > 0 && a.D > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.M > 0, "wgrad: non-positive extent");
    MMNN_REQUIRE((long)a.D * a.H * a.W < (1l << 30), "wgrad: volume too large for 32-bit voxel indices");
    MMNN_REQUIRE(taps != 27 || (long)33 * a.D * a.H * a.W < (1l << 31), "wgrad: volume too large for 32-bit element offsets of a 32-channel group");
    MMNN_REQUIRE(a.g0 && a.g1 && a.x && a.slab, "wgrad: null operand");
    MMNN_REQUIRE(a.nsplit >= 1 && a.nsplit <= 65535, "wgrad: bad split count %d", a.nsplit);
    MMNN_REQUIRE(a.slab_stride >= (long)taps * a.M * a.Cin, "wgrad: slab stride too small");
    MMNN_REQUIRE(taps == 27 ? a.M <= 32 : a.M <= 128, "wgrad batch: %d output channels exceed one block row", a.M);
    MMNN_REQUIRE(taps == 27 || wg1_wc(a.Cin, (long)a.D * a.H * a.W) == wg1_wc(f.Cin, (long)f.D * f.H * f.W), "wgrad batch: layer %d needs another channel-group width", i);
  }
  if (taps == 27) return wgrad3_launch_batched(host, dev, count, seed, stream);
  return wgrad1_launch_batched(host, dev, count, seed, wg1_wc(f.Cin, (long)f.D * f.H * f.W), stream);
}

int launch_wgrad(const WgradArgs& a, int taps, int pro_x, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.D > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.M > 0, "wgrad: non-positive extent");
  MMNN_REQUIRE((long)a.D * a.H * a.W < (1l << 30), "wgrad: volume too large for 32-bit voxel indices");
  MMNN_REQUIRE(taps != 27 || (long)33 * a.D * a.H * a.W < (1l << 31), "wgrad: volume too large for 32-bit element offsets of a 32-channel group");
  MMNN_REQUIRE(a.g0 && a.g1 && a.x && a.slab, "wgrad: null operand");
  MMNN_REQUIRE(a.nsplit >= 1 && a.nsplit <= 65535, "wgrad: bad split count %d", a.nsplit);
  MMNN_REQUIRE(taps == 1 || taps == 27, "wgrad: taps must be 1 or 27");
  MMNN_REQUIRE(taps != 27 || a.M <= 32, "wgrad: the 3x3x3 kernel handles at most 32 output channels (growth rate), got %d", a.M);
  MMNN_REQUIRE(a.slab_stride >= (long)taps * a.M * a.Cin, "wgrad: slab stride too small");
  MMNN_REQUIRE(pro_x == PRO_BNRELU || pro_x == PRO_NONE, "wgrad: unsupported input prologue %d", pro_x);
  return taps == 27 ? wgrad3_launch(a, pro_x, stream) : wgrad1_launch(a, pro_x, wg1_wc(a.Cin, (long)a.D * a.H * a.W), stream);
}

}  // namespace mmnn
