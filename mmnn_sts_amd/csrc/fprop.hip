// Host-side dispatch of the implicit-GEMM convolution template (fprop.hpp): picks a tile shape from the
// spatial extent and the number of output rows, validates every shape the kernel indexes with, and launches.
#include <stdlib.h>

#include "fprop.hpp"

namespace mmnn {

// capacity of the K-split scratch the plan provides (densenet.hip); the split is skipped when it would not fit
size_t kz_part_bytes = 0, kz_cnt_entries = 0;

template <int TAPS, int PRO, int EPI, int WM, int WN, int KS, int MT, int NT, int KC, int TD, int TH, int TW, bool SPEC = false>
static int launch_cfg(const FpropArgs& a, hipStream_t stream) {
  constexpr bool KZ_OK = (MT * NT == 1) || (TAPS == 27 && TW <= 16);   // only the small-extent tiles are ever short of blocks
  using C = FpropCfg<TAPS, PRO, EPI, WM, WN, KS, MT, NT, KC, TD, TH, TW, SPEC>;
  auto kern = fprop_kernel<TAPS, PRO, EPI, WM, WN, KS, MT, NT, KC, TD, TH, TW, SPEC>;
  size_t smem = C::smem_bytes(a.Cin);
  MMNN_REQUIRE(smem <= 160 * 1024, "fprop: %zu bytes of LDS needed (Cin=%d) exceeds 160 KiB", smem, a.Cin);
  {
    static const char* env = getenv("MMNN_FPROP_MIN_SMEM");   // experiment knob: force fewer blocks per CU
    if (env) { size_t v = (size_t)atol(env); if (v > smem && v <= 160 * 1024) smem = v; }
  }
  static size_t configured = 0;
  if (smem > configured) {
    MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    configured = smem;
  }
  long tiles;
  if (TAPS == 27) tiles = (long)a.N * cdiv(a.D, TD) * cdiv(a.H, TH) * cdiv(a.W, TW);
  else tiles = (long)a.N * cdiv((long)a.D * a.H * a.W, C::V_B);
  const int mtiles = cdiv(a.M, C::M_B);
  MMNN_REQUIRE(tiles > 0 && tiles < (1l << 31) && mtiles <= 65535, "fprop: grid out of range");
  // cross-block K-split: when the (voxel, row) tiles alone cannot fill the chip, slices of the channel axis become blocks too
  int kz = 1;
  static const bool kz_off = []() { const char* e = getenv("MMNN_NO_KZ"); return e && e[0] == '1'; }();   // debugging aid
  if (a.kz_part && a.kz_cnt && KZ_OK && !kz_off) {
    const int nch = cdiv(a.Cin, KC);
    while (kz * 2 <= nch && tiles * mtiles * kz * 2 <= 256 && kz < 8) kz *= 2;
    if (kz > 1) {
      const size_t need = (size_t)tiles * mtiles * kz * (WM * WN * MT * NT) * 1024 * sizeof(float);
      if (need > kz_part_bytes || (size_t)tiles * mtiles > kz_cnt_entries) kz = 1;
    }
  }
  MMNN_LAUNCH(kern, dim3((unsigned)tiles, (unsigned)mtiles, (unsigned)kz), dim3(C::NTHREADS), smem, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// Tile selection.  Template arguments: <TAPS, PRO, EPI, WM, WN, KS, MT, NT, KC, TD, TH, TW>; a block computes
// (WM*MT*32) output rows x (WN*NT*32) voxels with WM*WN*KS waves, KS wave groups splitting each channel chunk.
// Small extents get small voxel tiles + a deep K-split so that the late dense blocks (8^3, 4^3 voxels) still put
// hundreds of waves on the chip instead of a dozen.
template <int TAPS, int PRO, int EPI>
static int dispatch(const FpropArgs& a, hipStream_t s) {
  const long V = (long)a.D * a.H * a.W;
  if (TAPS == 1) {
    const long blocks_a = (long)a.N * cdiv(V, 128) * cdiv(a.M, 128);
    const long blocks_b = (long)a.N * cdiv(V, 64) * cdiv(a.M, 64);
    // 128-wide tile: KC = 8 / 16 / 32 measure the same (37 us at 32^3: a fixed ~17 us of prologue, output write and statistics,
    // then 0.15 us per input channel = 115 TFLOP/s).  64-wide tile: deep chunks, 4x fewer barriers than KC = 16 (25 -> 17 us at 16^3).
    if (blocks_a >= 192) return launch_cfg<1, PRO, EPI, 2, 2, 1, 2, 2, 16, 1, 1, 128>(a, s);
    if (blocks_b >= 192) return launch_cfg<1, PRO, EPI, 2, 2, 2, 1, 1, 64, 1, 1, 64>(a, s);
    return launch_cfg<1, PRO, EPI, 1, 1, 8, 1, 1, 128, 1, 1, 32>(a, s);   // few voxels: deep K chunks (the K loop is latency-bound)
  }
  if (a.M <= 32) {
    if (a.W > 16) return launch_cfg<27, PRO, EPI, 1, 4, 1, 1, 2, 8, 2, 4, 32, true>(a, s);
    if (a.W > 8) return launch_cfg<27, PRO, EPI, 1, 2, 4, 1, 1, 16, 1, 4, 16>(a, s);
    if (a.W > 4) return launch_cfg<27, PRO, EPI, 1, 1, 8, 1, 1, 16, 1, 4, 8>(a, s);
    return launch_cfg<27, PRO, EPI, 1, 1, 8, 1, 1, 16, 2, 4, 4>(a, s);
  }
  if (a.W > 16) {
    // KC = 2: the smallest chunk (one MFMA k-pair per tap) keeps the LDS footprint low enough for 3-4 blocks per CU, which
    // hides the staging latency better than loader waves or a deeper chunk do here (194 -> 173 us at block 1).
    static const char* e = getenv("MMNN_DGRAD_KC");   // experiment knob
    if (e && e[0] == '4') return launch_cfg<27, PRO, EPI, 2, 2, 1, 2, 2, 4, 1, 4, 32>(a, s);
    return launch_cfg<27, PRO, EPI, 2, 2, 1, 2, 2, 2, 1, 4, 32>(a, s);
  }
  if (a.W > 8) return launch_cfg<27, PRO, EPI, 2, 2, 2, 2, 1, 4, 1, 4, 16>(a, s);
  if (a.W > 4) return launch_cfg<27, PRO, EPI, 4, 1, 2, 1, 1, 4, 1, 4, 8>(a, s);
  return launch_cfg<27, PRO, EPI, 4, 1, 2, 1, 1, 4, 2, 4, 4>(a, s);
}

int launch_fprop(const FpropArgs& a, int taps, int pro, int epi, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.D > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.M > 0, "fprop: non-positive extent");
  MMNN_REQUIRE((long)a.D * a.H * a.W < (1l << 30), "fprop: volume too large for 32-bit voxel indices");
  MMNN_REQUIRE(a.in0 && a.w && a.out && a.w_ld >= a.M, "fprop: null operand or w_ld < M");
  MMNN_REQUIRE(pro != PRO_GRAD || a.in1, "fprop: PRO_GRAD needs the normalised tensor (in1)");
  MMNN_REQUIRE((epi != EPI_MASK_STORE && epi != EPI_MASK_ACCUM) || (a.ex && a.dgamma && a.dbeta), "fprop: mask epilogue operands missing");
#define MMNN_CASE(T, P, E) \
  if (taps == T && pro == P && epi == E) return dispatch<T, P, E>(a, stream);
  MMNN_CASE(1, PRO_BNRELU, EPI_STORE_STATS)    // dense-layer conv1 forward
  MMNN_CASE(1, PRO_NONE, EPI_STORE_STATS)      // transition conv forward (input already pooled)
  MMNN_CASE(1, PRO_GRAD, EPI_MASK_ACCUM)       // conv1 data gradient -> G of the block buffer
  MMNN_CASE(1, PRO_GRAD, EPI_STORE)            // transition conv data gradient
  MMNN_CASE(27, PRO_BNRELU, EPI_STORE_STATS)   // dense-layer conv2 forward
  MMNN_CASE(27, PRO_GRAD, EPI_MASK_STORE)      // conv2 data gradient
#undef MMNN_CASE
  set_error("fprop: unsupported (taps=%d, pro=%d, epi=%d)", taps, pro, epi);
  return 1;
}

}  // namespace mmnn
