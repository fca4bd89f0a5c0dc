// Weight-gradient kernels (autograd adjoint of the dense-layer convolutions wrt their weights, main.py:469 ->
// models/densenet.py:78,82,147) on fp32 MFMA, split over the voxel (reduction) axis into deterministic partial slabs.
//
//     dW[m][c][tap] = sum_{n,v}  dOut[n][m][v] * f(in[n][c][v + off(tap)])
//
// with dOut = BN-backward(G, X) evaluated on the fly (PRO_GRAD) and f = ReLU(BN(.)) (PRO_BNRELU) or identity.
// MFMA mapping: i = output channel m, j = input channel c, k = voxel.  Both operands are staged channel-major in LDS
// with an ODD row stride, so the 32 lanes of a half (32 different channels, same voxel) hit 32 different banks.
// Lanes 0-31 take voxel s of the tile, lanes 32-63 voxel s+32.
//
// Every block owns one (split, channel-group) pair, walks its share of spatial tiles and stores its partial result to
// slab[split]; slabs are summed by the gradient finaliser (finalize.hip) -- no float atomics, bit-reproducible.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "fprop.hpp"

namespace mmnn {

struct WgradArgs {
  int N, D, H, W;
  int M, Cin;
  // dOut operand: G (in g0) and the normalised tensor (in g1), BN-backward coefficients, channel dropout
  const float* g0; long g0_ns; int g0_coff;
  const float* g1; long g1_ns; int g1_coff;
  BnBwd gr;
  DropCfg drop;
  // input operand
  const float* x; long x_ns; int x_coff;
  BnFwd bn;                                   // PRO_BNRELU (ignored for PRO_NONE)
  // output slabs: 1x1: slab[split][m][c]   3x3x3: slab[split][tap][m][c]
  float* slab; long slab_stride; int nsplit;
  unsigned long long* trace;   // developer builds (MMNN_PHASE_TRACE): per-block phase cycle sums; null otherwise
};

int launch_wgrad(const WgradArgs& a, int taps, int pro_x, hipStream_t stream);
// `count` layers of identical extent (N, D, H, W), M and prologue in one launch.  host: the arguments (validated here);
// dev_table: the same `count` entries in device memory (already uploaded); seed: this step's dropout seed (overrides drop.seed).
// For taps == 1 every entry must select the same channel-group width (wgrad1_channel_width(Cin, V)).
int launch_wgrad_batched(const WgradArgs* host, const WgradArgs* dev_table, int count, uint64_t seed, int taps, int pro_x, hipStream_t stream);
int wgrad1_channel_width(int Cin, long V);   // input channels per block of the 1x1x1 kernel (128 or 256) at V voxels per sample
int wgrad_pick_splits(int taps, int N, int D, int H, int W, int M, int Cin, int batch = 1);   // batch: layers sharing the launch

#if defined(__HIPCC__)

#if !defined(MMNN_TRACE_TID)
#define MMNN_TRACE_TID 0     // thread whose phase stamps a trace build records (developer builds only; 320 = a 4-tap wave of wgrad3)
#endif

// Pointers read from a device-resident argument table (the batched kernels) are "generic" to the compiler, and every access
// through them becomes a FLAT instruction -- which counts on the LDS counter (lgkmcnt) as well as on vmcnt, so a wait for an LDS
// operand inside the MFMA loop also waits for the global loads in flight.  The tile loops therefore load through pointers that
// are explicitly in the global address space (kernel arguments passed by value get this automatically).
#define MMNN_GLOBAL __attribute__((address_space(1)))

// ----------------------------------------------------------------------------------------------------------------
// (r02: a loader / compute wave specialisation of this kernel -- waves 0-3 stage tile t+1 into a second LDS buffer and take one tap,
// waves 4-7 take 6,6,6,5 taps, one barrier per tile -- measured 10 % SLOWER at 32^3 (1000 vs 905 us for block 1's six layers), 13 %
// with s_setprio 1 on the compute waves: moving work between the two waves of a SIMD did not net, as MI355X_MICROARCH.md warns.)
// 3x3x3:  block = 8 waves sharing the 27 taps 3,3,3,3,3,4,4,4 (two waves per SIMD: 6/7/7/7 tap-units per SIMD, 96 % balanced;
// 9 waves of 3 taps would put 3 waves on one SIMD and 2 on the others), one 32-channel group of c, <= 4 accumulator tiles per wave.
// ----------------------------------------------------------------------------------------------------------------
template <int TD, int TH, int TW>
struct Wg3Cfg {
  static constexpr int VT = TD * TH * TW;            // 64 voxels per tile
  static constexpr int RS = TW + 2, HS = TH + 2, DS = TD + 2;
  static constexpr int XS = (DS * HS * RS) | 1;      // odd channel stride
  static constexpr int YS = VT + 1;
  static constexpr int NTHREADS = 8 * 64;
  static_assert(VT == 64, "tile must hold 64 voxels (two per MFMA k-step, 32 steps)");
  static_assert(32 % TW == 0, "tile width must divide 32");
  static constexpr int STAGE = 32 * XS + 32 * YS;   // floats of one staged tile: input halo box + dOut
  static size_t smem_bytes() { return sizeof(float) * (2 * STAGE + 3 * 32 + 2 * 32 + 3 * 32); }   // two tiles: see wgrad3_body
};

// MFMA over one staged 64-voxel tile for NTP taps: 32 k-steps; the operand reads of step s+1 are issued before the MFMAs of
// step s (two register sets).  Accumulators are passed by reference so that they stay in registers in both instantiations.
// filler(s) is called once per k-step, between the MFMA groups: the caller's staging work for the NEXT tile (BN+ReLU, LDS writes into
// the other buffer, global loads of the tile after) issues in the shadow of this wave's own MFMAs -- the matrix pipe is busy 64 cycles
// per instruction, an in-order wave has nothing else to issue meanwhile.
template <int NTP, int TD, int TH, int TW, class F>
__device__ __forceinline__ void wg3_mfma(f32x16 (&acc)[4], const float* yl, const float* x0, const float* x1, const float* x2, const float* x3, F&& filler) {
  constexpr int RS = Wg3Cfg<TD, TH, TW>::RS, HS = Wg3Cfg<TD, TH, TW>::HS;
  float a0, a1, b0[4], b1[4];
  auto rd = [&](int s, float& av, float (&bv)[4]) {
    const int wx = s % TW, hy = (s / TW) % TH, dz = s / (TW * TH);
    const int p0 = (dz * HS + hy) * RS + wx;
    av = yl[s];
    bv[0] = x0[p0]; bv[1] = x1[p0]; bv[2] = x2[p0];
    if (NTP == 4) bv[3] = x3[p0];
  };
  auto mm = [&](float av, const float (&bv)[4]) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[1], acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[2], acc[2], 0, 0, 0);
    if (NTP == 4) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[3], acc[3], 0, 0, 0);
  };
  rd(0, a0, b0);
#pragma unroll
  for (int s = 0; s < 32; s += 2) {
    rd(s + 1, a1, b1);
    __builtin_amdgcn_sched_group_barrier(0x100, NTP + 1, 0);
    mm(a0, b0);
    __builtin_amdgcn_sched_group_barrier(0x008, NTP, 0);
    filler(s);
    if (s + 2 < 32) rd(s + 2, a0, b0);
    __builtin_amdgcn_sched_group_barrier(0x100, NTP + 1, 0);
    mm(a1, b1);
    __builtin_amdgcn_sched_group_barrier(0x008, NTP, 0);
    filler(s + 1);
  }
}

template <int PRO_X, int TD, int TH, int TW>
__device__ __forceinline__ void wgrad3_body(const WgradArgs& a, const int split, const int cg) {
  using C = Wg3Cfg<TD, TH, TW>;
  constexpr int RS = C::RS, HS = C::HS, DS = C::DS, XS = C::XS, YS = C::YS, NTHREADS = C::NTHREADS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                 // [32 c][XS]   (tile buffer 0; buffer 1 follows at + C::STAGE)
  float* Ys = Xs + 32 * XS;         // [32 m][YS]
  float* gcoef = smem + 2 * C::STAGE;   // p,q,r for the 32 rows of dOut
  float* xcoef = gcoef + 96;        // a,b for the 32 input channels
  float* gbase = xcoef + 64;        // p,q,r before the per-sample dropout scale
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int V = a.D * a.H * a.W;
  const int c0 = cg * 32;
  const int nw = (a.W + TW - 1) / TW, nh = (a.H + TH - 1) / TH, nd = (a.D + TD - 1) / TD;
  const int tiles_per_n = nw * nh * nd;
  const int ntiles = a.N * tiles_per_n;
  const int t_begin = (int)((long)ntiles * split / a.nsplit), t_end = (int)((long)ntiles * (split + 1) / a.nsplit);

  // BN coefficients of both operands (one memory round trip over the fp64 statistics).  In the fast path they are computed
  // AFTER the first tile's loads were issued, so the two latencies overlap.
  auto coefficients = [&]() {
    if (tid < 32) {
      float ca = 0.f, cb = 0.f, mu, rs;
      if (PRO_X == PRO_BNRELU && c0 + tid < a.Cin) bn_fwd_coef(a.bn, c0 + tid, ca, cb, mu, rs);
      xcoef[tid] = ca; xcoef[32 + tid] = cb;
      float p = 0.f, q = 0.f, r = 0.f;
      if (tid < a.M) bn_bwd_coef(a.gr, tid, p, q, r);
      gbase[tid] = p; gbase[32 + tid] = q; gbase[64 + tid] = r;
    }
  };
  const int tap0 = (wave < 5) ? wave * 3 : 15 + (wave - 5) * 4;       // first tap of this wave
  const int ntap = (wave < 5) ? 3 : 4;
  // lane-half offset: voxel (s + 32*half) = voxel s shifted by 32/TW rows
  constexpr int ROWS_PER_HALF = 32 / TW;
  const int hrow = half * ROWS_PER_HALF;
  const int hoff = ((hrow / TH) * HS + (hrow % TH)) * RS;
  const float* xb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tap = min(tap0 + j, 26);
    xb[j] = Xs + l31 * XS + hoff + ((tap / 9) * HS + (tap / 3) % 3) * RS + tap % 3;
  }
  const float* yl = Ys + l31 * YS + 32 * half;

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- MFMA over one staged tile: 32 k-steps (2 voxels each) x 3 taps; operand reads of step s+1 issued before the
  // MFMAs of step s (two register sets) ----
  auto mfma_tile = [&](int boff, auto&& filler) {   // wave-uniform choice of the tap count; boff: float offset of the tile buffer
    if (ntap == 4) wg3_mfma<4, TD, TH, TW>(acc, yl + boff, xb[0] + boff, xb[1] + boff, xb[2] + boff, xb[3] + boff, filler);
    else wg3_mfma<3, TD, TH, TW>(acc, yl + boff, xb[0] + boff, xb[1] + boff, xb[2] + boff, xb[2] + boff, filler);
  };
  auto no_filler = [](int) {};
  auto tile_origin = [&](int tile, int& n, int& d0, int& h0, int& w0) {
    int b = tile;
    w0 = (b % nw) * TW; b /= nw;
    h0 = (b % nh) * TH; b /= nh;
    d0 = (b % nd) * TD; b /= nd;
    n = b;
  };
  const bool vec = ((a.W & 3) == 0) && ((TW & 3) == 0) && ((((uintptr_t)a.x | (uintptr_t)a.g0 | (uintptr_t)a.g1) & 15) == 0) && ((V & 3) == 0);

  if (vec) {
    // ===== fast path: 16-byte unconditional loads of tile t+1 issued before the MFMA loop of tile t (register staging) =====
    constexpr int ROWS = 32 * DS * HS, VPR = TW / 4;
    constexpr int XV_ITEMS = ROWS * VPR, XH_ITEMS = ROWS * 2, Y_ITEMS = 32 * 16;
    constexpr int XV_IT = (XV_ITEMS + NTHREADS - 1) / NTHREADS, XH_IT = (XH_ITEMS + NTHREADS - 1) / NTHREADS;
    constexpr int Y_IT = (Y_ITEMS + NTHREADS - 1) / NTHREADS;
    f32x4 xv[XV_IT], y0[Y_IT], y1[Y_IT];
    float xh[XH_IT];
    unsigned okv = 0, okh = 0, oky = 0;
    // Per-item descriptors that do not depend on the tile, computed ONCE per block: global offset relative to the tile origin,
    // LDS destination, local channel and the (dz, hy) the bounds checks need.  Per tile an item then costs one add and four
    // compares instead of a dozen divisions by constants and 64-bit multiplies -- the staging arithmetic of both waves of a SIMD
    // was 4.4k of the 19.9k cycles a tile takes (phase trace r02), and it delays the partner wave's MFMAs (VALU issue is shared).
    int xv_s[XV_IT], xv_l[XV_IT], xv_k[XV_IT];      // k packs cl | dz << 8 | hy << 16 | (4 * q) << 24, or -1 for "never valid"
    int xh_s[XH_IT], xh_l[XH_IT], xh_k[XH_IT];
    int y_s[Y_IT], y_k[Y_IT];                       // y_k packs m | dz << 8 | hy << 16 | wx << 24
#pragma unroll
    for (int i = 0; i < XV_IT; ++i) {
      const int it = tid + i * NTHREADS, r = it / VPR, q4 = 4 * (it % VPR);
      const int hy = r % HS, dz = (r / HS) % DS, cl = r / (HS * DS);
      const bool ok = (it < XV_ITEMS) && (c0 + cl < a.Cin);
      xv_s[i] = cl * V + (dz * a.H + hy) * a.W + q4;
      xv_l[i] = cl * XS + (dz * HS + hy) * RS + 1 + q4;
      xv_k[i] = ok ? (cl | (dz << 8) | (hy << 16) | (q4 << 24)) : -1;
    }
#pragma unroll
    for (int i = 0; i < XH_IT; ++i) {
      const int it = tid + i * NTHREADS, r = it >> 1, side = it & 1;
      const int hy = r % HS, dz = (r / HS) % DS, cl = r / (HS * DS);
      const bool ok = (it < XH_ITEMS) && (c0 + cl < a.Cin);
      xh_s[i] = cl * V + (dz * a.H + hy) * a.W + (side ? TW : -1);
      xh_l[i] = cl * XS + (dz * HS + hy) * RS + (side ? TW + 1 : 0);
      xh_k[i] = ok ? (cl | (dz << 8) | (hy << 16) | (side << 24)) : -1;
    }
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
      const int it = tid + i * NTHREADS, m = it >> 4, t = 4 * (it & 15);
      const int wx = t % TW, hy = (t / TW) % TH, dz = t / (TW * TH);
      const bool ok = (it < Y_ITEMS) && m < a.M;
      y_s[i] = m * V + (dz * a.H + hy) * a.W + wx;
      y_k[i] = ok ? (m | (dz << 8) | (hy << 16) | (wx << 24)) : -1;
    }
    auto load_tile = [&](int tile) {
      int n, d0, h0, w0;
      tile_origin(tile, n, d0, h0, w0);
      const MMNN_GLOBAL float* xn = (const MMNN_GLOBAL float*)a.x + (long)n * a.x_ns + (long)(a.x_coff + c0) * V;
      const MMNN_GLOBAL float* g0n = (const MMNN_GLOBAL float*)a.g0 + (long)n * a.g0_ns + (long)a.g0_coff * V;
      const MMNN_GLOBAL float* g1n = (const MMNN_GLOBAL float*)a.g1 + (long)n * a.g1_ns + (long)a.g1_coff * V;
      const int xbase = ((d0 - 1) * a.H + (h0 - 1)) * a.W + w0;      // tile origin of the halo box (may be negative: only used when in range)
      const int ybase = (d0 * a.H + h0) * a.W + w0;
      okv = okh = oky = 0;
#pragma unroll
      for (int i = 0; i < XV_IT; ++i) {
        const int k = xv_k[i];
        const int d = d0 - 1 + ((k >> 8) & 255), h = h0 - 1 + ((k >> 16) & 255), w = w0 + ((k >> 24) & 255);
        const bool ok = k >= 0 && (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && w < a.W;
        okv |= (ok ? 1u : 0u) << i;
        xv[i] = *reinterpret_cast<const MMNN_GLOBAL f32x4*>(xn + (ok ? xv_s[i] + xbase : 0));
      }
#pragma unroll
      for (int i = 0; i < XH_IT; ++i) {
        const int k = xh_k[i];
        const int d = d0 - 1 + ((k >> 8) & 255), h = h0 - 1 + ((k >> 16) & 255), w = ((k >> 24) & 1) ? w0 + TW : w0 - 1;
        const bool ok = k >= 0 && (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
        okh |= (ok ? 1u : 0u) << i;
        xh[i] = xn[ok ? xh_s[i] + xbase : 0];
      }
#pragma unroll
      for (int i = 0; i < Y_IT; ++i) {
        const int k = y_k[i];
        const int d = d0 + ((k >> 8) & 255), h = h0 + ((k >> 16) & 255), w = w0 + ((k >> 24) & 255);
        const bool ok = k >= 0 && d < a.D && h < a.H && w < a.W;
        const int g = ok ? y_s[i] + ybase : 0;
        oky |= (ok ? 1u : 0u) << i;
        y0[i] = *reinterpret_cast<const MMNN_GLOBAL f32x4*>(g0n + g);
        y1[i] = *reinterpret_cast<const MMNN_GLOBAL f32x4*>(g1n + g);
      }
    };
    // Tile-invariant coefficients of this thread's items, in registers: the per-tile stores then read nothing from LDS.
    float xv_a[XV_IT], xv_b[XV_IT], xh_a[XH_IT], xh_b[XH_IT];
    float yb_p[Y_IT], yb_q[Y_IT], yb_r[Y_IT], y_p[Y_IT], y_q[Y_IT], y_r[Y_IT];
    int ld_n = -1, st_n = -1;     // sample of the tile held in the staging registers / sample the dOut coefficients are scaled for
    auto item_coefficients = [&]() {
#pragma unroll
      for (int i = 0; i < XV_IT; ++i) { const int cl = xv_k[i] & 31; xv_a[i] = xcoef[cl]; xv_b[i] = xcoef[32 + cl]; }
#pragma unroll
      for (int i = 0; i < XH_IT; ++i) { const int cl = xh_k[i] & 31; xh_a[i] = xcoef[cl]; xh_b[i] = xcoef[32 + cl]; }
#pragma unroll
      for (int i = 0; i < Y_IT; ++i) { const int m = y_k[i] & 31; yb_p[i] = gbase[m]; yb_q[i] = gbase[32 + m]; yb_r[i] = gbase[64 + m]; }
    };
    // The staging work of one tile in NUNITS pieces (one item each), so that the MFMA loop can take one piece per k-step.
    constexpr int NUNITS = XV_IT + XH_IT + Y_IT + 1;
    auto store_unit = [&](int u, float* X, float* Y) {
      if (u == 0) {
        if (ld_n != st_n) {   // dropout scale depends on the sample (wave-uniform, at most once per block and sample)
#pragma unroll
          for (int i = 0; i < Y_IT; ++i) {
            const float sc = drop_scale(a.drop, ld_n, y_k[i] & 31);
            y_p[i] = yb_p[i] * sc; y_q[i] = yb_q[i] * sc; y_r[i] = yb_r[i] * sc;
          }
          st_n = ld_n;
        }
        return;
      }
      u -= 1;
#pragma unroll
      for (int i = 0; i < XV_IT; ++i) {
        if (i == u && tid + i * NTHREADS < XV_ITEMS) {
          const bool ok = (okv >> i) & 1u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float o = (PRO_X == PRO_BNRELU) ? fmaxf(fmaf(xv_a[i], xv[i][e], xv_b[i]), 0.f) : xv[i][e];
            X[xv_l[i] + e] = ok ? o : 0.f;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < XH_IT; ++i) {
        if (XV_IT + i == u && tid + i * NTHREADS < XH_ITEMS) {
          const float o = (PRO_X == PRO_BNRELU) ? fmaxf(fmaf(xh_a[i], xh[i], xh_b[i]), 0.f) : xh[i];
          X[xh_l[i]] = ((okh >> i) & 1u) ? o : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < Y_IT; ++i) {
        const int it = tid + i * NTHREADS;
        if (XV_IT + XH_IT + i == u && it < Y_ITEMS) {
          const int m = it >> 4, t = 4 * (it & 15);
          const bool ok = (oky >> i) & 1u;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            Y[m * YS + t + e] = ok ? fmaf(y_p[i], y0[i][e], fmaf(y_q[i], y1[i][e], y_r[i])) : 0.f;
        }
      }
    };
#if defined(MMNN_PHASE_TRACE)
    const bool tracing = a.trace != nullptr && tid == MMNN_TRACE_TID && split < 16 && cg == 0;
    unsigned long long tr[6] = {0, 0, 0, 0, 0, 0}, tprev = tracing ? __builtin_amdgcn_s_memtime() : 0;
    auto lap = [&](int k) { if (tracing) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr[k] += t - tprev; tprev = t; } };
#else
    auto lap = [&](int) {};
#endif
    // Software pipeline over the block's tiles, two LDS buffers: while the MFMAs of tile t run out of buffer t&1, the same waves
    // write tile t+1 (already in registers) into the other buffer and then issue the global loads of tile t+2 -- all of it between
    // their own MFMAs.  One barrier per tile.  (r02 phase trace of the former store / barrier / MFMA / barrier sequence: 5.6k of a
    // tile's 19.9k cycles had the matrix pipe idle, both waves of each SIMD staging in lockstep.)
    if (t_begin < t_end) {
      int n, d0, h0, w0;
      tile_origin(t_begin, n, d0, h0, w0);
      load_tile(t_begin); ld_n = n;
    }
    coefficients();
    __syncthreads();
    item_coefficients();
    if (t_begin < t_end) {
#pragma unroll
      for (int u = 0; u < NUNITS; ++u) store_unit(u, Xs, Ys);
      if (t_begin + 1 < t_end) {
        int n, d0, h0, w0;
        tile_origin(t_begin + 1, n, d0, h0, w0);
        load_tile(t_begin + 1); ld_n = n;
      }
    }
    __syncthreads();
    lap(0);
    constexpr int LOAD_AT = NUNITS + 1;      // k-step after which the registers are free again
    static_assert(LOAD_AT < 32, "staging pieces must fit the k-steps of a tile");
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int cur = (tile - t_begin) & 1;
      float* Xn = Xs + (cur ^ 1) * C::STAGE;
      float* Yn = Ys + (cur ^ 1) * C::STAGE;
      if (tile + 1 < t_end) {
        const int nxt = min(tile + 2, t_end - 1);     // the last tile is loaded twice (never stored the second time)
        int n2, d2, h2, w2;
        tile_origin(nxt, n2, d2, h2, w2);
        mfma_tile(cur * C::STAGE, [&](int s) {
          if (s < NUNITS) store_unit(s, Xn, Yn);
          if (s == LOAD_AT) { load_tile(nxt); ld_n = n2; }
        });
      } else {
        mfma_tile(cur * C::STAGE, no_filler);
      }
      lap(4);
      __syncthreads();
      lap(5);
    }
#if defined(MMNN_PHASE_TRACE)
    if (tracing) {
      for (int k = 0; k < 6; ++k) a.trace[split * 16 + k] = tr[k];
      a.trace[split * 16 + 6] = (unsigned long long)(t_end - t_begin);
      a.trace[10] = (27ull << 48) | (9ull << 40) | ((unsigned long long)a.M << 16) | (unsigned long long)a.Cin;
      a.trace[11] = ((unsigned long long)gridDim.x << 32) | ((unsigned long long)gridDim.y << 16) | gridDim.z;
    }
#endif
  } else {
  coefficients();
  int cur_n = -1;
  for (int tile = t_begin; tile < t_end; ++tile) {
    int b = tile;
    const int w0 = (b % nw) * TW; b /= nw;
    const int h0 = (b % nh) * TH; b /= nh;
    const int d0 = (b % nd) * TD; b /= nd;
    const int n = b;
    if (n != cur_n) {   // dropout scale depends on the sample: refresh the dOut coefficients
      __syncthreads();
      if (tid < 32) {
        const float s = drop_scale(a.drop, n, tid);
        gcoef[tid] = gbase[tid] * s; gcoef[32 + tid] = gbase[32 + tid] * s; gcoef[64 + tid] = gbase[64 + tid] * s;
      }
      cur_n = n;
      __syncthreads();
    }
    const float* xn = a.x + (long)n * a.x_ns + (long)(a.x_coff + c0) * V;
    const float* g0n = a.g0 + (long)n * a.g0_ns + (long)a.g0_coff * V;
    const float* g1n = a.g1 + (long)n * a.g1_ns + (long)a.g1_coff * V;
    // ---- stage the input halo box (zero padded AFTER the activation) ----
    constexpr int XITEMS = 32 * DS * HS * RS;
#pragma unroll 8
    for (int it = tid; it < XITEMS; it += NTHREADS) {
      const int q = it % RS;
      int row = it / RS;
      const int hy = row % HS; row /= HS;
      const int dz = row % DS;
      const int cl = row / DS;
      const int d = d0 + dz - 1, h = h0 + hy - 1, w = w0 + q - 1;
      // unconditional load from a clamped address (a load under a divergent branch costs one memory round trip EACH)
      const bool ok = c0 + cl < a.Cin && (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
      const float x = xn[ok ? (long)cl * V + ((long)d * a.H + h) * a.W + w : 0];
      const float o = (PRO_X == PRO_BNRELU) ? fmaxf(fmaf(xcoef[cl], x, xcoef[32 + cl]), 0.f) : x;
      Xs[cl * XS + (dz * HS + hy) * RS + q] = ok ? o : 0.f;
    }
    // ---- stage dOut for the tile's 64 voxels ----
#pragma unroll 4
    for (int it = tid; it < 32 * 64; it += NTHREADS) {
      const int t = it & 63, m = it >> 6;
      const int wx = t % TW, hy = (t / TW) % TH, dz = t / (TW * TH);
      const int d = d0 + dz, h = h0 + hy, w = w0 + wx;
      const bool ok = m < a.M && d < a.D && h < a.H && w < a.W;
      const long g = ok ? (long)m * V + ((long)d * a.H + h) * a.W + w : 0;
      const float o = fmaf(gcoef[m], g0n[g], fmaf(gcoef[32 + m], g1n[g], gcoef[64 + m]));
      Ys[m * YS + t] = ok ? o : 0.f;
    }
    __syncthreads();
    mfma_tile(0, no_filler);
    __syncthreads();
  }
  }
  // ---- partial result: slab[split][tap][m][c] ----
  float* out = a.slab + (long)split * a.slab_stride;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tap = tap0 + j;
    if (j < ntap) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = acc_row(r, half);
        if (m < a.M && c0 + l31 < a.Cin) out[((long)tap * a.M + m) * a.Cin + c0 + l31] = acc[j][r];
      }
    }
  }
}

template <int PRO_X, int TD, int TH, int TW>
__global__ void __launch_bounds__(512) wgrad3_kernel(const WgradArgs a) {
  wgrad3_body<PRO_X, TD, TH, TW>(a, blockIdx.x, blockIdx.y);
}

// Several layers in ONE launch (blockIdx.z = layer): the late dense blocks' weight gradients are a few dozen blocks per layer
// each, far too few for 256 CUs, and independent of one another once the data-gradient chain has passed their layers.  The
// argument table lives in device memory and is step-invariant; the per-step dropout seed arrives as a kernel argument.
template <int PRO_X, int TD, int TH, int TW>
__global__ void __launch_bounds__(512) wgrad3_batched_kernel(const WgradArgs* __restrict__ table, const uint64_t seed, const int nsplit,
                                                             const int cgroups, const int ngroups) {
  int z, split, cg;
  if (ngroups > 0) {
    // XCD-aware order (1-D grid): the `cgroups` blocks that share a (layer, split) pair -- they stage the SAME dOut tiles, only their
    // 32 input channels differ -- are dealt to ONE XCD back to back (workgroups go round-robin over the 8 XCDs: blocks b and b + 8
    // share one), so they run at the same time on neighbouring CUs and all but the first read dOut from that XCD's L2 instead of
    // from HBM.  (r02: each channel group re-read dOut from memory: 3.05x the algorithmic bytes of this kernel.)
    const int lin = blockIdx.x, x = lin & 7, j = lin >> 3;
    const int G = x + 8 * (j / cgroups);
    cg = j % cgroups;
    if (G >= ngroups) return;
    z = G / nsplit; split = G - z * nsplit;
  } else {
    z = blockIdx.z; split = blockIdx.x; cg = blockIdx.y;
  }
  WgradArgs a = table[z];
  if (split >= a.nsplit || cg * 32 >= a.Cin) return;   // grid = the largest layer of the batch
  a.drop.seed = seed;
  wgrad3_body<PRO_X, TD, TH, TW>(a, split, cg);
}

// ----------------------------------------------------------------------------------------------------------------
// 1x1x1:  block = WC waves, each owning 32 input channels x all 128 rows of one m-group (4 accumulator tiles).
// ----------------------------------------------------------------------------------------------------------------
template <int WC>
struct Wg1Cfg {
  static constexpr int VK = 64, S = VK + 1;
  static constexpr int NTHREADS = WC * 64;
  static size_t smem_bytes() { return sizeof(float) * (128 * S + 32 * WC * S + 3 * 128 + 2 * 32 * WC + 3 * 128); }
};

template <int PRO_X, int WC>
__device__ __forceinline__ void wgrad1_body(const WgradArgs& a, const int split, const int c0, const int m0) {
  using C = Wg1Cfg<WC>;
  constexpr int S = C::S, VK = C::VK, NTHREADS = C::NTHREADS, CB = 32 * WC;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [128 m][S]
  float* Bs = As + 128 * S;          // [CB c][S]
  float* gcoef = Bs + CB * S;        // p,q,r x 128
  float* xcoef = gcoef + 384;        // a,b x CB
  float* gbase = xcoef + 2 * CB;     // p,q,r before the per-sample dropout scale
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int V = a.D * a.H * a.W;
  const int chunks_per_n = (V + VK - 1) / VK;
  const int nchunks = a.N * chunks_per_n;
  const int k_begin = (int)((long)nchunks * split / a.nsplit), k_end = (int)((long)nchunks * (split + 1) / a.nsplit);

  auto coefficients = [&]() {   // see wgrad3_body: computed after the first chunk's loads were issued (fast path)
    for (int c = tid; c < CB; c += NTHREADS) {
      float ca = 0.f, cb = 0.f, mu, rs;
      if (PRO_X == PRO_BNRELU && c0 + c < a.Cin) bn_fwd_coef(a.bn, c0 + c, ca, cb, mu, rs);
      xcoef[c] = ca; xcoef[CB + c] = cb;
    }
    for (int m = tid; m < 128; m += NTHREADS) {
      float p = 0.f, q = 0.f, r = 0.f;
      if (m0 + m < a.M) bn_bwd_coef(a.gr, m0 + m, p, q, r);
      gbase[m] = p; gbase[128 + m] = q; gbase[256 + m] = r;
    }
  };
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float* al = As + l31 * S + 32 * half;
  const float* bl = Bs + (wave * 32 + l31) * S + 32 * half;
  const bool wave_has_channels = c0 + wave * 32 < a.Cin;
  const bool vec = (V & 3) == 0 && ((((uintptr_t)a.x | (uintptr_t)a.g0 | (uintptr_t)a.g1) & 15) == 0);

  // ---- MFMA over one staged chunk of 64 voxels: 32 k-steps x 4 output-channel tiles, operand reads one step ahead ----
  auto mfma_chunk = [&]() {
    auto rd = [&](int s, float (&av)[4], float& bv) {
      bv = bl[s];
#pragma unroll
      for (int t = 0; t < 4; ++t) av[t] = al[t * 32 * S + s];
    };
    auto mm = [&](const float (&av)[4], float bv) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[t], 0, 0, 0);
    };
    float a0[4], a1[4], b0, b1;
    rd(0, a0, b0);
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
      rd(s + 1, a1, b1);
      __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
      mm(a0, b0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      if (s + 2 < 32) rd(s + 2, a0, b0);
      __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
      mm(a1, b1);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  };

  if (vec) {
    // ===== fast path: unconditional 16-byte loads of chunk k+1 in flight during the MFMA loop of chunk k =====
    constexpr int A_ITEMS = 128 * (VK / 4), B_ITEMS = CB * (VK / 4);
    constexpr int A_IT = (A_ITEMS + NTHREADS - 1) / NTHREADS, B_IT = (B_ITEMS + NTHREADS - 1) / NTHREADS;
    f32x4 ga[A_IT], gb[A_IT], xb[B_IT];
    unsigned oka = 0, okb = 0;
    auto load_chunk = [&](int ch) {
      const int n = ch / chunks_per_n, v0 = (ch % chunks_per_n) * VK;
      const MMNN_GLOBAL float* xn = (const MMNN_GLOBAL float*)a.x + (long)n * a.x_ns + (long)(a.x_coff + c0) * V;      // (see MMNN_GLOBAL)
      const MMNN_GLOBAL float* g0n = (const MMNN_GLOBAL float*)a.g0 + (long)n * a.g0_ns + (long)(a.g0_coff + m0) * V;
      const MMNN_GLOBAL float* g1n = (const MMNN_GLOBAL float*)a.g1 + (long)n * a.g1_ns + (long)(a.g1_coff + m0) * V;
      oka = okb = 0;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int it = tid + i * NTHREADS;
        const int q = it % (VK / 4), m = it / (VK / 4);
        const int v = v0 + 4 * q;
        const bool ok = (it < A_ITEMS) && m0 + m < a.M && v < V;
        const long go = ok ? (long)m * V + v : 0;
        oka |= (ok ? 1u : 0u) << i;
        ga[i] = *reinterpret_cast<const MMNN_GLOBAL f32x4*>(g0n + go);
        gb[i] = *reinterpret_cast<const MMNN_GLOBAL f32x4*>(g1n + go);
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int it = tid + i * NTHREADS;
        const int q = it % (VK / 4), c = it / (VK / 4);
        const int v = v0 + 4 * q;
        const bool ok = (it < B_ITEMS) && c0 + c < a.Cin && v < V;
        okb |= (ok ? 1u : 0u) << i;
        xb[i] = *reinterpret_cast<const MMNN_GLOBAL f32x4*>(xn + (ok ? (long)c * V + v : 0));
      }
    };
    auto store_chunk = [&]() {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int it = tid + i * NTHREADS;
        if (it < A_ITEMS) {
          const int q = it % (VK / 4), m = it / (VK / 4);
          const bool ok = (oka >> i) & 1u;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            As[m * S + 4 * q + e] = ok ? fmaf(gcoef[m], ga[i][e], fmaf(gcoef[128 + m], gb[i][e], gcoef[256 + m])) : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int it = tid + i * NTHREADS;
        if (it < B_ITEMS) {
          const int q = it % (VK / 4), c = it / (VK / 4);
          const bool ok = (okb >> i) & 1u;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            Bs[c * S + 4 * q + e] = ok ? ((PRO_X == PRO_BNRELU) ? fmaxf(fmaf(xcoef[c], xb[i][e], xcoef[CB + c]), 0.f) : xb[i][e]) : 0.f;
        }
      }
    };
    int cur_n = -1;
#if defined(MMNN_PHASE_TRACE)
    const bool tracing = a.trace != nullptr && tid == 0 && split < 16 && c0 == 0 && m0 == 0;
    unsigned long long tr[6] = {0, 0, 0, 0, 0, 0}, tprev = tracing ? __builtin_amdgcn_s_memtime() : 0;
    auto lap = [&](int k) { if (tracing) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr[k] += t - tprev; tprev = t; } };
#else
    auto lap = [&](int) {};
#endif
    if (k_begin < k_end) load_chunk(k_begin);
    coefficients();
    lap(0);
    for (int ch = k_begin; ch < k_end; ++ch) {
      const int n = ch / chunks_per_n;
      if (n != cur_n) {
        for (int m = tid; m < 128; m += NTHREADS) {
          const float sc = drop_scale(a.drop, n, m0 + m);
          gcoef[m] = gbase[m] * sc; gcoef[128 + m] = gbase[128 + m] * sc; gcoef[256 + m] = gbase[256 + m] * sc;
        }
        cur_n = n;
        __syncthreads();
      }
      store_chunk();
      lap(1);
      __syncthreads();
      lap(2);
      if (ch + 1 < k_end) load_chunk(ch + 1);
      lap(3);
      if (wave_has_channels) mfma_chunk();   // waves whose 32 channels lie beyond Cin only help staging (Cin = 64 / 96 with WC = 4)
      lap(4);
      __syncthreads();
      lap(5);
    }
#if defined(MMNN_PHASE_TRACE)
    if (tracing) {
      for (int k = 0; k < 6; ++k) a.trace[split * 16 + k] = tr[k];
      a.trace[split * 16 + 6] = (unsigned long long)(k_end - k_begin);
      a.trace[10] = (1ull << 48) | (9ull << 40) | ((unsigned long long)a.M << 16) | (unsigned long long)a.Cin;
      a.trace[11] = ((unsigned long long)gridDim.x << 32) | ((unsigned long long)gridDim.y << 16) | gridDim.z;
    }
#endif
  } else {
  coefficients();
  int cur_n = -1;
  for (int ch = k_begin; ch < k_end; ++ch) {
    const int n = ch / chunks_per_n;
    const int v0 = (ch % chunks_per_n) * VK;
    if (n != cur_n) {
      __syncthreads();
      for (int m = tid; m < 128; m += NTHREADS) {
        const float s = drop_scale(a.drop, n, m0 + m);
        gcoef[m] = gbase[m] * s; gcoef[128 + m] = gbase[128 + m] * s; gcoef[256 + m] = gbase[256 + m] * s;
      }
      cur_n = n;
      __syncthreads();
    }
    const float* xn = a.x + (long)n * a.x_ns + (long)(a.x_coff + c0) * V;
    const float* g0n = a.g0 + (long)n * a.g0_ns + (long)(a.g0_coff + m0) * V;
    const float* g1n = a.g1 + (long)n * a.g1_ns + (long)(a.g1_coff + m0) * V;
    if (vec) {
#pragma unroll 4
      for (int it = tid; it < 128 * (VK / 4); it += NTHREADS) {
        const int q = it % (VK / 4), m = it / (VK / 4);
        const int v = v0 + 4 * q;
        const bool ok = m0 + m < a.M && v < V;
        const long go = ok ? (long)m * V + v : 0;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(g0n + go);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(g1n + go);
#pragma unroll
        for (int e = 0; e < 4; ++e) As[m * S + 4 * q + e] = ok ? fmaf(gcoef[m], g0[e], fmaf(gcoef[128 + m], g1[e], gcoef[256 + m])) : 0.f;
      }
#pragma unroll 4
      for (int it = tid; it < CB * (VK / 4); it += NTHREADS) {
        const int q = it % (VK / 4), c = it / (VK / 4);
        const int v = v0 + 4 * q;
        const bool ok = c0 + c < a.Cin && v < V;
        const f32x4 x = *reinterpret_cast<const f32x4*>(xn + (ok ? (long)c * V + v : 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[c * S + 4 * q + e] = ok ? ((PRO_X == PRO_BNRELU) ? fmaxf(fmaf(xcoef[c], x[e], xcoef[CB + c]), 0.f) : x[e]) : 0.f;
      }
    } else {
#pragma unroll 2
      for (int it = tid; it < 128 * VK; it += NTHREADS) {
        const int q = it % VK, m = it / VK;
        const int v = v0 + q;
        float o = 0.f;
        if (m0 + m < a.M && v < V) o = fmaf(gcoef[m], g0n[(long)m * V + v], fmaf(gcoef[128 + m], g1n[(long)m * V + v], gcoef[256 + m]));
        As[m * S + q] = o;
      }
#pragma unroll 2
      for (int it = tid; it < CB * VK; it += NTHREADS) {
        const int q = it % VK, c = it / VK;
        const int v = v0 + q;
        float o = 0.f;
        if (c0 + c < a.Cin && v < V) {
          const float x = xn[(long)c * V + v];
          o = (PRO_X == PRO_BNRELU) ? fmaxf(fmaf(xcoef[c], x, xcoef[CB + c]), 0.f) : x;
        }
        Bs[c * S + q] = o;
      }
    }
    __syncthreads();
    if (wave_has_channels) mfma_chunk();
    __syncthreads();
  }
  }
  float* out = a.slab + (long)split * a.slab_stride;
  const int c = c0 + wave * 32 + l31;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + t * 32 + acc_row(r, half);
      if (m < a.M && c < a.Cin) out[(long)m * a.Cin + c] = acc[t][r];
    }
}

template <int PRO_X, int WC>
__global__ void __launch_bounds__(WC * 64) wgrad1_kernel(const WgradArgs a) {
  wgrad1_body<PRO_X, WC>(a, blockIdx.x, blockIdx.y * (32 * WC), blockIdx.z * 128);
}

// several layers in one launch (blockIdx.z = layer; every layer of a batch has M <= 128): see wgrad3_batched_kernel
template <int PRO_X, int WC>
__global__ void __launch_bounds__(WC * 64) wgrad1_batched_kernel(const WgradArgs* __restrict__ table, const uint64_t seed, const int count,
                                                                 const int nsplit, const int members) {
  int z, split, cg;
  if (members > 0) {
    // XCD-aware order (1-D grid, see wgrad3_batched_kernel): all (layer, channel group) blocks of ONE voxel split go to one XCD back to
    // back.  They read the same voxel range: the channel groups of a layer share its 128-row dOut operand, and the layers share the
    // concat buffer's channels -- one HBM read per XCD instead of one per block.
    const int lin = blockIdx.x, x = lin & 7, j = lin >> 3;
    split = x + 8 * (j / members);
    if (split >= nsplit) return;
    int m = j % members;
    z = 0; cg = 0;
    for (int i = 0; i < count; ++i) {                  // member m -> (layer, channel group): a few dozen layers at most
      const int n = (table[i].Cin + 32 * WC - 1) / (32 * WC);
      if (m < n) { z = i; cg = m; break; }
      m -= n;
    }
  } else {
    z = blockIdx.z; split = blockIdx.x; cg = blockIdx.y;
  }
  WgradArgs a = table[z];
  if (split >= a.nsplit || cg * (32 * WC) >= a.Cin) return;
  a.drop.seed = seed;
  wgrad1_body<PRO_X, WC>(a, split, cg * (32 * WC), 0);
}

#endif  // __HIPCC__
}  // namespace mmnn
