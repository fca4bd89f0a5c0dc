// Dense-layer conv2 forward (models/densenet.py:80-85: ReLU(BN(t1)) -> 3x3x3 convolution to `growth` = 32 new channels -> channel dropout,
// + the batch statistics of the new channels for their consumers) for extents wider than 16 voxels, on the bf16 matrix pipe with fp32 accuracy:
// every fp32 operand is split into three bf16 pieces (x = hi + mid + lo, round to nearest at each step: 24 mantissa bits) and a product
// becomes six v_mfma_f32_32x32x16_bf16 (mid.mid, hi.lo, lo.hi, hi.mid, mid.hi, hi.hi) into one fp32 accumulator.  Measured against fp64
// (tools/microbench/bf16x3_gemm.hip, conv3_bf16x3.hip): 4.5e-7 rms -- the fp32 matrix instruction's own error is 1.0e-6 -- at 16x the
// rate per instruction cycle; and the bf16 instruction leaves the vector ALU free beside it, which v_mfma_f32_32x32x2_f32 does not
// (profiles/r03_microbench_mfma_acc_file.txt).  Same arguments, same results to fp32 rounding and the same statistics protocol as
// fprop_kernel<27, PRO_BNRELU, EPI_STORE_STATS> (fprop.hpp), which stays the kernel for every other shape.
//   MFMA mapping: i = output channel (32), j = voxel (32 consecutive w), k = input channel (16 per instruction); a tap is a row offset.
//   Workgroup: 4 waves, a 2 x 4 x 32 voxel tile.  In-block K-split over the taps: every wave multiplies all eight 32-voxel rows with taps
//   wv, wv + 4, ... (a weight operand serves 48 MFMAs, no two waves load the same weights); the four partial tiles are summed through LDS.
//   LDS: the halo tile of a 16-channel chunk as three bf16 planes [piece][channel half][halo voxel] of 16-byte entries, two buffers: chunk
//   ch + 1 is loaded at the head of chunk ch's tap loop, BN + ReLU + split between its MFMAs, written to the other buffer; one barrier per chunk.
#include <stdlib.h>

#include "fprop.hpp"

namespace mmnn {

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

namespace c3b {
constexpr int KC = 16, NT = 7, MAXC = 256;
// Tile geometry: TD x TH x TW voxels = ROWS rows of 32 voxels (voxel v = 32 row + column: w = v % TW, h = (v / TW) % TH, d = v / (TW TH)).
//   <2, 4, 32>: 256 voxels, extents wider than 16 (one workgroup per CU at 2 x 32^3).
//   <1, 2, 16>: 32 voxels (one row of two h-lines), extents of 9..16: 256 workgroups at 2 x 16^3 (opt-in, measured slower: see below).
template <int TD, int TH, int TW>
struct Geo {
  static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW, ROWS = TD * TH * TW / 32;
  static constexpr int ITEMS = (2 * HV + 255) / 256;
  static constexpr size_t OPER_BYTES = (size_t)2 * 6 * HV * 16;                            // two buffers of [3][2][HV] 16-byte entries
  static constexpr size_t SMEM = OPER_BYTES + sizeof(float) * (2 * MAXC + 2 * 4 * 32 + 32);   // + BN coefficients, statistics scratch, dropout scales
  static_assert(TD * TH * TW % 32 == 0 && (TW == 32 || TW == 16), "rows of 32 voxels");
  static_assert(ITEMS <= NT, "one staging item per tap slot");
  static_assert(4 * ROWS * 16 * 64 * sizeof(float) <= OPER_BYTES, "the partial tiles are summed in the operand buffers");
  static_assert(SMEM <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}
__device__ __forceinline__ uint32_t pack2(__bf16 lo, __bf16 hi) {
  return (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
}
__device__ __forceinline__ bf16x8_t as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8_t, v); }
}  // namespace c3b

template <int TD, int TH, int TW>
__global__ void __launch_bounds__(256) conv3_fwd_bf16x3_kernel(const FpropArgs a) {
  using namespace c3b;
  using G = Geo<TD, TH, TW>;
  constexpr int HH = G::HH, HW = G::HW, HV = G::HV, ROWS = G::ROWS, ITEMS = G::ITEMS;
  constexpr size_t OPER_BYTES = G::OPER_BYTES;
  extern __shared__ uint4 xs128[];                          // [2][3][2][HV]
  float* const coef = reinterpret_cast<float*>(reinterpret_cast<char*>(xs128) + OPER_BYTES);   // [2][MAXC]: a_c, b_c
  float* const sred = coef + 2 * MAXC;                      // [2][4][32]
  float* const dsc = sred + 2 * 4 * 32;                     // [32] dropout scale of the output channels
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = tid & 1;
  const int D = a.D, H = a.H, W = a.W, V = D * H * W, Cin = a.Cin, nchunk = Cin / KC;
  const int twn = (W + TW - 1) / TW, thn = (H + TH - 1) / TH, tdn = (D + TD - 1) / TD;
  int b = blockIdx.x;
  const int w0 = (b % twn) * TW; b /= twn;
  const int h0 = (b % thn) * TH; b /= thn;
  const int d0 = (b % tdn) * TD;
  const int n = b / tdn;
  const float* __restrict__ xin = a.in0 + (long)n * a.in0_ns + (long)a.in0_coff * V;
  const float* __restrict__ wgt = a.w;
  for (int c = tid; c < Cin; c += 256) {
    float ca_, cb_, mu, rs;
    bn_fwd_coef(a.bn_in, c, ca_, cb_, mu, rs);
    coef[c] = ca_; coef[MAXC + c] = cb_;
  }
  if (tid < 32) dsc[tid] = drop_scale(a.drop_out, n, tid);
  f32x16_t acc[ROWS];
#pragma unroll
  for (int t = 0; t < ROWS; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  float xr[ITEMS][8], ca[8], cb[8];
  auto item_pos = [&](int it, int& hv, int& o) -> bool {
    const int item = tid + it * 256;
    hv = item >> 1;
    const int hd = hv / (HH * HW), hh = (hv / HW) % HH, hw = hv % HW;
    const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
    o = (d * H + h) * W + w;
    return item < 2 * HV && (unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
  };
  auto issue = [&](int ch) {
    const int c0 = ch * KC + 8 * g;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ca[e] = coef[c0 + e]; cb[e] = coef[MAXC + c0 + e]; }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      int hv, o;
      const bool ok = item_pos(it, hv, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) xr[it][e] = ok ? xin[(long)(c0 + e) * V + o] : 0.f;
    }
  };
  auto commit_item = [&](int it, int buf) {
    int hv, o;
    const bool ok = item_pos(it, hv, o);
    if (tid + it * 256 >= 2 * HV) return;
    __bf16 ph[8], pm[8], pl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3(ok ? fmaxf(fmaf(ca[e], xr[it][e], cb[e]), 0.f) : 0.f, ph[e], pm[e], pl[e]);   // zero padding AFTER BN + ReLU
    uint4* dst = xs128 + buf * (6 * HV);
    dst[(0 * 2 + g) * HV + hv] = make_uint4(pack2(ph[0], ph[1]), pack2(ph[2], ph[3]), pack2(ph[4], ph[5]), pack2(ph[6], ph[7]));
    dst[(1 * 2 + g) * HV + hv] = make_uint4(pack2(pm[0], pm[1]), pack2(pm[2], pm[3]), pack2(pm[4], pm[5]), pack2(pm[6], pm[7]));
    dst[(2 * 2 + g) * HV + hv] = make_uint4(pack2(pl[0], pl[1]), pack2(pl[2], pl[3]), pack2(pl[4], pl[5]), pack2(pl[6], pl[7]));
  };
  // weights w[(c * 27 + tap) * w_ld + m] (fp32 panel of the library): lane = (row m, channel half); ring slot ti holds tap wv + 4 ti of the
  // chunk ahead as eight raw fp32 values, split into the three operand pieces where they are used
  float wr[NT][8];
  auto load_w = [&](int ti, int ch) {
    const int tap = wv + 4 * ti;
    if (tap < 27 && ch < nchunk) {
      const int c0 = ch * KC + 8 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 8; ++e) wr[ti][e] = wgt[((long)(c0 + e) * 27 + tap) * a.w_ld + (lane & 31)];
    }
  };
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) load_w(ti, 0);
  __syncthreads();                                          // coefficients
  issue(0);
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) commit_item(it, 0);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int cur = ch & 1;
    const bool more = ch + 1 < nchunk;
    if (more) issue(ch + 1);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      const int tap = wv + 4 * ti;
      if (tap < 27) {
        const int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
        __bf16 ph[8], pm[8], pl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3(wr[ti][e], ph[e], pm[e], pl[e]);
        bf16x8_t aw[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) { aw[0][e] = ph[e]; aw[1][e] = pm[e]; aw[2][e] = pl[e]; }
        load_w(ti, ch + 1);
#pragma unroll
        for (int t = 0; t < ROWS; ++t) {
          const int v = t * 32 + (lane & 31);
          const int hv = ((v / (TW * TH) + td) * HH + ((v / TW) % TH + th)) * HW + (v % TW + tw);
          bf16x8_t bb[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) bb[p] = as_bf16x8(xs128[cur * (6 * HV) + (p * 2 + (lane >> 5)) * HV + hv]);
          // smallest products first: their sum is formed before it meets the large one
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[1], bb[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[0], bb[2], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[2], bb[0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[0], bb[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[1], bb[0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[0], bb[0], acc[t], 0, 0, 0);
        }
      }
      if (more && ti < ITEMS) commit_item(ti, cur ^ 1);
    }
    __syncthreads();
  }
  // sum the four waves' partial tiles (fixed order) through LDS; wave wv finishes rows wv, wv + 4, ... of the tile.
  // register q of lane l: output channel 8 * (q / 4) + 4 * (l / 32) + q % 4, voxel column l % 32
  float* red = reinterpret_cast<float*>(xs128);            // [wave][row][q][lane]
#pragma unroll
  for (int t = 0; t < ROWS; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) red[((wv * ROWS + t) * 16 + q) * 64 + lane] = acc[t][q];
  __syncthreads();
  float* __restrict__ outn = a.out + (long)n * a.out_ns + (long)a.out_coff * V;
  const bool want_sums = a.st_out.sum != nullptr;
  float s0[16], s1[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) { s0[q] = 0.f; s1[q] = 0.f; }
#pragma unroll
  for (int tt = 0; tt < (ROWS + 3) / 4; ++tt) {
    const int t = wv + 4 * tt;
    if (t >= ROWS) break;
    const int v = t * 32 + (lane & 31);
    const int d = d0 + v / (TW * TH), h = h0 + (v / TW) % TH, w = w0 + v % TW;
    const bool ok = d < D && h < H && w < W;
    const int o = (d * H + h) * W + w;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float val = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) val += red[((w2 * ROWS + t) * 16 + q) * 64 + lane];
      const int m = 8 * (q / 4) + 4 * (lane >> 5) + q % 4;
      val *= dsc[m];
      if (ok) {
        outn[(long)m * V + o] = val;
        s0[q] += val;
        s1[q] += val * val;
      }
    }
  }
  if (want_sums) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float t0 = s0[q], t1 = s1[q];
      t0 += swz_xor<16>(t0); t1 += swz_xor<16>(t1);
      t0 += swz_xor<8>(t0);  t1 += swz_xor<8>(t1);
      t0 += swz_xor<4>(t0);  t1 += swz_xor<4>(t1);
      t0 += swz_xor<2>(t0);  t1 += swz_xor<2>(t1);
      t0 += swz_xor<1>(t0);  t1 += swz_xor<1>(t1);
      if ((lane & 31) == 0) {
        const int m = 8 * (q / 4) + 4 * (lane >> 5) + q % 4;
        sred[wv * 32 + m] = t0;
        sred[4 * 32 + wv * 32 + m] = t1;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int rep = blockIdx.x & ((a.nrep > 0 ? a.nrep : NREP) - 1);
      double v0 = 0.0, v1 = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) { v0 += (double)sred[w2 * 32 + tid]; v1 += (double)sred[4 * 32 + w2 * 32 + tid]; }
      atomicAdd(a.st_out.sum + (long)rep * a.st_out.stride + a.st_out.off + tid, v0);
      atomicAdd(a.st_out.sq + (long)rep * a.st_out.stride + a.st_out.off + tid, v1);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// conv2 DATA GRADIENT of the wide extents (autograd adjoint of the same layer, main.py:469): 32 -> 128 channels,
//   operand f(c, v) = p_c G[c][v] + q_c X[c][v] + r_c (BN backward of the layer's output channels on operand load, dropout scale folded in,
//   zero padding after it), epilogue: ReLU mask of conv2's input (pre = a_m T1 + b_m > 0), store, d beta / d gamma sums of norm2 --
// the contract of fprop_kernel<27, PRO_GRAD, EPI_MASK_STORE>.  Same tile and LDS planes as the forward; here the reduction is only 32
// channels x 27 taps, so BOTH 16-channel chunks are staged up-front (one buffer each), wave wv owns output rows 32 wv .. 32 wv + 31 for all eight
// voxel rows and all taps (no cross-wave sum), and its weights come through a three-tap ring of raw fp32 values split where they are used.
// ----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv3_dgrad_bf16x3_kernel(const FpropArgs a) {
  using namespace c3b;
  using G = Geo<2, 4, 32>;
  constexpr int TH = 4, TW = 32, HH = G::HH, HW = G::HW, HV = G::HV, ROWS = G::ROWS, ITEMS = G::ITEMS, PF = 3;
  extern __shared__ uint4 xs128[];                          // [2 chunks][3][2][HV]
  float* const coef = reinterpret_cast<float*>(reinterpret_cast<char*>(xs128) + G::OPER_BYTES);   // [3][32]: p, q, r (x dropout scale)
  float* const ecoef = coef + 96;                           // [4][128]: a_m, b_m, mean_m, rstd_m of norm2
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = tid & 1;
  const int D = a.D, H = a.H, W = a.W, V = D * H * W;
  const int twn = (W + TW - 1) / TW, thn = (H + TH - 1) / TH, tdn = (D + 1) / 2;
  int b = blockIdx.x;
  const int w0 = (b % twn) * TW; b /= twn;
  const int h0 = (b % thn) * TH; b /= thn;
  const int d0 = (b % tdn) * 2;
  const int n = b / tdn;
  const float* __restrict__ in0n = a.in0 + (long)n * a.in0_ns + (long)a.in0_coff * V;
  const float* __restrict__ in1n = a.in1 + (long)n * a.in1_ns + (long)a.in1_coff * V;
  const float* __restrict__ wgt = a.w;
  if (tid < 32) {
    float p_, q_, r_;
    bn_bwd_coef(a.gr_in, tid, p_, q_, r_);
    const float sc = drop_scale(a.drop_in, n, tid);
    coef[tid] = p_ * sc; coef[32 + tid] = q_ * sc; coef[64 + tid] = r_ * sc;
  }
  if (tid >= 128) {
    const int m = tid - 128;
    float ea, eb, mu, rs;
    bn_fwd_coef(a.ebn, m, ea, eb, mu, rs);
    ecoef[m] = ea; ecoef[128 + m] = eb; ecoef[256 + m] = mu; ecoef[384 + m] = rs;
  }
  float x0[2][ITEMS][8], x1[2][ITEMS][8];                  // both chunks' loads are in flight together (the accumulators are not live yet)
  auto item_pos = [&](int it, int& hv, int& o) -> bool {
    const int item = tid + it * 256;
    hv = item >> 1;
    const int hd = hv / (HH * HW), hh = (hv / HW) % HH, hw = hv % HW;
    const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
    o = (d * H + h) * W + w;
    return item < 2 * HV && (unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
  };
  auto issue = [&](int ch) {
    const int c0 = ch * KC + 8 * g;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      int hv, o;
      const bool ok = item_pos(it, hv, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        x0[ch][it][e] = ok ? in0n[(long)(c0 + e) * V + o] : 0.f;
        x1[ch][it][e] = ok ? in1n[(long)(c0 + e) * V + o] : 0.f;
      }
    }
  };
  auto commit_item = [&](int it, int buf) {
    int hv, o;
    const bool ok = item_pos(it, hv, o);
    if (tid + it * 256 >= 2 * HV) return;
    const int c0 = buf * KC + 8 * g;
    __bf16 ph[8], pm[8], pl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      split3(ok ? fmaf(coef[c0 + e], x0[buf][it][e], fmaf(coef[32 + c0 + e], x1[buf][it][e], coef[64 + c0 + e])) : 0.f, ph[e], pm[e], pl[e]);
    uint4* dst = xs128 + buf * (6 * HV);
    dst[(0 * 2 + g) * HV + hv] = make_uint4(pack2(ph[0], ph[1]), pack2(ph[2], ph[3]), pack2(ph[4], ph[5]), pack2(ph[6], ph[7]));
    dst[(1 * 2 + g) * HV + hv] = make_uint4(pack2(pm[0], pm[1]), pack2(pm[2], pm[3]), pack2(pm[4], pm[5]), pack2(pm[6], pm[7]));
    dst[(2 * 2 + g) * HV + hv] = make_uint4(pack2(pl[0], pl[1]), pack2(pl[2], pl[3]), pack2(pl[4], pl[5]), pack2(pl[6], pl[7]));
  };
  // weights w[(c * 27 + tap) * w_ld + m]: lane = (row m of this wave's tile, channel half); ring slot tap % PF
  float wr[PF][8];
  auto load_w = [&](int slot, int tap, int ch) {
    if (ch < 2) {
      const int c0 = ch * KC + 8 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 8; ++e) wr[slot][e] = wgt[((long)(c0 + e) * 27 + tap) * a.w_ld + wv * 32 + (lane & 31)];
    }
  };
#pragma unroll
  for (int t = 0; t < PF; ++t) load_w(t, t, 0);
  // both chunks are staged before the first MFMA: 112 staging registers beside 128 accumulators and the weight ring spill (a first build
  // that loaded chunk 1 under chunk 0's MFMAs: 168 spilled registers)
  issue(0);
  issue(1);
  __syncthreads();                                          // coefficients (their statistics loads travel beside the operand loads)
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) commit_item(it, 0);
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) commit_item(it, 1);
  __syncthreads();
  f32x16_t acc[ROWS];
#pragma unroll
  for (int t = 0; t < ROWS; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  for (int ch = 0; ch < 2; ++ch) {
#pragma unroll 1
    for (int tq = 0; tq < 9; ++tq) {                        // (td, th) pairs; the three tw of a pair are unrolled = the ring's three slots
    const int td = tq / 3, th = tq % 3;
#pragma unroll
    for (int tw = 0; tw < 3; ++tw) {
      const int tap = tq * 3 + tw;
      __bf16 ph[8], pm[8], pl[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split3(wr[tw][e], ph[e], pm[e], pl[e]);
      bf16x8_t aw[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) { aw[0][e] = ph[e]; aw[1][e] = pm[e]; aw[2][e] = pl[e]; }
      if (tq < 8) load_w(tw, tap + PF, ch); else load_w(tw, tw, ch + 1);
#pragma unroll
      for (int t = 0; t < ROWS; ++t) {
        const int hv = ((t / TH + td) * HH + (t % TH + th)) * HW + ((lane & 31) + tw);
        bf16x8_t bb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bb[p] = as_bf16x8(xs128[ch * (6 * HV) + (p * 2 + (lane >> 5)) * HV + hv]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[1], bb[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[0], bb[2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[2], bb[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[0], bb[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[1], bb[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[0], bb[0], acc[t], 0, 0, 0);
      }
    }
    }
  }
  // epilogue: register q of lane l = output row 32 wv + 8 (q / 4) + 4 (l / 32) + q % 4, voxel column l % 32 of row t
  const float* __restrict__ exn = a.ex + (long)n * a.ex_ns + (long)a.ex_coff * V;
  float* __restrict__ outn = a.out + (long)n * a.out_ns + (long)a.out_coff * V;
  float s0[16], s1[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) { s0[q] = 0.f; s1[q] = 0.f; }
  auto row_pos = [&](int t, int& o) -> bool {
    const int d = d0 + t / TH, h = h0 + t % TH, w = w0 + (lane & 31);
    o = (d * H + h) * W + w;
    return d < D && h < H && w < W;
  };
  float xn[16];
  {
    int o;
    const bool ok = row_pos(0, o);
#pragma unroll
    for (int q = 0; q < 16; ++q) xn[q] = ok ? exn[(long)(wv * 32 + 8 * (q / 4) + 4 * (lane >> 5) + q % 4) * V + o] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < ROWS; ++t) {
    int o;
    const bool ok = row_pos(t, o);
    float xe[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) xe[q] = xn[q];
    if (t + 1 < ROWS) {                                     // the next row's mask operand is in flight while this row is masked and stored
      int o2;
      const bool ok2 = row_pos(t + 1, o2);
#pragma unroll
      for (int q = 0; q < 16; ++q) xn[q] = ok2 ? exn[(long)(wv * 32 + 8 * (q / 4) + 4 * (lane >> 5) + q % 4) * V + o2] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = wv * 32 + 8 * (q / 4) + 4 * (lane >> 5) + q % 4;
      if (ok) {
        const float pre = fmaf(ecoef[m], xe[q], ecoef[128 + m]);
        const float z = pre > 0.f ? acc[t][q] : 0.f;
        const float xh = (xe[q] - ecoef[256 + m]) * ecoef[384 + m];
        outn[(long)m * V + o] = z;
        s0[q] += z;
        s1[q] += z * xh;
      }
    }
  }
  const int rep = blockIdx.x & ((a.nrep > 0 ? a.nrep : NREP) - 1);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float t0 = s0[q], t1 = s1[q];
    t0 += swz_xor<16>(t0); t1 += swz_xor<16>(t1);
    t0 += swz_xor<8>(t0);  t1 += swz_xor<8>(t1);
    t0 += swz_xor<4>(t0);  t1 += swz_xor<4>(t1);
    t0 += swz_xor<2>(t0);  t1 += swz_xor<2>(t1);
    t0 += swz_xor<1>(t0);  t1 += swz_xor<1>(t1);
    if ((lane & 31) == 0) {
      const int m = wv * 32 + 8 * (q / 4) + 4 * (lane >> 5) + q % 4;
      atomicAdd(a.dbeta + (long)rep * a.M + m, (double)t0);
      atomicAdd(a.dgamma + (long)rep * a.M + m, (double)t1);
    }
  }
}

bool conv3_dgrad_bf16x3_eligible(const FpropArgs& a) {
  static const int mode = [] { const char* e = getenv("MMNN_BF16X3_DGRAD"); return e ? atoi(e) : 1; }();
  static const int fmode = [] { const char* e = getenv("MMNN_BF16X3"); return e ? atoi(e) : 32; }();
  return mode != 0 && fmode != 0 && a.M == 128 && a.Cin == 32 && a.W > 16;
}

int launch_conv3_dgrad_bf16x3(const FpropArgs& a, hipStream_t stream) {
  using G = c3b::Geo<2, 4, 32>;
  constexpr size_t SMEM = G::OPER_BYTES + sizeof(float) * (96 + 512);
  static_assert(SMEM <= 160 * 1024, "LDS");
  MMNN_REQUIRE(conv3_dgrad_bf16x3_eligible(a), "conv3 dgrad bf16x3: shape not handled (M=%d, Cin=%d, W=%d)", a.M, a.Cin, a.W);
  MMNN_REQUIRE(a.in1 && a.ex && a.dgamma && a.dbeta, "conv3 dgrad bf16x3: operands missing");
  MMNN_REQUIRE((long)(a.M + 1) * a.D * a.H * a.W < (1l << 31), "conv3 dgrad bf16x3: volume too large for 32-bit element offsets");
  const long tiles = (long)a.N * cdiv(a.D, 2) * cdiv(a.H, 4) * cdiv(a.W, 32);
  MMNN_REQUIRE(tiles > 0 && tiles < (1l << 31), "conv3 dgrad bf16x3: grid out of range");
  static bool configured[MAX_DEVICES] = {false};
  bool& conf = configured[current_device_slot()];
  if (!conf) {
    MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_dgrad_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
    conf = true;
  }
  MMNN_LAUNCH(conv3_dgrad_bf16x3_kernel, dim3((unsigned)tiles), dim3(256), SMEM, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// Shapes this kernel takes (everything else stays on fprop_kernel): 32 output channels, input channels in chunks of 16, rows wider than
// 16 voxels.  MMNN_BF16X3=0 switches it off (A/B runs, debugging).  MMNN_BF16X3=16 also sends the 9..16-voxel extents here (the <1, 2, 16>
// tile): parity-green but SLOWER than fprop_kernel at 2 x 16^3 (40.6 vs 28.2 us per launch, profiles/r03_ab_experiments.txt) -- 256 workgroups
// of 32 voxels each stage a 6.75x halo and read all 442 KB of weights; those layers need a cross-workgroup K-split instead.  Off by default.
bool conv3_fwd_bf16x3_eligible(const FpropArgs& a) {
  static const int mode = [] { const char* e = getenv("MMNN_BF16X3"); return e ? atoi(e) : 32; }();
  return mode != 0 && a.M == 32 && a.Cin % c3b::KC == 0 && a.Cin >= c3b::KC && a.Cin <= c3b::MAXC && a.W > (mode == 16 ? 8 : 16);
}

template <int TD, int TH, int TW>
static int launch_geo(const FpropArgs& a, hipStream_t stream) {
  using G = c3b::Geo<TD, TH, TW>;
  const long tiles = (long)a.N * cdiv(a.D, TD) * cdiv(a.H, TH) * cdiv(a.W, TW);
  MMNN_REQUIRE(tiles > 0 && tiles < (1l << 31), "conv3 bf16x3: grid out of range");
  static bool configured[MAX_DEVICES] = {false};
  bool& conf = configured[current_device_slot()];
  if (!conf) {
    MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_fwd_bf16x3_kernel<TD, TH, TW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM));
    conf = true;
  }
  MMNN_LAUNCH((conv3_fwd_bf16x3_kernel<TD, TH, TW>), dim3((unsigned)tiles), dim3(256), G::SMEM, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int launch_conv3_fwd_bf16x3(const FpropArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(conv3_fwd_bf16x3_eligible(a), "conv3 bf16x3: shape not handled (M=%d, Cin=%d, W=%d)", a.M, a.Cin, a.W);
  MMNN_REQUIRE(a.drop_in.p <= 0.f, "conv3 bf16x3: no input dropout on this path");
  MMNN_REQUIRE((long)(a.Cin + 1) * a.D * a.H * a.W < (1l << 31), "conv3 bf16x3: volume too large for 32-bit element offsets");
  if (a.W > 16) return launch_geo<2, 4, 32>(a, stream);
  return launch_geo<1, 2, 16>(a, stream);
}

}  // namespace mmnn
