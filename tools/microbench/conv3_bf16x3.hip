// PROTOTYPE (developer microbenchmark, not part of the library): the block-1 conv2 forward -- 3x3x3, 128 -> 32 channels, 2 x 32^3 voxels, BN + ReLU
// on operand load, zero padding after it -- on the bf16 matrix pipe with three-piece operands (tools/microbench/bf16x3_gemm.hip), to see what
// the form is worth at kernel level before the production kernels (csrc/fprop.hpp, 137.8 us for this shape with the statistics epilogue) are
// rewritten.  No batch statistics, no dropout, fixed extents.  hipcc -O3 --offload-arch=gfx950
//   MFMA mapping: i = output channel (32), j = voxel (32 consecutive w), k = input channel (16 per v_mfma_f32_32x32x16_bf16; a tap is a row offset).
//   LDS: the halo tile of one 16-channel chunk, voxel-major rows of 16 channels, three bf16 planes (78 KB, two buffers); weights pre-split on the host into
//   [piece][tap][channel / 8][m][8] panels and read straight from L2 in operand layout (all four waves of a workgroup read the same 1 KB).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int N = 2, C = 128, M = 32, D = 32, H = 32, W = 32, V = D * H * W;
constexpr int TD = 2, TH = 4, TW = 32, HD = TD + 2, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW, KC = 16, NCHUNK = C / KC;
static_assert(W == TW && D % TD == 0 && H % TH == 0, "fixed extents");

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}
__device__ __forceinline__ uint32_t pack2(__bf16 lo, __bf16 hi) {
  return (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
}

// MODE 0: the kernel.  Timing-only variants (wrong results) that leave one part out: 1 no staging of chunks after the first, 2 no weight reloads,
// 3 operand reads from LDS hoisted out of the tile loop (one per tap), 4 = 1 + 2 + 3 (the MFMAs and the loop skeleton alone)
template <int MODE>
__global__ void __launch_bounds__(256) conv3_bf16x3(const float* __restrict__ x, const float* __restrict__ bn_a, const float* __restrict__ bn_b,
                                                    const bf16x8* __restrict__ wp, float* __restrict__ out) {
  extern __shared__ uint32_t xs32[];                       // [3][HV][KC / 2] channel pairs
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x, n = b / ((D / TD) * (H / TH)), r = b % ((D / TD) * (H / TH));
  const int d0 = (r / (H / TH)) * TD, h0 = (r % (H / TH)) * TH;
  // in-block K-split: every wave multiplies ALL eight 32-voxel rows of the tile (plane t / 4, row t % 4) with its own taps wv, wv + 4, ... --
  // the four waves then read different weights (no 4x redundant L2 traffic) and a weight operand is reused over eight MFMA groups
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t) for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  // staging: an item = (halo voxel, half of the chunk's 16 channels); lanes run over consecutive voxels (coalesced reads along w), a thread's
  // channel half is fixed (256 is even), its eight values per item become one 16-byte LDS write per piece.  The loads of chunk ch + 1 are
  // issued before the MFMAs of chunk ch and consumed after them.
  constexpr int ITEMS = (2 * HV + 255) / 256;
  uint4* xs128 = reinterpret_cast<uint4*>(xs32);            // [3][2][HV]: piece, channel half, halo voxel -> 8 bf16 (both halves of a wave read contiguous 512 bytes)
  const int g = tid & 1;
  float xr[ITEMS][8], ca[8], cb[8];
  auto item_pos = [&](int it, int& hv, long& o) -> bool {
    const int item = tid + it * 256;
    hv = item >> 1;
    const int hd = hv / (HH * HW), hh = (hv / HW) % HH, hw = hv % HW;
    const int d = d0 + hd - 1, h = h0 + hh - 1, w = hw - 1;
    o = ((long)(n * C) * D + d) * (H * W) + h * W + w;
    return item < 2 * HV && (unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
  };
  auto issue = [&](int ch) {
    const int c0 = ch * KC + 8 * g;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ca[e] = bn_a[c0 + e]; cb[e] = bn_b[c0 + e]; }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      int hv; long o;
      const bool ok = item_pos(it, hv, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) xr[it][e] = ok ? x[o + (long)(c0 + e) * V] : 0.f;
    }
  };
  auto commit_item = [&](int it, int buf) {
    int hv; long o;
    const bool ok = item_pos(it, hv, o);
    if (tid + it * 256 >= 2 * HV) return;
    __bf16 ph[8], pm[8], pl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3(ok ? fmaxf(fmaf(ca[e], xr[it][e], cb[e]), 0.f) : 0.f, ph[e], pm[e], pl[e]);
    uint4* dst = xs128 + buf * (6 * HV);
    dst[(0 * 2 + g) * HV + hv] = make_uint4(pack2(ph[0], ph[1]), pack2(ph[2], ph[3]), pack2(ph[4], ph[5]), pack2(ph[6], ph[7]));
    dst[(1 * 2 + g) * HV + hv] = make_uint4(pack2(pm[0], pm[1]), pack2(pm[2], pm[3]), pack2(pm[4], pm[5]), pack2(pm[6], pm[7]));
    dst[(2 * 2 + g) * HV + hv] = make_uint4(pack2(pl[0], pl[1]), pack2(pl[2], pl[3]), pack2(pl[4], pl[5]), pack2(pl[6], pl[7]));
  };
  constexpr int NT = 7;                                     // taps per wave and chunk (wave 3: six)
  bf16x8 aw[NT][3];
  auto load_w = [&](int ti, int ch) {
    const int tap = wv + 4 * ti;
    if (tap < 27 && ch < NCHUNK) {
#pragma unroll
      for (int p = 0; p < 3; ++p) aw[ti][p] = wp[((p * 27 + tap) * (C / 8) + ch * 2 + lane / 32) * M + lane % 32];
    }
  };
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) load_w(ti, 0);
  issue(0);
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) commit_item(it, 0);
  __syncthreads();
  for (int ch = 0; ch < NCHUNK; ++ch) {
    const int cur = ch & 1;
    const bool more = ch + 1 < NCHUNK;
    if (more && MODE != 1 && MODE != 4) issue(ch + 1);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      const int tap = wv + 4 * ti;
      if (tap < 27) {
        const int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
        bf16x8 a[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = aw[ti][p];
        if (MODE != 2 && MODE != 4) load_w(ti, ch + 1);                                 // the same tap of the next chunk: one whole chunk of MFMAs to arrive
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int hv = ((((MODE == 3 || MODE == 4) ? 0 : t / 4) + td) * HH + (((MODE == 3 || MODE == 4) ? 0 : t % 4) + th)) * HW + (lane % 32 + tw);
          bf16x8 bb[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) bb[p] = *reinterpret_cast<const bf16x8*>(xs128 + cur * (6 * HV) + (p * 2 + lane / 32) * HV + hv);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bb[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[2], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bb[0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bb[0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[0], acc[t], 0, 0, 0);
        }
      }
      if (more && MODE != 1 && MODE != 4) commit_item(ti, cur ^ 1);
    }
    __syncthreads();
  }
  // sum the four waves' partial tiles through LDS (the operand buffers are free now); wave wv finishes rows 2 wv and 2 wv + 1.
  // register q of lane l: output channel 8 * (q / 4) + 4 * (l / 32) + q % 4, voxel column l % 32
  float* red = reinterpret_cast<float*>(xs32);              // [wave][tile][q][lane]
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) red[((wv * 8 + t) * 16 + q) * 64 + lane] = acc[t][q];
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = 2 * wv + tt;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) v += red[((w2 * 8 + t) * 16 + q) * 64 + lane];
      const int m = 8 * (q / 4) + 4 * (lane / 32) + q % 4;
      out[((long)(n * M + m) * D + d0 + t / 4) * (H * W) + (h0 + t % 4) * W + lane % 32] = v;
    }
  }
}

static unsigned short bf16_rne(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  std::vector<float> x((size_t)N * C * V), w((size_t)C * 27 * M), a(C), bsh(C), o((size_t)N * M * V);
  srand(11);
  auto uni = [] { return (float)(rand() / (double)RAND_MAX * 2.0 - 1.0); };
  for (auto& v : x) v = uni();
  for (auto& v : w) v = 0.03f * uni();
  for (int c = 0; c < C; ++c) { a[c] = 0.5f + 0.5f * fabsf(uni()); bsh[c] = 0.3f * uni(); }
  // weights w[(c * 27 + tap) * M + m] (the library's layout) -> three bf16 planes [piece][tap][c / 8][m][c % 8]
  std::vector<unsigned short> wp((size_t)3 * 27 * C * M);
  for (int c = 0; c < C; ++c) for (int tap = 0; tap < 27; ++tap) for (int m = 0; m < M; ++m) {
    float v = w[((size_t)c * 27 + tap) * M + m];
    for (int p = 0; p < 3; ++p) {
      const unsigned short hb = bf16_rne(v);
      wp[((((size_t)p * 27 + tap) * (C / 8) + c / 8) * M + m) * 8 + c % 8] = hb;
      v -= bf16_f(hb);
    }
  }
  float *dx, *dw, *da, *db, *dout; void* dwp;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&da, C * 4)); CK(hipMalloc(&db, C * 4));
  CK(hipMalloc(&dout, o.size() * 4)); CK(hipMalloc(&dwp, wp.size() * 2));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(da, a.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bsh.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0xff, o.size() * 4));
  const size_t smem = (size_t)2 * 3 * HV * KC * 2;   // two buffers
  const int grid = N * (D / TD) * (H / TH);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_bf16x3<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(conv3_bf16x3<0>, dim3(grid), dim3(256), smem, 0, dx, da, db, static_cast<const bf16x8*>(dwp), dout);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
  // fp64 reference on a sample of voxels (corners, edges, interior), all 32 output channels
  double mx = 0, se = 0, sr = 0, scale = 0;
  int checked = 0;
  for (int s = 0; s < 1500; ++s) {
    int n = s & 1, d, h, ww;
    if (s < 64) { d = (s & 2) ? D - 1 : 0; h = (s & 4) ? H - 1 : 0; ww = (s & 8) ? W - 1 : 0; if (s & 16) d = 1 + rand() % (D - 2); if (s & 32) h = 1 + rand() % (H - 2); }
    else { d = rand() % D; h = rand() % H; ww = rand() % W; }
    for (int m = 0; m < M; ++m) {
      double ref = 0;
      for (int c = 0; c < C; ++c) for (int tap = 0; tap < 27; ++tap) {
        const int dd = d + tap / 9 - 1, hh = h + (tap / 3) % 3 - 1, w2 = ww + tap % 3 - 1;
        if (dd < 0 || dd >= D || hh < 0 || hh >= H || w2 < 0 || w2 >= W) continue;
        const float xv = fmaxf(fmaf(a[c], x[((size_t)(n * C + c) * D + dd) * (H * W) + hh * W + w2], bsh[c]), 0.f);
        ref += (double)w[((size_t)c * 27 + tap) * M + m] * (double)xv;
      }
      const double got = o[((size_t)(n * M + m) * D + d) * (H * W) + h * W + ww], e = got - ref;
      mx = fmax(mx, fabs(e)); se += e * e; sr += ref * ref; scale = fmax(scale, fabs(ref)); ++checked;
    }
  }
  printf("conv3 128 -> 32, 2 x 32^3, three bf16 pieces: %d outputs against fp64: max |err| / max |out| %.2e, rms err / rms out %.2e\n", checked, mx / scale, sqrt(se / sr));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(conv3_bf16x3<0>, dim3(grid), dim3(256), smem, 0, dx, da, db, static_cast<const bf16x8*>(dwp), dout);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(conv3_bf16x3<0>, dim3(grid), dim3(256), smem, 0, dx, da, db, static_cast<const bf16x8*>(dwp), dout);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / 20, flop = 2.0 * N * V * M * C * 27;
  printf("%.1f us per launch (%d workgroups of 4 waves, %zu bytes of LDS) = %.1f fp32-equivalent TFLOP/s; the library's fp32-MFMA kernel for this shape: 137.8 us = 105.2 TFLOP/s (with its statistics epilogue)\n",
         us, grid, smem, flop / (us * 1e-6) / 1e12);
  {
    auto timed = [&](const char* name, auto kern) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, 0, dx, da, db, static_cast<const bf16x8*>(dwp), dout);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, 0, dx, da, db, static_cast<const bf16x8*>(dwp), dout);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      printf("  timing-only variant: %-64s %7.1f us\n", name, t * 1e3 / 20);
    };
    timed("no staging after the first chunk", conv3_bf16x3<1>);
    timed("no weight reloads", conv3_bf16x3<2>);
    timed("one set of LDS operand reads per tap instead of eight", conv3_bf16x3<3>);
    timed("all three left out (MFMAs + loop skeleton + reduction)", conv3_bf16x3<4>);
  }
  return 0;
}
