// Stem kernels: conv0 (7x7x7, stride 2) forward on fp32 MFMA, BN+ReLU+max-pool forward/backward, conv0 weight
// gradient on fp32 MFMA.  Reference: models/densenet.py:199-202 and their autograd adjoints (main.py:469).
#include <stdlib.h>

#include "stem.hpp"

namespace mmnn {

// =====================================================================================================================
// conv0 forward.  Implicit GEMM: i = output channel (2 tiles of 32), j = 32 consecutive output voxels along W,
// k = two taps per MFMA.  One block = output tile 2 x 4 x 32; loop over kd, staging per kd the two needed input
// planes (W de-interleaved by parity so that the stride-2 gather of a tap is a contiguous, conflict-free LDS read)
// and the 49 (kh,kw) weight rows.  Lane halves take channel 2k / 2k+1 (even Cin) or taps kw / kw+1 (odd Cin).
// =====================================================================================================================
constexpr int SC_TD = 2, SC_TH = 4, SC_TW = 32;
constexpr int SC_ROWS = 2 * SC_TH + 5;     // 13 input rows
constexpr int SC_PO = 36;                  // parity plane stride (>= 35 entries used)
constexpr int SC_RS = 2 * SC_PO;           // 72
constexpr int SC_CS = SC_TD * SC_ROWS * SC_RS;
constexpr int SC_MAXC = 4;

template <bool PAIR_C>
__global__ void __launch_bounds__(256) stem_conv_kernel(const StemConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int krows = PAIR_C ? a.Cin * 49 : a.Cin * 56;
  float* Xs = smem;                           // [Cin][2][13][72]
  float* Ws = Xs + a.Cin * SC_CS;             // [krows][64]
  float* red = Ws + krows * 64;               // [4 waves][2][64] per-wave partial sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int Vi = a.D * a.H * a.W, Vo = a.Do * a.Ho * a.Wo;
  int b = blockIdx.x;
  const int nw = (a.Wo + SC_TW - 1) / SC_TW, nh = (a.Ho + SC_TH - 1) / SC_TH, nd = (a.Do + SC_TD - 1) / SC_TD;
  const int wo0 = (b % nw) * SC_TW; b /= nw;
  const int ho0 = (b % nh) * SC_TH; b /= nh;
  const int do0 = (b % nd) * SC_TD; b /= nd;
  const int n = b;
  const int rep = blockIdx.x & (NREP - 1);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int pos[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int jt = wave * 2 + j;
    pos[j] = ((jt / SC_TH) * SC_ROWS + 2 * (jt % SC_TH)) * SC_RS + l31;
  }
  const float* xn = a.x + (long)n * a.Cin * Vi;

  for (int kd = 0; kd < 7; ++kd) {
    __syncthreads();
    // ---- stage input planes d_in = 2*(do0+dz) + kd - 3 ----
    // 70 columns per row: the odd-Cin path pairs taps (kw, kw+1) across lane halves, so its zero-weight pad tap kw = 7
    // reads parity-1 entry l31 + 3 <= 34; that entry must hold a finite value (0 * garbage-NaN would poison the tile).
    const int items = a.Cin * SC_TD * SC_ROWS * 70;
#pragma unroll 8
    for (int it = tid; it < items; it += 256) {
      const int ci = it % 70;
      int row = it / 70;
      const int r = row % SC_ROWS; row /= SC_ROWS;
      const int dz = row % SC_TD;
      const int c = row / SC_TD;
      const int d = 2 * (do0 + dz) + kd - 3, h = 2 * ho0 + r - 3, w = 2 * wo0 + ci - 3;
      const bool ok = (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
      const float v = xn[ok ? (long)c * Vi + ((long)d * a.H + h) * a.W + w : 0];   // unconditional load, clamped address
      Xs[c * SC_CS + (dz * SC_ROWS + r) * SC_RS + (ci & 1) * SC_PO + (ci >> 1)] = ok ? v : 0.f;
    }
    // ---- stage this kd's weight rows ----
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wp + (long)kd * krows * 64);
    for (int it = tid; it < krows * 16; it += 256) reinterpret_cast<f32x4*>(Ws)[it] = wsrc[it];
    __syncthreads();

    if (PAIR_C) {
      for (int cp = 0; cp < a.Cin / 2; ++cp) {
        const float* xb = Xs + (2 * cp + half) * SC_CS;
        const float* wb = Ws + (2 * cp + half) * 49 * 64 + l31;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
          for (int kw = 0; kw < 7; ++kw) {
            const int toff = kh * SC_RS + (kw & 1) * SC_PO + (kw >> 1);
            const float a0 = wb[(kh * 7 + kw) * 64], a1 = wb[(kh * 7 + kw) * 64 + 32];
            const float b0 = xb[pos[0] + toff], b1 = xb[pos[1] + toff];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
          }
        }
      }
    } else {
      for (int c = 0; c < a.Cin; ++c) {
        const float* xb = Xs + c * SC_CS + half * SC_PO;
        const float* wb = Ws + (c * 56 + half) * 64 + l31;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
          for (int kp = 0; kp < 4; ++kp) {
            const int toff = kh * SC_RS + kp;
            const float a0 = wb[(kh * 8 + 2 * kp) * 64], a1 = wb[(kh * 8 + 2 * kp) * 64 + 32];
            const float b0 = xb[pos[0] + toff], b1 = xb[pos[1] + toff];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: store + batch statistics ----
  float* outn = a.out + (long)n * a.M * Vo;
  long vox[2];
  bool vok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int jt = wave * 2 + j;
    const int d = do0 + jt / SC_TH, h = ho0 + jt % SC_TH, w = wo0 + l31;
    vok[j] = d < a.Do && h < a.Ho && w < a.Wo;
    vox[j] = ((long)d * a.Ho + h) * a.Wo + w;
  }
  const bool want = a.st_out.sum != nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float s0[16], s1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = i * 32 + acc_row(r, half);
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (m < a.M && vok[j]) {
          const float v = acc[i][j][r];
          outn[(long)m * Vo + vox[j]] = v;
          t0 += v;
          t1 += v * v;
        }
      }
      s0[r] = t0; s1[r] = t1;
    }
    if (want) {
      const float r0 = half_reduce16(s0, lane), r1 = half_reduce16(s1, lane);
      if ((lane & 1) == 0) {
        const int m = i * 32 + acc_row((l31 >> 1) & 15, half);
        red[wave * 128 + m] = r0;
        red[wave * 128 + 64 + m] = r1;
      }
    }
  }
  if (want) {
    __syncthreads();
    if (tid < a.M) {
      double v0 = 0.0, v1 = 0.0;
      for (int j = 0; j < 4; ++j) { v0 += (double)red[j * 128 + tid]; v1 += (double)red[j * 128 + 64 + tid]; }
      atomicAdd(a.st_out.sum + (long)rep * a.st_out.stride + a.st_out.off + tid, v0);
      atomicAdd(a.st_out.sq + (long)rep * a.st_out.stride + a.st_out.off + tid, v1);
    }
  }
}

int launch_stem_conv(const StemConvArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.Cin > 0 && a.Cin <= SC_MAXC, "stem conv: in_channels %d outside [1,%d]", a.Cin, SC_MAXC);
  MMNN_REQUIRE(a.M > 0 && a.M <= 64, "stem conv: init_features %d outside [1,64]", a.M);
  MMNN_REQUIRE(a.Do == (a.D - 1) / 2 + 1 && a.Ho == (a.H - 1) / 2 + 1 && a.Wo == (a.W - 1) / 2 + 1, "stem conv: output extent mismatch");
  MMNN_REQUIRE((long)a.D * a.H * a.W < (1l << 30), "stem conv: volume too large");
  const int krows = stem_krows(a.Cin);
  const size_t smem = sizeof(float) * ((size_t)a.Cin * SC_CS + (size_t)krows * 64 + 512);
  const long blocks = (long)a.N * cdiv(a.Do, SC_TD) * cdiv(a.Ho, SC_TH) * cdiv(a.Wo, SC_TW);
  MMNN_REQUIRE(blocks < (1l << 31) && smem <= 160 * 1024, "stem conv: launch out of range");
  const bool pair_c = (a.Cin % 2 == 0);
  auto kern = pair_c ? stem_conv_kernel<true> : stem_conv_kernel<false>;
  MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  MMNN_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), smem, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// =====================================================================================================================
// BN + ReLU + max-pool(3, stride 2, pad 1) forward; records the winning tap for the backward pass.
// =====================================================================================================================
__global__ void __launch_bounds__(256) stem_pool_kernel(const StemPoolArgs a) {
  __shared__ float red[2][4];
  const int c = blockIdx.y, n = blockIdx.z;
  const int Vi = a.Di * a.Hi * a.Wi, Vo = a.Do * a.Ho * a.Wo;
  float ca, cb, mu, rs;
  bn_fwd_coef(a.bn, c, ca, cb, mu, rs);
  const float* xc = a.x + ((long)n * a.C + c) * Vi;
  const int p = blockIdx.x * 256 + threadIdx.x;
  float best = 0.f, s0 = 0.f, s1 = 0.f;
  if (p < Vo) {
    const int wo = p % a.Wo, ho = (p / a.Wo) % a.Ho, d_o = p / (a.Wo * a.Ho);
    best = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int d = 2 * d_o - 1 + kd;
      if ((unsigned)d >= (unsigned)a.Di) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * ho - 1 + kh;
        if ((unsigned)h >= (unsigned)a.Hi) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int w = 2 * wo - 1 + kw;
          if ((unsigned)w >= (unsigned)a.Wi) continue;
          const float v = fmaxf(fmaf(ca, xc[((long)d * a.Hi + h) * a.Wi + w], cb), 0.f);
          if (v > best) { best = v; bi = kd * 9 + kh * 3 + kw; }
        }
      }
    }
    a.out[(long)n * a.out_ns + (long)c * Vo + p] = best;
    a.idx[((long)n * a.C + c) * Vo + p] = (unsigned char)bi;
    s0 = best; s1 = best * best;
  }
  if (a.st_out.sum) {
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s0; red[1][wave] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int rep = blockIdx.x & (NREP - 1);
      atomicAdd(a.st_out.sum + (long)rep * a.st_out.stride + a.st_out.off + c, (double)red[0][0] + red[0][1] + red[0][2] + red[0][3]);
      atomicAdd(a.st_out.sq + (long)rep * a.st_out.stride + a.st_out.off + c, (double)red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
  }
}

// LDS-tiled version: one block = (n, c, 4 x 4 pooled rows, full width).  The 9 x 9 conv-output rows those windows cover are
// staged once as y = ReLU(a*x + b) (positions outside the tensor as -inf, so they never win), then every pooled voxel scans its
// 27 taps from LDS in the same (kd, kh, kw) order and with the same strict '>' as the direct kernel above: 1.27 global reads per
// input element instead of 3.4.
constexpr int PF_TD = 4, PF_TH = 4, PF_ROWS = 2 * PF_TH + 1, PF_PLANES = 2 * PF_TD + 1;

__global__ void __launch_bounds__(256) stem_pool_tiled_kernel(const StemPoolArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pf_smem[];
  __shared__ float red[2][4];
  const int c = blockIdx.y, n = blockIdx.z;
  const int Vi = a.Di * a.Hi * a.Wi, Vo = a.Do * a.Ho * a.Wo;
  const int nh = (a.Ho + PF_TH - 1) / PF_TH;
  const int pd0 = (blockIdx.x / nh) * PF_TD, ph0 = (blockIdx.x % nh) * PF_TH;
  const int RS = a.Wi + 2;                       // column 0: w = -1, column Wi + 1: w = Wi
  float ca, cb, mu, rs;
  bn_fwd_coef(a.bn, c, ca, cb, mu, rs);
  const float* xc = a.x + ((long)n * a.C + c) * Vi;
  const float NEG = -INFINITY;
  for (int r = threadIdx.x; r < PF_PLANES * PF_ROWS; r += 256) { pf_smem[r * RS] = NEG; pf_smem[r * RS + a.Wi + 1] = NEG; }
  if ((a.Wi & 3) == 0 && (((uintptr_t)xc & 15) == 0)) {
    const int wq = a.Wi / 4, items = PF_PLANES * PF_ROWS * wq;
    for (int it = threadIdx.x; it < items; it += 256) {
      const int q = it % wq, row = it / wq;
      const int d = 2 * pd0 - 1 + row / PF_ROWS, h = 2 * ph0 - 1 + row % PF_ROWS;
      const bool ok = (unsigned)d < (unsigned)a.Di && (unsigned)h < (unsigned)a.Hi;
      const f32x4 x = *reinterpret_cast<const f32x4*>(xc + (ok ? ((long)d * a.Hi + h) * a.Wi + 4 * q : 0));   // unconditional load
      float* dst = pf_smem + row * RS + 1 + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = ok ? fmaxf(fmaf(ca, x[e], cb), 0.f) : NEG;
    }
  } else {
    const int items = PF_PLANES * PF_ROWS * a.Wi;
    for (int it = threadIdx.x; it < items; it += 256) {
      const int w = it % a.Wi, row = it / a.Wi;
      const int d = 2 * pd0 - 1 + row / PF_ROWS, h = 2 * ph0 - 1 + row % PF_ROWS;
      const bool ok = (unsigned)d < (unsigned)a.Di && (unsigned)h < (unsigned)a.Hi;
      const float x = xc[ok ? ((long)d * a.Hi + h) * a.Wi + w : 0];
      pf_smem[row * RS + 1 + w] = ok ? fmaxf(fmaf(ca, x, cb), 0.f) : NEG;
    }
  }
  __syncthreads();
  float s0 = 0.f, s1 = 0.f;
  const int nout = PF_TD * PF_TH * a.Wo;
  for (int i = threadIdx.x; i < nout; i += 256) {
    const int wo = i % a.Wo, hl = (i / a.Wo) % PF_TH, dl = i / (a.Wo * PF_TH);
    const int d_o = pd0 + dl, ho = ph0 + hl;
    if (d_o >= a.Do || ho >= a.Ho) continue;
    const float* base = pf_smem + ((2 * dl) * PF_ROWS + 2 * hl) * RS + 2 * wo;   // tap (0,0,0) = (2d-1, 2h-1, 2w-1) -> column 2w
    float best = NEG;
    int bi = 0;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float v = base[(kd * PF_ROWS + kh) * RS + kw];
          if (v > best) { best = v; bi = kd * 9 + kh * 3 + kw; }
        }
    const long p = ((long)d_o * a.Ho + ho) * a.Wo + wo;
    a.out[(long)n * a.out_ns + (long)c * Vo + p] = best;
    a.idx[((long)n * a.C + c) * Vo + p] = (unsigned char)bi;
    s0 += best; s1 += best * best;
  }
  if (a.st_out.sum) {
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s0; red[1][wave] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int rep = blockIdx.x & (NREP - 1);
      atomicAdd(a.st_out.sum + (long)rep * a.st_out.stride + a.st_out.off + c, (double)red[0][0] + red[0][1] + red[0][2] + red[0][3]);
      atomicAdd(a.st_out.sq + (long)rep * a.st_out.stride + a.st_out.off + c, (double)red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
  }
}

int launch_stem_pool(const StemPoolArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.C > 0 && a.N <= 65535 && a.C <= 65535, "stem pool: bad extent");
  MMNN_REQUIRE(a.Do == (a.Di - 1) / 2 + 1 && a.Ho == (a.Hi - 1) / 2 + 1 && a.Wo == (a.Wi - 1) / 2 + 1, "stem pool: output extent mismatch");
  const int Vo = a.Do * a.Ho * a.Wo;
  const size_t smem = sizeof(float) * PF_PLANES * PF_ROWS * (size_t)(a.Wi + 2);
  static const bool tiled = [] { const char* e = getenv("MMNN_POOL_TILED"); return !(e && e[0] == '0'); }();   // =0: direct kernel (debugging)
  if (tiled && smem <= 64 * 1024) {   // default dynamic-LDS limit; wider rows (W > 200) take the direct kernel
    MMNN_LAUNCH(stem_pool_tiled_kernel, dim3(cdiv(a.Do, PF_TD) * cdiv(a.Ho, PF_TH), a.C, a.N), dim3(256), smem, stream, a);
    MMNN_HIP(hipGetLastError());
    return 0;
  }
  MMNN_LAUNCH(stem_pool_kernel, dim3(cdiv(Vo, 256), a.C, a.N), dim3(256), 0, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// =====================================================================================================================
// Backward of BN+ReLU+max-pool:  dz[q] = relu'(q) * sum over windows p whose argmax is q of dP[p],
// dP = BN-backward(G, pooled) on the fly;  also dgamma0 / dbeta0 sums.
// =====================================================================================================================
// One block = (n, c, 8 x 8 rows of the conv-output grid, full width).  The <= 5 x 5 x Wo pooled windows that can route a
// gradient into those rows are staged in LDS once (dP evaluated on the fly + winning tap), then every fine voxel gathers its
// <= 8 candidates from LDS with branch-free selects: deterministic (no atomics on the tensor), every global load batched.
constexpr int PB_TD = 8, PB_TH = 8, PB_PD = 5, PB_PH = 5;

__global__ void __launch_bounds__(256) stem_pool_bwd_kernel(const StemPoolBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pb_smem[];
  __shared__ float red[2][4];
  const int c = blockIdx.y, n = blockIdx.z;
  const int Vi = a.Di * a.Hi * a.Wi, Vo = a.Do * a.Ho * a.Wo;
  const int nh = (a.Hi + PB_TH - 1) / PB_TH;
  const int d0 = (blockIdx.x / nh) * PB_TD, h0 = (blockIdx.x % nh) * PB_TH;
  const int pd0 = d0 / 2, ph0 = h0 / 2;
  float* dP = pb_smem;                                               // [PB_PD][PB_PH][Wo]
  int* tapw = reinterpret_cast<int*>(pb_smem + PB_PD * PB_PH * a.Wo);   // winning tap of each window (or -1)
  float ca, cb, mu, rs, gp, gq, gr;
  bn_fwd_coef(a.bn, c, ca, cb, mu, rs);
  bn_bwd_coef(a.gr, c, gp, gq, gr);
  const float* xc = a.x + ((long)n * a.C + c) * Vi;
  const float* gc = a.g + (long)n * a.g_ns + (long)c * Vo;
  const float* pc = a.xp + (long)n * a.xp_ns + (long)c * Vo;
  const unsigned char* ic = a.idx + ((long)n * a.C + c) * Vo;
  const int nwin = PB_PD * PB_PH * a.Wo;
  for (int i = threadIdx.x; i < nwin; i += 256) {
    const int pw = i % a.Wo, ph = ph0 + (i / a.Wo) % PB_PH, pd = pd0 + i / (a.Wo * PB_PH);
    const bool ok = pd < a.Do && ph < a.Ho;
    const long p = ok ? ((long)pd * a.Ho + ph) * a.Wo + pw : 0;
    const float v = fmaf(gp, gc[p], fmaf(gq, pc[p], gr));
    const int t = ic[p];
    dP[i] = ok ? v : 0.f;
    tapw[i] = ok ? t : -1;
  }
  __syncthreads();
  float s0 = 0.f, s1 = 0.f;
  // gradient routed into fine voxel (d, h, w): gather its <= 8 candidate windows from LDS
  auto route = [&](int d, int h, int w) -> float {
    // candidate windows along each axis: even coordinate -> (p = q/2, k = 1); odd -> (p = (q+1)/2, k = 0) and (p = (q-1)/2, k = 2)
    const int od = d & 1, oh = h & 1, ow = w & 1;
    const int pdA = (d + 1) / 2 - pd0, kdA = od ? 0 : 1, pdB = (d - 1) / 2 - pd0;   // B valid only for odd coordinates (k = 2)
    const int phA = (h + 1) / 2 - ph0, khA = oh ? 0 : 1, phB = (h - 1) / 2 - ph0;
    const int pwA = (w + 1) / 2, kwA = ow ? 0 : 1, pwB = (w - 1) / 2;
    float z = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool bd = e & 4, bh = e & 2, bw = e & 1;
      const int pd = bd ? pdB : pdA, ph = bh ? phB : phA, pw = bw ? pwB : pwA;
      const int kd = bd ? 2 : kdA, kh = bh ? 2 : khA, kw = bw ? 2 : kwA;
      const bool valid = (!bd || od) && (!bh || oh) && (!bw || ow) && pd >= 0 && pd < PB_PD && ph >= 0 && ph < PB_PH && pw < a.Wo;
      const int li = valid ? (pd * PB_PH + ph) * a.Wo + pw : 0;
      const bool hit = valid && tapw[li] == kd * 9 + kh * 3 + kw;
      z += hit ? dP[li] : 0.f;
    }
    return z;
  };
  float* dzc = a.dz + ((long)n * a.C + c) * Vi;
  if ((a.Wi & 3) == 0 && ((((uintptr_t)xc | (uintptr_t)dzc) & 15) == 0)) {
    // four consecutive voxels of a row per item: one 16-byte load / store each
    const int wq = a.Wi / 4, nq4 = PB_TD * PB_TH * wq;
    for (int i = threadIdx.x; i < nq4; i += 256) {
      const int w0 = (i % wq) * 4, h = h0 + (i / wq) % PB_TH, d = d0 + i / (wq * PB_TH);
      const bool inq = d < a.Di && h < a.Hi;
      const long q = inq ? ((long)d * a.Hi + h) * a.Wi + w0 : 0;
      const f32x4 x = *reinterpret_cast<const f32x4*>(xc + q);
      f32x4 zv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float z = route(d, h, w0 + e);
        z = (inq && fmaf(ca, x[e], cb) > 0.f) ? z : 0.f;
        zv[e] = z;
        s0 += z; s1 += z * (x[e] - mu) * rs;
      }
      if (inq) *reinterpret_cast<f32x4*>(dzc + q) = zv;
    }
  } else {
    const int nq = PB_TD * PB_TH * a.Wi;
    for (int i = threadIdx.x; i < nq; i += 256) {
      const int w = i % a.Wi, h = h0 + (i / a.Wi) % PB_TH, d = d0 + i / (a.Wi * PB_TH);
      const bool inq = d < a.Di && h < a.Hi;
      const long q = inq ? ((long)d * a.Hi + h) * a.Wi + w : 0;
      const float x = xc[q];
      float z = route(d, h, w);
      z = (inq && fmaf(ca, x, cb) > 0.f) ? z : 0.f;
      if (inq) dzc[q] = z;
      s0 += z; s1 += z * (x - mu) * rs;
    }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = s0; red[1][wave] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int rep = blockIdx.x & (NREP - 1);
    atomicAdd(a.dbeta + (long)rep * a.C + c, (double)red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(a.dgamma + (long)rep * a.C + c, (double)red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

int launch_stem_pool_bwd(const StemPoolBwdArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.C > 0 && a.N <= 65535 && a.C <= 65535, "stem pool bwd: bad extent");
  const size_t smem = (size_t)PB_PD * PB_PH * a.Wo * 8;
  MMNN_REQUIRE(smem <= 64 * 1024, "stem pool bwd: row too wide (%d)", a.Wo);
  const int blocks = cdiv(a.Di, PB_TD) * cdiv(a.Hi, PB_TH);
  MMNN_LAUNCH(stem_pool_bwd_kernel, dim3(blocks, a.C, a.N), dim3(256), smem, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// =====================================================================================================================
// conv0 weight gradient.  i = output channel (2 tiles), j = 32 taps of the flattened (kd,kh,kw) index, k = output
// voxel.  11 waves per block (11 x 32 = 352 >= 343 taps), one input channel per block (blockIdx.y), tile of
// 2 x 2 x 16 output voxels; the input halo is staged with row stride == 7 and plane stride == 17 (mod 32) so that
// the 32 taps of a wave (LDS offset == tap index mod 32) fall in 32 different banks.
// =====================================================================================================================
constexpr int SW_TD = 2, SW_TH = 2, SW_TW = 16;
constexpr int SW_RS = 39;                                // >= 2*16+5 = 37, == 7 (mod 32)
constexpr int SW_ROWS = 2 * SW_TH + 5;                   // 9
constexpr int SW_PS = 369;                               // >= 9*39 = 351, == 17 (mod 32)
constexpr int SW_PLANES = 2 * SW_TD + 5;                 // 9
constexpr int SW_XN = SW_PLANES * SW_PS;
constexpr int SW_YS = 65;
constexpr int SW_THREADS = 11 * 64;

__global__ void __launch_bounds__(SW_THREADS) stem_wgrad_kernel(const StemWgradArgs a) {
  __shared__ float Xs[SW_XN];
  __shared__ float Ys[64 * SW_YS];
  __shared__ float gcoef[3 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int Vi = a.D * a.H * a.W, Vo = a.Do * a.Ho * a.Wo;
  const int split = blockIdx.x, c = blockIdx.y;
  const int nw = (a.Wo + SW_TW - 1) / SW_TW, nh = (a.Ho + SW_TH - 1) / SW_TH, nd = (a.Do + SW_TD - 1) / SW_TD;
  const int ntiles = a.N * nw * nh * nd;
  const int t_begin = (int)((long)ntiles * split / a.nsplit), t_end = (int)((long)ntiles * (split + 1) / a.nsplit);
  if (tid < 64) {
    float p = 0.f, q = 0.f, r = 0.f;
    if (tid < a.M) bn_bwd_coef(a.gr, tid, p, q, r);
    gcoef[tid] = p; gcoef[64 + tid] = q; gcoef[128 + tid] = r;
  }
  const int tap = wave * 32 + l31;                       // >= 343: padding lanes (results discarded)
  const int tkd = tap / 49, tkh = (tap / 7) % 7, tkw = tap % 7;
  const int tapoff = (tap < 343) ? tkd * SW_PS + tkh * SW_RS + tkw : 0;
  const float* xl = Xs + tapoff + half * (2 * SW_PS);    // voxel s+32 is one output depth slice further
  const float* yl = Ys + l31 * SW_YS + 32 * half;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- software pipeline: the (unconditional) global loads of tile t+1 are issued before the MFMA loop of tile t ----
  constexpr int X_ITEMS = SW_PLANES * SW_ROWS * 37, Y_ITEMS = 64 * 64;
  constexpr int X_IT = (X_ITEMS + SW_THREADS - 1) / SW_THREADS, Y_IT = (Y_ITEMS + SW_THREADS - 1) / SW_THREADS;
  float xr[X_IT], y0[Y_IT], y1[Y_IT];
  unsigned okx = 0, oky = 0;
  auto origin = [&](int tile, int& n, int& do0, int& ho0, int& wo0) {
    int b = tile;
    wo0 = (b % nw) * SW_TW; b /= nw;
    ho0 = (b % nh) * SW_TH; b /= nh;
    do0 = (b % nd) * SW_TD; b /= nd;
    n = b;
  };
  auto load_tile = [&](int tile) {
    int n, do0, ho0, wo0;
    origin(tile, n, do0, ho0, wo0);
    const float* xc = a.x + ((long)n * a.Cin + c) * Vi;
    okx = oky = 0;
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int it = tid + i * SW_THREADS;
      const int ci = it % 37, r = (it / 37) % SW_ROWS, pl = it / (37 * SW_ROWS);
      const int d = 2 * do0 + pl - 3, h = 2 * ho0 + r - 3, w = 2 * wo0 + ci - 3;
      const bool ok = it < X_ITEMS && (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
      okx |= (ok ? 1u : 0u) << i;
      xr[i] = xc[ok ? ((long)d * a.H + h) * a.W + w : 0];
    }
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
      const int it = tid + i * SW_THREADS;
      const int t = it & 63, m = it >> 6;
      const int wx = t % SW_TW, hy = (t / SW_TW) % SW_TH, dz = t / (SW_TW * SW_TH);
      const int d = do0 + dz, h = ho0 + hy, w = wo0 + wx;
      const bool ok = it < Y_ITEMS && m < a.M && d < a.Do && h < a.Ho && w < a.Wo;
      const long g = ok ? ((long)n * a.M + m) * Vo + ((long)d * a.Ho + h) * a.Wo + w : 0;
      oky |= (ok ? 1u : 0u) << i;
      y0[i] = a.dz[g];
      y1[i] = a.y[g];
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int it = tid + i * SW_THREADS;
      if (it < X_ITEMS) {
        const int ci = it % 37, r = (it / 37) % SW_ROWS, pl = it / (37 * SW_ROWS);
        Xs[pl * SW_PS + r * SW_RS + ci] = ((okx >> i) & 1u) ? xr[i] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
      const int it = tid + i * SW_THREADS;
      if (it < Y_ITEMS) {
        const int t = it & 63, m = it >> 6;
        Ys[m * SW_YS + t] = ((oky >> i) & 1u) ? fmaf(gcoef[m], y0[i], fmaf(gcoef[64 + m], y1[i], gcoef[128 + m])) : 0.f;
      }
    }
  };
  auto mfma_tile = [&]() {   // 32 k-steps x 2 output-channel tiles; operand reads one step ahead of the MFMAs
    auto rd = [&](int s, float (&av)[2], float& bv) {
      bv = xl[(2 * (s / SW_TW)) * SW_RS + 2 * (s % SW_TW)];      // s < 32: dz = 0, hy = s/16, wx = s%16
      av[0] = yl[s];
      av[1] = yl[32 * SW_YS + s];
    };
    auto mm = [&](const float (&av)[2], float bv) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv, acc[1], 0, 0, 0);
    };
    float a0[2], a1[2], b0, b1;
    rd(0, a0, b0);
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
      rd(s + 1, a1, b1);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      mm(a0, b0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (s + 2 < 32) rd(s + 2, a0, b0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      mm(a1, b1);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
  };
  __syncthreads();                                   // gcoef visible
  if (t_begin < t_end) load_tile(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    store_tile();
    __syncthreads();
    if (tile + 1 < t_end) load_tile(tile + 1);
    mfma_tile();
    __syncthreads();
  }
  float* out = a.slab + (long)split * a.slab_stride + (long)c * a.M * 352;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = t * 32 + acc_row(r, half);
      if (m < a.M) out[(long)m * 352 + tap] = acc[t][r];
    }
}

int stem_wgrad_pick_splits(int N, int Do, int Ho, int Wo) {
  const long ntiles = (long)N * cdiv(Do, SW_TD) * cdiv(Ho, SW_TH) * cdiv(Wo, SW_TW);
  long s = 128;
  if (s > ntiles / 2) s = ntiles / 2;
  return s < 1 ? 1 : (int)s;
}

int launch_stem_wgrad(const StemWgradArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.Cin > 0 && a.Cin <= 65535 && a.M > 0 && a.M <= 64, "stem wgrad: bad extent");
  MMNN_REQUIRE(a.nsplit >= 1 && a.slab_stride >= (long)a.Cin * a.M * 352, "stem wgrad: bad slab layout");
  MMNN_LAUNCH(stem_wgrad_kernel, dim3(a.nsplit, a.Cin), dim3(SW_THREADS), 0, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmnn
