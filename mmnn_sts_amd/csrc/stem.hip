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

// r03: software-pipelined over kd.  The workgroups that share a CU start together and take identical time per phase, so they stay in
// step for the whole launch: with one LDS buffer every one of them staged at the same moment, matrix pipe idle, and computed at the
// same moment (measured with the phases switched off one at a time: 327 us of MFMA loop + 98 us of staging = the 429 us of the whole
// kernel -- nothing overlapped, at two, three or four workgroups per CU alike).  Now a workgroup issues the global loads of slice kd+1
// before its MFMA loop of slice kd, writes them to the OTHER buffer after it, and meets ONE barrier per slice.
template <int CIN>
__global__ void __launch_bounds__(256, 2) stem_conv_kernel(const StemConvArgs a) {
  constexpr bool PAIR_C = (CIN % 2 == 0);
  constexpr int KROWS = PAIR_C ? CIN * 49 : CIN * 56;
  constexpr int XSZ = CIN * SC_CS, WSZ = KROWS * 64, BUFSZ = XSZ + WSZ;      // floats per buffer: [Cin][2][13][72] + [krows][64]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* red = smem;                          // [4 waves][2][64] per-wave partial sums: reuses buffer 0 after the last kd slice
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
  const float* xn = a.x + (long)n * CIN * Vi;

  // ---- staging descriptors.  A wave takes whole input rows (c, dz, r): everything about a row -- bounds, base address, LDS row
  // offset -- is wave-uniform and lives in scalar registers, and a lane only contributes its column: a byte offset and an LDS slot
  // worked out once per block.  (The r02 form -- one flat item index per thread, (column, row, plane, channel) by division, 64-bit
  // element offsets -- cost ~50 vector instructions per item, 3.6 per MFMA of the slice they feed, and vector instructions share the
  // SIMD's issue port with the matrix pipe: tools/microbench/mfma_valu_overlap.hip, 4 per MFMA = 79 cycles instead of 64.)
  // A row is 70 columns: lanes take column `lane` and, the first six, column `lane + 64` (the second load is issued by all lanes with
  // a clamped address: uniform control flow keeps all loads of a slice in flight together).
  // 70 columns per row: the odd-Cin path pairs taps (kw, kw+1) across lane halves, so its zero-weight pad tap kw = 7 reads parity-1
  // entry l31 + 3 <= 34; that entry must hold a finite value (0 * garbage-NaN would poison the tile). ----
  typedef const __attribute__((address_space(1))) float* gcf;
  auto ldg = [](gcf base, unsigned byte_off) { return *(gcf)((const __attribute__((address_space(1))) char*)base + byte_off); };
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const int wa = 2 * wo0 + lane - 3, wb_ = wa + 64;
  const bool oka = (unsigned)wa < (unsigned)a.W, okb = lane < 6 && (unsigned)wb_ < (unsigned)a.W;
  const unsigned offa = 4u * (unsigned)(oka ? wa : 2 * wo0), offb = 4u * (unsigned)(okb ? wb_ : 2 * wo0);   // clamped: the tile's first column
  const int dsta = (lane & 1) * SC_PO + (lane >> 1), dstb = lane < 6 ? dsta + 32 : -1;                      // column ci -> parity plane, ci >> 1
  constexpr int NROWS = CIN * SC_TD * SC_ROWS, RPW = (NROWS + 3) / 4;       // input rows of a slice; per wave
  constexpr int W_IT = (KROWS * 16 + 255) / 256;                            // 16-byte weight items per thread
  float va[RPW], vb[RPW];
  f32x4 wr[W_IT];
  // per row of this wave, once per block (scalar registers): byte offset of its first column at kd = 0, relative to the sample's first
  // channel (host check: Cin * D*H*W * 4 < 2^31); LDS row offset; whether h lies inside the volume; its dz.  Per slice a row then costs
  // an add and a few compares -- the first form of this staging re-derived (c, dz, r) and a 64-bit address per row and slice: ~60
  // scalar instructions a row, 860 per wave and slice, on the ONE scalar unit the CU's sixteen waves share.
  int rowoff0[RPW], lrow_[RPW];
  unsigned hokm = 0, dzm = 0;
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int row = min(uwave + 4 * u, NROWS - 1);      // a wave's last row may not exist: re-read the last one, not stored
    const int r = row % SC_ROWS, dz = (row / SC_ROWS) % SC_TD, c = row / (SC_ROWS * SC_TD);
    const int h = 2 * ho0 + r - 3;
    hokm |= ((unsigned)h < (unsigned)a.H ? 1u : 0u) << u;
    dzm |= (unsigned)dz << u;
    rowoff0[u] = 4 * (c * Vi + ((2 * (do0 + dz) - 3) * a.H + h) * a.W);
    lrow_[u] = c * SC_CS + (dz * SC_ROWS + r) * SC_RS;
  }
  const int plane4 = 4 * a.H * a.W;
  const int safe0 = 4 * ((2 * do0 * a.H + 2 * ho0) * a.W);   // a row that is always inside: the tile's own first row, channel 0
  gcf xg = (gcf)xn;
  unsigned rokm = 0;                                                        // bit u: row u of this wave lies inside the volume (wave-uniform)
  auto load_kd = [&](int kd) {
    const unsigned dok0 = (unsigned)(2 * do0 - 3 + kd) < (unsigned)a.D ? 1u : 0u, dok1 = (unsigned)(2 * do0 - 1 + kd) < (unsigned)a.D ? 1u : 0u;
    rokm = 0;
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const unsigned rok = ((hokm >> u) & 1u) & (((dzm >> u) & 1u) ? dok1 : dok0);
      rokm |= rok << u;
      const unsigned ro = (unsigned)(rok ? rowoff0[u] + kd * plane4 : safe0);
      va[u] = ldg(xg, ro + offa);
      vb[u] = ldg(xg, ro + offb);
    }
    // The weight loads are written out by hand: left to the compiler they sink below the MFMA loop, next to their first use, and the
    // slice's memory round trip is paid in full (ISA of the first pipelined build).  The compiler does not count them in vmcnt, so they
    // are issued AFTER the loads it does count (its waits then only ever wait longer than needed) and store_kd starts with vmcnt(0).
    gcf wsrc = (gcf)(a.wp + (long)kd * KROWS * 64);
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const unsigned off = 16u * (unsigned)min(tid + i * 256, KROWS * 16 - 1);
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wr[i]) : "v"(off), "s"(wsrc) : "memory");
    }
  };
  auto store_kd = [&](int buf) {
    float* Xs = smem + buf * BUFSZ;
    float* Ws = Xs + XSZ;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int row = uwave + 4 * u;
      if (row < NROWS) {
        const int lrow = lrow_[u];
        const bool rok = (rokm >> u) & 1u;
        Xs[lrow + dsta] = (rok && oka) ? va[u] : 0.f;
        if (dstb >= 0) Xs[lrow + dstb] = (rok && okb) ? vb[u] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i)
      if (tid + i * 256 < KROWS * 16) reinterpret_cast<f32x4*>(Ws)[tid + i * 256] = wr[i];
  };

  load_kd(0);
  store_kd(0);
  __syncthreads();
  for (int kd = 0; kd < 7; ++kd) {
    if (kd + 1 < 7) load_kd(kd + 1);
    const float* Xs = smem + (kd & 1) * BUFSZ;
    const float* Ws = Xs + XSZ;
    if (PAIR_C) {
      for (int cp = 0; cp < CIN / 2; ++cp) {
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
      for (int c = 0; c < CIN; ++c) {
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
    if (kd + 1 < 7) store_kd((kd + 1) & 1);     // the other buffer: its readers (slice kd - 1) are past the barrier below
    __syncthreads();
  }

  // ---- epilogue: store + batch statistics ----
  // (`red` lives in buffer 0, whose last readers -- slice 6 -- are past the loop's final barrier)
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
  MMNN_REQUIRE((long)a.D * a.H * a.W < (1l << 30) && 4l * a.Cin * a.D * a.H * a.W < (1l << 31), "stem conv: volume too large");
  const int krows = stem_krows(a.Cin);
  const size_t smem = 2 * sizeof(float) * ((size_t)a.Cin * SC_CS + (size_t)krows * 64);   // two buffers (each >= 512 floats: room for the epilogue's partial sums)
  const long blocks = (long)a.N * cdiv(a.Do, SC_TD) * cdiv(a.Ho, SC_TH) * cdiv(a.Wo, SC_TW);
  MMNN_REQUIRE(blocks < (1l << 31) && smem <= 160 * 1024, "stem conv: launch out of range");
  auto kern = a.Cin == 1 ? stem_conv_kernel<1> : a.Cin == 2 ? stem_conv_kernel<2> : a.Cin == 3 ? stem_conv_kernel<3> : stem_conv_kernel<4>;
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

// Two LDS buffers and ONE barrier per tile (r03): a block owns its CU (11 waves, 112 registers), so with a single buffer every wave
// sat through the store / barrier / load-issue phases together with the matrix pipe idle -- 18.4k cycles per tile against 12.3k of
// MFMAs on the busiest SIMD.  Now the waves write tile t+1 into the other buffer between the two halves of their own MFMAs of tile t,
// out of step with one another, and the loads of tile t+2 go out right after the barrier.
__global__ void __launch_bounds__(SW_THREADS) stem_wgrad_kernel(const StemWgradArgs a) {
  __shared__ float Xs[2 * SW_XN];
  __shared__ float Ys[2 * 64 * SW_YS];
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

  // ---- staging.  The (unconditional) global loads of tile t+1 are issued one barrier ahead of their LDS stores.  Everything a
  // thread needs to know about its items is worked out ONCE: with 11 waves a CU the vector ALU is shared with the matrix pipe, and
  // the r02 form (item -> (plane, row, column) by division, 64-bit element offsets, bounds per item; 489 vector instructions per
  // tile and wave = 7.6 per MFMA) held the kernel at 92 cycles per MFMA instead of 64 (tools/microbench/mfma_valu_overlap.hip:
  // 8 vector instructions per MFMA = 92).  Now: per-item LDS offset and 32-bit byte offset from a per-tile uniform base
  // (`global_load_dword v, v_off, s[base]`), validity = two compares against per-tile scalars, tile origin advanced by carries. ----
  typedef const __attribute__((address_space(1))) float* gcf;
  auto ldg = [](gcf base, unsigned byte_off) { return *(gcf)((const __attribute__((address_space(1))) char*)base + byte_off); };
  constexpr int XT = 19 * 37;                      // threads staging the input halo: 19 of its 81 (plane, row) pairs x 37 columns a pass
  constexpr int X_IT = 5, Y_IT = 6;                // 81 / 19 passes;  64 x 64 gradient values / 704 threads
  static_assert(X_IT * 19 >= SW_PLANES * SW_ROWS && Y_IT * SW_THREADS >= 64 * 64 && SW_THREADS >= XT, "staging passes");
  constexpr int X_DUMMY = SW_XN - 1, Y_DUMMY = 64 * SW_YS - 1;   // padding words: where items that do not exist write their zero
  static_assert(X_DUMMY > (SW_PLANES - 1) * SW_PS + (SW_ROWS - 1) * SW_RS + 36, "dummy word inside the used halo");
  const int xci = tid % 37, xrow0 = tid / 37;
  int x_pl[X_IT], x_r[X_IT], x_dst[X_IT];
  unsigned x_rel[X_IT];
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const int row = xrow0 + 19 * i, pl = row / SW_ROWS, r = row % SW_ROWS;
    const bool live = tid < XT && row < SW_PLANES * SW_ROWS;
    x_pl[i] = live ? pl : -1000;                   // fails every bounds test below
    x_r[i] = r;
    x_dst[i] = live ? pl * SW_PS + r * SW_RS + xci : X_DUMMY;
    x_rel[i] = 4u * (unsigned)((pl * a.H + r) * a.W + xci);      // from the halo's corner (2 do0 - 3, 2 ho0 - 3, 2 wo0 - 3)
  }
  const unsigned x_safe = 4u * (unsigned)((3 * a.H + 3) * a.W + 3);   // the tile's own origin voxel: always inside the tensor
  const int yt = tid & 63, ywx = yt % SW_TW, yhy = (yt / SW_TW) % SW_TH, ydz = yt / (SW_TW * SW_TH);
  unsigned y_rel[Y_IT];
  int y_dst[Y_IT];
  unsigned y_mok = 0;
#pragma unroll
  for (int i = 0; i < Y_IT; ++i) {
    const int m = wave + 11 * i;                   // item tid + i * 704 -> (m, t) = (it >> 6, it & 63)
    const bool ok = m < a.M && m < 64;
    y_mok |= (ok ? 1u : 0u) << i;
    y_rel[i] = ok ? 4u * (unsigned)(m * Vo + (ydz * a.Ho + yhy) * a.Wo + ywx) : 0u;
    y_dst[i] = m < 64 ? m * SW_YS + yt : Y_DUMMY;
  }
  float xr[X_IT], y0[Y_IT], y1[Y_IT];
  unsigned okx = 0, oky = 0;
  int ln, ld0, lh0, lw0;                            // origin of the NEXT tile to load
  {
    int b = t_begin;
    lw0 = (b % nw) * SW_TW; b /= nw;
    lh0 = (b % nh) * SW_TH; b /= nh;
    ld0 = (b % nd) * SW_TD; b /= nd;
    ln = b;
  }
  auto load_tile = [&]() {
    const int dbase = 2 * ld0 - 3, hbase = 2 * lh0 - 3, wbase = 2 * lw0 - 3;
    gcf xb = (gcf)(a.x + ((long)ln * a.Cin + c) * Vi + ((long)dbase * a.H + hbase) * a.W + wbase);
    const long yo = (long)ln * a.M * Vo + ((long)ld0 * a.Ho + lh0) * a.Wo + lw0;
    gcf yb0 = (gcf)(a.dz + yo), yb1 = (gcf)(a.y + yo);
    const bool wok = (unsigned)(wbase + xci) < (unsigned)a.W;
    okx = 0;
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const bool ok = wok && (unsigned)(dbase + x_pl[i]) < (unsigned)a.D && (unsigned)(hbase + x_r[i]) < (unsigned)a.H;
      okx |= (ok ? 1u : 0u) << i;
      xr[i] = ldg(xb, ok ? x_rel[i] : x_safe);
    }
    const bool vok = ld0 + ydz < a.Do && lh0 + yhy < a.Ho && lw0 + ywx < a.Wo;
    oky = vok ? y_mok : 0u;
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
      const unsigned o = vok ? y_rel[i] : 0u;
      y0[i] = ldg(yb0, o);
      y1[i] = ldg(yb1, o);
    }
    // next tile, W fastest
    lw0 += SW_TW;
    if (lw0 >= nw * SW_TW) { lw0 = 0; lh0 += SW_TH; if (lh0 >= nh * SW_TH) { lh0 = 0; ld0 += SW_TD; if (ld0 >= nd * SW_TD) { ld0 = 0; ++ln; } } }
  };
  auto store_tile = [&](int buf) {
    float* xs = Xs + buf * SW_XN;
    float* ys = Ys + buf * (64 * SW_YS);
#pragma unroll
    for (int i = 0; i < X_IT; ++i) xs[x_dst[i]] = ((okx >> i) & 1u) ? xr[i] : 0.f;
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
      const int m = min(wave + 11 * i, 63);
      ys[y_dst[i]] = ((oky >> i) & 1u) ? fmaf(gcoef[m], y0[i], fmaf(gcoef[64 + m], y1[i], gcoef[128 + m])) : 0.f;
    }
  };
  // k-steps [S0, S0 + 16) of a tile x 2 output-channel tiles; operand reads one step ahead of the MFMAs
  auto mfma_half = [&](int buf, int S0) {
    const float* xb = xl + buf * SW_XN;
    const float* yb = yl + buf * (64 * SW_YS);
    auto rd = [&](int s, float (&av)[2], float& bv) {
      bv = xb[(2 * (s / SW_TW)) * SW_RS + 2 * (s % SW_TW)];      // s < 32: dz = 0, hy = s/16, wx = s%16
      av[0] = yb[s];
      av[1] = yb[32 * SW_YS + s];
    };
    auto mm = [&](const float (&av)[2], float bv) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv, acc[1], 0, 0, 0);
    };
    float a0[2], a1[2], b0, b1;
    rd(S0, a0, b0);
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
      rd(S0 + s + 1, a1, b1);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      mm(a0, b0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (s + 2 < 16) rd(S0 + s + 2, a0, b0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      mm(a1, b1);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
  };
  __syncthreads();                                   // gcoef visible
  if (t_begin < t_end) {
    load_tile();
    store_tile(0);
    __syncthreads();
    if (t_begin + 1 < t_end) load_tile();
  }
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int cur = (tile - t_begin) & 1;
    mfma_half(cur, 0);
    if (tile + 1 < t_end) store_tile(cur ^ 1);       // loaded one barrier ago; the other buffer's readers are past that barrier too
    mfma_half(cur, 16);
    __syncthreads();
    if (tile + 2 < t_end) load_tile();
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

int stem_wgrad_pick_splits(int N, int Do, int Ho, int Wo, int Cin) {
  const long ntiles = (long)N * cdiv(Do, SW_TD) * cdiv(Ho, SW_TH) * cdiv(Wo, SW_TW);
  // grid = splits x input channels, one 11-wave block per CU: 256 blocks in all (r03: a fixed 128 left half the chip idle for the
  // single-channel encoders -- 472 us for half the work of the 2-channel fusion stem's 497 us)
  long s = Cin >= 1 && Cin <= 8 ? 256 / Cin : 32;
  if (s > ntiles / 2) s = ntiles / 2;
  return s < 1 ? 1 : (int)s;
}

int launch_stem_wgrad(const StemWgradArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.Cin > 0 && a.Cin <= 65535 && a.M > 0 && a.M <= 64, "stem wgrad: bad extent");
  MMNN_REQUIRE(a.nsplit >= 1 && a.slab_stride >= (long)a.Cin * a.M * 352, "stem wgrad: bad slab layout");
  // the kernel addresses a tile's operands with 32-bit byte offsets from a per-tile base
  MMNN_REQUIRE(4l * a.M * a.Do * a.Ho * a.Wo < (1l << 32) && 4l * (9l * a.H + 9) * a.W < (1l << 32), "stem wgrad: extent too large for 32-bit tile offsets");
  MMNN_LAUNCH(stem_wgrad_kernel, dim3(a.nsplit, a.Cin), dim3(SW_THREADS), 0, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmnn
