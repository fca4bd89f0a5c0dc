// Persistent forward of a small-extent dense block (see blockfwd.hpp).  Reference arithmetic: models/densenet.py:46-120.
//
// Workgroup = 8 waves; the grid stays resident for the whole block.  Per layer:
//   phase A  conv1 (1x1x1, cin -> mid) on v_mfma_f32_32x32x2_f32: one 32 x 32 (rows x voxels) tile per workgroup, the channel axis split
//            over the 8 waves, operands streamed from L2 straight into MFMA register layout (no LDS staging: nothing is shared between
//            waves), BN+ReLU of norm1 applied on the fly, partial tiles summed through LDS, T1 + its batch statistics written;
//   ---- grid barrier ----
//   phase B  conv2 (3x3x3, mid -> growth) on v_mfma_f32_16x16x4_f32: one 16 x 16 tile per workgroup (twice as many tiles as 32 x 32 would
//            give), each wave owns mid/8 channels: it stages THEIR halo box of relu(norm2(T1)) into wave-private LDS and runs all 27 taps;
//            partial tiles summed through LDS, channel dropout, the growth new channels + their statistics written to the concat buffer;
//   ---- grid barrier ----
// Everything one workgroup writes and another reads inside the launch leaves with write-through (sc1) stores and is read with sc1 loads
// behind the barrier's drained ticket (MI355X_MICROARCH.md, inter-workgroup visibility: "ONE lane of each storing workgroup ... an
// agent-scope atomic add / an sc1 load poll of that counter").  Every spin is bounded: a barrier that cannot complete sets sync[1] and
// the grid drains.
#include "blockfwd.hpp"

namespace mmnn {

constexpr int BF_THREADS = 512, BF_WAVES = 8;
constexpr int BF_MAXC = 1024;          // channels of the concat buffer
constexpr int BF_MAXMID = 128;
constexpr int BF_SPIN_LIMIT = 300000;  // polls of ~1 us each

typedef float f32x4v __attribute__((ext_vector_type(4)));

// Pointers that come out of the device-resident layer table are "generic" to the compiler: every access through them would be a FLAT
// instruction with a 64-bit per-lane address (two registers per load in flight, and FLAT loads also count on the LDS counter).  All of
// them are global memory: say so (cf. MMNN_GLOBAL in wgrad.hpp), and uniform base + 32-bit lane offset becomes the addressing mode.
#define BF_G __attribute__((address_space(1)))
typedef const BF_G float* gcf;
typedef BF_G float* gf;
typedef const BF_G double* gcd;
typedef BF_G double* gd;
// The layer table is read with vector loads (the kernel also stores to global memory, so the compiler may not use the scalar cache for
// it): its fields arrive as per-lane values.  They are the same in every lane; say so, and they live in scalar registers.
template <class T>
__device__ __forceinline__ T* uni(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
// uniform base + 32-bit BYTE offset per lane = the `global_load_dword v, v_off, s[base]` addressing mode: one register per address in flight
// (an element index would have to be widened to 64 bits before the shift: two registers and two instructions per load)
__device__ __forceinline__ float ldg(gcf base, unsigned byte_off) { return *(gcf)((const BF_G char*)base + byte_off); }
__device__ __forceinline__ float ldg_sc1(gcf base, unsigned byte_off) {
  return __hip_atomic_load((gcf)((const BF_G char*)base + byte_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(gcf p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(gcd p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(gf p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

static inline int bf_box_stride(int W) {
  const int TW = W < 16 ? W : 16, TH = 16 / TW;
  return (3 * (TH + 2) * (TW + 2)) | 1;
}
static inline size_t bf_smem_bytes(int W, int mid) {
  return sizeof(float) * ((size_t)4 * BF_MAXC + 2 * BF_MAXC + (size_t)BF_WAVES * 1024 + (size_t)mid * bf_box_stride(W) + BF_WAVES * 64 + 192);
}

bool block_fwd_supported(int N, int D, int H, int W, int ctot, int mid, int growth) {
  const long V = (long)D * H * W;
  if (W != 4 && W != 8 && W != 16) return false;                 // 16-voxel tiles of whole rows
  const int TW = W < 16 ? W : 16, TH = 16 / TW;
  if (H % TH != 0 || V % 32 != 0) return false;
  if (ctot > BF_MAXC || ctot % 2 != 0 || mid > BF_MAXMID || mid % 32 != 0 || growth % 16 != 0 || growth > 32 || (mid / BF_WAVES) % 4 != 0 || mid / BF_WAVES > 16) return false;
  const long tilesA = (long)(mid / 32) * (N * V / 32), tilesB = (long)(growth / 16) * (N * V / 16);
  if (tilesA > 256 || tilesB > 256) return false;                // one tile per resident workgroup
  return bf_smem_bytes(W, mid) <= 160 * 1024;
}

template <int WDIM>     // W of the block: 4, 8 or 16 -- fixes the 16-voxel tile shape and the halo-box size at compile time
__global__ void __launch_bounds__(BF_THREADS) block_fwd_kernel(const BlockFwdArgs a) {
  int wg = blockIdx.x;
  if (a.xcd_local) {
    if (wg & 7) return;
    wg >>= 3;
  }
  if (wg >= a.nwg) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double* mr = reinterpret_cast<double*>(smem);     // [BF_MAXC][2]: mean, rstd of the concat channels -- layer-invariant, kept for the block
  float* coefa = smem + 4 * BF_MAXC;                // y = relu(a*x + b) of the current phase's input channels
  float* coefb = coefa + BF_MAXC;
  float* red = coefb + BF_MAXC;                     // [8 waves][1024]: partial accumulator tiles
  float* box = red + BF_WAVES * 1024;               // phase B: [mid][PS] halo boxes (wave w: channels [w*cpw, (w+1)*cpw))
  __shared__ int s_fail;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = a.D * a.H * a.W;
  constexpr int TW = WDIM < 16 ? WDIM : 16, TH = 16 / TW;
  constexpr int BH = TH + 2, BW = TW + 2, P = 3 * BH * BW, PS = P | 1;
  float* dump = box + a.mid * PS;                   // [8][64]: landing area of the prefetch LDS-DMA (never read)
  int* boxoff = reinterpret_cast<int*>(dump + BF_WAVES * 64);   // [P <= 162] voxel offset of a halo-box position inside the sample, -1 outside
  const int mtA = a.mid / 32, tilesA = mtA * (a.N * V / 32);
  const int mtB = a.growth / 16, tilesB = mtB * (a.N * V / 16);
  const int rep = wg & (a.nrep - 1);
  const int cpw = a.mid / BF_WAVES;                 // T1 channels per wave in phase B
  unsigned round = 0;
  int known = 0;                                    // concat channels whose (mean, rstd) are in `mr`
  // phase-B tile of this workgroup: the same for every layer, so its halo-box geometry is worked out once
  const int tps = V / 16;                           // 16-voxel tiles per sample
  const int b_mt = wg % mtB, b_vt = wg / mtB;
  const int b_n = b_vt / tps, b_tv = b_vt - b_n * tps;
  const int b_w0 = (b_tv % (a.W / TW)) * TW, b_h0 = ((b_tv / (a.W / TW)) % (a.H / TH)) * TH, b_d0 = b_tv / ((a.W / TW) * (a.H / TH));
  if (wg < tilesB && tid < P) {
    const int bw = tid % BW, bh = (tid / BW) % BH, bd = tid / (BW * BH);
    const int d = b_d0 + bd - 1, h = b_h0 + bh - 1, w = b_w0 + bw - 1;
    const bool ok = (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
    boxoff[tid] = ok ? (d * a.H + h) * a.W + w : -1;
  }
  __syncthreads();

  // ---- grid barrier: every wave drains its stores / atomics, one lane arrives, polls with sc1 loads, bounded ----
  // `pf` / `pf_bytes`: weights of the NEXT phase; waves 1..7 pull them into this XCD's L2 by LDS-DMA (no registers, nothing waits for
  // them) right after the arrival, while wave 0 polls.
  auto barrier = [&](const float* pf, long pf_bytes) -> bool {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      s_fail = 0;
      __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave > 0 && pf != nullptr) {
      const long lines = pf_bytes >> 7;
      const int wgx = a.xcd_local ? wg : (wg >> 3), per = a.xcd_local ? a.nwg : ((a.nwg + 7) >> 3);
      // a wave-instruction fetches one dword from each of 64 consecutive 128-byte lines
      for (long ln0 = ((long)wgx * (BF_WAVES - 1) + (wave - 1)) * 64; ln0 < lines; ln0 += (long)per * (BF_WAVES - 1) * 64) {
        long ln = ln0 + lane;
        if (ln >= lines) ln = lines - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pf + ln * 32),
                                         (__attribute__((address_space(3))) void*)(dump + wave * 64), 4, 0, 0);
      }
    }
    if (tid == 0) {
      const unsigned target = (round + 1) * (unsigned)a.nwg;
      int spins = 0;
      while (__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > BF_SPIN_LIMIT || __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          s_fail = 1;
          __hip_atomic_store(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
    ++round;
    // (read through readfirstlane: a per-lane LDS value would make the layer loop's exit "divergent", the loop counter a vector value,
    // and every pointer of the per-layer table a 64-bit per-lane address)
    return __builtin_amdgcn_readfirstlane(s_fail) == 0;
  };

  for (int l = 0; l < a.nlayers; ++l) {
    const BlkLayer L = a.layers[l];
    const int cin = __builtin_amdgcn_readfirstlane(L.cin);
    const int L_layer_id = __builtin_amdgcn_readfirstlane(L.layer_id);
    const gcf Lw1 = (gcf)uni(L.w1), Lw2 = (gcf)uni(L.w2), Lg1 = (gcf)uni(L.g1), Lb1 = (gcf)uni(L.b1), Lg2 = (gcf)uni(L.g2), Lb2 = (gcf)uni(L.b2);
    const gcf Lrm1 = (gcf)uni(L.rm1), Lrv1 = (gcf)uni(L.rv1), Lrm2 = (gcf)uni(L.rm2), Lrv2 = (gcf)uni(L.rv2);
    const gf Lt1 = (gf)uni(L.t1);
    const gd Ls1 = (gd)uni(L.st_t1_sum), Lq1 = (gd)uni(L.st_t1_sq);
    const gf ax = (gf)a.x;
    const gd asx = (gd)a.st_x_sum, aqx = (gd)a.st_x_sq;
    // =========================== phase A: T1 = conv1(relu(norm1(concat[0:cin)))) ===========================
    // One memory round trip per phase: the inputs of the BN coefficients are requested first, then EVERY operand of the wave's
    // K slice (<= 64 channel pairs: 128 loads per lane in flight), then the coefficients are formed while the operands arrive.
    // (First version: three register sets of 8 pairs refilled as they drained -- eight dependent L2 round trips of ~2 us.)
    if (wg < tilesA) {
      const int mt = wg % mtA, vt = wg / mtA;
      const int g0i = vt * 32, n = g0i / V, v0 = g0i - n * V;         // V % 32 == 0: a tile lies inside one sample
      int lva = lane;                                                 // opaque per layer: see phase B
      asm volatile("" : "+v"(lva));
      const int half = lva >> 5, l31 = lva & 31;
      const int pairs = cin >> 1, ppw = (pairs + BF_WAVES - 1) / BF_WAVES;
      const int p0 = wave * ppw;
      const int np = max(0, min(ppw, pairs - p0));                    // channel pairs of this wave (<= 64: cin <= 1024)
      // ---- (1) coefficient inputs: cin <= 1024 = 2 channels per thread ----
      const int c0 = tid, c1 = tid + BF_THREADS;
      const bool v0c = c0 < cin, v1c = c1 < cin;
      const int k0 = v0c ? c0 : 0, k1 = v1c ? c1 : 0;
      const float g0 = Lg1[k0], e0 = Lb1[k0], g1 = Lg1[k1], e1 = Lb1[k1];
      const bool n0 = a.training && v0c && c0 >= known, n1 = a.training && v1c && c1 >= known;   // statistics final since the last layer
      double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
      float rm0 = 0.f, rv0 = 1.f, rm1_ = 0.f, rv1_ = 1.f;
      if (a.training) {
        if (n0 || n1) {
          for (int r = 0; r < a.nrep; ++r) {
            s0 += ld_sc1(asx + (long)r * a.ctot + k0); q0 += ld_sc1(aqx + (long)r * a.ctot + k0);
            s1 += ld_sc1(asx + (long)r * a.ctot + k1); q1 += ld_sc1(aqx + (long)r * a.ctot + k1);
          }
        }
      } else {
        rm0 = Lrm1[k0]; rv0 = Lrv1[k0]; rm1_ = Lrm1[k1]; rv1_ = Lrv1[k1];
      }
      // ---- (2) all operands of this wave's K slice.  Uniform base pointers + 32-bit per-lane element offsets. ----
      const gcf xb = ax + (long)n * a.x_ns;                       // element (c * V + v0 + l31)
      const gcf wb = Lw1;                                         // element (c * mid + 32 * mt + l31)
      const unsigned xo = (unsigned)(v0 + l31), wo = (unsigned)(32 * mt + l31);
      constexpr int CH = 16, NSET = 2;                                // 2 x 16 pairs in flight, refilled as they drain
      float wa[NSET][CH], xa[NSET][CH];
      const int nch = (np + CH - 1) / CH;
      auto load = [&](int ch, float (&wv)[CH], float (&xv)[CH]) {     // unconditional, clamped (see fprop.hpp on loads under branches)
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          int p = ch * CH + i;
          p = p < np ? p : (np > 0 ? np - 1 : 0);
          const int c = 2 * (p0 + p) + half;
          const int cc = c < cin ? c : 0;
          wv[i] = ldg(wb, 4u * ((unsigned)cc * (unsigned)a.mid + wo));
          xv[i] = ldg_sc1(xb, 4u * ((unsigned)cc * (unsigned)V + xo));
        }
      };
      if (nch > 0) load(0, wa[0], xa[0]);
      if (nch > 1) load(1, wa[1], xa[1]);
      // ---- (3) coefficients (fp64, as bn_fwd_coef); (mean, rstd) of a concat channel never change: cached for the later layers ----
      {
        auto finish = [&](bool valid, bool fresh, int c, double sm, double sq, float rm, float rv, float g, float e) {
          if (!valid) return;
          double mean, rstd;
          if (!a.training) {
            mean = (double)rm; rstd = rsqrt_var((double)rv + (double)a.eps);
          } else if (fresh) {
            mean = sm * a.inv_count;
            double var = sq * a.inv_count - mean * mean;
            if (var < 0.0) var = 0.0;
            rstd = rsqrt_var(var + (double)a.eps);
            mr[2 * c] = mean; mr[2 * c + 1] = rstd;
          } else {
            mean = mr[2 * c]; rstd = mr[2 * c + 1];
          }
          coefa[c] = (float)((double)g * rstd);
          coefb[c] = (float)((double)e - mean * (double)g * rstd);
        };
        finish(v0c, n0, c0, s0, q0, rm0, rv0, g0, e0);
        finish(v1c, n1, c1, s1, q1, rm1_, rv1_, g1, e1);
      }
      known = cin;
      __syncthreads();
      // ---- (4) MFMAs, in channel order ----
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      auto comp = [&](int ch, const float (&wv)[CH], const float (&xv)[CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int p = ch * CH + i;
          const int c = 2 * (p0 + min(p, np > 0 ? np - 1 : 0)) + half;
          const int cc = c < cin ? c : 0;
          const float y = fmaxf(fmaf(coefa[cc], xv[i], coefb[cc]), 0.f);
          const float w = (p < np && c < cin) ? wv[i] : 0.f;           // padding pairs / an odd last channel multiply by zero weights
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w, y, acc, 0, 0, 0);
        }
      };
      if (nch > 0) comp(0, wa[0], xa[0]);
      if (nch > 2) load(2, wa[0], xa[0]);
      if (nch > 1) comp(1, wa[1], xa[1]);
      if (nch > 3) load(3, wa[1], xa[1]);
      if (nch > 2) comp(2, wa[0], xa[0]);
      if (nch > 3) comp(3, wa[1], xa[1]);
      // ---- sum the 8 partial tiles (wave order: reproducible), each wave finishes 2 of the 16 accumulator registers ----
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lva] = acc[r];
      __syncthreads();
      const gf t1o = Lt1 + (long)n * a.mid * V;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < BF_WAVES; ++k) v += red[k * 1024 + r * 64 + lva];
        const int m = 32 * mt + acc_row(r, half);
        st_sc1(t1o + ((unsigned)m * (unsigned)V + xo), v);
        if (a.training) {
          float s0 = v, s1 = v * v;
          s0 += swz_xor<1>(s0); s1 += swz_xor<1>(s1);
          s0 += swz_xor<2>(s0); s1 += swz_xor<2>(s1);
          s0 += swz_xor<4>(s0); s1 += swz_xor<4>(s1);
          s0 += swz_xor<8>(s0); s1 += swz_xor<8>(s1);
          s0 += swz_xor<16>(s0); s1 += swz_xor<16>(s1);
          if (l31 == 0) {
            atomicAdd((double*)(Ls1 + (long)rep * a.mid + m), (double)s0);
            atomicAdd((double*)(Lq1 + (long)rep * a.mid + m), (double)s1);
          }
        }
      }
    }
    if (!barrier((const float*)Lw2, (long)a.mid * 27 * a.growth * 4)) return;

    // =========================== phase B: concat[cin : cin+growth) = dropout(conv2(relu(norm2(T1)))) ===========================
    // Same order: coefficient inputs, then every T1 value of the wave's halo boxes and every weight of its channels, then the rest.
    if (wg < tilesB) {
      const int mt = b_mt, n = b_n, d0 = b_d0, h0 = b_h0, w0 = b_w0;
      // Nothing per-lane in this phase depends on the layer (box addresses, weight offsets, validity masks): left alone, the compiler
      // computes all ~300 of those values once, before the layer loop, and spills them.  An opaque copy of the lane index per layer makes
      // them values of the iteration, formed where they are used.
      int lv = lane;
      asm volatile("" : "+v"(lv));
      // ---- (1) coefficient inputs of norm2: one channel per thread (mid <= 128) ----
      const bool vc = tid < a.mid;
      const int kc = vc ? tid : 0;
      const float gg = Lg2[kc], ee = Lb2[kc];
      double s2 = 0.0, q2 = 0.0;
      float rmm = 0.f, rvv = 1.f;
      if (a.training) {
        if (vc)
          for (int r = 0; r < a.nrep; ++r) { s2 += ld_sc1(Ls1 + (long)r * a.mid + kc); q2 += ld_sc1(Lq1 + (long)r * a.mid + kc); }
      } else {
        rmm = Lrm2[kc]; rvv = Lrv2[kc];
      }
      // ---- (2) this wave's channels: the T1 values of their halo boxes, then all their weights ----
      const gcf t1n = Lt1 + (long)n * a.mid * V;
      float* mybox = box + wave * cpw * PS;
      constexpr int MAXI = (16 * P + 63) / 64;                        // box values per lane: 16 channels * P positions / 64 lanes (27 .. 41)
      const int items = cpw * P;
      float tv[MAXI];
      {
        int cl = 0, pp = lv;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
          const bool valid = lv + 64 * i < items;
          const int off = valid ? boxoff[pp] : -1;
          tv[i] = ldg_sc1(t1n, 4u * ((unsigned)(wave * cpw + (valid ? cl : 0)) * (unsigned)V + (unsigned)(off >= 0 ? off : 0)));
          pp += 64;
          if (pp >= P) { pp -= P; ++cl; }
        }
      }
      const int kq = lv >> 4, ij = lv & 15;
      const int hy = ij / TW, wx = ij - hy * TW;
      const int centre = (1 * BH + hy + 1) * BW + wx + 1;
      const gcf w2b = Lw2;                                        // element ((c * 27 + tap) * growth + 16 * mt + ij)
      const unsigned w2o = (unsigned)(16 * mt + ij);
      const int nq = cpw / 4;                                         // channel quads per wave: 4 for mid = 128
      float wq[2][27];                                                // quad 0 travels with the box values; the others are double-buffered
      auto loadq = [&](int q, float (&wv)[27]) {
        const int c = wave * cpw + 4 * q + kq;
#pragma unroll
        for (int t = 0; t < 27; ++t) wv[t] = ldg(w2b, 4u * (((unsigned)c * 27u + (unsigned)t) * (unsigned)a.growth + w2o));
      };
      if (nq > 0) loadq(0, wq[0]);
      // ---- (3) coefficients -> LDS ----
      if (vc) {
        double mean, rstd;
        if (a.training) {
          mean = s2 * a.inv_count;
          double var = q2 * a.inv_count - mean * mean;
          if (var < 0.0) var = 0.0;
          rstd = rsqrt_var(var + (double)a.eps);
        } else {
          mean = (double)rmm; rstd = rsqrt_var((double)rvv + (double)a.eps);
        }
        coefa[kc] = (float)((double)gg * rstd);
        coefb[kc] = (float)((double)ee - mean * (double)gg * rstd);
      }
      __syncthreads();
      // ---- (4) halo boxes of relu(norm2(T1)), zero outside the volume (padding is applied AFTER the activation) ----
      {
        int cl = 0, pp = lv;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
          if (lv + 64 * i < items) {
            const int c = wave * cpw + cl;
            mybox[cl * PS + pp] = boxoff[pp] >= 0 ? fmaxf(fmaf(coefa[c], tv[i], coefb[c]), 0.f) : 0.f;
          }
          pp += 64;
          if (pp >= P) { pp -= P; ++cl; }
        }
      }
      if (nq > 1) loadq(1, wq[1]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // the box is wave-private: LDS executes a wave's accesses in order
      __builtin_amdgcn_wave_barrier();
      // ---- (5) 27 taps x cpw/4 channel quads of v_mfma_f32_16x16x4_f32: lane = (k = lane >> 4: channel of the quad, i/j = lane & 15) ----
      f32x4v acc4 = {0.f, 0.f, 0.f, 0.f};
      auto compq = [&](int q, const float (&wv)[27]) {
        const float* bx = mybox + (4 * q + kq) * PS + centre;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
          const int off = ((t / 9 - 1) * BH + ((t / 3) % 3 - 1)) * BW + (t % 3 - 1);
          acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], bx[off], acc4, 0, 0, 0);
        }
      };
      if (nq > 0) compq(0, wq[0]);
      if (nq > 2) loadq(2, wq[0]);
      if (nq > 1) compq(1, wq[1]);
      if (nq > 3) loadq(3, wq[1]);
      if (nq > 2) compq(2, wq[0]);
      if (nq > 3) compq(3, wq[1]);
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * 256 + r * 64 + lv] = acc4[r];
      __syncthreads();
      if (wave < 4) {                                                 // wave r finishes accumulator register r: rows 4 * (lane >> 4) + r
        const int r = wave;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < BF_WAVES; ++k) v += red[k * 256 + r * 64 + lv];
        const int row = 16 * mt + 4 * kq + r;                         // output channel within the layer's growth new ones
        DropCfg dc; dc.seed = a.seed; dc.p = a.training ? a.drop_p : 0.f; dc.layer = L_layer_id;
        v *= drop_scale(dc, n, row);
        const int vox = ((d0 * a.H) + h0 + hy) * a.W + w0 + wx;
        st_sc1(ax + (long)n * a.x_ns + (long)(cin + row) * V + vox, v);
        if (a.training) {
          float s0 = v, s1 = v * v;
          s0 += swz_xor<1>(s0); s1 += swz_xor<1>(s1);
          s0 += swz_xor<2>(s0); s1 += swz_xor<2>(s1);
          s0 += swz_xor<4>(s0); s1 += swz_xor<4>(s1);
          s0 += swz_xor<8>(s0); s1 += swz_xor<8>(s1);
          if (ij == 0) {
            atomicAdd((double*)(asx + (long)rep * a.ctot + cin + row), (double)s0);
            atomicAdd((double*)(aqx + (long)rep * a.ctot + cin + row), (double)s1);
          }
        }
      }
    }
    const bool last = l + 1 == a.nlayers;
    const float* nw1 = last ? nullptr : uni(a.layers[l + 1].w1);
    const int ncin = last ? 0 : __builtin_amdgcn_readfirstlane(a.layers[l + 1].cin);
    if (!barrier(nw1, (long)ncin * a.mid * 4)) return;
  }
}

int launch_block_fwd(BlockFwdArgs a, hipStream_t stream) {
  const long V = (long)a.D * a.H * a.W;
  MMNN_REQUIRE(block_fwd_supported(a.N, a.D, a.H, a.W, a.ctot, a.mid, a.growth), "block_fwd: unsupported extent / widths");
  MMNN_REQUIRE(a.x && a.layers && a.sync && a.st_x_sum && a.st_x_sq && a.nlayers >= 1 && a.nrep >= 1 && (a.nrep & (a.nrep - 1)) == 0, "block_fwd: bad arguments");
  const int tilesA = (a.mid / 32) * (int)(a.N * V / 32), tilesB = (a.growth / 16) * (int)(a.N * V / 16);
  a.nwg = tilesA > tilesB ? tilesA : tilesB;
  a.xcd_local = a.nwg <= 32 ? 1 : 0;
  const size_t smem = bf_smem_bytes(a.W, a.mid);
  static size_t configured[32] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  size_t& conf = configured[dev % 32];
  void (*kern)(const BlockFwdArgs) = a.W == 4 ? block_fwd_kernel<4> : (a.W == 8 ? block_fwd_kernel<8> : block_fwd_kernel<16>);
  if (smem > conf) {
    for (auto k : {block_fwd_kernel<4>, block_fwd_kernel<8>, block_fwd_kernel<16>})
      MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bf_smem_bytes(16, a.mid)));
    conf = bf_smem_bytes(16, a.mid);
  }
  const unsigned grid = (unsigned)(a.xcd_local ? 8 * a.nwg : a.nwg);
  MMNN_LAUNCH(kern, dim3(grid), dim3(BF_THREADS), smem, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

}  // namespace mmnn
