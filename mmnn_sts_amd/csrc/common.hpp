// Shared device/host helpers for the MI355X (gfx950) kernels of the MMNN_STS fusion path.
//
// Conventions used by every kernel in this directory:
//  * activations are fp32, channel-major per sample:  buf[n][c][v],  v = (d*H + h)*W + w  (NCDHW as the reference's
//    collate produces it, utils/utils.py:112-117).  A dense block's concat (models/densenet.py:87-89) is ONE
//    pre-allocated buffer; every layer writes its growth_rate channels into its slice ("in_coff/out_coff").
//  * batch-norm statistics are accumulated by the PRODUCER of a tensor (epilogue) as fp64 sums in NREP replicas
//    (replica = blockIdx & 7, i.e. blocks that share an XCD share a replica) and turned into per-channel affine
//    coefficients by the CONSUMER's prologue.  fp64 atomics make the result independent of arrival order at fp32
//    resolution.
//  * wave = 64 lanes; MFMA = v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmnn {

constexpr int NREP = 8;   // replicas of each atomically accumulated statistic

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- host-side error channel (C-ABI: int status + thread-local message) -------------------------------------------
void set_error(const char* fmt, ...);
const char* last_error();
#define MMNN_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      ::mmnn::set_error(__VA_ARGS__);           \
      return 1;                                 \
    }                                           \
  } while (0)
#define MMNN_HIP(expr)                                                                        \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::mmnn::set_error("HIP error %d (%s) at %s:%d", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                               \
    }                                                                                         \
  } while (0)

// Every kernel launch goes through MMNN_LAUNCH.  With MMNN_POISON_LDS=1 in the environment (tests only) each launch is
// preceded by a kernel that fills the LDS of every CU with NaN bit patterns, so that a kernel which multiplies a
// zero-padded operand with an LDS entry it never staged shows up as NaN instead of depending on its predecessor.
void debug_poison_lds(hipStream_t stream);
#define MMNN_LAUNCH(kern, grid, block, smem, stream, ...)            \
  do {                                                               \
    ::mmnn::debug_poison_lds(stream);                                \
    hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__); \
  } while (0)

// ---- descriptors ----------------------------------------------------------------------------------------------------
// Statistics of a tensor's channels: sum[r*stride + off + c], sq[...] for replica r.
struct StatPtr {
  double* sum;
  double* sq;
  int stride;   // channels per replica row
  int off;      // first channel of the tensor inside the row
  int nrep;     // replicas the writers of this tensor actually use (power of two <= NREP; the others stay zero): the small-extent layers
                // have a few dozen writer blocks per launch and are latency chains -- their consumers read 1 or 2 replicas instead of 8
  int pad_;     // explicit: argument tables that embed a StatPtr are compared bytewise (densenet.hip: wg_shadow)
};

// Forward batch-norm of channel c:  y = a_c*x + b_c  (then ReLU where the layer has one).
// training: batch statistics (biased variance) from `st`;  eval: running statistics.
struct BnFwd {
  StatPtr st;
  const float* rmean;
  const float* rvar;
  const float* gamma;
  const float* beta;
  double inv_count;   // 1 / (N*V)
  float eps;
  int training;
};

// Backward through the normalisation of a tensor X with batch statistics `st`, given per-voxel upstream G and the
// per-channel sums S1 = sum(G), S2 = sum(G*xhat):   dX = gamma_c * rstd * (G - S1/n - xhat*S2/n) = p*G + q*X + r.
// `gamma` may be null (G already carries the consumers' gammas: dense-block concat channels).
struct BnBwd {
  StatPtr st;        // statistics of X
  StatPtr s;         // s.sum = S1 replicas, s.sq = S2 replicas
  const float* gamma;
  double inv_count;
  float eps;
};

// Channel dropout (nn.Dropout3d semantics, models/densenet.py:84-85): scale of channel c of sample n in layer `layer`.
struct DropCfg {
  uint64_t seed;
  float p;        // 0 => disabled
  int layer;
};

#if defined(__HIPCC__)
__device__ __forceinline__ double stat_total(const double* base, int stride, int idx, int nrep = NREP) {
  double t = 0.0;
  if (nrep == NREP) {
#pragma unroll
    for (int r = 0; r < NREP; ++r) t += base[(long)r * stride + idx];
  } else if (nrep <= 2) {
    // both loads unconditional (replica 0 twice when there is one): a loop over a run-time count waits for each load in turn
    const double t0 = base[idx], t1 = base[(nrep > 1 ? (long)stride : 0l) + idx];
    t = (t + t0) + (nrep > 1 ? t1 : 0.0);
  } else {
    for (int r = 0; r < nrep; ++r) t += base[(long)r * stride + idx];
  }
  return t;
}

// 1 / sqrt(v) in double for v = variance + eps (>= eps > 0, far inside the float range): the hardware's single-precision
// reciprocal square root refined by two Newton steps in fp64 (relative error ~1e-7 -> 1e-14 -> below 1 ulp).  The library
// sequence for 1.0 / sqrt(double) is ~100 quarter-rate instructions, and this sits in the prologue of every small-extent launch.
__device__ __forceinline__ double rsqrt_var(double v) {
  double y = (double)__builtin_amdgcn_rsqf((float)v);
  y = y * (1.5 - 0.5 * v * y * y);
  y = y * (1.5 - 0.5 * v * y * y);
  return y;
}

__device__ __forceinline__ void bn_fwd_coef(const BnFwd& s, int c, float& a, float& b, float& mean_f, float& rstd_f) {
  double mean, var;
  if (s.training) {
    mean = stat_total(s.st.sum, s.st.stride, s.st.off + c, s.st.nrep) * s.inv_count;
    var = stat_total(s.st.sq, s.st.stride, s.st.off + c, s.st.nrep) * s.inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
  } else {
    mean = (double)s.rmean[c];
    var = (double)s.rvar[c];
  }
  double rstd = rsqrt_var(var + (double)s.eps);
  double g = (double)s.gamma[c];
  a = (float)(g * rstd);
  b = (float)((double)s.beta[c] - mean * g * rstd);
  mean_f = (float)mean;
  rstd_f = (float)rstd;
}

__device__ __forceinline__ void bn_bwd_coef(const BnBwd& s, int c, float& p, float& q, float& r) {
  double mean = stat_total(s.st.sum, s.st.stride, s.st.off + c, s.st.nrep) * s.inv_count;
  double var = stat_total(s.st.sq, s.st.stride, s.st.off + c, s.st.nrep) * s.inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  double rstd = rsqrt_var(var + (double)s.eps);
  double m1 = stat_total(s.s.sum, s.s.stride, s.s.off + c, s.s.nrep) * s.inv_count;
  double m2 = stat_total(s.s.sq, s.s.stride, s.s.off + c, s.s.nrep) * s.inv_count;
  double g = s.gamma ? (double)s.gamma[c] : 1.0;
  p = (float)(g * rstd);
  q = (float)(-g * rstd * rstd * m2);
  r = (float)(g * rstd * rstd * m2 * mean - g * rstd * m1);
}

// The same coefficients in two steps for the latency-bound (small-extent) launches: `*_issue` only LOADS -- its results are not
// touched until `*_finish`, so the caller can put the first chunk's operand loads between the two and the statistics' memory
// round trip runs beside theirs instead of after it.  Training-mode statistics with at most two replicas (StatPtr::nrep <= 2: every
// small-extent dense block); `c` must be a valid channel (callers clamp and discard).  Bit-identical to bn_fwd_coef / bn_bwd_coef.
struct BnFwdRaw { double s0, s1, q0, q1; float g, b; };
struct BnBwdRaw { double s0, s1, q0, q1, a0, a1, b0, b1; float g; };

__device__ __forceinline__ void bn_fwd_issue(const BnFwd& s, int c, BnFwdRaw& r) {
  const long i = s.st.off + c, i1 = (s.st.nrep > 1 ? (long)s.st.stride : 0l) + i;
  r.s0 = s.st.sum[i]; r.s1 = s.st.sum[i1];
  r.q0 = s.st.sq[i];  r.q1 = s.st.sq[i1];
  r.g = s.gamma[c]; r.b = s.beta[c];
}
// (`pin`: an empty volatile asm that "rewrites" the loaded values where the arithmetic is meant to start.  Without it the compiler
// hoists the first conversions and additions up to the loads and waits for them there -- ahead of the operand loads they were issued
// early to overlap with.)
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ void bn_fwd_finish(const BnFwd& s, BnFwdRaw& r, float& a, float& b, float& mean_f, float& rstd_f) {
  pin(r.s0); pin(r.s1); pin(r.q0); pin(r.q1); pin(r.g); pin(r.b);
  const bool two = s.st.nrep > 1;
  const double mean = ((0.0 + r.s0) + (two ? r.s1 : 0.0)) * s.inv_count;
  double var = ((0.0 + r.q0) + (two ? r.q1 : 0.0)) * s.inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = rsqrt_var(var + (double)s.eps);
  const double g = (double)r.g;
  a = (float)(g * rstd);
  b = (float)((double)r.b - mean * g * rstd);
  mean_f = (float)mean;
  rstd_f = (float)rstd;
}
__device__ __forceinline__ void bn_bwd_issue(const BnBwd& s, int c, BnBwdRaw& r) {
  const long i = s.st.off + c, i1 = (s.st.nrep > 1 ? (long)s.st.stride : 0l) + i;
  const long j = s.s.off + c, j1 = (s.s.nrep > 1 ? (long)s.s.stride : 0l) + j;
  r.s0 = s.st.sum[i]; r.s1 = s.st.sum[i1];
  r.q0 = s.st.sq[i];  r.q1 = s.st.sq[i1];
  r.a0 = s.s.sum[j];  r.a1 = s.s.sum[j1];
  r.b0 = s.s.sq[j];   r.b1 = s.s.sq[j1];
  r.g = s.gamma ? s.gamma[c] : 1.f;
}
__device__ __forceinline__ void bn_bwd_finish(const BnBwd& s, BnBwdRaw& r, float& p, float& q, float& rr) {
  pin(r.s0); pin(r.s1); pin(r.q0); pin(r.q1); pin(r.a0); pin(r.a1); pin(r.b0); pin(r.b1); pin(r.g);
  const bool two = s.st.nrep > 1, two_s = s.s.nrep > 1;
  const double mean = ((0.0 + r.s0) + (two ? r.s1 : 0.0)) * s.inv_count;
  double var = ((0.0 + r.q0) + (two ? r.q1 : 0.0)) * s.inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = rsqrt_var(var + (double)s.eps);
  const double m1 = ((0.0 + r.a0) + (two_s ? r.a1 : 0.0)) * s.inv_count;
  const double m2 = ((0.0 + r.b0) + (two_s ? r.b1 : 0.0)) * s.inv_count;
  const double g = (double)r.g;
  p = (float)(g * rstd);
  q = (float)(-g * rstd * rstd * m2);
  rr = (float)(g * rstd * rstd * m2 * mean - g * rstd * m1);
}

// counter-based uniform in [0,1): splitmix64 of (seed, layer, n, c)
__device__ __host__ __forceinline__ float drop_scale(const DropCfg& d, int n, int c) {
  if (d.p <= 0.f) return 1.f;
  uint64_t x = d.seed + 0x9E3779B97F4A7C15ull * (uint64_t)(((uint64_t)(uint32_t)d.layer << 40) ^ ((uint64_t)(uint32_t)n << 20) ^ (uint64_t)(uint32_t)c);
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  float u = (float)(x >> 40) * (1.0f / 16777216.0f);
  return u < d.p ? 0.f : 1.f / (1.f - d.p);
}

// ---- cross-lane helpers ---------------------------------------------------------------------------------------------
template <int XOR>
__device__ __forceinline__ float swz_xor(float v) {   // lane ^ XOR within each 32-lane half (ds_swizzle bit mode)
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (XOR << 10) | 0x1F));
}

// Sum each of 16 per-lane values over the 32 lanes of a wave half.  On return, lane L of a half holds the total of
// register index (L >> 1) & 15 (lanes 2t and 2t+1 hold the same total).   16 + 8 + 4 + 2 + 1 = 31 exchanges.
__device__ __forceinline__ float half_reduce16(const float v[16], int lane) {
  float w8[8], w4[4], w2[2], w1;
  const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float send = b4 ? v[r] : v[r + 8];
    float keep = b4 ? v[r + 8] : v[r];
    w8[r] = keep + swz_xor<16>(send);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float send = b3 ? w8[r] : w8[r + 4];
    float keep = b3 ? w8[r + 4] : w8[r];
    w4[r] = keep + swz_xor<8>(send);
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float send = b2 ? w4[r] : w4[r + 2];
    float keep = b2 ? w4[r + 2] : w4[r];
    w2[r] = keep + swz_xor<4>(send);
  }
  {
    float send = b1 ? w2[0] : w2[1];
    float keep = b1 ? w2[1] : w2[0];
    w1 = keep + swz_xor<2>(send);
  }
  return w1 + swz_xor<1>(w1);
}

// row of the 32x32 accumulator tile held in register r by a lane of half h (C/D layout of v_mfma_f32_32x32x2_f32)
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Every 64-byte line of the kernel-argument segment in ONE scalar-cache round trip, first thing in a kernel.  The argument
// structs here are 200-560 bytes; the compiler places each `s_load` next to its first use, every line it touches for the first
// time is a miss (the segment was written by the host a moment ago: cold in every cache) and the misses come one after the other --
// five dependent waits before the first vector load of the convolution kernels, ~2 us of a 12 us small-extent launch
// (tools/phase_trace.py, phase "load0 issue").  One asm statement: the destination registers must not be reused while a load
// is in flight, so the wait sits inside it.  Twelve loads; the ones past the last line of the struct re-read that line.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
  constexpr int L = (BYTES + 63) / 64;
  static_assert(L >= 1 && L <= 12, "argument struct larger than 768 bytes");
#define MMNN_KOFF(i) ((i) < L ? (i) * 64 : (L - 1) * 64)
  const auto k = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11;
  asm volatile(
      "s_load_dword %0, %12, %13\n\ts_load_dword %1, %12, %14\n\ts_load_dword %2, %12, %15\n\ts_load_dword %3, %12, %16\n\t"
      "s_load_dword %4, %12, %17\n\ts_load_dword %5, %12, %18\n\ts_load_dword %6, %12, %19\n\ts_load_dword %7, %12, %20\n\t"
      "s_load_dword %8, %12, %21\n\ts_load_dword %9, %12, %22\n\ts_load_dword %10, %12, %23\n\ts_load_dword %11, %12, %24\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10),
        "=&s"(t11)
      : "s"(k), "n"(MMNN_KOFF(0)), "n"(MMNN_KOFF(1)), "n"(MMNN_KOFF(2)), "n"(MMNN_KOFF(3)), "n"(MMNN_KOFF(4)), "n"(MMNN_KOFF(5)),
        "n"(MMNN_KOFF(6)), "n"(MMNN_KOFF(7)), "n"(MMNN_KOFF(8)), "n"(MMNN_KOFF(9)), "n"(MMNN_KOFF(10)), "n"(MMNN_KOFF(11))
      : "memory");
#undef MMNN_KOFF
}
#endif  // __HIPCC__

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace mmnn
