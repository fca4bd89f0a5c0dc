// Kernels of the 3-D ResNet-18 variant (reference models/resnet.py:5-227: BasicStem :5-13, BasicBlock :60-93, Conv3DSimple :95-112,
// Resnet18 :114-199, r3d_18 :202-227).  The net is 8 / 16 channels wide behind a 64-channel stem -- far too narrow for 32x32 MFMA
// tiles -- so these are direct (VALU) kernels for any kernel extent / stride / padding: coalesced NCDHW voxel rows across the
// lanes, the filter bank of an output-channel group staged once per block in LDS and read as broadcasts, per-channel batch-norm
// sums by wave shuffles + fp64 atomics, deterministic slab reduction for the weight gradients.
#include <algorithm>

#include "../../include/mmnn_sts.h"
#include "common.hpp"

namespace mmnn {

struct ConvGeom {
  int N, Cin, D, H, W;        // input
  int Cout, Do, Ho, Wo;       // output
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
};

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// element-wise dropout keep-scale of element `idx` (nn.Dropout semantics, models/resnet.py:128,163-169)
__device__ __forceinline__ float elem_drop_scale(uint64_t seed, long idx, float p) {
  if (p <= 0.f) return 1.f;
  const float u = (float)(mix64(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(idx + 1))) >> 40) * (1.0f / 16777216.0f);
  return u < p ? 0.f : 1.f / (1.f - p);
}

// ---- forward: one lane = one output voxel x CO output channels (a group); weights of the group in LDS [Cin*taps][CO] ----------
template <int CO>
__global__ void __launch_bounds__(256) conv3d_fwd_kernel(const ConvGeom g, const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int taps = g.kd * g.kh * g.kw, K = g.Cin * taps;
  const int co0 = blockIdx.y * CO, n = blockIdx.z;
  for (int e = threadIdx.x; e < K * CO; e += 256) {
    const int co = e % CO, k = e / CO;
    wl[e] = (co0 + co < g.Cout) ? w[(long)(co0 + co) * K + k] : 0.f;
  }
  __syncthreads();
  const long Vo = (long)g.Do * g.Ho * g.Wo, Vi = (long)g.D * g.H * g.W;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= Vo) return;
  const int ow = (int)(v % g.Wo), oh = (int)((v / g.Wo) % g.Ho), od = (int)(v / ((long)g.Wo * g.Ho));
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  const float* xn = x + (long)n * g.Cin * Vi;
  for (int ci = 0; ci < g.Cin; ++ci) {
    const float* xc = xn + (long)ci * Vi;
    int k = ci * taps;
    for (int a = 0; a < g.kd; ++a) {
      const int id = od * g.sd - g.pd + a;
      for (int b = 0; b < g.kh; ++b) {
        const int ih = oh * g.sh - g.ph + b;
        const bool rowok = (unsigned)id < (unsigned)g.D && (unsigned)ih < (unsigned)g.H;
        for (int c = 0; c < g.kw; ++c, ++k) {
          const int iw = ow * g.sw - g.pw + c;
          const bool ok = rowok && (unsigned)iw < (unsigned)g.W;
          const float xv = ok ? xc[((long)id * g.H + ih) * g.W + iw] : 0.f;
          const float* wk = wl + (long)k * CO;
#pragma unroll
          for (int o = 0; o < CO; ++o) acc[o] = fmaf(xv, wk[o], acc[o]);
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < CO; ++o)
    if (co0 + o < g.Cout) y[((long)n * g.Cout + co0 + o) * Vo + v] = acc[o];
}

// ---- data gradient: one lane = one INPUT voxel x CI input channels; dx[ci][vi] = sum_{co,tap} dy[co][vo(vi,tap)] w[co][ci][tap] --
template <int CI>
__global__ void __launch_bounds__(256) conv3d_dgrad_kernel(const ConvGeom g, const float* __restrict__ dy, const float* __restrict__ w,
                                                           float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [Cout*taps][CI]
  const int taps = g.kd * g.kh * g.kw;
  const int ci0 = blockIdx.y * CI, n = blockIdx.z;
  for (int e = threadIdx.x; e < g.Cout * taps * CI; e += 256) {
    const int ci = e % CI, t = (e / CI) % taps, co = e / (CI * taps);
    wl[e] = (ci0 + ci < g.Cin) ? w[((long)co * g.Cin + ci0 + ci) * taps + t] : 0.f;
  }
  __syncthreads();
  const long Vo = (long)g.Do * g.Ho * g.Wo, Vi = (long)g.D * g.H * g.W;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= Vi) return;
  const int iw = (int)(v % g.W), ih = (int)((v / g.W) % g.H), id = (int)(v / ((long)g.W * g.H));
  float acc[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) acc[c] = 0.f;
  const float* dyn = dy + (long)n * g.Cout * Vo;
  for (int a = 0; a < g.kd; ++a) {
    const int td = id + g.pd - a;
    if (td < 0 || td % g.sd) continue;
    const int od = td / g.sd;
    if (od >= g.Do) continue;
    for (int b = 0; b < g.kh; ++b) {
      const int th = ih + g.ph - b;
      if (th < 0 || th % g.sh) continue;
      const int oh = th / g.sh;
      if (oh >= g.Ho) continue;
      for (int c = 0; c < g.kw; ++c) {
        const int tw = iw + g.pw - c;
        if (tw < 0 || tw % g.sw) continue;
        const int ow = tw / g.sw;
        if (ow >= g.Wo) continue;
        const long vo = ((long)od * g.Ho + oh) * g.Wo + ow;
        const int t = (a * g.kh + b) * g.kw + c;
        for (int co = 0; co < g.Cout; ++co) {
          const float d = dyn[(long)co * Vo + vo];
          const float* wk = wl + ((long)co * taps + t) * CI;
#pragma unroll
          for (int i = 0; i < CI; ++i) acc[i] = fmaf(d, wk[i], acc[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CI; ++i)
    if (ci0 + i < g.Cin) dx[((long)n * g.Cin + ci0 + i) * Vi + v] = acc[i];
}

// ---- weight gradient: block (split, ci); thread = (co, tap) pairs; partial sums over the split's output voxels -> slab --------
__global__ void __launch_bounds__(256) conv3d_wgrad_kernel(const ConvGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ slab, int nsplit) {
  const int taps = g.kd * g.kh * g.kw, per = g.Cout * taps;
  const int split = blockIdx.x, ci = blockIdx.y;
  const long Vo = (long)g.Do * g.Ho * g.Wo, Vi = (long)g.D * g.H * g.W;
  const long tot = (long)g.N * Vo;
  const long v_begin = tot * split / nsplit, v_end = tot * (split + 1) / nsplit;
  for (int e = threadIdx.x; e < per; e += 256) {
    const int t = e % taps, co = e / taps;
    const int c = t % g.kw, b = (t / g.kw) % g.kh, a = t / (g.kw * g.kh);
    float acc = 0.f;
    for (long q = v_begin; q < v_end; ++q) {
      const int n = (int)(q / Vo);
      const long v = q - (long)n * Vo;
      const int ow = (int)(v % g.Wo), oh = (int)((v / g.Wo) % g.Ho), od = (int)(v / ((long)g.Wo * g.Ho));
      const int id = od * g.sd - g.pd + a, ih = oh * g.sh - g.ph + b, iw = ow * g.sw - g.pw + c;
      const bool ok = (unsigned)id < (unsigned)g.D && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      const float xv = ok ? x[((long)n * g.Cin + ci) * Vi + ((long)id * g.H + ih) * g.W + iw] : 0.f;
      acc = fmaf(dy[((long)n * g.Cout + co) * Vo + v], xv, acc);
    }
    slab[(long)split * g.Cout * g.Cin * taps + ((long)co * g.Cin + ci) * taps + t] = acc;
  }
}
__global__ void __launch_bounds__(256) slab_sum_kernel(const float* __restrict__ slab, long count, int nsplit, float* __restrict__ out,
                                                       int accumulate) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slab[(long)k * count + e];     // fixed order: bit-reproducible
    out[e] = accumulate ? out[e] + s : s;
  }
}

// ---- batch norm (training: batch statistics; eval: running statistics) + optional residual add + ReLU + dropout ----------------
// sums[0][c] = sum x, sums[1][c] = sum x^2 (fp64, zero on entry)
__global__ void __launch_bounds__(256) bn_stats_kernel(int N, int C, long V, const float* __restrict__ x, double* __restrict__ sums) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* p = x + ((long)n * C + c) * V;
  float s = 0.f, q = 0.f;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const float t = p[v];
    s += t; q = fmaf(t, t, q);
  }
  const double sd = wave_sum_d((double)s), qd = wave_sum_d((double)q);
  if ((threadIdx.x & 63) == 0) { atomicAdd(sums + c, sd); atomicAdd(sums + C + c, qd); }
}
// save[0][c] = mean, save[1][c] = rstd; running statistics updated with the UNBIASED variance (torch.nn.BatchNorm3d)
__global__ void bn_finalize_kernel(int C, double count, const double* __restrict__ sums, float eps, float momentum, int training,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ save) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double mean, var;
  if (training) {
    mean = sums[c] / count;
    var = sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
    rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
  } else {
    mean = rmean[c]; var = rvar[c];
  }
  save[c] = (float)mean;
  save[C + c] = (float)(1.0 / sqrt(var + (double)eps));
}
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(int N, int C, long V, const float* __restrict__ x, const float* __restrict__ save,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ res, int relu, float drop_p, uint64_t seed,
                                                         float* __restrict__ out) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float a = gamma[c] * save[C + c], b = beta[c] - save[c] * a;
  const long base = ((long)n * C + c) * V;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    float t = fmaf(a, x[base + v], b);
    if (res) t += res[base + v];
    if (relu) t = fmaxf(t, 0.f);
    out[base + v] = t * elem_drop_scale(seed, base + v, drop_p);
  }
}
// backward, pass 1: dz = dout * keep_scale * [relu: out > 0];  sums[0][c] = sum dz, sums[1][c] = sum dz * xhat   (fp64, zero on entry)
__device__ __forceinline__ float bn_dz(const float* dout, const float* out, long i, int relu, float drop_p, uint64_t seed) {
  float d = dout[i] * elem_drop_scale(seed, i, drop_p);
  if (relu && !(out[i] > 0.f)) d = 0.f;     // `out` is the stored (post-ReLU, post-dropout) result: 0 where the ReLU clipped
  return d;
}
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(int N, int C, long V, const float* __restrict__ x, const float* __restrict__ out,
                                                            const float* __restrict__ dout, const float* __restrict__ save, int relu,
                                                            float drop_p, uint64_t seed, double* __restrict__ sums) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float mean = save[c], rstd = save[C + c];
  const long base = ((long)n * C + c) * V;
  float s = 0.f, q = 0.f;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const float d = bn_dz(dout, out, base + v, relu, drop_p, seed);
    s += d; q = fmaf(d, (x[base + v] - mean) * rstd, q);
  }
  const double sd = wave_sum_d((double)s), qd = wave_sum_d((double)q);
  if ((threadIdx.x & 63) == 0) { atomicAdd(sums + c, sd); atomicAdd(sums + C + c, qd); }
}
// pass 2: dx = gamma * rstd * (dz - S1/n - xhat * S2/n) (training) or gamma * rstd * dz (eval);  dres = dz;  dgamma = S2, dbeta = S1
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(int N, int C, long V, const float* __restrict__ x, const float* __restrict__ out,
                                                           const float* __restrict__ dout, const float* __restrict__ save,
                                                           const float* __restrict__ gamma, const double* __restrict__ sums, double inv_count,
                                                           int training, int relu, float drop_p, uint64_t seed, float* __restrict__ dx,
                                                           float* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float mean = save[c], rstd = save[C + c], gr = gamma[c] * rstd;
  const float m1 = training ? (float)(sums[c] * inv_count) : 0.f, m2 = training ? (float)(sums[C + c] * inv_count) : 0.f;
  if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) { dgamma[c] = (float)sums[C + c]; dbeta[c] = (float)sums[c]; }
  const long base = ((long)n * C + c) * V;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const float d = bn_dz(dout, out, base + v, relu, drop_p, seed);
    const float xh = (x[base + v] - mean) * rstd;
    dx[base + v] = gr * (d - m1 - xh * m2);
    if (dres) dres[base + v] = d;
  }
}

// ---- head: AdaptiveAvgPool3d(1) -> flatten -> Linear -> sigmoid  (models/resnet.py:152-167) --------------------------------------
__global__ void __launch_bounds__(256) gap_fc_sigmoid_fwd_kernel(int N, int C, long V, int O, const float* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                                 float* __restrict__ pooled, float* __restrict__ y) {
  __shared__ float red[4];
  __shared__ float pl[256];
  const int n = blockIdx.x;
  for (int c = 0; c < C; ++c) {
    const float* p = x + ((long)n * C + c) * V;
    float s = 0.f;
    for (long v = threadIdx.x; v < V; v += 256) s += p[v];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { pl[c] = (red[0] + red[1] + red[2] + red[3]) / (float)V; pooled[(long)n * C + c] = pl[c]; }
    __syncthreads();
  }
  for (int o = threadIdx.x; o < O; o += 256) {
    float s = b[o];
    for (int c = 0; c < C; ++c) s = fmaf(pl[c], w[(long)o * C + c], s);
    y[(long)n * O + o] = 1.f / (1.f + expf(-s));
  }
}
// dlogit = dy * y (1 - y);  dw[o][c] = sum_n dlogit[n][o] pooled[n][c];  db[o] = sum_n dlogit;  dx[n][c][v] = sum_o dlogit w[o][c] / V
__global__ void __launch_bounds__(256) gap_fc_sigmoid_bwd_kernel(int N, int C, long V, int O, const float* __restrict__ w,
                                                                 const float* __restrict__ pooled, const float* __restrict__ y,
                                                                 const float* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db,
                                                                 float* __restrict__ dx) {
  if (blockIdx.y == 0 && blockIdx.x == 0) {
    for (int e = threadIdx.x; e < O * C; e += 256) {
      const int o = e / C, c = e % C;
      float s = 0.f;
      for (int n = 0; n < N; ++n) { const float yv = y[(long)n * O + o]; s = fmaf(dy[(long)n * O + o] * yv * (1.f - yv), pooled[(long)n * C + c], s); }
      dw[e] = s;
    }
    for (int o = threadIdx.x; o < O; o += 256) {
      float s = 0.f;
      for (int n = 0; n < N; ++n) { const float yv = y[(long)n * O + o]; s += dy[(long)n * O + o] * yv * (1.f - yv); }
      db[o] = s;
    }
  }
  const int c = blockIdx.y % C, n = blockIdx.y / C;
  float g = 0.f;
  for (int o = 0; o < O; ++o) { const float yv = y[(long)n * O + o]; g = fmaf(dy[(long)n * O + o] * yv * (1.f - yv), w[(long)o * C + c], g); }
  g /= (float)V;
  float* p = dx + ((long)n * C + c) * V;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) p[v] = g;
}

static int geom_from(ConvGeom& g, const mmnn_conv3d_desc* d) {
  MMNN_REQUIRE(d, "conv3d: null descriptor");
  g.N = d->n; g.Cin = d->c_in; g.D = d->d; g.H = d->h; g.W = d->w; g.Cout = d->c_out;
  g.kd = d->kernel[0]; g.kh = d->kernel[1]; g.kw = d->kernel[2];
  g.sd = d->stride[0]; g.sh = d->stride[1]; g.sw = d->stride[2];
  g.pd = d->padding[0]; g.ph = d->padding[1]; g.pw = d->padding[2];
  MMNN_REQUIRE(g.N > 0 && g.Cin > 0 && g.Cout > 0 && g.D > 0 && g.H > 0 && g.W > 0, "conv3d: non-positive extent");
  MMNN_REQUIRE(g.kd > 0 && g.kh > 0 && g.kw > 0 && g.sd > 0 && g.sh > 0 && g.sw > 0 && g.pd >= 0 && g.ph >= 0 && g.pw >= 0, "conv3d: bad kernel / stride / padding");
  g.Do = (g.D + 2 * g.pd - g.kd) / g.sd + 1; g.Ho = (g.H + 2 * g.ph - g.kh) / g.sh + 1; g.Wo = (g.W + 2 * g.pw - g.kw) / g.sw + 1;
  MMNN_REQUIRE(g.Do > 0 && g.Ho > 0 && g.Wo > 0, "conv3d: empty output");
  MMNN_REQUIRE(g.N <= 65535 && g.Cin <= 65535 && g.Cout <= 65535, "conv3d: batch / channel count beyond the grid limits");
  return 0;
}
static int wgrad_splits(const ConvGeom& g) {
  const long tot = (long)g.N * g.Do * g.Ho * g.Wo;
  long s = 2048 / std::max(1, g.Cin);            // about 2048 blocks
  s = std::max<long>(1, std::min<long>(s, 512));
  return (int)std::max<long>(1, std::min(s, tot));
}
template <typename K>
static int set_smem(K kern, size_t smem) {
  MMNN_REQUIRE(smem <= 160 * 1024, "conv3d: filter bank of %zu bytes exceeds the 160 KiB LDS", smem);
  if (smem > 48 * 1024) MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  return 0;
}

}  // namespace mmnn

using namespace mmnn;

extern "C" {

int mmnn_conv3d_out_shape(const mmnn_conv3d_desc* d, int32_t* od, int32_t* oh, int32_t* ow) {
  ConvGeom g;
  if (int rc = geom_from(g, d)) return rc;
  MMNN_REQUIRE(od && oh && ow, "conv3d_out_shape: null output");
  *od = g.Do; *oh = g.Ho; *ow = g.Wo;
  return 0;
}

int mmnn_conv3d_forward(const mmnn_conv3d_desc* d, const float* x, const float* w, float* y, void* stream) {
  ConvGeom g;
  if (int rc = geom_from(g, d)) return rc;
  MMNN_REQUIRE(x && w && y, "conv3d_forward: null buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long Vo = (long)g.Do * g.Ho * g.Wo;
  const int taps = g.kd * g.kh * g.kw;
  if (g.Cout <= 8) {
    const size_t smem = sizeof(float) * (size_t)g.Cin * taps * 8;
    if (int rc = set_smem(conv3d_fwd_kernel<8>, smem)) return rc;
    MMNN_LAUNCH(conv3d_fwd_kernel<8>, dim3(cdiv(Vo, 256), 1, g.N), dim3(256), smem, s, g, x, w, y);
  } else {
    const size_t smem = sizeof(float) * (size_t)g.Cin * taps * 16;
    if (int rc = set_smem(conv3d_fwd_kernel<16>, smem)) return rc;
    MMNN_LAUNCH(conv3d_fwd_kernel<16>, dim3(cdiv(Vo, 256), cdiv(g.Cout, 16), g.N), dim3(256), smem, s, g, x, w, y);
  }
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_conv3d_backward_data(const mmnn_conv3d_desc* d, const float* dy, const float* w, float* dx, void* stream) {
  ConvGeom g;
  if (int rc = geom_from(g, d)) return rc;
  MMNN_REQUIRE(dy && w && dx, "conv3d_backward_data: null buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long Vi = (long)g.D * g.H * g.W;
  const int taps = g.kd * g.kh * g.kw;
  if (g.Cin <= 8) {
    const size_t smem = sizeof(float) * (size_t)g.Cout * taps * 8;
    if (int rc = set_smem(conv3d_dgrad_kernel<8>, smem)) return rc;
    MMNN_LAUNCH(conv3d_dgrad_kernel<8>, dim3(cdiv(Vi, 256), 1, g.N), dim3(256), smem, s, g, dy, w, dx);
  } else {
    const size_t smem = sizeof(float) * (size_t)g.Cout * taps * 16;
    if (int rc = set_smem(conv3d_dgrad_kernel<16>, smem)) return rc;
    MMNN_LAUNCH(conv3d_dgrad_kernel<16>, dim3(cdiv(Vi, 256), cdiv(g.Cin, 16), g.N), dim3(256), smem, s, g, dy, w, dx);
  }
  MMNN_HIP(hipGetLastError());
  return 0;
}

int64_t mmnn_conv3d_wgrad_workspace_bytes(const mmnn_conv3d_desc* d) {
  ConvGeom g;
  if (geom_from(g, d)) return -1;
  return (int64_t)sizeof(float) * wgrad_splits(g) * g.Cout * g.Cin * g.kd * g.kh * g.kw;
}

int mmnn_conv3d_backward_weight(const mmnn_conv3d_desc* d, const float* x, const float* dy, float* dw, void* workspace, int32_t accumulate,
                                void* stream) {
  ConvGeom g;
  if (int rc = geom_from(g, d)) return rc;
  MMNN_REQUIRE(x && dy && dw && workspace, "conv3d_backward_weight: null buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ns = wgrad_splits(g);
  const long count = (long)g.Cout * g.Cin * g.kd * g.kh * g.kw;
  float* slab = static_cast<float*>(workspace);
  MMNN_LAUNCH(conv3d_wgrad_kernel, dim3(ns, g.Cin), dim3(256), 0, s, g, x, dy, slab, ns);
  MMNN_LAUNCH(slab_sum_kernel, dim3((unsigned)std::min<long>(1024, cdiv(count, 256))), dim3(256), 0, s, (const float*)slab, count, ns, dw, (int)accumulate);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_bn3d_forward(int32_t n, int32_t c, int64_t v, const float* x, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int32_t training, int32_t relu, const float* residual,
                      float dropout_prob, uint64_t seed, float* out, float* save, double* stat_ws, void* stream) {
  MMNN_REQUIRE(n > 0 && c > 0 && v > 0 && n <= 65535 && c <= 65535 && x && gamma && beta && running_mean && running_var && out && save && stat_ws,
               "bn3d_forward: bad arguments");
  MMNN_REQUIRE(!(training && (long)n * v < 2), "bn3d_forward: Expected more than 1 value per channel when training");
  MMNN_REQUIRE(dropout_prob >= 0.f && dropout_prob < 1.f, "bn3d_forward: dropout probability must be in [0, 1)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int gx = (int)std::min<long>(64, cdiv(v, 256));
  if (training) {
    MMNN_HIP(hipMemsetAsync(stat_ws, 0, sizeof(double) * 2 * c, s));
    MMNN_LAUNCH(bn_stats_kernel, dim3(gx, c, n), dim3(256), 0, s, (int)n, (int)c, (long)v, x, stat_ws);
  }
  MMNN_LAUNCH(bn_finalize_kernel, dim3(cdiv(c, 64)), dim3(64), 0, s, (int)c, (double)n * (double)v, (const double*)stat_ws, eps, momentum,
              (int)training, running_mean, running_var, save);
  MMNN_LAUNCH(bn_act_fwd_kernel, dim3(gx, c, n), dim3(256), 0, s, (int)n, (int)c, (long)v, x, (const float*)save, gamma, beta, residual, (int)relu,
              training ? dropout_prob : 0.f, seed, out);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_bn3d_backward(int32_t n, int32_t c, int64_t v, const float* x, const float* out, const float* dout, const float* gamma,
                       const float* save, int32_t training, int32_t relu, float dropout_prob, uint64_t seed, float* dx, float* dresidual,
                       float* dgamma, float* dbeta, double* stat_ws, void* stream) {
  MMNN_REQUIRE(n > 0 && c > 0 && v > 0 && n <= 65535 && c <= 65535 && x && out && dout && gamma && save && dx && dgamma && dbeta && stat_ws,
               "bn3d_backward: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int gx = (int)std::min<long>(64, cdiv(v, 256));
  const float p = training ? dropout_prob : 0.f;
  MMNN_HIP(hipMemsetAsync(stat_ws, 0, sizeof(double) * 2 * c, s));
  MMNN_LAUNCH(bn_bwd_reduce_kernel, dim3(gx, c, n), dim3(256), 0, s, (int)n, (int)c, (long)v, x, out, dout, save, (int)relu, p, seed, stat_ws);
  MMNN_LAUNCH(bn_bwd_apply_kernel, dim3(gx, c, n), dim3(256), 0, s, (int)n, (int)c, (long)v, x, out, dout, save, gamma, (const double*)stat_ws,
              1.0 / ((double)n * (double)v), (int)training, (int)relu, p, seed, dx, dresidual, dgamma, dbeta);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_gap_fc_sigmoid_forward(int32_t n, int32_t c, int64_t v, int32_t o, const float* x, const float* w, const float* b, float* pooled,
                                float* y, void* stream) {
  MMNN_REQUIRE(n > 0 && c > 0 && c <= 256 && v > 0 && o > 0 && x && w && b && pooled && y, "gap_fc_sigmoid_forward: bad arguments (<= 256 channels)");
  MMNN_LAUNCH(gap_fc_sigmoid_fwd_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), (int)n, (int)c, (long)v, (int)o, x, w, b, pooled, y);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_gap_fc_sigmoid_backward(int32_t n, int32_t c, int64_t v, int32_t o, const float* w, const float* pooled, const float* y, const float* dy,
                                 float* dw, float* db, float* dx, void* stream) {
  MMNN_REQUIRE(n > 0 && c > 0 && v > 0 && o > 0 && (long)n * c <= 65535 && w && pooled && y && dy && dw && db && dx, "gap_fc_sigmoid_backward: bad arguments");
  const int gx = (int)std::min<long>(64, cdiv(v, 256));
  MMNN_LAUNCH(gap_fc_sigmoid_bwd_kernel, dim3(gx, n * c), dim3(256), 0, static_cast<hipStream_t>(stream), (int)n, (int)c, (long)v, (int)o, w, pooled, y,
              dy, dw, db, dx);
  MMNN_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
