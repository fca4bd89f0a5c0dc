// Small single-block kernels for the "tail" of the fusion model -- the pieces with ~0 FLOPs whose cost is launch
// count, fused so that a training step spends a handful of launches there instead of ~200 eager ops:
//   * DenseNet.features          models/densenet.py:234-247   ReLU -> global average pool -> Linear -> Dropout
//   * MLP.backbone / MLP.features models/mlp.py:19-51          [Linear -> BatchNorm1d -> (ReLU, Dropout1d)] x k
//   * fusion heads                models/multimodal.py:62-77   cat -> Linear ; per-modality Linear ; stack
//   * Cox partial likelihood, blended  losses/losses.py:6-9, utils/utils.py:24-29, losses/GradientBlender.py:197-205
#include <string.h>

#include <algorithm>

#include "../../include/mmnn_sts.h"
#include "common.hpp"

namespace mmnn {

// =====================================================================================================================
// DenseNet.features
// =====================================================================================================================
struct GapArgs {
  int N, C, V, F;
  const float* h;        // [N][C][V]  (norm5 output)
  const float* w;        // [F][C]
  const float* b;        // [F]
  float* pooled;         // [N][C]  saved: mean_v relu(h)
  float* out;            // [N][F]
  float p; uint64_t seed; int training;
};

__device__ __forceinline__ float elem_drop_scale(uint64_t seed, float p, int training, long idx) {
  if (!training || p <= 0.f) return 1.f;
  DropCfg d; d.seed = seed; d.p = p; d.layer = 0x7F0000 + (int)(idx >> 20);
  return drop_scale(d, (int)((idx >> 10) & 1023), (int)(idx & 1023));
}

__global__ void __launch_bounds__(256) gap_kernel(const GapArgs a) {   // grid (C/4 groups, N): one wave per (n,c)
  const int n = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + wave;
  if (c >= a.C) return;
  const float* hc = a.h + ((long)n * a.C + c) * a.V;
  float s = 0.f;
  for (int v = lane; v < a.V; v += 64) s += fmaxf(hc[v], 0.f);
  s = wave_sum(s);
  if (lane == 0) a.pooled[(long)n * a.C + c] = s / (float)a.V;
}

__global__ void __launch_bounds__(256) gap_linear_kernel(const GapArgs a) {   // a wave per output element (r03: over a grid, not one block:
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;                 // 256 dependent dot products in a row were 31 us of every step)
  for (int o = blockIdx.x * 4 + wave; o < a.N * a.F; o += 4 * gridDim.x) {
    const int n = o / a.F, f = o % a.F;
    float s = 0.f;
    for (int c = lane; c < a.C; c += 64) s += a.pooled[(long)n * a.C + c] * a.w[(long)f * a.C + c];
    s = wave_sum(s);
    if (lane == 0) a.out[o] = (s + a.b[f]) * elem_drop_scale(a.seed, a.p, a.training, o);
  }
}

struct GapBwdArgs {
  int N, C, V, F;
  const float* h; const float* w; const float* pooled;
  const float* dout;     // [N][F]
  float* dw; float* db;  // [F][C], [F]
  float* dh;             // [N][C][V]
  float p; uint64_t seed; int training; int accumulate;
};

__global__ void __launch_bounds__(256) gap_bwd_kernel(const GapBwdArgs a) {   // grid (C, N)
  const int c = blockIdx.x, n = blockIdx.y;
  float dg = 0.f;
  for (int f = 0; f < a.F; ++f)
    dg += a.dout[n * a.F + f] * elem_drop_scale(a.seed, a.p, a.training, n * a.F + f) * a.w[(long)f * a.C + c];
  dg /= (float)a.V;
  const float* hc = a.h + ((long)n * a.C + c) * a.V;
  float* dc = a.dh + ((long)n * a.C + c) * a.V;
  for (int v = threadIdx.x; v < a.V; v += 256) dc[v] = hc[v] > 0.f ? dg : 0.f;
  if (n == 0) {   // parameter gradients: this block owns column c of dW (and db when c == 0)
    for (int f = threadIdx.x; f < a.F; f += 256) {
      float s = 0.f, sb = 0.f;
      for (int m = 0; m < a.N; ++m) {
        const float d = a.dout[m * a.F + f] * elem_drop_scale(a.seed, a.p, a.training, m * a.F + f);
        s += d * a.pooled[(long)m * a.C + c];
        sb += d;
      }
      float* pw = a.dw + (long)f * a.C + c;
      *pw = a.accumulate ? *pw + s : s;
      if (c == 0) a.db[f] = a.accumulate ? a.db[f] + sb : sb;
    }
  }
}

// =====================================================================================================================
// Linear -> BatchNorm1d -> ReLU/Dropout1d stack (one block; batch statistics over the N rows)
// =====================================================================================================================
constexpr int MLP_MAXL = 8;
struct MlpArgs {
  int N, nl;
  int din[MLP_MAXL], dout[MLP_MAXL], relu_first[MLP_MAXL];
  const float* w[MLP_MAXL]; const float* b[MLP_MAXL]; const float* gamma[MLP_MAXL]; const float* beta[MLP_MAXL];
  float* rmean[MLP_MAXL]; float* rvar[MLP_MAXL]; long long* nbt[MLP_MAXL];
  float* dw[MLP_MAXL]; float* db[MLP_MAXL]; float* dgamma[MLP_MAXL]; float* dbeta[MLP_MAXL];
  const float* x;          // [N][din0]
  float* out;              // [N][dout_last]
  float* saved;            // per layer: xhat [N][dout], act [N][dout], rstd [dout]
  const float* dy;         // backward: [N][dout_last]
  float* dx;               // backward: [N][din0] (may be null)
  float* scratch;          // backward: 2 * N * maxdim floats
  float p, eps, momentum; uint64_t seed; int training, layer0, accumulate;
};

__device__ __forceinline__ long mlp_saved_off(const MlpArgs& a, int i) {
  long o = 0;
  for (int j = 0; j < i; ++j) o += 2l * a.N * a.dout[j] + a.dout[j];
  return o;
}
__device__ __forceinline__ float row_drop(const MlpArgs& a, int i, int n) {
  DropCfg d; d.seed = a.seed; d.p = a.training ? a.p : 0.f; d.layer = 0x7E0000 + a.layer0 + i;
  return drop_scale(d, n, 0);
}

__global__ void __launch_bounds__(256) mlp_fwd_kernel(const MlpArgs a) {
  const int tid = threadIdx.x;
  const float* x = a.x;
  for (int i = 0; i < a.nl; ++i) {
    const int D = a.din[i], O = a.dout[i];
    float* xhat = a.saved + mlp_saved_off(a, i);
    float* act = xhat + (long)a.N * O;
    float* rstd = act + (long)a.N * O;
    float* dst = (i == a.nl - 1) ? a.out : act;
    for (int e = tid; e < a.N * O; e += 256) {       // z = x W^T + b   (kept in xhat for now)
      const int n = e / O, o = e % O;
      float s = a.b[i][o];
#pragma unroll 8
      for (int k = 0; k < D; ++k) s = fmaf(x[n * D + k], a.w[i][o * D + k], s);     // (unrolled: eight independent loads in flight, same summation order)
      xhat[e] = s;
    }
    __syncthreads();
    for (int o = tid; o < O; o += 256) {             // statistics per feature
      double mean, var;
      if (a.training) {
        double s = 0.0, q = 0.0;
        for (int n = 0; n < a.N; ++n) { const double z = xhat[n * O + o]; s += z; q += z * z; }
        mean = s / a.N; var = q / a.N - mean * mean;
        if (var < 0.0) var = 0.0;
        const double unb = a.N > 1 ? var * a.N / (a.N - 1.0) : var;
        a.rmean[i][o] = (float)((1.0 - a.momentum) * a.rmean[i][o] + a.momentum * mean);
        a.rvar[i][o] = (float)((1.0 - a.momentum) * a.rvar[i][o] + a.momentum * unb);
      } else {
        mean = a.rmean[i][o]; var = a.rvar[i][o];
      }
      const float r = (float)(1.0 / sqrt(var + (double)a.eps));
      rstd[o] = r;
      for (int n = 0; n < a.N; ++n) xhat[n * O + o] = (float)(((double)xhat[n * O + o] - mean) * r);
    }
    __syncthreads();
    for (int e = tid; e < a.N * O; e += 256) {
      const int n = e / O, o = e % O;
      const float y = fmaf(a.gamma[i][o], xhat[e], a.beta[i][o]);
      const float m = row_drop(a, i, n);
      dst[e] = a.relu_first[i] ? fmaxf(y, 0.f) * m : fmaxf(y * m, 0.f);
      if (dst != act) act[e] = dst[e];
    }
    __syncthreads();
    x = act;
  }
  if (a.training && tid < a.nl && a.nbt[tid]) *a.nbt[tid] += 1;      // nn.BatchNorm1d.num_batches_tracked
}

__global__ void __launch_bounds__(256) mlp_bwd_kernel(const MlpArgs a) {
  const int tid = threadIdx.x;
  int maxd = a.din[0];
  for (int i = 0; i < a.nl; ++i) maxd = max(maxd, a.dout[i]);
  float* g = a.scratch;                   // upstream gradient of the current layer's output [N][O]
  float* dz = a.scratch + (long)a.N * maxd;
  const int OL = a.dout[a.nl - 1];
  for (int e = tid; e < a.N * OL; e += 256) g[e] = a.dy[e];
  __syncthreads();
  for (int i = a.nl - 1; i >= 0; --i) {
    const int D = a.din[i], O = a.dout[i];
    const float* xhat = a.saved + mlp_saved_off(a, i);
    const float* rstd = xhat + 2l * a.N * O;
    const float* xin = (i == 0) ? a.x : (a.saved + mlp_saved_off(a, i - 1) + (long)a.N * a.dout[i - 1]);
    for (int e = tid; e < a.N * O; e += 256) {       // through ReLU / Dropout1d -> dy
      const int n = e / O, o = e % O;
      const float y = fmaf(a.gamma[i][o], xhat[e], a.beta[i][o]);
      const float m = row_drop(a, i, n);
      const float pre = a.relu_first[i] ? y : y * m;
      g[e] = pre > 0.f ? g[e] * m : 0.f;
    }
    __syncthreads();
    for (int o = tid; o < O; o += 256) {             // BN backward per feature
      double s0 = 0.0, s1 = 0.0;
      for (int n = 0; n < a.N; ++n) { s0 += g[n * O + o]; s1 += (double)g[n * O + o] * xhat[n * O + o]; }
      a.dbeta[i][o] = (a.accumulate ? a.dbeta[i][o] : 0.f) + (float)s0;
      a.dgamma[i][o] = (a.accumulate ? a.dgamma[i][o] : 0.f) + (float)s1;
      const float k = a.gamma[i][o] * rstd[o];
      const float m0 = (float)(s0 / a.N), m1 = (float)(s1 / a.N);
      float sb = 0.f;
      for (int n = 0; n < a.N; ++n) {
        const float v = k * (g[n * O + o] - m0 - xhat[n * O + o] * m1);
        dz[n * O + o] = v;
        sb += v;
      }
      a.db[i][o] = (a.accumulate ? a.db[i][o] : 0.f) + sb;
    }
    __syncthreads();
    for (int e = tid; e < O * D; e += 256) {         // dW
      const int o = e / D, k = e % D;
      float s = 0.f;
      for (int n = 0; n < a.N; ++n) s = fmaf(dz[n * O + o], xin[n * D + k], s);
      a.dw[i][e] = (a.accumulate ? a.dw[i][e] : 0.f) + s;
    }
    float* gnext = (i == 0) ? a.dx : g;
    __syncthreads();
    if (gnext) {
      for (int e = tid; e < a.N * D; e += 256) {     // dx = dz W
        const int n = e / D, k = e % D;
        float s = 0.f;
#pragma unroll 8
        for (int o = 0; o < O; ++o) s = fmaf(dz[n * O + o], a.w[i][o * D + k], s);
        gnext[e] = s;
      }
    }
    __syncthreads();
  }
}

// =====================================================================================================================
// fusion heads: out[0] = [fi, fc] Wf^T + bf ; out[1] = fi Wi^T + bi ; out[2] = fc Wc^T + bc
// =====================================================================================================================
struct HeadsArgs {
  int N, F, C, blend;
  const float* fi; const float* fc;
  const float* wf; const float* bf; const float* wi; const float* bi; const float* wc; const float* bc;
  float* out;                              // [H][N][C], H = 3 (blend) or 1
  const float* dout;
  float* dfi; float* dfc; float* dwf; float* dbf; float* dwi; float* dbi; float* dwc; float* dbc;
  int accumulate;
};

__global__ void __launch_bounds__(256) heads_fwd_kernel(const HeadsArgs a) {
  const int H = a.blend ? 3 : 1;
  for (int e = threadIdx.x; e < H * a.N * a.C; e += 256) {
    const int c = e % a.C, n = (e / a.C) % a.N, h = e / (a.C * a.N);
    float s;
    if (h == 0) {
      s = a.bf[c];
      for (int k = 0; k < a.F; ++k) s = fmaf(a.fi[n * a.F + k], a.wf[c * 2 * a.F + k], s);
      for (int k = 0; k < a.F; ++k) s = fmaf(a.fc[n * a.F + k], a.wf[c * 2 * a.F + a.F + k], s);
    } else {
      const float* f = (h == 1) ? a.fi : a.fc;
      const float* w = (h == 1) ? a.wi : a.wc;
      s = (h == 1) ? a.bi[c] : a.bc[c];
      for (int k = 0; k < a.F; ++k) s = fmaf(f[n * a.F + k], w[c * a.F + k], s);
    }
    a.out[e] = s;
  }
}

__global__ void __launch_bounds__(256) heads_bwd_kernel(const HeadsArgs a) {
  const int N = a.N, F = a.F, C = a.C;
  const float* d0 = a.dout;
  const float* d1 = a.blend ? a.dout + N * C : nullptr;
  const float* d2 = a.blend ? a.dout + 2 * N * C : nullptr;
  auto put = [&](float* p, float v) { *p = a.accumulate ? *p + v : v; };
  for (int e = threadIdx.x; e < N * F; e += 256) {          // feature gradients
    const int n = e / F, k = e % F;
    float si = 0.f, sc = 0.f;
    for (int c = 0; c < C; ++c) {
      si = fmaf(d0[n * C + c], a.wf[c * 2 * F + k], si);
      sc = fmaf(d0[n * C + c], a.wf[c * 2 * F + F + k], sc);
      if (a.blend) {
        si = fmaf(d1[n * C + c], a.wi[c * F + k], si);
        sc = fmaf(d2[n * C + c], a.wc[c * F + k], sc);
      }
    }
    a.dfi[e] = si;
    a.dfc[e] = sc;
  }
  for (int e = threadIdx.x; e < C * 2 * F; e += 256) {      // fused head weight
    const int c = e / (2 * F), k = e % (2 * F);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(d0[n * C + c], k < F ? a.fi[n * F + k] : a.fc[n * F + k - F], s);
    put(a.dwf + e, s);
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int n = 0; n < N; ++n) { s0 += d0[n * C + c]; if (a.blend) { s1 += d1[n * C + c]; s2 += d2[n * C + c]; } }
    put(a.dbf + c, s0);
    if (a.blend) { put(a.dbi + c, s1); put(a.dbc + c, s2); }
  }
  if (a.blend) {
    for (int e = threadIdx.x; e < C * F; e += 256) {
      const int c = e / F, k = e % F;
      float s1 = 0.f, s2 = 0.f;
      for (int n = 0; n < N; ++n) { s1 = fmaf(d1[n * C + c], a.fi[n * F + k], s1); s2 = fmaf(d2[n * C + c], a.fc[n * F + k], s2); }
      put(a.dwi + e, s1);
      put(a.dwc + e, s2);
    }
  }
}

// =====================================================================================================================
// Cox partial likelihood as the reference calls it (losses/losses.py:8-9 passes (log_h, events, duration) into pycox's
// (log_h, durations, events)): `key` is the sort key (descending, stable), `wgt` the weights.  One block per
// (head, target) problem.   loss = -sum_i w_i (h_i - lcs_i) / sum_i w_i,  lcs_i = log(cumsum_i exp(h - max h) + eps) + max h
// =====================================================================================================================
struct CoxArgs {
  int H, N, C;
  const float* preds;        // [H][N][C]
  const void* key; int key_dt;   // [N][C], element type MMNN_DT_*: read as fp64 (holds every int64 duration < 2^53 and every fp32 exactly)
  const void* wgt; int wgt_dt;   // [N][C]
  const float* hw;           // [H] blend weights (null: all 1)
  float* head_loss;          // [H]  (sum over targets)
  float* loss;               // [1]  sum_h hw[h] * head_loss[h]
  float* grad;               // [H][N][C]  d loss / d preds
  float* scratch;            // [4*N]
  float eps;
};

__device__ __forceinline__ double load_f64(const void* p, int dt, long i) {
  switch (dt) {
    case MMNN_DT_F32: return (double)static_cast<const float*>(p)[i];
    case MMNN_DT_I64: return (double)static_cast<const long long*>(p)[i];
    case MMNN_DT_I32: return (double)static_cast<const int*>(p)[i];
    case MMNN_DT_U8: return (double)static_cast<const unsigned char*>(p)[i];
    default: return static_cast<const double*>(p)[i];
  }
}

__global__ void __launch_bounds__(256) cox_kernel(const CoxArgs a) {   // ONE block: problems in order => reproducible sums
  const int N = a.N, tid = threadIdx.x;
  __shared__ float sh_g, sh_w;
  __shared__ float pl[64];                              // per-problem losses (H*C <= 64)
  __shared__ float lds_scratch[4 * 1024];               // small batches: the serial part below is a chain of dependent accesses
  float* const scratch = N <= 1024 ? lds_scratch : a.scratch;
  for (int pb = 0; pb < a.H * a.C; ++pb) {
    const int h = pb / a.C, c = pb % a.C;
    float* hs = scratch;                                // sorted log-hazards
    float* ws = hs + N;                                 // sorted weights
    float* cs = ws + N;                                 // cumsum / later suffix sums
    int* pos = reinterpret_cast<int*>(cs + N);          // original index of sorted slot
    __syncthreads();
    for (int i = tid; i < N; i += 256) {                // stable descending rank
      const double ki = load_f64(a.key, a.key_dt, i * a.C + c);
      int r = 0;
      for (int j = 0; j < N; ++j) {
        const double kj = load_f64(a.key, a.key_dt, j * a.C + c);
        r += (kj > ki) || (kj == ki && j < i);
      }
      hs[r] = a.preds[((long)h * N + i) * a.C + c];
      ws[r] = (float)load_f64(a.wgt, a.wgt_dt, i * a.C + c);
      pos[r] = i;
    }
    __syncthreads();
    if (tid == 0) {
      float g = -INFINITY;
      double W = 0.0;
      for (int i = 0; i < N; ++i) { g = fmaxf(g, hs[i]); W += load_f64(a.wgt, a.wgt_dt, pos[i] * a.C + c); }
      float run = 0.f, num = 0.f;
      for (int i = 0; i < N; ++i) {
        run += expf(hs[i] - g);
        cs[i] = run + a.eps;
        num += (hs[i] - (logf(cs[i]) + g)) * ws[i];
      }
      pl[pb] = -num / (float)W;
      sh_g = g; sh_w = (float)W;
      float suf = 0.f;                                  // suffix sums of w_i / (cumsum_i + eps)
      for (int i = N - 1; i >= 0; --i) { suf += ws[i] / cs[i]; cs[i] = suf; }
    }
    __syncthreads();
    const float g = sh_g, W = sh_w, w8 = a.hw ? a.hw[h] : 1.f;
    for (int k = tid; k < N; k += 256) {
      const float d = -(ws[k] - expf(hs[k] - g) * cs[k]) / W;
      a.grad[((long)h * N + pos[k]) * a.C + c] = w8 * d;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float total = 0.f;
    for (int h = 0; h < a.H; ++h) {
      float l = 0.f;
      for (int c = 0; c < a.C; ++c) l += pl[h * a.C + c];
      a.head_loss[h] = l;
      total += (a.hw ? a.hw[h] : 1.f) * l;
    }
    *a.loss = total;
  }
}

// autograd adjoint of the blended loss: out = saved * dloss + saved * dheads[h] / head_weights[h]  (saved = d loss / d preds as written by
// cox_kernel, i.e. already scaled by head_weights[h]; either upstream gradient may be absent)
__global__ void __launch_bounds__(256) cox_bwd_kernel(long total, long per_head, const float* saved, const float* hw, const float* dloss,
                                                      const float* dheads, float* out) {
  const float dl = dloss ? dloss[0] : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    float f = dl;
    if (dheads) {
      const long h = i / per_head;
      f += dheads[h] / (hw ? hw[h] : 1.f);
    }
    out[i] = saved[i] * f;
  }
}

// =====================================================================================================================
// small dense layer  y = x W^T + b   (class_layers.out, MLP.output_head.dense6: models/densenet.py:250-256, mlp.py:53-57)
// =====================================================================================================================
__global__ void __launch_bounds__(256) linear_fwd_kernel(int N, int D, int O, const float* x, const float* w, const float* b, float* y) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < N * O; e += gridDim.x * 256) {
    const int n = e / O, o = e % O;
    float s = b ? b[o] : 0.f;
    for (int k = 0; k < D; ++k) s = fmaf(x[(long)n * D + k], w[(long)o * D + k], s);
    y[e] = s;
  }
}
__global__ void __launch_bounds__(256) linear_bwd_kernel(int N, int D, int O, const float* x, const float* w, const float* dy, float* dx,
                                                         float* dw, float* db, int accumulate) {
  const int t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  for (int e = t; e < N * D; e += stride) {
    const int n = e / D, k = e % D;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(dy[n * O + o], w[(long)o * D + k], s);
    if (dx) dx[e] = s;
  }
  for (int e = t; e < O * D; e += stride) {
    const int o = e / D, k = e % D;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(dy[n * O + o], x[(long)n * D + k], s);
    dw[e] = accumulate ? dw[e] + s : s;
  }
  if (db)
    for (int o = t; o < O; o += stride) {
      float s = 0.f;
      for (int n = 0; n < N; ++n) s += dy[n * O + o];
      db[o] = accumulate ? db[o] + s : s;
    }
}

// =====================================================================================================================
// element-wise binary cross entropy on logits with per-class positive weights -- nn.BCEWithLogitsLoss(pos_weight, reduction
// ='none') as the classification trainer builds it (main.py:147-153), for `criterion` (utils/utils.py:20-22) and
// GradientBlender.computeLossClassification (losses/GradientBlender.py:150-179).
//   l = pw_c * y * softplus(-x) + (1 - y) * softplus(x),   dl/dx = (1 - y) * sigmoid(x) - pw_c * y * sigmoid(-x)
// =====================================================================================================================
__global__ void __launch_bounds__(256) bce_logits_kernel(long total, int C, const float* x, const float* y, const float* pw, float* loss,
                                                          float* dldx) {
  for (long e = blockIdx.x * 256l + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const float xv = x[e], yv = y[e], w = pw ? pw[e % C] : 1.f;
    const float ax = fabsf(xv), l1p = log1pf(expf(-ax));          // stable softplus: softplus(t) = max(t, 0) + log1p(exp(-|t|))
    const float sp_pos = fmaxf(xv, 0.f) + l1p, sp_neg = fmaxf(-xv, 0.f) + l1p;
    const float sg = 1.f / (1.f + expf(-xv));
    loss[e] = w * yv * sp_neg + (1.f - yv) * sp_pos;
    if (dldx) dldx[e] = (1.f - yv) * sg - w * yv * (1.f - sg);
  }
}

}  // namespace mmnn

using namespace mmnn;

extern "C" {

int mmnn_bce_logits(int64_t total, int32_t c, const float* logits, const float* targets, const float* pos_weight, float* loss,
                    float* dloss_dlogits, void* stream) {
  MMNN_REQUIRE(total > 0 && c > 0 && total % c == 0 && logits && targets && loss, "bce_logits: bad arguments");
  const int blocks = (int)std::min<long>(1024, (total + 255) / 256);
  MMNN_LAUNCH(bce_logits_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), (long)total, (int)c, logits, targets,
              pos_weight, loss, dloss_dlogits);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_gap_linear_forward(int32_t n, int32_t c, int32_t v, int32_t f, const float* h, const float* w, const float* b, float* pooled,
                            float* out, float p, uint64_t seed, int32_t training, void* stream) {
  MMNN_REQUIRE(n > 0 && c > 0 && v > 0 && f > 0 && n <= 65535 && h && w && b && pooled && out, "gap_linear_forward: bad arguments");
  GapArgs a{n, c, v, f, h, w, b, pooled, out, p, seed, training};
  hipStream_t s = static_cast<hipStream_t>(stream);
  MMNN_LAUNCH(gap_kernel, dim3(cdiv(c, 4), n), dim3(256), 0, s, a);
  MMNN_LAUNCH(gap_linear_kernel, dim3(std::min(cdiv((long)n * f, 4), 1024)), dim3(256), 0, s, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_gap_linear_backward(int32_t n, int32_t c, int32_t v, int32_t f, const float* h, const float* w, const float* pooled,
                             const float* dout, float* dw, float* db, float* dh, float p, uint64_t seed, int32_t training,
                             int32_t accumulate, void* stream) {
  MMNN_REQUIRE(n > 0 && c > 0 && v > 0 && f > 0 && n <= 65535 && c <= 65535 && h && w && pooled && dout && dw && db && dh, "gap_linear_backward: bad arguments");
  GapBwdArgs a{n, c, v, f, h, w, pooled, dout, dw, db, dh, p, seed, training, accumulate};
  MMNN_LAUNCH(gap_bwd_kernel, dim3(c, n), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int64_t mmnn_mlp_saved_floats(const mmnn_mlp_desc* d) {
  if (!d || d->num_layers < 1 || d->num_layers > MLP_MAXL) return -1;
  int64_t o = 0;
  for (int i = 0; i < d->num_layers; ++i) o += 2ll * d->n * d->out_dim[i] + d->out_dim[i];
  return o;
}

static int mlp_fill(MlpArgs& a, const mmnn_mlp_desc* d, const mmnn_mlp_params* p) {
  MMNN_REQUIRE(d && p && d->num_layers >= 1 && d->num_layers <= MLP_MAXL && d->n >= 1, "mlp: bad descriptor");
  memset(&a, 0, sizeof(a));
  a.N = d->n; a.nl = d->num_layers;
  for (int i = 0; i < a.nl; ++i) {
    MMNN_REQUIRE(d->in_dim[i] > 0 && d->out_dim[i] > 0 && (i == 0 || d->in_dim[i] == d->out_dim[i - 1]), "mlp: layer %d dims inconsistent", i);
    a.din[i] = d->in_dim[i]; a.dout[i] = d->out_dim[i]; a.relu_first[i] = d->relu_first[i];
    a.w[i] = p->weight[i]; a.b[i] = p->bias[i]; a.gamma[i] = p->gamma[i]; a.beta[i] = p->beta[i];
    a.rmean[i] = p->running_mean[i]; a.rvar[i] = p->running_var[i]; a.nbt[i] = reinterpret_cast<long long*>(p->num_batches_tracked[i]);
    a.dw[i] = p->grad_weight[i]; a.db[i] = p->grad_bias[i]; a.dgamma[i] = p->grad_gamma[i]; a.dbeta[i] = p->grad_beta[i];
    MMNN_REQUIRE(a.w[i] && a.b[i] && a.gamma[i] && a.beta[i] && a.rmean[i] && a.rvar[i], "mlp: null parameter pointer in layer %d", i);
  }
  a.p = d->dropout_prob; a.eps = d->eps; a.momentum = d->momentum; a.seed = d->seed; a.training = d->training; a.layer0 = d->first_layer_id;
  return 0;
}

int mmnn_mlp_forward(const mmnn_mlp_desc* d, const mmnn_mlp_params* p, const float* x, float* out, float* saved, void* stream) {
  MlpArgs a;
  if (int rc = mlp_fill(a, d, p)) return rc;
  MMNN_REQUIRE(x && out && saved, "mlp_forward: null buffer");
  MMNN_REQUIRE(!(d->training && d->n < 2), "mlp_forward: batch norm in training mode needs more than 1 value per channel (N=%d)", d->n);
  a.x = x; a.out = out; a.saved = saved;
  MMNN_LAUNCH(mlp_fwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_mlp_backward(const mmnn_mlp_desc* d, const mmnn_mlp_params* p, const float* x, const float* saved, const float* dy, float* dx,
                      float* scratch, int32_t accumulate, void* stream) {
  MlpArgs a;
  if (int rc = mlp_fill(a, d, p)) return rc;
  MMNN_REQUIRE(x && saved && dy && scratch, "mlp_backward: null buffer");
  for (int i = 0; i < a.nl; ++i) MMNN_REQUIRE(a.dw[i] && a.db[i] && a.dgamma[i] && a.dbeta[i], "mlp_backward: null gradient pointer in layer %d", i);
  a.x = x; a.saved = const_cast<float*>(saved); a.dy = dy; a.dx = dx; a.scratch = scratch; a.accumulate = accumulate;
  MMNN_LAUNCH(mlp_bwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_fusion_heads_forward(int32_t n, int32_t f, int32_t c, int32_t blend, const float* fi, const float* fc, const float* wf,
                              const float* bf, const float* wi, const float* bi, const float* wc, const float* bc, float* out,
                              void* stream) {
  MMNN_REQUIRE(n > 0 && f > 0 && c > 0 && fi && fc && wf && bf && out && (!blend || (wi && bi && wc && bc)), "fusion_heads_forward: bad arguments");
  HeadsArgs a;
  memset(&a, 0, sizeof(a));
  a.N = n; a.F = f; a.C = c; a.blend = blend; a.fi = fi; a.fc = fc; a.wf = wf; a.bf = bf; a.wi = wi; a.bi = bi; a.wc = wc; a.bc = bc; a.out = out;
  MMNN_LAUNCH(heads_fwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_fusion_heads_backward(int32_t n, int32_t f, int32_t c, int32_t blend, const float* fi, const float* fc, const float* wf,
                               const float* wi, const float* wc, const float* dout, float* dfi, float* dfc, float* dwf, float* dbf,
                               float* dwi, float* dbi, float* dwc, float* dbc, int32_t accumulate, void* stream) {
  MMNN_REQUIRE(n > 0 && f > 0 && c > 0 && fi && fc && wf && dout && dfi && dfc && dwf && dbf, "fusion_heads_backward: bad arguments");
  MMNN_REQUIRE(!blend || (wi && wc && dwi && dbi && dwc && dbc), "fusion_heads_backward: blend operands missing");
  HeadsArgs a;
  memset(&a, 0, sizeof(a));
  a.N = n; a.F = f; a.C = c; a.blend = blend; a.fi = fi; a.fc = fc; a.wf = wf; a.wi = wi; a.wc = wc; a.dout = dout;
  a.dfi = dfi; a.dfc = dfc; a.dwf = dwf; a.dbf = dbf; a.dwi = dwi; a.dbi = dbi; a.dwc = dwc; a.dbc = dbc; a.accumulate = accumulate;
  MMNN_LAUNCH(heads_bwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_linear_forward(int32_t n, int32_t d, int32_t o, const float* x, const float* w, const float* b, float* y, void* stream) {
  MMNN_REQUIRE(n > 0 && d > 0 && o > 0 && x && w && y, "linear_forward: bad arguments");
  int g = cdiv((long)n * o, 256);
  if (g > 64) g = 64;
  MMNN_LAUNCH(linear_fwd_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), n, d, o, x, w, b, y);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_linear_backward(int32_t n, int32_t d, int32_t o, const float* x, const float* w, const float* dy, float* dx, float* dw,
                         float* db, int32_t accumulate, void* stream) {
  MMNN_REQUIRE(n > 0 && d > 0 && o > 0 && x && w && dy && dw, "linear_backward: bad arguments");
  int g = cdiv(std::max((long)n * d, (long)o * d), 256);
  if (g > 64) g = 64;
  MMNN_LAUNCH(linear_bwd_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), n, d, o, x, w, dy, dx, dw, db, accumulate);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_cox_blend_loss_typed(int32_t heads, int32_t n, int32_t c, const float* preds, const void* sort_key, int32_t sort_key_dtype,
                              const void* weight, int32_t weight_dtype, const float* head_weights, float* loss, float* head_losses,
                              float* grad_preds, float* scratch, void* stream) {
  MMNN_REQUIRE(heads > 0 && n > 0 && c > 0 && heads * c <= 64 && preds && sort_key && weight && loss && head_losses && grad_preds && scratch,
               "cox_blend_loss: bad arguments (heads*targets must be <= 64)");
  MMNN_REQUIRE(sort_key_dtype >= MMNN_DT_F64 && sort_key_dtype <= MMNN_DT_U8 && weight_dtype >= MMNN_DT_F64 && weight_dtype <= MMNN_DT_U8,
               "cox_blend_loss: unknown element type (%d, %d)", sort_key_dtype, weight_dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  CoxArgs a;
  a.H = heads; a.N = n; a.C = c; a.preds = preds; a.key = sort_key; a.key_dt = sort_key_dtype;
  a.wgt = weight; a.wgt_dt = weight_dtype; a.hw = head_weights; a.head_loss = head_losses; a.loss = loss;
  a.grad = grad_preds; a.scratch = scratch; a.eps = 1e-7f;
  MMNN_LAUNCH(cox_kernel, dim3(1), dim3(256), 0, s, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

int mmnn_cox_blend_loss(int32_t heads, int32_t n, int32_t c, const float* preds, const double* sort_key, const double* weight,
                        const float* head_weights, float* loss, float* head_losses, float* grad_preds, float* scratch, void* stream) {
  return mmnn_cox_blend_loss_typed(heads, n, c, preds, sort_key, MMNN_DT_F64, weight, MMNN_DT_F64, head_weights, loss, head_losses, grad_preds,
                                   scratch, stream);
}

int mmnn_cox_blend_backward(int32_t heads, int32_t n, int32_t c, const float* grad_saved, const float* head_weights, const float* dloss,
                            const float* dheads, float* grad_preds, void* stream) {
  MMNN_REQUIRE(heads > 0 && n > 0 && c > 0 && grad_saved && grad_preds && (dloss || dheads), "cox_blend_backward: bad arguments");
  const long total = (long)heads * n * c;
  int gx = (int)((total + 255) / 256);
  if (gx > 1024) gx = 1024;
  MMNN_LAUNCH(cox_bwd_kernel, dim3(gx), dim3(256), 0, static_cast<hipStream_t>(stream), total, (long)n * c, grad_saved, head_weights, dloss, dheads,
              grad_preds);
  MMNN_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
