// Grad-CAM of the fusion model on the last conv2 of the image backbone: the arithmetic of MultiModalGradCAM
// (utils/utils.py:293-344 of the reference) after the eval-mode forward, as two launches:
//   gradcam_heat_kernel      one block: closed-form d out[0,cls] / d act (tiny GEMV through fused head -> feature layer -> GAP -> ReLU mask
//                            -> norm5 scale), channel-pooled gradient (:308-311), IN-PLACE cumulative weighting of the activations across
//                            classes (:313-314), channel mean, min-max normalisation (:316-323)
//   gradcam_upsample_kernel  F.interpolate(mode='trilinear', align_corners=False) of every class map to the input extent (:339),
//                            HBM-bound: 16-byte stores of the (D, H, W) map, source map in LDS
#include "../../include/mmnn_sts.h"
#include "common.hpp"

namespace mmnn {

struct GradcamArgs {
  int ctot, g, v, classes, F, head_ld;
  const float* h5; const float* act_in; const float* w_head; const float* w_feat; const float* gamma5; const float* rvar5;
  float eps;
  float* act; float* grads; float* heat;
};

constexpr int GC_THREADS = 1024;
constexpr int GC_MAX_G = 64, GC_MAX_CLASSES = 16;

__device__ __forceinline__ float block_reduce(float val, float* red, bool is_max, bool is_min) {
  // all 1024 threads call; red: 16 floats of LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float other = __shfl_xor(val, o, 64);
    val = is_max ? fmaxf(val, other) : (is_min ? fminf(val, other) : val + other);
  }
  __syncthreads();
  if (lane == 0) red[wave] = val;
  __syncthreads();
  float r = red[0];
  for (int k = 1; k < GC_THREADS / 64; ++k) r = is_max ? fmaxf(r, red[k]) : (is_min ? fminf(r, red[k]) : r + red[k]);
  return r;
}

__global__ void __launch_bounds__(GC_THREADS) gradcam_heat_kernel(const GradcamArgs a) {
  __shared__ float chan[GC_MAX_CLASSES * GC_MAX_G];   // d out[0,cls] / d act[c][.] where the ReLU passes
  __shared__ float cnt[GC_MAX_G];                      // voxels of channel c where norm5's output is positive
  __shared__ float red[GC_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = a.ctot - a.g;
  // ---- ReLU mask population per captured channel (features.relu, models/densenet.py:236) ----
  for (int c = wave; c < a.g; c += GC_THREADS / 64) {
    const float* hc = a.h5 + (long)(c0 + c) * a.v;
    float n = 0.f;
    for (int i = lane; i < a.v; i += 64) n += hc[i] > 0.f ? 1.f : 0.f;
    n = wave_sum(n);
    if (lane == 0) cnt[c] = n;
  }
  // ---- chan[cls][c] = sum_f Whead[cls][f] * Wfeat[f][c'] * gamma5[c'] / sqrt(rvar5[c'] + eps) / V ----
  for (int i = tid; i < a.classes * a.g; i += GC_THREADS) {
    const int cls = i / a.g, c = i % a.g, cc = c0 + c;
    float s = 0.f;
    for (int f = 0; f < a.F; ++f) s = fmaf(a.w_head[(long)cls * a.head_ld + f], a.w_feat[(long)f * a.ctot + cc], s);
    const float a5 = a.gamma5[cc] / sqrtf(a.rvar5[cc] + a.eps);
    chan[i] = s * a5 / (float)a.v;
  }
  __syncthreads();
  // ---- per class, in order: weight the activations IN PLACE by the pooled gradient (cumulative across classes) ----
  for (int cls = 0; cls < a.classes; ++cls) {
    float lo = 3.4e38f, hi = -3.4e38f;
    for (int i = tid; i < a.v; i += GC_THREADS) {
      float s = 0.f;
      for (int c = 0; c < a.g; ++c) {
        const float pooled = chan[cls * a.g + c] * cnt[c] / (float)a.v;       // mean over (0,2,3,4) of mask * chan
        const float prev = cls == 0 ? a.act_in[(long)c * a.v + i] : a.act[(long)c * a.v + i];
        const float cur = prev * pooled;
        a.act[(long)c * a.v + i] = cur;
        s += cur;
      }
      s /= (float)a.g;
      a.heat[(long)cls * a.v + i] = s;
      lo = fminf(lo, s); hi = fmaxf(hi, s);
    }
    lo = block_reduce(lo, red, false, true);
    float top = -3.4e38f;
    for (int i = tid; i < a.v; i += GC_THREADS) {     // heat -= min ; heat /= max(heat)
      const float t = a.heat[(long)cls * a.v + i] - lo;
      a.heat[(long)cls * a.v + i] = t;
      top = fmaxf(top, t);
    }
    top = block_reduce(top, red, true, false);
    for (int i = tid; i < a.v; i += GC_THREADS) a.heat[(long)cls * a.v + i] /= top;
    (void)hi;
  }
  // ---- the hooked gradient of the LAST class (`self.grads`) ----
  if (a.grads) {
    const int cls = a.classes - 1;
    for (long i = tid; i < (long)a.g * a.v; i += GC_THREADS) {
      const int c = (int)(i / a.v);
      a.grads[i] = a.h5[(long)(c0 + c) * a.v + (i % a.v)] > 0.f ? chan[cls * a.g + c] : 0.f;
    }
  }
}

struct UpsampleArgs {
  int d, h, w, D, H, W, classes;
  const float* heat;     // [classes][d*h*w]
  float* maps;           // [classes][D][H][W]
  float sd, sh, sw;      // in / out per axis
};

__device__ __forceinline__ void src_index(float scale, int dst, int in, int& i0, int& step, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;     // area_pixel_compute_source_index, align_corners = false
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  step = i0 < in - 1 ? 1 : 0;
  l1 = s - (float)i0;
}

constexpr int UP_LDS_FLOATS = 8192;

__global__ void __launch_bounds__(256) gradcam_upsample_kernel(const UpsampleArgs a) {
  __shared__ float src[UP_LDS_FLOATS];
  const int cls = blockIdx.y;
  const int v = a.d * a.h * a.w;
  const float* hm = a.heat + (long)cls * v;
  const bool in_lds = v <= UP_LDS_FLOATS;
  if (in_lds) {
    for (int i = threadIdx.x; i < v; i += 256) src[i] = hm[i];
    __syncthreads();
  }
  const long total = (long)a.D * a.H * a.W;
  float* out = a.maps + (long)cls * total;
  const bool vec = (a.W & 3) == 0 && (((uintptr_t)out & 15) == 0);
  // a block = 4 output rows (od, oh) x 64 lanes along W, 4 consecutive outputs per lane: the d / h interpolation terms are
  // computed once per thread, the stores are 16 bytes wide and contiguous across the wave
  const int ry = threadIdx.x >> 6, qx = threadIdx.x & 63;
  const int rows = a.D * a.H;
  for (int r = blockIdx.x * 4 + ry; r < rows; r += gridDim.x * 4) {
    const int od = r / a.H, oh = r - od * a.H;
    int d0, dp, h0, hp; float ld, lh;
    src_index(a.sd, od, a.d, d0, dp, ld);
    src_index(a.sh, oh, a.h, h0, hp, lh);
    const int b00 = (d0 * a.h + h0) * a.w, b01 = b00 + hp * a.w, b10 = b00 + dp * a.h * a.w, b11 = b10 + hp * a.w;
    const float h0l = 1.f - lh, d0l = 1.f - ld;
    float* orow = out + (long)r * a.W;
    for (int q = qx; q * 4 < a.W; q += 64) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ow = q * 4 + e;
        int w0, wp; float lw;
        src_index(a.sw, ow < a.W ? ow : a.W - 1, a.w, w0, wp, lw);
        float x000, x001, x010, x011, x100, x101, x110, x111;
        if (in_lds) {
          x000 = src[b00 + w0]; x001 = src[b00 + w0 + wp]; x010 = src[b01 + w0]; x011 = src[b01 + w0 + wp];
          x100 = src[b10 + w0]; x101 = src[b10 + w0 + wp]; x110 = src[b11 + w0]; x111 = src[b11 + w0 + wp];
        } else {
          x000 = hm[b00 + w0]; x001 = hm[b00 + w0 + wp]; x010 = hm[b01 + w0]; x011 = hm[b01 + w0 + wp];
          x100 = hm[b10 + w0]; x101 = hm[b10 + w0 + wp]; x110 = hm[b11 + w0]; x111 = hm[b11 + w0 + wp];
        }
        const float w0l = 1.f - lw;
        o[e] = d0l * (h0l * (w0l * x000 + lw * x001) + lh * (w0l * x010 + lw * x011)) +
               ld * (h0l * (w0l * x100 + lw * x101) + lh * (w0l * x110 + lw * x111));
      }
      if (vec) {
        *reinterpret_cast<f32x4*>(orow + q * 4) = f32x4{o[0], o[1], o[2], o[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (q * 4 + e < a.W) orow[q * 4 + e] = o[e];
      }
    }
  }
}

}  // namespace mmnn

using namespace mmnn;

extern "C" int mmnn_gradcam(const mmnn_gradcam_desc* d, const float* h5, const float* act_in, const float* w_head, const float* w_feat,
                            const float* gamma5, const float* running_var5, float* act, float* grads, float* heat, float* maps,
                            void* stream) {
  MMNN_REQUIRE(d && h5 && act_in && w_head && w_feat && gamma5 && running_var5 && act && heat && maps, "gradcam: null argument");
  MMNN_REQUIRE(d->growth >= 1 && d->growth <= GC_MAX_G && d->growth <= d->c_total, "gradcam: captured layer width %d not in 1..%d", d->growth, GC_MAX_G);
  MMNN_REQUIRE(d->classes >= 1 && d->classes <= GC_MAX_CLASSES, "gradcam: %d classes not in 1..%d", d->classes, GC_MAX_CLASSES);
  MMNN_REQUIRE(d->d >= 1 && d->h >= 1 && d->w >= 1 && d->out_d >= 1 && d->out_h >= 1 && d->out_w >= 1, "gradcam: bad extent");
  MMNN_REQUIRE(d->features >= 1 && d->head_ld >= d->features, "gradcam: fused head narrower than the image features");
  const long v = (long)d->d * d->h * d->w;
  MMNN_REQUIRE(v * d->c_total < (1l << 31), "gradcam: activation too large for 32-bit indices");
  hipStream_t st = static_cast<hipStream_t>(stream);
  GradcamArgs a;
  a.ctot = d->c_total; a.g = d->growth; a.v = (int)v; a.classes = d->classes; a.F = d->features; a.head_ld = d->head_ld;
  a.h5 = h5; a.act_in = act_in; a.w_head = w_head; a.w_feat = w_feat; a.gamma5 = gamma5; a.rvar5 = running_var5; a.eps = d->eps;
  a.act = act; a.grads = grads; a.heat = heat;
  MMNN_LAUNCH(gradcam_heat_kernel, dim3(1), dim3(GC_THREADS), 0, st, a);
  MMNN_HIP(hipGetLastError());
  UpsampleArgs u;
  u.d = d->d; u.h = d->h; u.w = d->w; u.D = d->out_d; u.H = d->out_h; u.W = d->out_w; u.classes = d->classes;
  u.heat = heat; u.maps = maps;
  u.sd = (float)d->d / (float)d->out_d; u.sh = (float)d->h / (float)d->out_h; u.sw = (float)d->w / (float)d->out_w;
  MMNN_REQUIRE((long)d->out_d * d->out_h < (1l << 31), "gradcam: output extent too large");
  long gx = ((long)d->out_d * d->out_h + 3) / 4;
  if (gx > 65536) gx = 65536;
  MMNN_LAUNCH(gradcam_upsample_kernel, dim3((unsigned)gx, (unsigned)d->classes), dim3(256), 0, st, u);
  MMNN_HIP(hipGetLastError());
  return 0;
}
