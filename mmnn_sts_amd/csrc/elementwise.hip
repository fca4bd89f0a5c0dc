#include "elementwise.hpp"

#include <algorithm>

namespace mmnn {

// block-wide sum of two floats -> thread 0 (256 threads)
__device__ __forceinline__ void block_sum2(float& s0, float& s1, float (*red)[4]) {
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = s0; red[1][wave] = s1; }
  __syncthreads();
  s0 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  s1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bnrelu_avgpool_kernel(const PoolFwdArgs a) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int Do = a.D / 2, Ho = a.H / 2, Wo = a.W / 2;
  const int Vo = Do * Ho * Wo, V = a.D * a.H * a.W;
  float ca, cb, mu, rs;
  bn_fwd_coef(a.bn, c, ca, cb, mu, rs);
  const float* xc = a.x + (long)n * a.x_ns + (long)c * V;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < Vo; p += gridDim.x * 256) {
    const int wo = p % Wo, ho = (p / Wo) % Ho, d_o = p / (Wo * Ho);
    float s = 0.f;
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const float* row = xc + ((long)(2 * d_o + kd) * a.H + 2 * ho + kh) * a.W + 2 * wo;
        s += fmaxf(fmaf(ca, row[0], cb), 0.f) + fmaxf(fmaf(ca, row[1], cb), 0.f);
      }
    a.out[((long)n * a.C + c) * Vo + p] = s * 0.125f;
  }
}

int launch_bnrelu_avgpool(const PoolFwdArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.C > 0 && a.D >= 2 && a.H >= 2 && a.W >= 2, "avgpool: extent too small (%d,%d,%d)", a.D, a.H, a.W);
  MMNN_REQUIRE(a.N <= 65535 && a.C <= 65535, "avgpool: grid out of range");
  const int Vo = (a.D / 2) * (a.H / 2) * (a.W / 2);
  MMNN_LAUNCH(bnrelu_avgpool_kernel, dim3(cdiv(Vo, 256), a.C, a.N), dim3(256), 0, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_apply_kernel(const BnApplyArgs a) {
  const int c = blockIdx.y, n = blockIdx.z;
  float ca, cb, mu, rs;
  bn_fwd_coef(a.bn, c, ca, cb, mu, rs);
  const float* xc = a.x + (long)n * a.x_ns + (long)c * a.V;
  float* oc = a.out + ((long)n * a.C + c) * a.V;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < a.V; v += gridDim.x * 256) oc[v] = fmaf(ca, xc[v], cb);
}

int launch_bn_apply(const BnApplyArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.C > 0 && a.V > 0 && a.N <= 65535 && a.C <= 65535, "bn_apply: bad extent");
  int gx = cdiv(a.V, 256);
  if (gx > 64) gx = 64;
  MMNN_LAUNCH(bn_apply_kernel, dim3(gx, a.C, a.N), dim3(256), 0, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) relu_mask_kernel(const MaskArgs a) {
  const int c = blockIdx.y, n = blockIdx.z;
  float ca, cb, mu, rs;
  bn_fwd_coef(a.bn, c, ca, cb, mu, rs);
  const float* xc = a.x + (long)n * a.x_ns + (long)c * a.V;
  unsigned char* oc = a.out + ((long)n * a.C + c) * a.V;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < a.V; v += gridDim.x * 256) oc[v] = fmaf(ca, xc[v], cb) > 0.f ? 1 : 0;
}

int launch_relu_mask(const MaskArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.C > 0 && a.V > 0 && a.N <= 65535 && a.C <= 65535 && a.out, "relu_mask: bad arguments");
  int gx = cdiv(a.V, 256);
  if (gx > 64) gx = 64;
  MMNN_LAUNCH(relu_mask_kernel, dim3(gx, a.C, a.N), dim3(256), 0, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) consumer_bwd_kernel(const ConsumerBwdArgs a) {
  __shared__ float red[2][4];
  const int c = blockIdx.y, n = blockIdx.z;
  const int V = a.D * a.H * a.W;
  const int Do = a.D / 2, Ho = a.H / 2, Wo = a.W / 2;
  float ca, cb, mu, rs;
  bn_fwd_coef(a.bn, c, ca, cb, mu, rs);
  const float gam = a.bn.gamma[c];
  const float* xc = a.x + (long)n * a.x_ns + (long)c * V;
  float* gc = a.g + (long)n * a.g_ns + (long)c * V;
  const float* dyc = a.dy + ((long)n * a.C + c) * (a.mode == 0 ? V : Do * Ho * Wo);
  float s0 = 0.f, s1 = 0.f;
  if (a.mode == 1 && (a.W & 3) == 0 && ((((uintptr_t)xc | (uintptr_t)gc) & 15) == 0) && (Wo & 1) == 0 &&
      (long)V * (a.W >> 2) < (4l << 32) && (long)a.D * a.H * a.H < (1l << 32)) {
    // four consecutive voxels of a row per thread (r03): 16-byte accesses, and ONE (row, quad) decomposition by multiply-high per four
    // elements -- per element the run-time divisions by W and H were ~60 vector instructions, and the 16.8 M elements of block 1's
    // transition made this an ALU-bound kernel (4 launches, 98 us per step).  Per-row sums in the same order as the scalar loop would
    // give them is not required: the totals go into fp64 atomics.
    const int wq_n = a.W >> 2, rows = a.D * a.H, items = rows * wq_n;
    const unsigned mg_q = wq_n > 1 ? (unsigned)((0x100000000ull + wq_n - 1) / wq_n) : 0u;      // exact for item * wq_n < 2^32
    const unsigned mg_h = a.H > 1 ? (unsigned)((0x100000000ull + a.H - 1) / a.H) : 0u;
    for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
      const int row = mg_q ? (int)__umulhi((unsigned)it, mg_q) : it;
      const int w = (it - row * wq_n) << 2;
      const int d = mg_h ? (int)__umulhi((unsigned)row, mg_h) : row;
      const int h = row - d * a.H;
      const int v = row * a.W + w;
      const f32x4 x = *reinterpret_cast<const f32x4*>(xc + v);
      const int pw = w >> 1, ph = h >> 1, pd = d >> 1;
      float dy0 = 0.f, dy1 = 0.f;
      if (pw < Wo && ph < Ho && pd < Do) {           // pw even, Wo even: pw + 1 < Wo as well
        const float* q = dyc + ((long)pd * Ho + ph) * Wo + pw;
        dy0 = 0.125f * q[0]; dy1 = 0.125f * q[1];
      }
      f32x4 g;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaf(ca, x[e], cb) > 0.f ? (e < 2 ? dy0 : dy1) : 0.f;
        g[e] = gam * z;
        s0 += z;
        s1 += z * (x[e] - mu) * rs;
      }
      *reinterpret_cast<f32x4*>(gc + v) = g;
    }
  } else
  for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
    const float x = xc[v];
    float z;
    if (a.mode == 0) {
      z = dyc[v];
    } else {
      const int w = v % a.W, h = (v / a.W) % a.H, d = v / (a.W * a.H);
      const int pw = w >> 1, ph = h >> 1, pd = d >> 1;
      z = 0.f;
      if (pw < Wo && ph < Ho && pd < Do && fmaf(ca, x, cb) > 0.f) z = 0.125f * dyc[((long)pd * Ho + ph) * Wo + pw];
    }
    gc[v] = gam * z;
    s0 += z;
    s1 += z * (x - mu) * rs;
  }
  block_sum2(s0, s1, red);
  if (threadIdx.x == 0) {
    const int rep = blockIdx.x & ((a.nrep > 0 ? a.nrep : NREP) - 1);
    atomicAdd(a.dbeta + (long)rep * a.C + c, (double)s0);
    atomicAdd(a.dgamma + (long)rep * a.C + c, (double)s1);
    atomicAdd(a.s_acc.sum + (long)rep * a.s_acc.stride + a.s_acc.off + c, (double)gam * s0);
    atomicAdd(a.s_acc.sq + (long)rep * a.s_acc.stride + a.s_acc.off + c, (double)gam * s1);
  }
}

int launch_consumer_bwd(const ConsumerBwdArgs& a, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.C > 0 && a.N <= 65535 && a.C <= 65535, "consumer_bwd: bad extent");
  MMNN_REQUIRE(a.mode == 0 || a.mode == 1, "consumer_bwd: bad mode");
  const int V = a.D * a.H * a.W;
  // every block starts with the coefficient chain of its channel (statistics loads + fp64 arithmetic, ~3 us): few, long blocks -- 8192
  // voxels each, eight 16-byte items per thread (r03: 1024 voxels per block meant 16 384 blocks and eight rounds of that chain at 32^3)
  int gx = cdiv(V, 8192);
  if (gx > 64) gx = 64;
  MMNN_LAUNCH(consumer_bwd_kernel, dim3(gx, a.C, a.N), dim3(256), 0, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) running_stats_kernel(const RunStatJob* jobs, float momentum, long long* nbt) {
  const RunStatJob j = jobs[blockIdx.x];
  if (nbt && threadIdx.x == 0) nbt[blockIdx.x] += 1;      // num_batches_tracked of BN number blockIdx.x (jobs are in module order)
  for (int c = threadIdx.x; c < j.C; c += 256) {
    const double mean = stat_total(j.sum, j.stride, j.off + c) / j.count;
    double var = stat_total(j.sq, j.stride, j.off + c) / j.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double unb = j.count > 1.0 ? var * j.count / (j.count - 1.0) : var;
    j.rmean[c] = (float)((1.0 - momentum) * (double)j.rmean[c] + momentum * mean);
    j.rvar[c] = (float)((1.0 - momentum) * (double)j.rvar[c] + momentum * unb);
  }
}

int launch_running_stats(const RunStatJob* jobs_dev, int njobs, float momentum, long long* nbt, hipStream_t stream) {
  if (njobs <= 0) return 0;
  MMNN_LAUNCH(running_stats_kernel, dim3(njobs), dim3(256), 0, stream, jobs_dev, momentum, nbt);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_kernel(const PackJob* jobs) {
  const PackJob j = jobs[blockIdx.y];
  if (j.kind <= 2) {
    // Kinds 0-2 are (batched) transpositions  dst[b][j'][i] = src[b][i][j]  with  j' = j  or  J-1-j:
    //   0: [M][C] -> [C][M]      1: [M][C*27] -> [C*27][M]      2: M times [C][27] -> [27 (taps reversed)][C].
    // 32 x 32 tiles through LDS, so that both the 45 MB read and the 56 MB written per step are contiguous 128-byte rows (r03: the
    // element-per-thread form read with a stride of C or 27 floats between neighbouring lanes: 59 us per step, 1.7 TB/s).
    __shared__ float tile[32][33];
    const int B = j.kind == 2 ? j.M : 1, I = j.kind == 2 ? j.C : j.M, J = j.kind == 0 ? j.C : (j.kind == 1 ? j.C * 27 : 27);
    const bool flip = j.kind == 2;
    const int ti_n = (I + 31) / 32, tj_n = (J + 31) / 32;
    const long ntiles = (long)B * ti_n * tj_n;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                    // 32 x 8 threads, four rows each
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int tj = (int)(t % tj_n), ti = (int)((t / tj_n) % ti_n), b = (int)(t / ((long)tj_n * ti_n));
      const float* src = j.src + (long)b * I * J;
      float* dst = j.dst + (long)b * I * J;
      __syncthreads();                                                         // previous tile's readers are done
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * 32 + ty + 8 * r, jj = tj * 32 + tx;
        tile[ty + 8 * r][tx] = (i < I && jj < J) ? src[(long)i * J + jj] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = tj * 32 + ty + 8 * r, i = ti * 32 + tx;
        if (i < I && jj < J) dst[(long)(flip ? J - 1 - jj : jj) * I + i] = tile[tx][ty + 8 * r];
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < j.count; i += (long)gridDim.x * 256) {
    float v = 0.f;
    if (j.kind == 0) {                      // dst[c][m] = w[m][c]
      const int m = (int)(i % j.M), c = (int)(i / j.M);
      v = j.src[(long)m * j.C + c];
    } else if (j.kind == 1) {               // dst[c*27+t][m] = w[m][c][t]
      const int m = (int)(i % j.M);
      const long k = i / j.M;
      v = j.src[(long)m * j.C * 27 + k];
    } else if (j.kind == 2) {               // dst[m*27+t][c] = w[m][c][26-t]
      const int c = (int)(i % j.C);
      const long k = i / j.C;
      const int t = (int)(k % 27), m = (int)(k / 27);
      v = j.src[((long)m * j.C + c) * 27 + 26 - t];
    } else if (j.kind == 3) {               // dst[kd][c*49 + kh*7 + kw][64] = w[m][c][kd][kh][kw]
      const int m = (int)(i % 64);
      long k = i / 64;
      const int r = (int)(k % (j.C * 49)), kd = (int)(k / (j.C * 49));
      const int c = r / 49, hw = r % 49;
      if (m < j.M) v = j.src[(((long)m * j.C + c) * 7 + kd) * 49 + hw];
    } else {                                // dst[kd][c*56 + kh*8 + kw][64], kw == 7 is zero padding
      const int m = (int)(i % 64);
      long k = i / 64;
      const int r = (int)(k % (j.C * 56)), kd = (int)(k / (j.C * 56));
      const int c = r / 56, kh = (r % 56) / 8, kw = r % 8;
      if (m < j.M && kw < 7) v = j.src[((((long)m * j.C + c) * 7 + kd) * 7 + kh) * 7 + kw];
    }
    j.dst[i] = v;
  }
}

int launch_pack(const PackJob* jobs_dev, int njobs, long max_count, hipStream_t stream) {
  if (njobs <= 0) return 0;
  MMNN_REQUIRE(njobs <= 65535, "pack: too many jobs");
  int gx = cdiv(max_count, 256 * 4);
  if (gx < 1) gx = 1;
  if (gx > 256) gx = 256;
  MMNN_LAUNCH(pack_kernel, dim3(gx, njobs), dim3(256), 0, stream, jobs_dev);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) finalize_kernel(const GradJob* jobs, float* grad, int accumulate) {
  const GradJob j = jobs[blockIdx.y];
  float* dst = grad + j.dst_off;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < j.count; i += (long)gridDim.x * 256) {
    float v;
    long di = i;                              // destination index (kind 1 walks the SOURCE order: see below)
    if (j.kind == 3) {
      const double* s = static_cast<const double*>(j.src);
      double t = 0.0;
      for (int r = 0; r < NREP; ++r) t += s[(long)r * j.stride + j.off + i];
      v = (float)t;
    } else {
      const float* s = static_cast<const float*>(j.src);
      long si;
      if (j.kind == 0) {
        si = i;
      } else if (j.kind == 1) {             // dst[m][c][tap] <- slab[tap][m][c]
        // threads follow the slabs' order (tap-major, (m, c) contiguous): the nsplit reads of an element are the 0.5 GB this kernel
        // moves and are now coalesced; the one write per element is the strided side (r03: in destination order a wave's 64 reads
        // touched 27 different lines, 12 useful bytes each)
        const long mc_n = (long)j.M * j.C;
        const int tap = (int)(i / mc_n);
        const long mc = i - (long)tap * mc_n;
        si = i;
        di = mc * 27 + tap;
      } else {                              // dst[m][c][tap343] <- slab[c][m][352]
        const int tap = (int)(i % 343);
        const long mc = i / 343;
        const int c = (int)(mc % j.C), m = (int)(mc / j.C);
        si = ((long)c * j.M + m) * 352 + tap;
      }
      // eight slabs' loads in flight per round trip, added in slab order as before (a plain loop over the run-time count waits for
      // every load in turn: 0.5 GB of slabs took 179 us, 2.8 TB/s)
      float t = 0.f;
      int sp = 0;
      for (; sp + 8 <= j.nsplit; sp += 8) {
        float u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = s[(long)(sp + k) * j.stride + si];
#pragma unroll
        for (int k = 0; k < 8; ++k) t += u[k];
      }
      for (; sp < j.nsplit; ++sp) t += s[(long)sp * j.stride + si];
      v = t;
    }
    dst[di] = accumulate ? dst[di] + v : v;
  }
}

int launch_finalize(const GradJob* jobs_dev, int njobs, long max_count, float* grad, int accumulate, hipStream_t stream) {
  if (njobs <= 0) return 0;
  MMNN_REQUIRE(njobs <= 65535 && grad, "finalize: bad arguments");
  int gx = cdiv(max_count, 256 * 2);
  if (gx < 1) gx = 1;
  if (gx > 128) gx = 128;
  MMNN_LAUNCH(finalize_kernel, dim3(gx, njobs), dim3(256), 0, stream, jobs_dev, grad, accumulate);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sgd_kernel(float* p, const float* g, float* buf, long n, float lr, float mom, float wd, int nesterov, int first) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) == 0)) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 bv = first ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<f32x4*>(buf + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = fmaf(wd, pv[e], gv[e]);
        const float b = first ? d : fmaf(mom, bv[e], d);
        bv[e] = b;
        d = nesterov ? fmaf(mom, b, d) : b;
        pv[e] = fmaf(-lr, d, pv[e]);
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(buf + i) = bv;
    } else {
      for (long j = i; j < n && j < i + 4; ++j) {
        float d = fmaf(wd, p[j], g[j]);
        const float b = first ? d : fmaf(mom, buf[j], d);
        buf[j] = b;
        d = nesterov ? fmaf(mom, b, d) : b;
        p[j] = fmaf(-lr, d, p[j]);
      }
    }
  }
}

int launch_sgd(float* p, const float* g, float* buf, long n, float lr, float momentum, float weight_decay, int nesterov, int first_step,
               hipStream_t stream) {
  MMNN_REQUIRE(p && g && buf && n > 0, "sgd: bad arguments");
  int gx = cdiv(n, 1024);
  if (gx > 2048) gx = 2048;
  MMNN_LAUNCH(sgd_kernel, dim3(gx), dim3(256), 0, stream, p, g, buf, n, lr, momentum, weight_decay, nesterov, first_step);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// ---- the same update over a LIST of small tensors (the ~30 parameter tensors outside the backbone) in one launch: the table travels
// as a kernel argument (no upload), blockIdx.y = tensor -------------------------------------------------------------------------------
struct MultiTable {
  int n;
  float* p[MMNN_MULTI_MAX]; const float* g[MMNN_MULTI_MAX]; long count[MMNN_MULTI_MAX]; long off[MMNN_MULTI_MAX]; int first[MMNN_MULTI_MAX];
};

__global__ void __launch_bounds__(256) sgd_multi_kernel(const MultiTable t, float* buf, float lr, float mom, float wd, int nesterov) {
  const int k = blockIdx.y;
  float* p = t.p[k]; const float* g = t.g[k]; float* b = buf + t.off[k];
  const bool first = t.first[k] != 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < t.count[k]; i += (long)gridDim.x * 256) {
    float d = fmaf(wd, p[i], g[i]);
    const float m = first ? d : fmaf(mom, b[i], d);
    b[i] = m;
    d = nesterov ? fmaf(mom, m, d) : m;
    p[i] = fmaf(-lr, d, p[i]);
  }
}

int launch_sgd_multi(const mmnn_tensor_ref* refs, int n, float* buf, float lr, float momentum, float weight_decay, int nesterov, hipStream_t stream) {
  MMNN_REQUIRE(refs && buf && n >= 1 && n <= MMNN_MULTI_MAX, "sgd_multi: 1..%d tensors per call, got %d", MMNN_MULTI_MAX, n);
  MultiTable t;
  t.n = n;
  long most = 0;
  for (int i = 0; i < n; ++i) {
    MMNN_REQUIRE(refs[i].param && refs[i].grad && refs[i].count > 0 && refs[i].flat_offset >= 0, "sgd_multi: bad tensor %d", i);
    t.p[i] = refs[i].param; t.g[i] = refs[i].grad; t.count[i] = refs[i].count; t.off[i] = refs[i].flat_offset; t.first[i] = refs[i].first_step;
    most = std::max(most, (long)refs[i].count);
  }
  int gx = cdiv(most, 256);
  if (gx > 64) gx = 64;
  MMNN_LAUNCH(sgd_multi_kernel, dim3(gx, n), dim3(256), 0, stream, t, buf, lr, momentum, weight_decay, nesterov);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// gather (scatter == 0: flat[off_k + i] = grad_k[i]) / scatter (grad_k[i] = flat[off_k + i]) of a list of tensors: the coalesced bucket
// of the small gradients for the data-parallel all-reduce, without torch.cat / per-tensor copies
__global__ void __launch_bounds__(256) multi_copy_kernel(const MultiTable t, float* flat, int scatter) {
  const int k = blockIdx.y;
  float* g = const_cast<float*>(t.g[k]); float* f = flat + t.off[k];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < t.count[k]; i += (long)gridDim.x * 256) {
    if (scatter) g[i] = f[i]; else f[i] = g[i];
  }
}

int launch_multi_copy(const mmnn_tensor_ref* refs, int n, float* flat, int scatter, hipStream_t stream) {
  MMNN_REQUIRE(refs && flat && n >= 1 && n <= MMNN_MULTI_MAX, "multi_copy: 1..%d tensors per call, got %d", MMNN_MULTI_MAX, n);
  MultiTable t;
  t.n = n;
  long most = 0;
  for (int i = 0; i < n; ++i) {
    MMNN_REQUIRE(refs[i].grad && refs[i].count > 0 && refs[i].flat_offset >= 0, "multi_copy: bad tensor %d", i);
    t.p[i] = nullptr; t.g[i] = refs[i].grad; t.count[i] = refs[i].count; t.off[i] = refs[i].flat_offset; t.first[i] = 0;
    most = std::max(most, (long)refs[i].count);
  }
  int gx = cdiv(most, 256);
  if (gx > 64) gx = 64;
  MMNN_LAUNCH(multi_copy_kernel, dim3(gx, n), dim3(256), 0, stream, t, flat, scatter);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// ---- debugging aid: poison the LDS of every CU (see MMNN_LAUNCH in common.hpp) -----------------------------------------
__global__ void __launch_bounds__(256) poison_lds_kernel(int words) {
  extern __shared__ unsigned poison[];
  volatile unsigned* q = poison;   // volatile: the stores have no reader in this kernel
  for (int i = threadIdx.x; i < words; i += 256) q[i] = 0xFFFFFFFFu;   // quiet NaN as fp32, and as either half of an fp64
}

void debug_poison_lds(hipStream_t stream) {
  static const int on = [] { const char* e = getenv("MMNN_POISON_LDS"); return (e && e[0] == '1') ? 1 : 0; }();
  if (!on) return;
  constexpr int BYTES = 160 * 1024;   // the whole LDS of a CU: one block per CU at a time
  static const bool ready =
      hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES) == hipSuccess;
  if (!ready) return;
  hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), BYTES, stream, BYTES / 4);
}

}  // namespace mmnn
