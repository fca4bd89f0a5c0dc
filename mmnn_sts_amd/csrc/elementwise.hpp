// HBM-bound helper kernels of the DenseNet path: transition pooling, final norm, BN running statistics,
// weight packing, gradient finalisation.
#pragma once
#include "../../include/mmnn_sts.h"
#include "common.hpp"

namespace mmnn {

// ---- transition (models/densenet.py:145-148): A_p = avgpool2(ReLU(BN(X)))  (pool commuted in front of the 1x1x1 conv)
struct PoolFwdArgs {
  int N, C, D, H, W;          // input extent; output extent is floor(D/2) ...
  const float* x; long x_ns;  // block buffer [N][Ctot][V], channels [0, C)
  BnFwd bn;
  float* out;                 // [N][C][Vo]
};
int launch_bnrelu_avgpool(const PoolFwdArgs& a, hipStream_t stream);

// ---- y = a_c * x + b_c (norm5, models/densenet.py:221-224)
struct BnApplyArgs {
  int N, C, V;
  const float* x; long x_ns;
  BnFwd bn;
  float* out;                 // [N][C][V]
};
int launch_bn_apply(const BnApplyArgs& a, hipStream_t stream);

// ---- first contribution to a block buffer's G:  z = mask * upstream,  G = gamma*z,  dgamma/dbeta and S1/S2 sums.
//   mode 0: upstream = dy[n][c][v]                 (norm5 backward, no ReLU)
//   mode 1: upstream = dy[n][c][parent(v)] / 8,    ReLU mask from a*x+b > 0   (transition: un-pool + ReLU + BN)
struct ConsumerBwdArgs {
  int N, C, D, H, W;          // extent of x / G
  int mode;
  const float* dy;            // mode 0: [N][C][V];  mode 1: [N][C][Vo] pooled grid
  const float* x; long x_ns;
  BnFwd bn;
  float* g; long g_ns;
  double* dgamma; double* dbeta;   // [NREP][C]
  StatPtr s_acc;              // S1 / S2 of the block buffer
  int nrep;                   // replicas the fp64 atomics are spread over (0: NREP)
};
int launch_consumer_bwd(const ConsumerBwdArgs& a, hipStream_t stream);

// ---- introspection (tests, GradCAM): mask[n][c][v] = (a_c*x + b_c > 0), the exact ReLU decision the kernels take
struct MaskArgs {
  int N, C, V;
  const float* x; long x_ns;
  BnFwd bn;
  unsigned char* out;         // [N][C][V]
};
int launch_relu_mask(const MaskArgs& a, hipStream_t stream);

// ---- running statistics: rm = (1-mom)*rm + mom*mean ; rv = (1-mom)*rv + mom*var*n/(n-1)   for a table of BN layers
struct RunStatJob {
  const double* sum; const double* sq; int stride; int off; int C;
  float* rmean; float* rvar;
  double count;
};
int launch_running_stats(const RunStatJob* jobs_dev, int njobs, float momentum, long long* nbt, hipStream_t stream);   // nbt: optional [njobs] step counters, +1 each

// ---- weight packing (once per forward): dst[k][m] layouts consumed by fprop / stem kernels
struct PackJob {
  const float* src; float* dst;
  int kind;      // 0: conv1 fwd  dst[c][m] = w[m][c]
                 // 1: conv2 fwd  dst[(c*27+t)][m] = w[m][c][t]
                 // 2: conv2 dgrad dst[(m*27+t)][c] = w[m][c][26-t]
                 // 3: stem, even Cin: dst[kd][c*49 + kh*7+kw][64] ; 4: stem, odd Cin: dst[kd][c*56 + kh*8 + kw][64]
  int M, C;      // weight is [M][C][taps]
  long count;    // number of dst elements
};
int launch_pack(const PackJob* jobs_dev, int njobs, long max_count, hipStream_t stream);

// ---- gradient finalisation: one launch turns every partial result into the flat fp32 gradient buffer
struct GradJob {
  int kind;      // 0: slab sum, dst[i] = sum_s slab[s*stride + i]                       (conv1 / transition conv)
                 // 1: conv2 slab [s][tap][m][c] -> dst[m][c][tap]
                 // 2: stem slab [s][c][m][352]  -> dst[m][c][343]
                 // 3: fp64 replicas -> fp32   dst[i] = sum_r src[r*stride + off + i]       (BN gamma / beta)
  const void* src; long stride; int nsplit; int off;
  int M, C;
  long dst_off; long count;
};
int launch_finalize(const GradJob* jobs_dev, int njobs, long max_count, float* grad, int accumulate, hipStream_t stream);

// ---- SGD with momentum / Nesterov / weight decay over a flat buffer (torch.optim.SGD semantics, main.py:410-413)
int launch_sgd(float* p, const float* g, float* buf, long n, float lr, float momentum, float weight_decay, int nesterov, int first_step,
               hipStream_t stream);

// the same update / a flat gather-scatter over a list of small tensors, one launch each (table passed as a kernel argument)
int launch_sgd_multi(const mmnn_tensor_ref* refs, int n, float* buf, float lr, float momentum, float weight_decay, int nesterov, hipStream_t stream);
int launch_multi_copy(const mmnn_tensor_ref* refs, int n, float* flat, int scatter, hipStream_t stream);

}  // namespace mmnn
