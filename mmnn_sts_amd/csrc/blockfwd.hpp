// Forward of ALL layers of a small-extent dense block (models/densenet.py:92-120: _DenseBlock of _DenseLayers, :46-89) in ONE persistent
// launch.  The 8^3 / 4^3 blocks are chains of memory round trips, not arithmetic: per layer two dependent kernels of 12-17 us each whose
// MFMA loops take 1.7 us.  Here the workgroups stay resident, a layer costs two grid barriers (0.8 us among the 32 CUs of one
// XCD, 3.6 us chip-wide: tools/microbench/grid_barrier.hip) and the per-launch fixed costs -- kernel arguments, descriptors, the cold
// weight fetch, the cross-workgroup K-split hand-off -- disappear: every output tile is reduced inside one workgroup and the next
// phase's weights are pulled into L2 while the barrier is waited for.
#pragma once
#include "common.hpp"

namespace mmnn {

struct BlkLayer {            // one dense layer; the table lives in device memory (plan workspace)
  int cin;                   // channels seen by norm1 / conv1 (= the layer's offset of its new channels in the concat)
  int layer_id;              // dropout stream id
  const float* w1;           // packed conv1 weights [cin][mid]            (PackJob kind 0)
  const float* w2;           // packed conv2 weights [(c*27 + tap)][growth] (PackJob kind 1)
  const float* g1; const float* b1;       // norm1 gamma / beta [cin]
  const float* g2; const float* b2;       // norm2 gamma / beta [mid]
  const float* rm1; const float* rv1;     // running statistics (eval mode)
  const float* rm2; const float* rv2;
  float* t1;                 // bottleneck tensor [N][mid][V] (kept for the backward)
  double* st_t1_sum; double* st_t1_sq;    // its batch statistics, [NREP][mid] replicas
};

struct BlockFwdArgs {
  int N, D, H, W;
  int cin0, ctot, mid, growth, nlayers;
  float* x; long x_ns;                    // concat buffer [N][ctot][V]
  double* st_x_sum; double* st_x_sq;      // statistics of the concat channels, [NREP][ctot]
  int nrep;                               // replicas in use (StatPtr::nrep)
  const BlkLayer* layers;
  unsigned* sync;                         // [0] arrival counter (zero on entry), [1] error flag (set when a barrier gave up)
  double inv_count; float eps; int training;
  uint64_t seed; float drop_p;
  int xcd_local;                          // 1: grid = 8 * nwg and only blockIdx % 8 == 0 take part (one XCD: cheap barriers)
  int nwg;                                // participating workgroups
};

// can this block run as one persistent launch?  (else: the per-layer kernels of fprop.hpp)
bool block_fwd_supported(int N, int D, int H, int W, int ctot, int mid, int growth);
int launch_block_fwd(BlockFwdArgs a, hipStream_t stream);

}  // namespace mmnn
