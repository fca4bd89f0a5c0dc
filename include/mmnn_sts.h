/* C-ABI of libmmnn_sts.so -- the MI355X (gfx950) native compute path of the MMNN_STS multimodal-fusion
 * training step.  Plain C types only: device pointers, sizes, a HIP stream passed as void*.
 *
 * The reference (DigITs-AIML/MMNN_STS) has no FFI of its own: its hot path is the PyTorch module tree
 *   models/densenet.py:151-271  (DenseNet.backbone / .features),  models/mlp.py:7-63,
 *   models/multimodal.py:9-90,  losses/GradientBlender.py:181-205,  losses/losses.py:6-9, utils/utils.py:24-29
 * driven by main.py:460-469 (`model(inputs)`, `computeLoss`, `loss.backward()`).  Each entry point below names the
 * reference code whose arithmetic it replaces; the Python mirror in mmnn_sts_amd/ binds them with ctypes
 * (see INTEGRATION.md for the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - every function returns 0 on success; on failure a non-zero status and mmnn_last_error() (thread-local text).
 *     1 = invalid argument / shape (-> ValueError), 2 = HIP runtime error (-> RuntimeError).  Never aborts.
 *   - the caller (PyTorch) owns every device buffer; the library never allocates, frees or retains device memory.
 *     Workspace sizes come from pure query functions.  All tensors are contiguous fp32, NCDHW.
 *   - re-entrant: no global mutable state besides the thread-local error text; a plan handle must not be used from
 *     two threads at once (autograd calls backward from another thread than forward, sequentially: fine).
 */
#ifndef MMNN_STS_H
#define MMNN_STS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int mmnn_version(void);
const char* mmnn_last_error(void);

/* ---- 3-D DenseNet backbone: models/densenet.py:196-231 (conv0 .. norm5) ------------------------------------------ */
typedef struct {
  int32_t in_channels;      /* models/densenet.py:176 */
  int32_t init_features;    /* :179  (<= 64)  */
  int32_t growth_rate;      /* :180  (<= 32)  */
  int32_t bn_size;          /* :182 */
  int32_t num_blocks;       /* len(block_config), :181 */
  int32_t block_config[8];
  float eps;                /* BatchNorm eps (1e-5) */
  float momentum;           /* BatchNorm momentum (0.1) */
  float dropout_prob;       /* :185, nn.Dropout3d after conv2 (:84-85) */
} mmnn_densenet_config;

/* A plan fixes (config, batch, input extent); it owns host-side tables only.  NULL on error. */
void* mmnn_densenet_plan_create(const mmnn_densenet_config* cfg, int32_t n, int32_t d, int32_t h, int32_t w);
void mmnn_densenet_plan_destroy(void* plan);
int64_t mmnn_densenet_param_count(const void* plan);      /* floats in the flat parameter buffer, PyTorch
                                                             named_parameters() order of `backbone` */
int64_t mmnn_densenet_runstat_count(const void* plan);    /* floats in the flat running-stat buffer:
                                                             (running_mean, running_var) per BN in module order */
int64_t mmnn_densenet_workspace_bytes(const void* plan);
int mmnn_densenet_out_shape(const void* plan, int32_t* c, int32_t* d, int32_t* h, int32_t* w);

/* backbone(x): replaces DenseNet.backbone.forward (models/densenet.py:267-268).  training != 0: batch statistics,
 * running-stat update (momentum), channel dropout keyed by `seed`; training == 0: running statistics, no dropout.
 * out: [n][C][d'][h'][w'] = norm5 output. */
int mmnn_densenet_forward(void* plan, const float* params, float* runstats, const float* x, void* workspace, float* out,
                          int32_t training, uint64_t seed, void* stream);
/* autograd adjoint of the above wrt every backbone parameter (main.py:469); needs the workspace of the matching
 * training forward untouched.  grad_params: flat, same layout as params; accumulate != 0 adds into it. */
int mmnn_densenet_backward(void* plan, const float* params, const float* x, void* workspace, const float* grad_out,
                           float* grad_params, int32_t accumulate, uint64_t seed, void* stream);
/* byte offset of a named workspace region (tests / GradCAM): "x","g","t1","conv0","st_x",... ; -1 if unknown */
int64_t mmnn_densenet_ws_offset(const void* plan, const char* name, int32_t i, int32_t j);

#ifdef __cplusplus
}
#endif
#endif
