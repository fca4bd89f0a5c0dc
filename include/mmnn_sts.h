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
 *   - threading: the library keeps no mutable state of its own besides the thread-local error text and per-device
 *     "kernel attribute already set" flags (idempotent; written with the value every writer would write).  A PLAN, however, is
 *     a stateful object: it owns host-side job tables (one pinned staging buffer), two lazily created side streams and a pool
 *     of events for its backward, and optional timers.  One plan must therefore not be used from two threads at once, and it
 *     belongs to the device that was current when its first forward ran.  Sequential use from different threads is fine
 *     (autograd calls backward from another thread than forward).  Different plans are independent of each other.
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
/* The same backward cut at dense-block boundaries, for data parallelism (there is no counterpart upstream: main.py:336 is single
 * process; the accumulate-then-step rule it implements is main.py:403-407,478-481): runs dense blocks hi_block, hi_block-1, ...,
 * lo_block (0-based).  Calls must walk the blocks downwards without gaps, the first one starting at the last block; the call with
 * lo_block == 0 also runs the stem.  When a call has completed in stream order, the gradients of every parameter of the blocks it
 * covered (and of the transition / norm5 behind each, and of the stem when lo_block == 0) are final in grad_params, so their SUM
 * all-reduce can run while the next call's kernels execute.  mmnn_densenet_backward == one call with (num_blocks-1, 0). */
int mmnn_densenet_backward_range(void* plan, const float* params, const float* x, void* workspace, const float* grad_out,
                                 float* grad_params, int32_t accumulate, uint64_t seed, int32_t hi_block, int32_t lo_block, void* stream);
/* [begin, end) of the flat parameter / gradient buffer owned by dense block `block` (its layers + the transition or norm5 that
 * follows it); block -1: the stem (conv0, norm0).  The ranges tile [0, param_count) in the order stem, block 0, block 1, ... */
int mmnn_densenet_block_param_range(const void* plan, int32_t block, int64_t* begin, int64_t* end);
/* introspection: the ReLU decisions (a*x+b > 0, uint8 [n][C][V]) of one BN+ReLU site of the last training forward.
 * kind 0: relu0 (models/densenet.py:201); 1: denselayer relu1 (:77); 2: relu2 (:81); 3: transition relu (:146).
 * Used by the gradient parity tests (ReLU is not differentiable at 0: a reference must take the same branch). */
int mmnn_densenet_relu_mask(void* plan, const float* params, void* workspace, int32_t kind, int32_t block, int32_t layer,
                            uint8_t* out, void* stream);
/* measurement: time every launch of one kernel class (-1: of every class) with HIP events recorded on the launch stream.
 * kernel_class 0 none, 1 conv2 fwd, 2 conv2 dgrad, 3 conv2 wgrad, 4 conv1 fwd, 5 conv1 dgrad, 6 conv1 wgrad, 7 stem conv,
 * 8 stem wgrad; block >= 0 restricts to one dense block (0-based).  read_timer synchronises the recorded events and returns
 * the accumulated device time and launch count since set_timer (read_timer: all recorded classes and blocks together;
 * read_timer_class: one class (0: all) of one dense block (< 0: all)). */
int mmnn_densenet_set_timer(void* plan, int32_t kernel_class, int32_t block);
int mmnn_densenet_read_timer(void* plan, double* total_ms, int64_t* launches);
int mmnn_densenet_read_timer_class(void* plan, int32_t kernel_class, int32_t block, double* total_ms, int64_t* launches);
/* plan options.  "persistent_forward" (0/1, default 0; experiment): run the forward of a small-extent dense block (8^3 / 4^3 voxels) as ONE
 * resident launch with grid barriers between the layers (csrc/blockfwd.hip) instead of two kernels per layer -- same results up to summation
 * order; measured slower in round 3, kept for the record and under test.  "no_kz" (0/1): never split the channel axis of a small-extent convolution over several workgroups (the tests' reference
 * for the cross-workgroup hand-off).  "side_streams" (0, 1 or 2; default 0): run the weight-gradient kernels of the backward on that many side streams
 * beside the data-gradient chain instead of on the caller's stream -- same results.  "single_stream" (0/1): force 0 side streams
 * (un-overlapped kernel durations for profiling).  "params_version" (any non-zero
 * number the caller changes whenever it changed a parameter; 0 = unknown, the default): the forward re-packs the weights only
 * when the version, the parameter buffer or the workspace differs from the last packed one. */
int mmnn_densenet_set_option(void* plan, const char* name, int64_t value);
/* nn.BatchNorm3d.num_batches_tracked of every BN of the backbone (module order, int64 [runstat_count / 2 channels ... one per BN]): when set
 * (non-NULL device pointer to `bn_count` int64 values), every training forward adds 1 to each of them in its running-statistics kernel;
 * NULL (the default) leaves the counters to the caller. */
int mmnn_densenet_set_batch_counters(void* plan, int64_t* num_batches_tracked, int32_t bn_count);
/* byte offset of a named workspace region (tests / GradCAM): "x","g","t1","conv0","st_x",... ; -1 if unknown */
int64_t mmnn_densenet_ws_offset(const void* plan, const char* name, int32_t i, int32_t j);

/* ---- DenseNet.features: ReLU -> AdaptiveAvgPool3d(1) -> flatten -> Linear -> Dropout (models/densenet.py:234-247) ---- */
/* h [n][c][v] (norm5 output), w [f][c], b [f] -> out [n][f]; pooled [n][c] is saved for the backward. */
int mmnn_gap_linear_forward(int32_t n, int32_t c, int32_t v, int32_t f, const float* h, const float* w, const float* b,
                            float* pooled, float* out, float dropout_prob, uint64_t seed, int32_t training, void* stream);
int mmnn_gap_linear_backward(int32_t n, int32_t c, int32_t v, int32_t f, const float* h, const float* w, const float* pooled,
                             const float* dout, float* dw, float* db, float* dh, float dropout_prob, uint64_t seed,
                             int32_t training, int32_t accumulate, void* stream);

/* ---- [Linear -> BatchNorm1d -> ReLU / Dropout1d] stacks: MLP.backbone, MLP.features (models/mlp.py:19-51) ---------- */
#define MMNN_MLP_MAX_LAYERS 8
typedef struct {
  int32_t n;                                  /* batch rows */
  int32_t num_layers;
  int32_t in_dim[MMNN_MLP_MAX_LAYERS];
  int32_t out_dim[MMNN_MLP_MAX_LAYERS];
  int32_t relu_first[MMNN_MLP_MAX_LAYERS];    /* 1: dense-bn-relu-drop (mlp.py:21-24); 0: dense-bn-drop-relu (:25-49) */
  float dropout_prob;                         /* nn.Dropout1d on a 2-D input: whole ROWS are dropped */
  float eps, momentum;
  uint64_t seed;
  int32_t training;
  int32_t first_layer_id;                     /* dropout stream id of layer 0 of this stack */
} mmnn_mlp_desc;
typedef struct {
  const float* weight[MMNN_MLP_MAX_LAYERS];   /* [out][in] */
  const float* bias[MMNN_MLP_MAX_LAYERS];
  const float* gamma[MMNN_MLP_MAX_LAYERS];
  const float* beta[MMNN_MLP_MAX_LAYERS];
  float* running_mean[MMNN_MLP_MAX_LAYERS];
  float* running_var[MMNN_MLP_MAX_LAYERS];
  float* grad_weight[MMNN_MLP_MAX_LAYERS];    /* backward only */
  float* grad_bias[MMNN_MLP_MAX_LAYERS];
  float* grad_gamma[MMNN_MLP_MAX_LAYERS];
  float* grad_beta[MMNN_MLP_MAX_LAYERS];
  int64_t* num_batches_tracked[MMNN_MLP_MAX_LAYERS];   /* optional (NULL: not maintained): nn.BatchNorm1d's step counter, +1 per training forward */
} mmnn_mlp_params;
int64_t mmnn_mlp_saved_floats(const mmnn_mlp_desc* d);      /* size of `saved` */
int mmnn_mlp_forward(const mmnn_mlp_desc* d, const mmnn_mlp_params* p, const float* x, float* out, float* saved, void* stream);
/* scratch: 2 * n * max(dim) floats; dx may be NULL */
int mmnn_mlp_backward(const mmnn_mlp_desc* d, const mmnn_mlp_params* p, const float* x, const float* saved, const float* dy,
                      float* dx, float* scratch, int32_t accumulate, void* stream);

/* ---- fusion heads (models/multimodal.py:62-77): out[0] = cat(fi,fc) Wf^T + bf; blend: out[1] = image head, out[2] = clinical */
int mmnn_fusion_heads_forward(int32_t n, int32_t f, int32_t c, int32_t blend, const float* fi, const float* fc, const float* wf,
                              const float* bf, const float* wi, const float* bi, const float* wc, const float* bc, float* out,
                              void* stream);
int mmnn_fusion_heads_backward(int32_t n, int32_t f, int32_t c, int32_t blend, const float* fi, const float* fc, const float* wf,
                               const float* wi, const float* wc, const float* dout, float* dfi, float* dfc, float* dwf, float* dbf,
                               float* dwi, float* dbi, float* dwc, float* dbc, int32_t accumulate, void* stream);

/* ---- small dense layer y = x W^T + b: class_layers.out (models/densenet.py:250-256), MLP.output_head (mlp.py:53-57) - */
int mmnn_linear_forward(int32_t n, int32_t d, int32_t o, const float* x, const float* w, const float* b, float* y, void* stream);
int mmnn_linear_backward(int32_t n, int32_t d, int32_t o, const float* x, const float* w, const float* dy, float* dx, float* dw,
                         float* db, int32_t accumulate, void* stream);

/* ---- Cox partial likelihood summed over targets and blended over heads (losses/losses.py:6-9 -> pycox CoxPHLoss,
 * utils/utils.py:24-29, losses/GradientBlender.py:197-205).  preds [heads][n][c]; sort_key / weight [n][c] fp64: pycox's
 * `durations` / `events` arguments (the reference passes events / durations there, in that order; its datasets build them as
 * int64 or float32 tensors, data/ImageDatasets.py:462 -- both are exact in fp64, fractional durations included).  Stable
 * descending sort.
 * Writes loss = sum_h head_weights[h] * head_losses[h] (head_weights NULL: all 1), head_losses[h] = sum_c cox(h, c), and
 * grad_preds = d loss / d preds.  scratch: 4 * n floats. */
int mmnn_cox_blend_loss(int32_t heads, int32_t n, int32_t c, const float* preds, const double* sort_key, const double* weight,
                        const float* head_weights, float* loss, float* head_losses, float* grad_preds, float* scratch, void* stream);

/* The same with sort_key / weight in the element type the caller's tensors already have (the reference's datasets produce int64 and
 * float32, data/ImageDatasets.py:462): no conversion pass.  Every supported type is exact in the kernel's fp64 arithmetic. */
#define MMNN_DT_F64 0
#define MMNN_DT_F32 1
#define MMNN_DT_I64 2
#define MMNN_DT_I32 3
#define MMNN_DT_U8 4
int mmnn_cox_blend_loss_typed(int32_t heads, int32_t n, int32_t c, const float* preds, const void* sort_key, int32_t sort_key_dtype,
                              const void* weight, int32_t weight_dtype, const float* head_weights, float* loss, float* head_losses,
                              float* grad_preds, float* scratch, void* stream);
/* autograd adjoint of (loss, head_losses) wrt preds (main.py:469): grad_preds = grad_saved * dloss[0] + grad_saved[h] * dheads[h] /
 * head_weights[h], where grad_saved is what mmnn_cox_blend_loss wrote.  dloss (1 float) / dheads ([heads]) are device pointers, either may
 * be NULL (that output took no part in the graph). */
int mmnn_cox_blend_backward(int32_t heads, int32_t n, int32_t c, const float* grad_saved, const float* head_weights, const float* dloss,
                            const float* dheads, float* grad_preds, void* stream);

/* ---- element-wise binary cross entropy on logits with per-class positive weights: nn.BCEWithLogitsLoss(pos_weight=...,
 * reduction='none') of the classification trainer (main.py:147-153), the loss behind `criterion` (utils/utils.py:20-22) and
 * GradientBlender.computeLossClassification (losses/GradientBlender.py:150-179).  logits / targets / loss / dloss_dlogits hold
 * `total` floats whose fastest axis is the class axis of length c; pos_weight [c] or NULL; dloss_dlogits may be NULL. */
int mmnn_bce_logits(int64_t total, int32_t c, const float* logits, const float* targets, const float* pos_weight, float* loss,
                    float* dloss_dlogits, void* stream);

/* ---- 3-D ResNet-18 variant (models/resnet.py:5-227).  Generic direct kernels: the net is 8 / 16 channels wide. ---------------- */
typedef struct {
  int32_t n, c_in, d, h, w;      /* input  [n][c_in][d][h][w]  */
  int32_t c_out;                 /* weight [c_out][c_in][kernel...], no bias (Conv3DSimple :95-112, BasicStem :9-11, downsample :176-178) */
  int32_t kernel[3], stride[3], padding[3];
} mmnn_conv3d_desc;
int mmnn_conv3d_out_shape(const mmnn_conv3d_desc* d, int32_t* od, int32_t* oh, int32_t* ow);
int mmnn_conv3d_forward(const mmnn_conv3d_desc* d, const float* x, const float* w, float* y, void* stream);
/* autograd adjoints (main.py:469): dx [n][c_in][d][h][w];  dw [c_out][c_in][kernel...] (accumulate != 0: added to);
 * workspace for the deterministic partial-sum slabs: mmnn_conv3d_wgrad_workspace_bytes(d) bytes */
int mmnn_conv3d_backward_data(const mmnn_conv3d_desc* d, const float* dy, const float* w, float* dx, void* stream);
int64_t mmnn_conv3d_wgrad_workspace_bytes(const mmnn_conv3d_desc* d);
int mmnn_conv3d_backward_weight(const mmnn_conv3d_desc* d, const float* x, const float* dy, float* dw, void* workspace, int32_t accumulate,
                                void* stream);
/* nn.BatchNorm3d [+ residual add] [+ ReLU] [+ element-wise nn.Dropout]: the tail of BasicStem (:12-13), of both halves of a BasicBlock
 * (:74-77, :89-91: `out += residual; out = relu(out)`), of a downsample branch (:179) and the dropout after each stage (:159-166).
 * x / residual / out [n][c][v]; save [2][c] (mean, rstd) and stat_ws (2*c doubles) are caller-provided scratch; training != 0: batch
 * statistics + running-stat update (unbiased variance), training == 0: running statistics, no dropout. */
int mmnn_bn3d_forward(int32_t n, int32_t c, int64_t v, const float* x, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int32_t training, int32_t relu, const float* residual,
                      float dropout_prob, uint64_t seed, float* out, float* save, double* stat_ws, void* stream);
/* adjoint: dx wrt the BN input, dresidual (may be NULL) wrt the added tensor, dgamma / dbeta [c] (overwritten) */
int mmnn_bn3d_backward(int32_t n, int32_t c, int64_t v, const float* x, const float* out, const float* dout, const float* gamma,
                       const float* save, int32_t training, int32_t relu, float dropout_prob, uint64_t seed, float* dx, float* dresidual,
                       float* dgamma, float* dbeta, double* stat_ws, void* stream);
/* AdaptiveAvgPool3d(1) -> flatten -> Linear(c, o) -> sigmoid (models/resnet.py:152-167); pooled [n][c] is kept for the backward */
int mmnn_gap_fc_sigmoid_forward(int32_t n, int32_t c, int64_t v, int32_t o, const float* x, const float* w, const float* b, float* pooled,
                                float* y, void* stream);
int mmnn_gap_fc_sigmoid_backward(int32_t n, int32_t c, int64_t v, int32_t o, const float* w, const float* pooled, const float* y, const float* dy,
                                 float* dw, float* db, float* dx, void* stream);

/* ---- optimizer step over a flat buffer: torch.optim.SGD(momentum, nesterov, weight_decay) as main.py:410-413 uses it.
 * d = g + wd*p; buf = first_step ? d : momentum*buf + d; p -= lr * (nesterov ? d + momentum*buf : buf) */
int mmnn_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                  int32_t nesterov, int32_t first_step, void* stream);

/* The same update for a LIST of small tensors (the ~30 parameter tensors of the MLP, the feature layer and the heads) in ONE launch.
 * momentum_buf is one flat buffer; tensor i owns [flat_offset, flat_offset + count).  first_step is per tensor (torch creates a
 * parameter's momentum buffer at its first step WITH a gradient; tensors without gradient are simply not listed). */
#define MMNN_MULTI_MAX 64
typedef struct {
  float* param;            /* may be NULL for mmnn_multi_copy */
  const float* grad;
  int64_t count;
  int64_t flat_offset;     /* position of this tensor inside the flat momentum / bucket buffer (floats) */
  int32_t first_step;
  int32_t reserved;
} mmnn_tensor_ref;
int mmnn_sgd_step_multi(const mmnn_tensor_ref* refs, int32_t n, float* momentum_buf, float lr, float momentum, float weight_decay,
                        int32_t nesterov, void* stream);
/* scatter == 0: flat[flat_offset + i] = grad[i] (gather the small gradients into ONE all-reduce bucket); scatter != 0: the way back */
int mmnn_multi_copy(const mmnn_tensor_ref* refs, int32_t n, float* flat, int32_t scatter, void* stream);

/* ---- Grad-CAM of the fusion model on the last Conv3d of the image backbone: MultiModalGradCAM.forward after its model forward
 * (utils/utils.py:293-344), batch size 1 (:334).  Per class, in order: d out[0,cls] / d act in closed form (fused head -> feature_layer ->
 * average pool -> ReLU mask -> eval-mode norm5 scale, restricted to the captured layer = the LAST `growth` channels of the concat),
 * channel-pooled gradient (:308-311), the activations weighted by it IN PLACE and therefore cumulatively across classes (:313-314),
 * channel mean, min-max normalisation (:316-323), trilinear up-sampling to the input extent (:339, align_corners = False). */
typedef struct {
  int32_t c_total;          /* channels of the norm5 output */
  int32_t growth;           /* channels of the captured layer (<= 64) */
  int32_t d, h, w;          /* extent of the captured activations */
  int32_t classes;          /* rows of the fused head = Grad-CAM targets (<= 16) */
  int32_t features;         /* width of DenseNet.features' output = the image half of the fused head's input */
  int32_t head_ld;          /* row stride (floats) of the fused head weight (2 * features in the fusion model) */
  int32_t out_d, out_h, out_w;   /* extent of the attention maps = of the input volume */
  float eps;                /* norm5 eps */
} mmnn_gradcam_desc;
/* h5 [c_total][v]: eval-mode norm5 output; act_in [growth][v]: output of the last conv2 (the last channels of the block buffer);
 * w_head [classes][head_ld]; w_feat [features][c_total]; gamma5 / running_var5 [c_total].
 * Writes act [growth][v] (the weighted activations after the last class = `.features`), grads [growth][v] (the gradient of the last class
 * = `.grads`; may be NULL), heat [classes][v] (the normalised low-resolution maps) and maps [classes][out_d][out_h][out_w]. */
int mmnn_gradcam(const mmnn_gradcam_desc* d, const float* h5, const float* act_in, const float* w_head, const float* w_feat,
                 const float* gamma5, const float* running_var5, float* act, float* grads, float* heat, float* maps, void* stream);

/* ---- measurement aid (bench.py): MHz the chip sustains under a chip-wide v_mfma_f32_32x32x2_f32 load (one wave per SIMD, every CU), from
 * the known cycle count of an MFMA loop and HIP events around it.  Synchronises the stream.  scratch: >= 1 float of device memory. */
int mmnn_measure_mfma_clock(double* mhz, float* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif
