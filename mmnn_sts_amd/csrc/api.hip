// extern "C" surface of libmmnn_sts.so (declared in include/mmnn_sts.h).
#include "../../include/mmnn_sts.h"

#include <new>

#include "densenet.hpp"

using namespace mmnn;

extern "C" {

int mmnn_version(void) { return 300; }
const char* mmnn_last_error(void) { return last_error(); }

void* mmnn_densenet_plan_create(const mmnn_densenet_config* cfg, int32_t n, int32_t d, int32_t h, int32_t w) {
  if (!cfg) { set_error("plan_create: null config"); return nullptr; }
  if (cfg->num_blocks < 1 || cfg->num_blocks > MAX_BLOCKS) { set_error("plan_create: num_blocks %d out of range", cfg->num_blocks); return nullptr; }
  NetCfg c;
  c.in_channels = cfg->in_channels; c.init_features = cfg->init_features; c.growth = cfg->growth_rate; c.bn_size = cfg->bn_size;
  c.nblocks = cfg->num_blocks;
  for (int i = 0; i < MAX_BLOCKS; ++i) c.block_layers[i] = i < cfg->num_blocks ? cfg->block_config[i] : 0;
  c.eps = cfg->eps; c.momentum = cfg->momentum; c.dropout_p = cfg->dropout_prob;
  Plan* p = new (std::nothrow) Plan();
  if (!p) { set_error("plan_create: out of host memory"); return nullptr; }
  if (plan_build(*p, c, n, d, h, w) != 0) { delete p; return nullptr; }
  return p;
}

void mmnn_densenet_plan_destroy(void* plan) {
  if (!plan) return;
  Plan* p = static_cast<Plan*>(plan);
  plan_free(*p);
  delete p;
}

int64_t mmnn_densenet_param_count(const void* plan) { return plan ? static_cast<const Plan*>(plan)->n_params : -1; }
int64_t mmnn_densenet_runstat_count(const void* plan) { return plan ? static_cast<const Plan*>(plan)->n_runstats : -1; }
int64_t mmnn_densenet_workspace_bytes(const void* plan) { return plan ? (int64_t) static_cast<const Plan*>(plan)->ws_bytes : -1; }

int mmnn_densenet_out_shape(const void* plan, int32_t* c, int32_t* d, int32_t* h, int32_t* w) {
  MMNN_REQUIRE(plan && c && d && h && w, "out_shape: null argument");
  const Plan* p = static_cast<const Plan*>(plan);
  const int b = p->cfg.nblocks - 1;
  *c = p->ctot_b[b]; *d = p->Db[b]; *h = p->Hb[b]; *w = p->Wb[b];
  return 0;
}

int mmnn_densenet_forward(void* plan, const float* params, float* runstats, const float* x, void* workspace, float* out,
                          int32_t training, uint64_t seed, void* stream) {
  MMNN_REQUIRE(plan, "forward: null plan");
  return plan_forward(*static_cast<Plan*>(plan), params, runstats, x, static_cast<char*>(workspace), out, training, seed,
                      static_cast<hipStream_t>(stream));
}

int mmnn_densenet_backward(void* plan, const float* params, const float* x, void* workspace, const float* grad_out,
                           float* grad_params, int32_t accumulate, uint64_t seed, void* stream) {
  MMNN_REQUIRE(plan, "backward: null plan");
  return plan_backward(*static_cast<Plan*>(plan), params, x, static_cast<char*>(workspace), grad_out, grad_params, accumulate, seed,
                       static_cast<hipStream_t>(stream));
}

int mmnn_densenet_backward_range(void* plan, const float* params, const float* x, void* workspace, const float* grad_out,
                                 float* grad_params, int32_t accumulate, uint64_t seed, int32_t hi_block, int32_t lo_block, void* stream) {
  MMNN_REQUIRE(plan, "backward_range: null plan");
  return plan_backward_range(*static_cast<Plan*>(plan), params, x, static_cast<char*>(workspace), grad_out, grad_params, accumulate, seed,
                             hi_block, lo_block, static_cast<hipStream_t>(stream));
}

int mmnn_densenet_block_param_range(const void* plan, int32_t block, int64_t* begin, int64_t* end) {
  MMNN_REQUIRE(plan && begin && end, "block_param_range: null argument");
  long b = 0, e = 0;
  const int rc = plan_block_param_range(*static_cast<const Plan*>(plan), block, &b, &e);
  *begin = b; *end = e;
  return rc;
}

int mmnn_densenet_relu_mask(void* plan, const float* params, void* workspace, int32_t kind, int32_t block, int32_t layer,
                            uint8_t* out, void* stream) {
  MMNN_REQUIRE(plan && out, "relu_mask: null argument");
  return plan_relu_mask(*static_cast<Plan*>(plan), params, static_cast<char*>(workspace), kind, block, layer, out,
                        static_cast<hipStream_t>(stream));
}

int mmnn_densenet_set_timer(void* plan, int32_t kernel_class, int32_t block) {
  MMNN_REQUIRE(plan, "set_timer: null plan");
  return plan_set_timer(*static_cast<Plan*>(plan), kernel_class, block);
}

int mmnn_densenet_read_timer(void* plan, double* total_ms, int64_t* launches) {
  return mmnn_densenet_read_timer_class(plan, 0, -1, total_ms, launches);
}

int mmnn_densenet_read_timer_class(void* plan, int32_t kernel_class, int32_t block, double* total_ms, int64_t* launches) {
  MMNN_REQUIRE(plan, "read_timer: null plan");
  long n = 0;
  int rc = plan_read_timer(*static_cast<Plan*>(plan), kernel_class, block, total_ms, &n);
  if (launches) *launches = n;
  return rc;
}

int mmnn_densenet_set_option(void* plan, const char* name, int64_t value) {
  MMNN_REQUIRE(plan && name, "set_option: null argument");
  return plan_set_option(*static_cast<Plan*>(plan), name, (long)value);
}

int mmnn_densenet_set_batch_counters(void* plan, int64_t* num_batches_tracked, int32_t bn_count) {
  MMNN_REQUIRE(plan, "set_batch_counters: null plan");
  Plan* p = static_cast<Plan*>(plan);
  MMNN_REQUIRE(num_batches_tracked == nullptr || bn_count == p->n_bn, "set_batch_counters: the backbone has %d batch norms, got %d counters", p->n_bn, bn_count);
  p->nbt = reinterpret_cast<long long*>(num_batches_tracked); p->nbt_count = num_batches_tracked ? bn_count : 0;
  return 0;
}

int64_t mmnn_densenet_ws_offset(const void* plan, const char* name, int32_t i, int32_t j) {
  if (!plan || !name) return -1;
  return plan_ws_offset(*static_cast<const Plan*>(plan), name, i, j);
}

int mmnn_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                  int32_t nesterov, int32_t first_step, void* stream) {
  return launch_sgd(params, grads, momentum_buf, n, lr, momentum, weight_decay, nesterov, first_step, static_cast<hipStream_t>(stream));
}

int mmnn_sgd_step_multi(const mmnn_tensor_ref* refs, int32_t n, float* momentum_buf, float lr, float momentum, float weight_decay,
                        int32_t nesterov, void* stream) {
  return launch_sgd_multi(refs, n, momentum_buf, lr, momentum, weight_decay, nesterov, static_cast<hipStream_t>(stream));
}

int mmnn_multi_copy(const mmnn_tensor_ref* refs, int32_t n, float* flat, int32_t scatter, void* stream) {
  return launch_multi_copy(refs, n, flat, scatter, static_cast<hipStream_t>(stream));
}

}  // extern "C"
