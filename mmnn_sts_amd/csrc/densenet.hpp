// Host-side plan of the 3-D DenseNet backbone (models/densenet.py:196-231): workspace layout, parameter layout and
// the launch sequences of forward / backward.  PyTorch owns every device buffer; the plan owns only small host tables.
#pragma once
#include <string>
#include <vector>

#include "common.hpp"
#include "blockfwd.hpp"
#include "elementwise.hpp"

namespace mmnn {

constexpr int MAX_BLOCKS = 8;

struct NetCfg {
  int in_channels, init_features, growth, bn_size, nblocks;
  int block_layers[MAX_BLOCKS];
  float eps, momentum, dropout_p;
};

struct LayerOff {          // offsets (floats) into the flat parameter / running-stat buffers
  long n1w, n1b, c1, n2w, n2b, c2;
  long r1m, r1v, r2m, r2v;
  int cin;                 // channels seen by norm1 / conv1
};
struct TransOff { long nw, nb, cw; long rm, rv; int cin, cout; };

struct Plan {
  NetCfg cfg;
  int N, D, H, W;
  int D0, H0, W0;                         // conv0 output extent
  int Db[MAX_BLOCKS], Hb[MAX_BLOCKS], Wb[MAX_BLOCKS], Vb[MAX_BLOCKS];
  int cin_b[MAX_BLOCKS], ctot_b[MAX_BLOCKS];
  int nrep_b[MAX_BLOCKS];                 // statistic replicas used by the kernels of a block (StatPtr::nrep)
  int mid;                                // bn_size * growth
  // parameters
  long p_conv0, p_n0w, p_n0b, r_n0m, r_n0v, p_n5w, p_n5b, r_n5m, r_n5v;
  std::vector<std::vector<LayerOff>> layers;
  std::vector<TransOff> trans;
  long n_params, n_runstats;
  int n_bn;
  // workspace (byte offsets)
  size_t ws_bytes;
  size_t o_conv0, o_idx, o_x[MAX_BLOCKS], o_g[MAX_BLOCKS], o_ap[MAX_BLOCKS], o_dz2[MAX_BLOCKS], o_dap, o_dz0;
  std::vector<std::vector<size_t>> o_t1;
  // fp64 statistics: one zero-filled region for forward sums, one for backward sums
  size_t o_fstat, fstat_bytes, o_bstat, bstat_bytes;
  size_t o_st_conv0, o_st_x[MAX_BLOCKS];          // inside fstat: [2][NREP][C]
  std::vector<std::vector<size_t>> o_st_t1;
  size_t o_s_x[MAX_BLOCKS];                       // inside bstat: S1/S2 [2][NREP][ctot]
  size_t o_dg_n0, o_dg_n5;                        // dgamma/dbeta [2][NREP][C]
  std::vector<std::vector<size_t>> o_dg_n1, o_dg_n2;
  std::vector<size_t> o_dg_tr;
  // packed weights
  size_t o_pk_conv0;
  std::vector<std::vector<size_t>> o_pk_c1, o_pk_c2f, o_pk_c2b;
  std::vector<size_t> o_pk_tr;
  // weight-gradient slabs
  size_t o_sl_conv0; int ns_conv0;
  std::vector<std::vector<size_t>> o_sl_c1, o_sl_c2;
  std::vector<std::vector<int>> ns_c1, ns_c2;
  std::vector<size_t> o_sl_tr; std::vector<int> ns_tr;
  // device job tables (inside the workspace) + pinned host staging
  size_t o_kz_part, o_kz_cnt;                       // cross-block K-split scratch (fprop.hpp)
  size_t o_jobs_run, o_jobs_pack, o_jobs_grad, o_jobs_blk;   // o_jobs_blk: BlkLayer table of the persistent block forward
  size_t o_blk_sync = 0;                            // 2 words per dense block: arrival counter, error flag (zeroed with kz_cnt)
  bool persist_b[MAX_BLOCKS] = {false};             // block CAN run as ONE persistent forward launch (blockfwd.hpp)
  bool persistent = false;                          // option "persistent_forward" (experiment, default off): use the persistent launch where supported
  int n_layers = 0;
  int wg_group[MAX_BLOCKS] = {0};                  // dense layers per weight-gradient launch, by block
  size_t o_wg_table = 0;                            // device tables of the weight-gradient arguments (batched launches)
  void* wg_pinned = nullptr; std::vector<char> wg_shadow; bool wg_uploaded = false;
  void* host_jobs = nullptr; size_t host_jobs_bytes = 0;
  // cached identity of the buffers the tables were built for
  const float* tab_params = nullptr; float* tab_run = nullptr; char* tab_ws = nullptr;
  int n_run_jobs = 0, n_pack_jobs = 0, n_grad_jobs = 0; long max_pack = 0, max_grad = 0;
  int gj_begin[MAX_BLOCKS + 1] = {0};   // gradient jobs of block b: [gj_begin[b], gj_begin[b+1])  (jobs 0..2: the stem)
  long gj_max[MAX_BLOCKS] = {0};        // largest job of the block
  // backward runs the weight-gradient kernels on a second stream beside the data-gradient chain (host objects only)
  hipStream_t side = nullptr, side2 = nullptr; bool side_tried = false;   // conv2 / conv1 weight-gradient streams
  std::vector<hipEvent_t> sync_ev; size_t sync_used = 0;
  // optional live timing of one kernel class with HIP events (bench.py roofline leg)
  unsigned timer_mask = 0; int timer_block = -1;   // bit k: class k of TimerKind is timed; block < 0: every block
  std::vector<hipEvent_t> timer_ev;                // start/stop pairs recorded since the last read
  std::vector<int> timer_tag;                      // class of each pair
  size_t timer_used = 0;
  double timer_ms[16 * MAX_BLOCKS] = {0}; long timer_count[16 * MAX_BLOCKS] = {0};   // [class][block]
  bool single_stream = false;                      // option "single_stream": backward on the caller's stream only (overrides side_streams)
  int side_streams = 0;                            // option "side_streams": 0 (default), 1 or 2 streams for the weight-gradient kernels
  long params_version = 0, packed_version = 0; const float* packed_params = nullptr; const char* packed_ws = nullptr;   // option "params_version"
  long pack_launches = 0;
  long long* nbt = nullptr; int nbt_count = 0;     // mmnn_densenet_set_batch_counters
  bool no_kz = false;                              // option "no_kz": no cross-workgroup K-split (tests)
  int bwd_next = -1;                               // plan_backward_range: the block the next partial call must start at
  unsigned long long* trace_base = nullptr; mutable int trace_seq = 0; int trace_slots = 0;   // developer aid: per-launch phase stamps
};

enum TimerKind { T_NONE = 0, T_CONV2_FWD = 1, T_CONV2_DGRAD = 2, T_CONV2_WGRAD = 3, T_CONV1_FWD = 4, T_CONV1_DGRAD = 5,
                 T_CONV1_WGRAD = 6, T_STEM_CONV = 7, T_STEM_WGRAD = 8, T_BLOCK_FWD = 9, T_COUNT = 10 };
int plan_set_timer(Plan& p, int kind, int block);                              // kind -1: every class
int plan_read_timer(Plan& p, int kind, int block, double* total_ms, long* count);   // kind 0 / block < 0: all of them together
int plan_set_option(Plan& p, const char* name, long value);

int plan_build(Plan& p, const NetCfg& cfg, int N, int D, int H, int W);
void plan_free(Plan& p);
int plan_forward(Plan& p, const float* params, float* runstats, const float* x, char* ws, float* out, int training,
                 uint64_t seed, hipStream_t stream);
int plan_backward(Plan& p, const float* params, const float* x, char* ws, const float* grad_out, float* grad_params,
                  int accumulate, uint64_t seed, hipStream_t stream);
int plan_backward_range(Plan& p, const float* params, const float* x, char* ws, const float* grad_out, float* grad_params,
                        int accumulate, uint64_t seed, int hi, int lo, hipStream_t stream);
int plan_block_param_range(const Plan& p, int block, long* begin, long* end);   // block -1: the stem
long plan_ws_offset(const Plan& p, const char* name, int i, int j);
// ReLU decisions of one BN+ReLU site after a training forward: kind 0 = relu0 (stem), 1 = layer relu1, 2 = layer relu2,
// 3 = transition relu.  out: uint8 [N][C][V] of that site.
int plan_relu_mask(Plan& p, const float* params, char* ws, int kind, int b, int l, unsigned char* out, hipStream_t stream);

}  // namespace mmnn
