// Launch plan of the DenseNet backbone forward / backward (see densenet.hpp).
#include "densenet.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>

#include "fprop.hpp"
#include "stem.hpp"
#include "wgrad.hpp"

namespace mmnn {

static size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
// cross-block K-split scratch of the plan's workspace: <= 512 blocks x (up to 4) 32x32 partial tiles, + per-tile counters
constexpr size_t KZ_PART_BYTES = (size_t)512 * 4 * 1024 * sizeof(float);
constexpr unsigned KZ_CNT_ENTRIES = 4096;
constexpr size_t BLK_SYNC_BYTES = 256;     // 2 words per dense block (MAX_BLOCKS = 8), zeroed together with the K-split counters

namespace {
struct Carver {
  size_t cur = 0;
  size_t take(size_t bytes) {
    size_t o = cur;
    cur = align_up(cur + bytes);
    return o;
  }
};
}  // namespace

int plan_build(Plan& p, const NetCfg& cfg, int N, int D, int H, int W) {
  MMNN_REQUIRE(cfg.nblocks >= 1 && cfg.nblocks <= MAX_BLOCKS, "plan: 1..%d dense blocks supported, got %d", MAX_BLOCKS, cfg.nblocks);
  MMNN_REQUIRE(cfg.in_channels >= 1 && cfg.in_channels <= 4, "plan: in_channels must be 1..4, got %d", cfg.in_channels);
  MMNN_REQUIRE(cfg.init_features >= 1 && cfg.init_features <= 64, "plan: init_features must be <= 64, got %d", cfg.init_features);
  MMNN_REQUIRE(cfg.growth >= 1 && cfg.growth <= 32, "plan: growth_rate must be <= 32, got %d", cfg.growth);
  MMNN_REQUIRE(cfg.bn_size >= 1 && cfg.dropout_p >= 0.f && cfg.dropout_p < 1.f, "plan: bad bn_size / dropout");
  MMNN_REQUIRE(N >= 1 && D >= 1 && H >= 1 && W >= 1, "plan: bad input extent");
  p.cfg = cfg; p.N = N; p.D = D; p.H = H; p.W = W;
  p.mid = cfg.bn_size * cfg.growth;
  p.D0 = (D - 1) / 2 + 1; p.H0 = (H - 1) / 2 + 1; p.W0 = (W - 1) / 2 + 1;
  int d = (p.D0 - 1) / 2 + 1, h = (p.H0 - 1) / 2 + 1, w = (p.W0 - 1) / 2 + 1;
  int c = cfg.init_features;
  for (int b = 0; b < cfg.nblocks; ++b) {
    MMNN_REQUIRE(cfg.block_layers[b] >= 1, "plan: empty dense block %d", b);
    MMNN_REQUIRE(d >= 1 && h >= 1 && w >= 1, "plan: input too small, block %d has no voxels", b + 1);
    p.Db[b] = d; p.Hb[b] = h; p.Wb[b] = w; p.Vb[b] = d * h * w;
    p.cin_b[b] = c;
    // replicas of the fp64 statistics (StatPtr::nrep): block 0 receives the stem's statistics (always NREP replicas)
    p.nrep_b[b] = (b == 0 || (long)N * d * h * w >= 32768) ? NREP : ((long)N * d * h * w >= 4096 ? 2 : 1);
    p.ctot_b[b] = c + cfg.block_layers[b] * cfg.growth;
    c = p.ctot_b[b];
    if (b != cfg.nblocks - 1) {
      MMNN_REQUIRE(c % 2 == 0, "plan: transition halves an odd channel count");
      c /= 2; d /= 2; h /= 2; w /= 2;
    }
  }
  // ---- parameter layout (PyTorch named_parameters order of `backbone`) ----
  long po = 0, ro = 0;
  int nbn = 0;
  auto bn = [&](int ch, long& w_, long& b_, long& m_, long& v_) { w_ = po; po += ch; b_ = po; po += ch; m_ = ro; ro += ch; v_ = ro; ro += ch; ++nbn; };
  p.p_conv0 = po; po += (long)cfg.init_features * cfg.in_channels * 343;
  bn(cfg.init_features, p.p_n0w, p.p_n0b, p.r_n0m, p.r_n0v);
  p.layers.assign(cfg.nblocks, {});
  p.trans.clear();
  for (int b = 0; b < cfg.nblocks; ++b) {
    int ci = p.cin_b[b];
    for (int l = 0; l < cfg.block_layers[b]; ++l) {
      LayerOff lo;
      lo.cin = ci;
      bn(ci, lo.n1w, lo.n1b, lo.r1m, lo.r1v);
      lo.c1 = po; po += (long)p.mid * ci;
      bn(p.mid, lo.n2w, lo.n2b, lo.r2m, lo.r2v);
      lo.c2 = po; po += (long)cfg.growth * p.mid * 27;
      p.layers[b].push_back(lo);
      ci += cfg.growth;
    }
    if (b != cfg.nblocks - 1) {
      TransOff t;
      t.cin = ci; t.cout = ci / 2;
      bn(ci, t.nw, t.nb, t.rm, t.rv);
      t.cw = po; po += (long)t.cout * ci;
      p.trans.push_back(t);
    } else {
      bn(ci, p.p_n5w, p.p_n5b, p.r_n5m, p.r_n5v);
    }
  }
  p.n_params = po; p.n_runstats = ro; p.n_bn = nbn;

  // ---- workspace ----
  Carver cv;
  const int nb = cfg.nblocks;
  const size_t F = sizeof(float);
  const long V0 = (long)p.D0 * p.H0 * p.W0;
  p.o_conv0 = cv.take((size_t)N * cfg.init_features * V0 * F);
  p.o_idx = cv.take((size_t)N * cfg.init_features * p.Vb[0]);
  p.o_t1.assign(nb, {});
  size_t dap = 0;
  for (int b = 0; b < nb; ++b) {
    p.o_x[b] = cv.take((size_t)N * p.ctot_b[b] * p.Vb[b] * F);
    p.o_g[b] = cv.take((size_t)N * p.ctot_b[b] * p.Vb[b] * F);
    for (int l = 0; l < cfg.block_layers[b]; ++l) p.o_t1[b].push_back(cv.take((size_t)N * p.mid * p.Vb[b] * F));
    if (b != nb - 1) {
      p.o_ap[b] = cv.take((size_t)N * p.ctot_b[b] * p.Vb[b + 1] * F);
      dap = std::max(dap, (size_t)N * p.ctot_b[b] * p.Vb[b + 1] * F);
    } else {
      p.o_ap[b] = 0;
    }
  }
  p.o_dap = cv.take(dap ? dap : 256);
  p.o_dz0 = cv.take((size_t)N * cfg.init_features * V0 * F);
  // forward statistics (fp64, zeroed at the start of every training forward)
  const size_t PAIR = 2 * NREP * sizeof(double);
  p.o_fstat = cv.cur;
  p.o_st_conv0 = cv.take(PAIR * cfg.init_features);
  p.o_st_t1.assign(nb, {});
  for (int b = 0; b < nb; ++b) {
    p.o_st_x[b] = cv.take(PAIR * p.ctot_b[b]);
    for (int l = 0; l < cfg.block_layers[b]; ++l) p.o_st_t1[b].push_back(cv.take(PAIR * p.mid));
  }
  p.fstat_bytes = cv.cur - p.o_fstat;
  // backward statistics (fp64, zeroed at the start of every backward)
  p.o_bstat = cv.cur;
  p.o_dg_n0 = cv.take(PAIR * cfg.init_features);
  p.o_dg_n1.assign(nb, {}); p.o_dg_n2.assign(nb, {}); p.o_dg_tr.clear();
  for (int b = 0; b < nb; ++b) {
    p.o_s_x[b] = cv.take(PAIR * p.ctot_b[b]);
    for (int l = 0; l < cfg.block_layers[b]; ++l) {
      p.o_dg_n1[b].push_back(cv.take(PAIR * p.layers[b][l].cin));
      p.o_dg_n2[b].push_back(cv.take(PAIR * p.mid));
    }
    if (b != nb - 1) p.o_dg_tr.push_back(cv.take(PAIR * p.ctot_b[b]));
  }
  p.o_dg_n5 = cv.take(PAIR * p.ctot_b[nb - 1]);
  p.bstat_bytes = cv.cur - p.o_bstat;
  // packed weights
  p.o_pk_conv0 = cv.take((size_t)7 * stem_krows(cfg.in_channels) * 64 * F);
  p.o_pk_c1.assign(nb, {}); p.o_pk_c2f.assign(nb, {}); p.o_pk_c2b.assign(nb, {}); p.o_pk_tr.clear();
  for (int b = 0; b < nb; ++b) {
    for (int l = 0; l < cfg.block_layers[b]; ++l) {
      p.o_pk_c1[b].push_back(cv.take((size_t)p.mid * p.layers[b][l].cin * F));
      p.o_pk_c2f[b].push_back(cv.take((size_t)cfg.growth * p.mid * 27 * F));
      p.o_pk_c2b[b].push_back(cv.take((size_t)cfg.growth * p.mid * 27 * F));
    }
    if (b != nb - 1) p.o_pk_tr.push_back(cv.take((size_t)p.trans[b].cin * p.trans[b].cout * F));
  }
  // weight-gradient slabs
  p.ns_conv0 = stem_wgrad_pick_splits(N, p.D0, p.H0, p.W0, cfg.in_channels);
  p.o_sl_conv0 = cv.take((size_t)p.ns_conv0 * cfg.in_channels * cfg.init_features * 352 * F);
  p.o_sl_c1.assign(nb, {}); p.o_sl_c2.assign(nb, {}); p.ns_c1.assign(nb, {}); p.ns_c2.assign(nb, {});
  p.o_sl_tr.clear(); p.ns_tr.clear();
  // Layers per weight-gradient launch (MMNN_WGRAD_GROUP="a,b,c" for >= 32768 / >= 4096 / fewer voxels per batch; 0 = the whole
  // dense block, the default).  With the backward on one stream the weight gradients of a block can all wait for the end of its
  // data-gradient chain: one launch per kernel variant, and the voxel splits -- hence the partial slabs `finalize` has to read
  // back -- shrink by the number of layers that share the launch.
  {
    int g[3] = {0, 0, 0};
    if (const char* e = getenv("MMNN_WGRAD_GROUP")) sscanf(e, "%d,%d,%d", &g[0], &g[1], &g[2]);
    for (int b = 0; b < nb; ++b) {
      const long nvox = (long)N * p.Vb[b];
      const int want = g[nvox >= 32768 ? 0 : nvox >= 4096 ? 1 : 2];
      p.wg_group[b] = (want < 1 || want > cfg.block_layers[b]) ? cfg.block_layers[b] : want;
    }
  }
  for (int b = 0; b < nb; ++b) {
    for (int l = 0; l < cfg.block_layers[b]; ++l) {
      const int ci = p.layers[b][l].cin;
      // conv1: a launch covers the layers of the group that use the same channel-group width; `pairs` = its (layer, channel
      // group) pairs, so that every block of the launch walks the same number of voxel chunks
      int pairs = 0;
      {
        const int g0 = (cfg.block_layers[b] - 1 - l) / p.wg_group[b];          // groups are formed from the LAST layer downwards
        const int cw = wgrad1_channel_width(ci, p.Vb[b]);
        for (int k = 0; k < cfg.block_layers[b]; ++k)
          if ((cfg.block_layers[b] - 1 - k) / p.wg_group[b] == g0 && wgrad1_channel_width(p.layers[b][k].cin, p.Vb[b]) == cw)
            pairs += cdiv(p.layers[b][k].cin, cw);
      }
      const int s1 = wgrad_pick_splits(1, N, p.Db[b], p.Hb[b], p.Wb[b], p.mid, ci, pairs);
      const int s2 = wgrad_pick_splits(27, N, p.Db[b], p.Hb[b], p.Wb[b], cfg.growth, p.mid, p.wg_group[b]);
      p.ns_c1[b].push_back(s1); p.ns_c2[b].push_back(s2);
      p.o_sl_c1[b].push_back(cv.take((size_t)s1 * p.mid * ci * F));
      p.o_sl_c2[b].push_back(cv.take((size_t)s2 * 27 * cfg.growth * p.mid * F));
    }
    if (b != nb - 1) {
      const int s = wgrad_pick_splits(1, N, p.Db[b + 1], p.Hb[b + 1], p.Wb[b + 1], p.trans[b].cout, p.trans[b].cin);
      p.ns_tr.push_back(s);
      p.o_sl_tr.push_back(cv.take((size_t)s * p.trans[b].cout * p.trans[b].cin * F));
    }
  }
  // dZ2 (gradient wrt the norm2 output) of every layer has its own buffer: the conv1 weight gradients that read it run on a
  // side stream, several layers behind the data-gradient chain, and must never make the chain wait for a buffer.
  for (int b = 0; b < nb; ++b) p.o_dz2[b] = cv.take((size_t)cfg.block_layers[b] * N * p.mid * p.Vb[b] * F);
  // cross-block K-split scratch: <= 256 blocks x one 32x32 (or 4 x 32x32) partial tile each, + per-tile counters
  p.o_kz_part = cv.take(KZ_PART_BYTES);
  p.o_kz_cnt = cv.take(KZ_CNT_ENTRIES * sizeof(unsigned) + BLK_SYNC_BYTES);   // + the persistent block kernels' barrier words
  p.o_blk_sync = p.o_kz_cnt + KZ_CNT_ENTRIES * sizeof(unsigned);
  // job tables
  int nlayers = 0;
  for (int b = 0; b < nb; ++b) nlayers += cfg.block_layers[b];
  p.n_run_jobs = nbn;
  p.n_pack_jobs = 1 + 3 * nlayers + (nb - 1);
  p.n_grad_jobs = 3 + 6 * nlayers + 3 * (nb - 1) + 2;
  p.o_jobs_run = cv.take(sizeof(RunStatJob) * p.n_run_jobs);
  p.o_jobs_pack = cv.take(sizeof(PackJob) * p.n_pack_jobs);
  p.o_jobs_grad = cv.take(sizeof(GradJob) * p.n_grad_jobs);
  p.o_jobs_blk = cv.take(sizeof(BlkLayer) * nlayers);
  {
    // r03 EXPERIMENT, off by default (plan option "persistent_forward" / MMNN_PERSISTENT=1): measured SLOWER than the per-layer
    // kernels at 2 x 2 x 128^3 -- 31-43 us per layer against 26-28 (profiles/r03_ab_experiments.txt).
    static const bool env_on = [] { const char* e = getenv("MMNN_PERSISTENT"); return e && e[0] == '1'; }();
    p.persistent = env_on;
    for (int b = 0; b < nb; ++b)
      p.persist_b[b] = (p.cin_b[b] % 2 == 0) && block_fwd_supported(N, p.Db[b], p.Hb[b], p.Wb[b], p.ctot_b[b], p.mid, cfg.growth);
  }
  p.n_layers = nlayers;
  p.o_wg_table = cv.take(sizeof(WgradArgs) * 2 * nlayers);     // [conv2 of every layer][conv1 of every layer], see plan_backward
  p.ws_bytes = cv.cur;
  p.host_jobs_bytes = p.o_wg_table - p.o_jobs_run;
  if (p.host_jobs) { (void)hipHostFree(p.host_jobs); p.host_jobs = nullptr; }   // (re)allocated lazily by the first forward
  p.tab_params = nullptr; p.tab_run = nullptr; p.tab_ws = nullptr;
  return 0;
}

void plan_free(Plan& p) {
  if (p.host_jobs) (void)hipHostFree(p.host_jobs);
  p.host_jobs = nullptr;
  if (p.wg_pinned) (void)hipHostFree(p.wg_pinned);
  p.wg_pinned = nullptr;
  for (hipEvent_t e : p.timer_ev) (void)hipEventDestroy(e);
  p.timer_ev.clear();
  for (hipEvent_t e : p.sync_ev) (void)hipEventDestroy(e);
  p.sync_ev.clear();
  if (p.side) (void)hipStreamDestroy(p.side);
  if (p.side2) (void)hipStreamDestroy(p.side2);
  p.side = p.side2 = nullptr;
}

static StatPtr statptr(char* ws, size_t o, int C, int off, int nrep = NREP) {
  StatPtr s;
  s.sum = reinterpret_cast<double*>(ws + o);
  s.sq = s.sum + (long)NREP * C;
  s.stride = C;
  s.off = off;
  s.nrep = nrep;
  s.pad_ = 0;
  return s;
}
static float* fptr(char* ws, size_t o) { return reinterpret_cast<float*>(ws + o); }

// capacities travel with the arguments: no process-global state
static void set_kz(FpropArgs& a, const Plan& p, char* ws) {
  a.kz_part = p.no_kz ? nullptr : fptr(ws, p.o_kz_part); a.kz_cnt = reinterpret_cast<unsigned*>(ws + p.o_kz_cnt);
  a.kz_part_bytes = KZ_PART_BYTES; a.kz_cnt_entries = KZ_CNT_ENTRIES;
  // developer aid: every convolution launch gets its own 64 x 16 slot of the phase-trace buffer, in launch order
  a.trace = (p.trace_base && p.trace_seq < p.trace_slots) ? p.trace_base + (size_t)(p.trace_seq++) * 64 * 16 : nullptr;
}

static BnFwd bnfwd(const Plan& p, StatPtr st, const float* params, float* run, long w, long b, long rm, long rv, double count, int training) {
  BnFwd f;
  f.st = st;
  f.rmean = run + rm; f.rvar = run + rv;
  f.gamma = params + w; f.beta = params + b;
  f.inv_count = 1.0 / count;
  f.eps = p.cfg.eps;
  f.training = training;
  return f;
}

// (re)build the device job tables when the buffers they point into change
// returns true when the tables were (re)built and have to be uploaded
static bool build_tables(Plan& p, const float* params, float* run, char* ws) {
  if (p.tab_params == params && p.tab_run == run && p.tab_ws == ws) return false;
  char* hj = static_cast<char*>(p.host_jobs);
  RunStatJob* rj = reinterpret_cast<RunStatJob*>(hj);
  PackJob* pj = reinterpret_cast<PackJob*>(hj + (p.o_jobs_pack - p.o_jobs_run));
  GradJob* gj = reinterpret_cast<GradJob*>(hj + (p.o_jobs_grad - p.o_jobs_run));
  const NetCfg& c = p.cfg;
  const int nb = c.nblocks;
  int ir = 0, ip = 0, ig = 0;
  p.max_pack = 0; p.max_grad = 0;
  auto run_job = [&](size_t o, int C, int off, int n, long rm, long rv, double count) {
    StatPtr s = statptr(ws, o, C, off);
    RunStatJob j; j.sum = s.sum; j.sq = s.sq; j.stride = C; j.off = off; j.C = n; j.rmean = run ? run + rm : nullptr; j.rvar = run ? run + rv : nullptr; j.count = count;
    rj[ir++] = j;
  };
  auto pack_job = [&](long src, size_t dst, int kind, int M, int C, long count) {
    PackJob j; j.src = params + src; j.dst = fptr(ws, dst); j.kind = kind; j.M = M; j.C = C; j.count = count;
    pj[ip++] = j; p.max_pack = std::max(p.max_pack, count);
  };
  auto grad_slab = [&](int kind, size_t o, long stride, int ns, int M, int C, long dst, long count) {
    GradJob j; j.kind = kind; j.src = ws + o; j.stride = stride; j.nsplit = ns; j.off = 0; j.M = M; j.C = C; j.dst_off = dst; j.count = count;
    gj[ig++] = j; p.max_grad = std::max(p.max_grad, count);
  };
  auto grad_bn = [&](size_t o, int C, long dw, long db) {   // pair block: [dbeta replicas][dgamma replicas]
    StatPtr s = statptr(ws, o, C, 0);
    GradJob j; j.kind = 3; j.stride = C; j.nsplit = 0; j.off = 0; j.M = 0; j.C = C; j.count = C;
    j.src = s.sq; j.dst_off = dw; gj[ig++] = j;     // dgamma
    j.src = s.sum; j.dst_off = db; gj[ig++] = j;    // dbeta
    p.max_grad = std::max(p.max_grad, (long)C);
  };
  {   // per-layer table of the persistent block forward
    BlkLayer* bl = reinterpret_cast<BlkLayer*>(hj + (p.o_jobs_blk - p.o_jobs_run));
    int id = 0;
    for (int b = 0; b < nb; ++b)
      for (int l = 0; l < c.block_layers[b]; ++l, ++id) {
        const LayerOff& lo = p.layers[b][l];
        BlkLayer& e = bl[id];
        memset(&e, 0, sizeof(e));
        e.cin = lo.cin; e.layer_id = id;
        e.w1 = fptr(ws, p.o_pk_c1[b][l]); e.w2 = fptr(ws, p.o_pk_c2f[b][l]);
        e.g1 = params + lo.n1w; e.b1 = params + lo.n1b; e.g2 = params + lo.n2w; e.b2 = params + lo.n2b;
        e.rm1 = run + lo.r1m; e.rv1 = run + lo.r1v; e.rm2 = run + lo.r2m; e.rv2 = run + lo.r2v;
        e.t1 = fptr(ws, p.o_t1[b][l]);
        const StatPtr st = statptr(ws, p.o_st_t1[b][l], p.mid, 0, p.nrep_b[b]);
        e.st_t1_sum = st.sum; e.st_t1_sq = st.sq;
      }
  }
  const double cnt0 = (double)p.N * p.D0 * p.H0 * p.W0;
  run_job(p.o_st_conv0, c.init_features, 0, c.init_features, p.r_n0m, p.r_n0v, cnt0);
  pack_job(p.p_conv0, p.o_pk_conv0, (c.in_channels % 2 == 0) ? 3 : 4, c.init_features, c.in_channels, (long)7 * stem_krows(c.in_channels) * 64);
  grad_slab(2, p.o_sl_conv0, (long)c.in_channels * c.init_features * 352, p.ns_conv0, c.init_features, c.in_channels, p.p_conv0,
            (long)c.init_features * c.in_channels * 343);
  grad_bn(p.o_dg_n0, c.init_features, p.p_n0w, p.p_n0b);
  for (int b = 0; b < nb; ++b) {
    const double cnt = (double)p.N * p.Vb[b];
    p.gj_begin[b] = ig;
    const long max_before = p.max_grad;
    p.max_grad = 0;
    for (int l = 0; l < c.block_layers[b]; ++l) {
      const LayerOff& lo = p.layers[b][l];
      run_job(p.o_st_x[b], p.ctot_b[b], 0, lo.cin, lo.r1m, lo.r1v, cnt);
      run_job(p.o_st_t1[b][l], p.mid, 0, p.mid, lo.r2m, lo.r2v, cnt);
      pack_job(lo.c1, p.o_pk_c1[b][l], 0, p.mid, lo.cin, (long)p.mid * lo.cin);
      pack_job(lo.c2, p.o_pk_c2f[b][l], 1, c.growth, p.mid, (long)c.growth * p.mid * 27);
      pack_job(lo.c2, p.o_pk_c2b[b][l], 2, c.growth, p.mid, (long)c.growth * p.mid * 27);
      grad_bn(p.o_dg_n1[b][l], lo.cin, lo.n1w, lo.n1b);
      grad_slab(0, p.o_sl_c1[b][l], (long)p.mid * lo.cin, p.ns_c1[b][l], p.mid, lo.cin, lo.c1, (long)p.mid * lo.cin);
      grad_bn(p.o_dg_n2[b][l], p.mid, lo.n2w, lo.n2b);
      grad_slab(1, p.o_sl_c2[b][l], (long)27 * c.growth * p.mid, p.ns_c2[b][l], c.growth, p.mid, lo.c2, (long)c.growth * p.mid * 27);
    }
    if (b != nb - 1) {
      const TransOff& t = p.trans[b];
      run_job(p.o_st_x[b], p.ctot_b[b], 0, t.cin, t.rm, t.rv, cnt);
      pack_job(t.cw, p.o_pk_tr[b], 0, t.cout, t.cin, (long)t.cout * t.cin);
      grad_bn(p.o_dg_tr[b], t.cin, t.nw, t.nb);
      grad_slab(0, p.o_sl_tr[b], (long)t.cout * t.cin, p.ns_tr[b], t.cout, t.cin, t.cw, (long)t.cout * t.cin);
    } else {
      run_job(p.o_st_x[b], p.ctot_b[b], 0, p.ctot_b[b], p.r_n5m, p.r_n5v, cnt);
      grad_bn(p.o_dg_n5, p.ctot_b[b], p.p_n5w, p.p_n5b);
    }
    p.gj_max[b] = p.max_grad;
    p.max_grad = std::max(p.max_grad, max_before);
  }
  p.gj_begin[nb] = ig;
  p.n_run_jobs = ir; p.n_pack_jobs = ip; p.n_grad_jobs = ig;
  p.tab_params = params; p.tab_run = run; p.tab_ws = ws;
  return true;
}

// ---- live kernel timing ---------------------------------------------------------------------------------------------
namespace {
struct ScopedTimer {   // records a start/stop event pair around one launch when the plan's timer selects it
  Plan& p; hipStream_t s; bool on;
  ScopedTimer(Plan& p_, int kind, int block, hipStream_t s_) : p(p_), s(s_), on(false) {
    if (!((p.timer_mask >> kind) & 1u) || (p.timer_block >= 0 && p.timer_block != block)) return;
    if (p.timer_used + 2 > p.timer_ev.size()) {
      if (p.timer_ev.size() >= 16384) return;   // read_timer() was not called for a long time: stop recording
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      p.timer_ev.push_back(a); p.timer_ev.push_back(b);
      p.timer_tag.push_back(0);
    }
    p.timer_tag[p.timer_used / 2] = kind * MAX_BLOCKS + (block >= 0 && block < MAX_BLOCKS ? block : 0);
    on = hipEventRecord(p.timer_ev[p.timer_used], s) == hipSuccess;
  }
  ~ScopedTimer() {
    if (!on) return;
    (void)hipEventRecord(p.timer_ev[p.timer_used + 1], s);
    p.timer_used += 2;
  }
};
}  // namespace

int plan_set_timer(Plan& p, int kind, int block) {
  MMNN_REQUIRE(kind >= -1 && kind < T_COUNT, "set_timer: unknown kernel class %d", kind);
  p.timer_mask = kind < 0 ? ~1u : (kind == T_NONE ? 0u : 1u << kind);   // -1: every class
  p.timer_block = block; p.timer_used = 0;
  for (int k = 0; k < T_COUNT * MAX_BLOCKS; ++k) { p.timer_ms[k] = 0.0; p.timer_count[k] = 0; }
  return 0;
}

int plan_read_timer(Plan& p, int kind, int block, double* total_ms, long* count) {
  MMNN_REQUIRE(kind >= T_NONE && kind < T_COUNT && block < MAX_BLOCKS, "read_timer: unknown kernel class %d / block %d", kind, block);
  for (size_t i = 0; i + 1 < p.timer_used; i += 2) {
    MMNN_HIP(hipEventSynchronize(p.timer_ev[i + 1]));
    float ms = 0.f;
    MMNN_HIP(hipEventElapsedTime(&ms, p.timer_ev[i], p.timer_ev[i + 1]));
    const int k = p.timer_tag[i / 2];
    p.timer_ms[k] += ms; p.timer_count[k] += 1;
  }
  p.timer_used = 0;
  double ms = 0.0; long n = 0;
  for (int k = 1; k < T_COUNT; ++k)         // class 0: all recorded classes together; block < 0: all blocks together
    for (int b = 0; b < MAX_BLOCKS; ++b)
      if ((kind == T_NONE || kind == k) && (block < 0 || block == b)) { ms += p.timer_ms[k * MAX_BLOCKS + b]; n += p.timer_count[k * MAX_BLOCKS + b]; }
  if (total_ms) *total_ms = ms;
  if (count) *count = n;
  return 0;
}

int plan_set_option(Plan& p, const char* name, long value) {
  const std::string s(name ? name : "");
  if (s == "single_stream") { p.single_stream = value != 0; return 0; }
  if (s == "side_streams") { p.side_streams = value < 0 ? 0 : (value > 2 ? 2 : (int)value); p.side_tried = false; return 0; }
  if (s == "trace_buffer") { p.trace_base = reinterpret_cast<unsigned long long*>(value); p.trace_seq = 0; return 0; }   // device pointer, 0 = off
  if (s == "trace_slots") { p.trace_slots = (int)value; return 0; }
  if (s == "params_version") { p.params_version = value; return 0; }
  if (s == "no_kz") { p.no_kz = value != 0; return 0; }
  if (s == "persistent_forward") { p.persistent = value != 0; return 0; }
  set_error("set_option: unknown option '%s'", s.c_str());
  return 1;
}

static DropCfg dropcfg(const Plan& p, uint64_t seed, int layer, int training) {
  DropCfg d;
  d.seed = seed; d.layer = layer;
  d.p = training ? p.cfg.dropout_p : 0.f;
  return d;
}

int plan_forward(Plan& p, const float* params, float* run, const float* x, char* ws, float* out, int training, uint64_t seed,
                 hipStream_t stream) {
  MMNN_REQUIRE(params && run && x && ws && out, "forward: null buffer");
  MMNN_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)params & 15) == 0 && ((uintptr_t)x & 15) == 0, "forward: buffers must be 256/16-byte aligned");
  const NetCfg& c = p.cfg;
  const int nb = c.nblocks, N = p.N;
  if (!p.host_jobs) {   // pinned staging for the job tables: the only memory the plan owns
    MMNN_HIP(hipHostMalloc(&p.host_jobs, p.host_jobs_bytes, hipHostMallocDefault));
    memset(p.host_jobs, 0, p.host_jobs_bytes);
  }
  static const bool host_timing = [] { const char* e = getenv("MMNN_HOST_TIMING"); return e && e[0] == '1'; }();   // developer aid
  static double ht[6] = {0, 0, 0, 0, 0, 0}; static long ht_n = 0;
  auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
  double t_prev = host_timing ? now() : 0.0;
  auto lap = [&](int k) { if (host_timing) { const double t = now(); ht[k] += t - t_prev; t_prev = t; } };
  // The job tables hold pointers and counts only: they change when a buffer moves, in practice once.  Uploading them on every
  // forward cost 0.85 ms of HOST time per step (hipMemcpyAsync from the pinned staging buffer returns only when the copy has
  // been handed to the idle stream), which kept the enqueueing thread from ever running ahead of the GPU.
  if (p.tab_params != params || p.tab_run != run || p.tab_ws != ws) {
    if (p.tab_ws != nullptr) MMNN_HIP(hipStreamSynchronize(stream));   // an earlier upload from the same staging buffer may still be in flight
    build_tables(p, params, run, ws);
    lap(0);
    MMNN_HIP(hipMemcpyAsync(ws + p.o_jobs_run, p.host_jobs, p.host_jobs_bytes, hipMemcpyHostToDevice, stream));
  }
  p.trace_seq = 0;
  lap(1);
  if (training) MMNN_HIP(hipMemsetAsync(ws + p.o_fstat, 0, p.fstat_bytes, stream));
  MMNN_HIP(hipMemsetAsync(ws + p.o_kz_cnt, 0, KZ_CNT_ENTRIES * sizeof(unsigned) + BLK_SYNC_BYTES, stream));
  lap(2);
  // The [k][m] weight panels only change when the parameters do.  A caller that can vouch for a version number (option
  // "params_version", non-zero) gets the repack skipped while it stays the same -- 31 of 32 forwards under the reference's
  // accumulate-to-64 rule (main.py:403-407).  Version 0 (the default) repacks on every forward.
  int rc = 0;
  if (p.params_version == 0 || p.params_version != p.packed_version || p.packed_params != params || p.packed_ws != ws) {
    rc = launch_pack(reinterpret_cast<const PackJob*>(ws + p.o_jobs_pack), p.n_pack_jobs, p.max_pack, stream);
    if (rc) return rc;
    p.packed_version = p.params_version; p.packed_params = params; p.packed_ws = ws;
    ++p.pack_launches;
  }
  lap(3);

  const double cnt0 = (double)N * p.D0 * p.H0 * p.W0;
  {  // stem
    StemConvArgs a;
    a.N = N; a.Cin = c.in_channels; a.D = p.D; a.H = p.H; a.W = p.W; a.Do = p.D0; a.Ho = p.H0; a.Wo = p.W0; a.M = c.init_features;
    a.x = x; a.wp = fptr(ws, p.o_pk_conv0); a.out = fptr(ws, p.o_conv0);
    a.st_out = statptr(ws, p.o_st_conv0, c.init_features, 0);
    if (!training) a.st_out.sum = nullptr;
    { ScopedTimer t(p, T_STEM_CONV, -1, stream); rc = launch_stem_conv(a, stream); }
    if (rc) return rc;
    StemPoolArgs q;
    q.N = N; q.C = c.init_features; q.Di = p.D0; q.Hi = p.H0; q.Wi = p.W0; q.Do = p.Db[0]; q.Ho = p.Hb[0]; q.Wo = p.Wb[0];
    q.x = fptr(ws, p.o_conv0);
    q.bn = bnfwd(p, statptr(ws, p.o_st_conv0, c.init_features, 0), params, run, p.p_n0w, p.p_n0b, p.r_n0m, p.r_n0v, cnt0, training);
    q.out = fptr(ws, p.o_x[0]); q.out_ns = (long)p.ctot_b[0] * p.Vb[0];
    q.idx = reinterpret_cast<unsigned char*>(ws + p.o_idx);
    q.st_out = statptr(ws, p.o_st_x[0], p.ctot_b[0], 0, p.nrep_b[0]);
    if (!training) q.st_out.sum = nullptr;
    if ((rc = launch_stem_pool(q, stream))) return rc;
  }
  lap(4);
  int layer_id = 0;
  for (int b = 0; b < nb; ++b) {
    const double cnt = (double)N * p.Vb[b];
    const long xns = (long)p.ctot_b[b] * p.Vb[b];
    const bool persistent = p.persist_b[b] && p.persistent;
    if (persistent) {   // every layer of the block in ONE resident launch (blockfwd.hpp)
      BlockFwdArgs q;
      memset(&q, 0, sizeof(q));
      q.N = N; q.D = p.Db[b]; q.H = p.Hb[b]; q.W = p.Wb[b];
      q.cin0 = p.cin_b[b]; q.ctot = p.ctot_b[b]; q.mid = p.mid; q.growth = c.growth; q.nlayers = c.block_layers[b];
      q.x = fptr(ws, p.o_x[b]); q.x_ns = xns;
      const StatPtr sx = statptr(ws, p.o_st_x[b], p.ctot_b[b], 0, p.nrep_b[b]);
      q.st_x_sum = sx.sum; q.st_x_sq = sx.sq; q.nrep = p.nrep_b[b];
      q.layers = reinterpret_cast<const BlkLayer*>(ws + p.o_jobs_blk) + layer_id;
      q.sync = reinterpret_cast<unsigned*>(ws + p.o_blk_sync) + 2 * b;
      q.inv_count = 1.0 / cnt; q.eps = c.eps; q.training = training;
      q.seed = seed; q.drop_p = training ? c.dropout_p : 0.f;
      { ScopedTimer t(p, T_BLOCK_FWD, b, stream); rc = launch_block_fwd(q, stream); }
      if (rc) return rc;
      layer_id += c.block_layers[b];
    }
    for (int l = 0; l < c.block_layers[b] && !persistent; ++l, ++layer_id) {
      const LayerOff& lo = p.layers[b][l];
      FpropArgs a;
      memset(&a, 0, sizeof(a));
      set_kz(a, p, ws);
      a.N = N; a.D = p.Db[b]; a.H = p.Hb[b]; a.W = p.Wb[b];
      // conv1: ReLU(BN(concat)) -> T1
      a.Cin = lo.cin; a.M = p.mid; a.nrep = p.nrep_b[b];
      a.in0 = fptr(ws, p.o_x[b]); a.in0_ns = xns; a.in0_coff = 0;
      a.bn_in = bnfwd(p, statptr(ws, p.o_st_x[b], p.ctot_b[b], 0, p.nrep_b[b]), params, run, lo.n1w, lo.n1b, lo.r1m, lo.r1v, cnt, training);
      a.w = fptr(ws, p.o_pk_c1[b][l]); a.w_ld = p.mid;
      a.out = fptr(ws, p.o_t1[b][l]); a.out_ns = (long)p.mid * p.Vb[b]; a.out_coff = 0;
      a.st_out = statptr(ws, p.o_st_t1[b][l], p.mid, 0, p.nrep_b[b]);
      if (!training) a.st_out.sum = nullptr;
      a.pf_ptr = fptr(ws, p.o_pk_c2f[b][l]); a.pf_bytes = (unsigned)(sizeof(float) * c.growth * p.mid * 27);
      { ScopedTimer t(p, T_CONV1_FWD, b, stream); rc = launch_fprop(a, 1, PRO_BNRELU, EPI_STORE_STATS, stream); }
      if (rc) return rc;
      // conv2: ReLU(BN(T1)) -> growth new channels of the concat buffer (+ channel dropout)
      FpropArgs e;
      memset(&e, 0, sizeof(e));
      set_kz(e, p, ws);
      e.N = N; e.D = p.Db[b]; e.H = p.Hb[b]; e.W = p.Wb[b];
      e.Cin = p.mid; e.M = c.growth; e.nrep = p.nrep_b[b];
      e.in0 = fptr(ws, p.o_t1[b][l]); e.in0_ns = (long)p.mid * p.Vb[b]; e.in0_coff = 0;
      e.bn_in = bnfwd(p, statptr(ws, p.o_st_t1[b][l], p.mid, 0, p.nrep_b[b]), params, run, lo.n2w, lo.n2b, lo.r2m, lo.r2v, cnt, training);
      e.w = fptr(ws, p.o_pk_c2f[b][l]); e.w_ld = c.growth;
      e.out = fptr(ws, p.o_x[b]); e.out_ns = xns; e.out_coff = lo.cin;
      e.drop_out = dropcfg(p, seed, layer_id, training);
      e.st_out = statptr(ws, p.o_st_x[b], p.ctot_b[b], lo.cin, p.nrep_b[b]);
      if (!training) e.st_out.sum = nullptr;
      if (l + 1 < c.block_layers[b]) {
        e.pf_ptr = fptr(ws, p.o_pk_c1[b][l + 1]); e.pf_bytes = (unsigned)(sizeof(float) * p.mid * p.layers[b][l + 1].cin);
      } else if (b != nb - 1) {
        e.pf_ptr = fptr(ws, p.o_pk_tr[b]); e.pf_bytes = (unsigned)(sizeof(float) * p.trans[b].cin * p.trans[b].cout);
      }
      { ScopedTimer t(p, T_CONV2_FWD, b, stream); rc = launch_fprop(e, 27, PRO_BNRELU, EPI_STORE_STATS, stream); }
      if (rc) return rc;
    }
    if (b != nb - 1) {
      const TransOff& t = p.trans[b];
      PoolFwdArgs q;
      q.N = N; q.C = t.cin; q.D = p.Db[b]; q.H = p.Hb[b]; q.W = p.Wb[b];
      q.x = fptr(ws, p.o_x[b]); q.x_ns = xns;
      q.bn = bnfwd(p, statptr(ws, p.o_st_x[b], p.ctot_b[b], 0, p.nrep_b[b]), params, run, t.nw, t.nb, t.rm, t.rv, cnt, training);
      q.out = fptr(ws, p.o_ap[b]);
      if ((rc = launch_bnrelu_avgpool(q, stream))) return rc;
      FpropArgs a;
      memset(&a, 0, sizeof(a));
      set_kz(a, p, ws);
      a.N = N; a.D = p.Db[b + 1]; a.H = p.Hb[b + 1]; a.W = p.Wb[b + 1];
      a.Cin = t.cin; a.M = t.cout; a.nrep = p.nrep_b[b + 1];
      a.in0 = fptr(ws, p.o_ap[b]); a.in0_ns = (long)t.cin * p.Vb[b + 1]; a.in0_coff = 0;
      a.w = fptr(ws, p.o_pk_tr[b]); a.w_ld = t.cout;
      a.out = fptr(ws, p.o_x[b + 1]); a.out_ns = (long)p.ctot_b[b + 1] * p.Vb[b + 1]; a.out_coff = 0;
      a.st_out = statptr(ws, p.o_st_x[b + 1], p.ctot_b[b + 1], 0, p.nrep_b[b + 1]);
      if (!training) a.st_out.sum = nullptr;
      a.pf_ptr = fptr(ws, p.o_pk_c1[b + 1][0]); a.pf_bytes = (unsigned)(sizeof(float) * p.mid * p.layers[b + 1][0].cin);
      if ((rc = launch_fprop(a, 1, PRO_NONE, EPI_STORE_STATS, stream))) return rc;
    } else {
      BnApplyArgs q;
      q.N = N; q.C = p.ctot_b[b]; q.V = p.Vb[b];
      q.x = fptr(ws, p.o_x[b]); q.x_ns = xns;
      q.bn = bnfwd(p, statptr(ws, p.o_st_x[b], p.ctot_b[b], 0, p.nrep_b[b]), params, run, p.p_n5w, p.p_n5b, p.r_n5m, p.r_n5v, cnt, training);
      q.out = out;
      if ((rc = launch_bn_apply(q, stream))) return rc;
    }
  }
  if (training) {
    if ((rc = launch_running_stats(reinterpret_cast<const RunStatJob*>(ws + p.o_jobs_run), p.n_run_jobs, c.momentum,
                                   p.nbt_count == p.n_run_jobs ? p.nbt : nullptr, stream))) return rc;
  }
  lap(5);
  if (host_timing && ++ht_n % 20 == 0) {
    fprintf(stderr, "[mmnn host timing, us per forward] tables %.1f  memcpy %.1f  memsets %.1f  pack %.1f  stem %.1f  layers %.1f\n", ht[0] / 20, ht[1] / 20,
            ht[2] / 20, ht[3] / 20, ht[4] / 20, ht[5] / 20);
    for (double& v : ht) v = 0.0;
  }
  return 0;
}

// Arguments of the two weight-gradient launches of dense layer (b, l): conv2 (3x3x3, `w2`) and conv1 (1x1x1, `w1`).
static void layer_wgrad_args(const Plan& p, const float* params, float* run, char* ws, int b, int l, int layer_id, uint64_t seed,
                             WgradArgs& w2, WgradArgs& w1) {
  const NetCfg& c = p.cfg;
  const int N = p.N;
  const double cnt = (double)N * p.Vb[b];
  const long xns = (long)p.ctot_b[b] * p.Vb[b], tns = (long)p.mid * p.Vb[b];
  const LayerOff& lo = p.layers[b][l];
  const BnFwd bn1 = bnfwd(p, statptr(ws, p.o_st_x[b], p.ctot_b[b], 0, p.nrep_b[b]), params, run, lo.n1w, lo.n1b, lo.r1m, lo.r1v, cnt, 1);
  const BnFwd bn2 = bnfwd(p, statptr(ws, p.o_st_t1[b][l], p.mid, 0, p.nrep_b[b]), params, run, lo.n2w, lo.n2b, lo.r2m, lo.r2v, cnt, 1);
  const StatPtr dg2 = statptr(ws, p.o_dg_n2[b][l], p.mid, 0, p.nrep_b[b]);
  float* dz2 = fptr(ws, p.o_dz2[b]) + (long)l * N * tns;
  memset(&w2, 0, sizeof(w2));
  w2.N = N; w2.D = p.Db[b]; w2.H = p.Hb[b]; w2.W = p.Wb[b];
  w2.M = c.growth; w2.Cin = p.mid;
  w2.g0 = fptr(ws, p.o_g[b]); w2.g0_ns = xns; w2.g0_coff = lo.cin;
  w2.g1 = fptr(ws, p.o_x[b]); w2.g1_ns = xns; w2.g1_coff = lo.cin;
  w2.gr.st = statptr(ws, p.o_st_x[b], p.ctot_b[b], lo.cin, p.nrep_b[b]);       // BN-backward of the layer's concat slice (gammas folded into G)
  w2.gr.s = statptr(ws, p.o_s_x[b], p.ctot_b[b], lo.cin, p.nrep_b[b]);
  w2.gr.gamma = nullptr; w2.gr.inv_count = 1.0 / cnt; w2.gr.eps = c.eps;
  w2.drop.seed = seed; w2.drop.layer = layer_id; w2.drop.p = c.dropout_p;
  w2.x = fptr(ws, p.o_t1[b][l]); w2.x_ns = tns; w2.x_coff = 0;
  w2.bn = bn2;
  w2.slab = fptr(ws, p.o_sl_c2[b][l]); w2.slab_stride = (long)27 * c.growth * p.mid; w2.nsplit = p.ns_c2[b][l];
  memset(&w1, 0, sizeof(w1));
  w1.N = N; w1.D = p.Db[b]; w1.H = p.Hb[b]; w1.W = p.Wb[b];
  w1.M = p.mid; w1.Cin = lo.cin;
  w1.g0 = dz2; w1.g0_ns = tns; w1.g0_coff = 0;
  w1.g1 = fptr(ws, p.o_t1[b][l]); w1.g1_ns = tns; w1.g1_coff = 0;
  w1.gr.st = statptr(ws, p.o_st_t1[b][l], p.mid, 0, p.nrep_b[b]);             // BN-backward of T1 (single consumer norm2): S1 = dbeta2, S2 = dgamma2
  w1.gr.s = dg2;
  w1.gr.gamma = params + lo.n2w; w1.gr.inv_count = 1.0 / cnt; w1.gr.eps = c.eps;
  w1.x = fptr(ws, p.o_x[b]); w1.x_ns = xns; w1.x_coff = 0;
  w1.bn = bn1;
  w1.slab = fptr(ws, p.o_sl_c1[b][l]); w1.slab_stride = (long)p.mid * lo.cin; w1.nsplit = p.ns_c1[b][l];
}

int plan_backward(Plan& p, const float* params, const float* x, char* ws, const float* grad_out, float* grad_params, int accumulate,
                  uint64_t seed, hipStream_t stream) {
  return plan_backward_range(p, params, x, ws, grad_out, grad_params, accumulate, seed, p.cfg.nblocks - 1, 0, stream);
}

int plan_block_param_range(const Plan& p, int block, long* begin, long* end) {
  const int nb = p.cfg.nblocks;
  MMNN_REQUIRE(block >= -1 && block < nb && begin && end, "block_param_range: block %d out of range [-1, %d)", block, nb);
  if (block < 0) { *begin = 0; *end = p.layers[0][0].n1w; return 0; }          // stem: conv0, norm0
  *begin = p.layers[block][0].n1w;                                              // the block's layers, then its transition / norm5
  *end = (block + 1 < nb) ? p.layers[block + 1][0].n1w : p.n_params;
  return 0;
}

// Backward of dense blocks hi, hi-1, ..., lo (0-based; the whole backward = one call with hi = nblocks-1, lo = 0).  A caller that
// splits it must walk the blocks downwards without gaps, starting at the last block; the call for `lo == 0` also runs the stem.  When
// a call returns (in stream order) the gradients of every parameter of blocks [lo, hi] -- plus the stem's when lo == 0 -- are FINAL in
// grad_params: a data-parallel caller can start reducing that range while the next call's kernels run.
int plan_backward_range(Plan& p, const float* params, const float* x, char* ws, const float* grad_out, float* grad_params, int accumulate,
                        uint64_t seed, int hi, int lo, hipStream_t stream) {
  MMNN_REQUIRE(params && x && ws && grad_out && grad_params, "backward: null buffer");
  MMNN_REQUIRE(p.tab_ws == ws && p.tab_params == params, "backward: must follow a training forward on the same buffers");
  const NetCfg& c = p.cfg;
  const int nb = c.nblocks, N = p.N;
  MMNN_REQUIRE(hi >= lo && lo >= 0 && hi < nb, "backward: bad block range [%d, %d] of %d blocks", lo, hi, nb);
  MMNN_REQUIRE(hi == nb - 1 || p.bwd_next == hi, "backward: block range [%d, %d] out of order (expected to continue at block %d)", lo, hi, p.bwd_next);
  float* run = p.tab_run;
  int rc;
  const bool first_call = hi == nb - 1;
  if (first_call) {
    MMNN_HIP(hipMemsetAsync(ws + p.o_bstat, 0, p.bstat_bytes, stream));
    MMNN_HIP(hipMemsetAsync(ws + p.o_kz_cnt, 0, KZ_CNT_ENTRIES * sizeof(unsigned) + BLK_SYNC_BYTES, stream));
  }
  p.bwd_next = lo - 1;
  // Two streams: the data-gradient chain (conv2 dgrad -> conv1 dgrad -> next layer) is the critical path; the weight-gradient
  // kernels only consume its products, so they run beside it on `side`, ordered by events.  Matters for the late dense blocks
  // whose kernels fill a fraction of the chip.  Falls back to one stream if the side stream cannot be created.
  // Streams.  The weight-gradient kernels only consume products of the data-gradient chain, so they CAN run beside it on side
  // streams (option "side_streams" / MMNN_SIDE_STREAMS = 1 or 2).  Default 0: since the small blocks' weight gradients go out as
  // batched launches that fill the chip, overlapping buys nothing any more and costs event hand-offs plus contention with the
  // chain (r02, 2x2x128^3: 10.7 ms with two side streams, 10.0 with one, 9.9 with none; same ranking at 64^3; one side stream is
  // 2 % ahead at 96^3).  Block 1's kernels cannot share a CU anyway (two waves per SIMD each).
  static const int env_side = [] { const char* e = getenv("MMNN_SIDE_STREAMS"); return e ? atoi(e) : -1; }();
  static const bool env_single = [] { const char* e = getenv("MMNN_SINGLE_STREAM"); return e && e[0] == '1'; }();
  const int want_side = (env_single || p.single_stream) ? 0 : (env_side >= 0 ? env_side : p.side_streams);
  const bool single = want_side <= 0;
  if (!single && (!p.side || (want_side >= 2 && !p.side2)) && !p.side_tried) {
    p.side_tried = true;
    if (!p.side && hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking) != hipSuccess) p.side = nullptr;
    if (want_side >= 2 && p.side && !p.side2 && hipStreamCreateWithFlags(&p.side2, hipStreamNonBlocking) != hipSuccess) p.side2 = nullptr;
  }
  hipStream_t side = (p.side && !single) ? p.side : stream;        // conv2 weight gradients (+ the big gradient finalise)
  hipStream_t side2 = (p.side2 && !single && want_side >= 2) ? p.side2 : side;   // conv1 weight gradients
  const bool two = side != stream;
  p.sync_used = 0;
  auto next_event = [&]() -> hipEvent_t {
    if (p.sync_used == p.sync_ev.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      p.sync_ev.push_back(e);
    }
    return p.sync_ev[p.sync_used++];
  };
  // `to` waits for everything enqueued on `from` so far
  auto order = [&](hipStream_t from, hipStream_t to) -> int {
    if (!two) return 0;
    hipEvent_t e = next_event();
    MMNN_REQUIRE(e != nullptr, "backward: cannot create a synchronisation event");
    MMNN_HIP(hipEventRecord(e, from));
    MMNN_HIP(hipStreamWaitEvent(to, e, 0));
    return 0;
  };
  // Weight gradients are launched in groups of `grp` layers: one event record on the main stream per group.  Every record
  // costs the chain ~6 us (the next kernel waits for the barrier packet's signal instead of being chained by the command
  // processor), which is as long as a whole small-block kernel, so blocks 2-4 batch several layers per record.
  // Device-resident argument tables of every layer's two weight-gradient launches (for the batched launches below).  Their
  // content does not depend on the step (the dropout seed travels as a kernel argument), so they are uploaded only when a
  // buffer moved -- in practice once.
  if (first_call || !p.wg_uploaded) {
    const size_t bytes = sizeof(WgradArgs) * 2 * p.n_layers;
    if (!p.wg_pinned) {
      MMNN_HIP(hipHostMalloc(&p.wg_pinned, bytes, hipHostMallocDefault));
      p.wg_shadow.assign(bytes, 0);
      p.wg_uploaded = false;
    }
    std::vector<WgradArgs> tab(2 * p.n_layers);
    int id = 0;
    for (int b = 0; b < nb; ++b)
      for (int l = 0; l < c.block_layers[b]; ++l, ++id) layer_wgrad_args(p, params, run, ws, b, l, id, 0, tab[id], tab[p.n_layers + id]);
    if (!p.wg_uploaded || memcmp(tab.data(), p.wg_shadow.data(), bytes) != 0) {
      if (p.wg_uploaded) MMNN_HIP(hipStreamSynchronize(stream));   // an earlier upload from the pinned buffer may still be in flight
      memcpy(p.wg_pinned, tab.data(), bytes);
      memcpy(p.wg_shadow.data(), tab.data(), bytes);
      MMNN_HIP(hipMemcpyAsync(ws + p.o_wg_table, p.wg_pinned, bytes, hipMemcpyHostToDevice, stream));
      p.wg_uploaded = true;
    }
  }
  const WgradArgs* dev_w2 = reinterpret_cast<const WgradArgs*>(ws + p.o_wg_table);
  const WgradArgs* dev_w1 = dev_w2 + p.n_layers;
  struct PendingW { WgradArgs w2, w1; int b, id; };
  std::vector<PendingW> pend;        // consecutive layers in DESCENDING layer id
  static const bool no_batch = [] { const char* e = getenv("MMNN_NO_WGRAD_BATCH"); return e && e[0] == '1'; }();   // debugging aid
  hipStream_t side_all = side, side2_all = side2;
  auto flush = [&]() -> int {
    if (pend.empty()) return 0;
    int rc2;
    // MMNN_SIDE_FROM_BLOCK=b (with side streams on): only the weight gradients of dense blocks >= b (0-based) leave the main stream
    static const int side_from = [] { const char* e = getenv("MMNN_SIDE_FROM_BLOCK"); return e ? atoi(e) : 0; }();
    const bool on_side = two && pend[0].b >= side_from;
    hipStream_t side = on_side ? side_all : stream, side2 = on_side ? side2_all : stream;
    if (on_side) {
      hipEvent_t e = next_event();
      MMNN_REQUIRE(e != nullptr, "backward: cannot create a synchronisation event");
      MMNN_HIP(hipEventRecord(e, stream));
      MMNN_HIP(hipStreamWaitEvent(side, e, 0));
      if (side2 != side) MMNN_HIP(hipStreamWaitEvent(side2, e, 0));
    }
    // The layers of a group are independent: ONE launch per kernel variant covers them all (blockIdx.z = layer).  pend holds
    // descending layer ids, so a run [i, j) of it is the ascending table range [pend[j-1].id, pend[i].id].
    const int np = (int)pend.size();
    std::vector<WgradArgs> host(np);
    if (np > 1 && !no_batch) {
      for (int i = 0; i < np; ++i) host[np - 1 - i] = pend[i].w2;
      { ScopedTimer t(p, T_CONV2_WGRAD, pend[0].b, side); rc2 = launch_wgrad_batched(host.data(), dev_w2 + pend[np - 1].id, np, seed, 27, PRO_BNRELU, side); }
      if (rc2) return rc2;
      for (int i = 0; i < np;) {       // conv1: runs of equal channel-group width (monotonic in the layer index)
        int j = i + 1;
        const long vb = (long)pend[i].w1.D * pend[i].w1.H * pend[i].w1.W;
        while (j < np && wgrad1_channel_width(pend[j].w1.Cin, vb) == wgrad1_channel_width(pend[i].w1.Cin, vb)) ++j;
        if (j - i > 1) {
          for (int k = i; k < j; ++k) host[j - 1 - k] = pend[k].w1;
          ScopedTimer t(p, T_CONV1_WGRAD, pend[i].b, side2);
          rc2 = launch_wgrad_batched(host.data(), dev_w1 + pend[j - 1].id, j - i, seed, 1, PRO_BNRELU, side2);
        } else {
          ScopedTimer t(p, T_CONV1_WGRAD, pend[i].b, side2);
          rc2 = launch_wgrad(pend[i].w1, 1, PRO_BNRELU, side2);
        }
        if (rc2) return rc2;
        i = j;
      }
    } else {
      for (const PendingW& q : pend) {
        { ScopedTimer t(p, T_CONV2_WGRAD, q.b, side); rc2 = launch_wgrad(q.w2, 27, PRO_BNRELU, side); }
        if (rc2) return rc2;
      }
      for (const PendingW& q : pend) {
        { ScopedTimer t(p, T_CONV1_WGRAD, q.b, side2); rc2 = launch_wgrad(q.w1, 1, PRO_BNRELU, side2); }
        if (rc2) return rc2;
      }
    }
    pend.clear();
    return 0;
  };
  auto sptr = [&](int b, int off) { return statptr(ws, p.o_s_x[b], p.ctot_b[b], off, p.nrep_b[b]); };
  auto concat_grad = [&](int b, int off) {   // BN-backward of concat channels [off, ...) of block b (gammas folded into G)
    BnBwd g;
    g.st = statptr(ws, p.o_st_x[b], p.ctot_b[b], off, p.nrep_b[b]);
    g.s = sptr(b, off);
    g.gamma = nullptr;
    g.inv_count = 1.0 / ((double)N * p.Vb[b]);
    g.eps = c.eps;
    return g;
  };
  if (first_call) {  // norm5: first contribution to the last block's G
    const int b = nb - 1;
    ConsumerBwdArgs a;
    a.N = N; a.C = p.ctot_b[b]; a.D = p.Db[b]; a.H = p.Hb[b]; a.W = p.Wb[b]; a.mode = 0; a.nrep = p.nrep_b[b];
    a.dy = grad_out;
    a.x = fptr(ws, p.o_x[b]); a.x_ns = (long)p.ctot_b[b] * p.Vb[b];
    a.bn = bnfwd(p, statptr(ws, p.o_st_x[b], p.ctot_b[b], 0, p.nrep_b[b]), params, run, p.p_n5w, p.p_n5b, p.r_n5m, p.r_n5v, (double)N * p.Vb[b], 1);
    a.g = fptr(ws, p.o_g[b]); a.g_ns = a.x_ns;
    StatPtr dg = statptr(ws, p.o_dg_n5, p.ctot_b[b], 0, p.nrep_b[b]);
    a.dbeta = dg.sum; a.dgamma = dg.sq;
    a.s_acc = sptr(b, 0);
    if ((rc = launch_consumer_bwd(a, stream))) return rc;
  }
  int layer_id = 0;
  for (int b = 0; b <= hi; ++b) layer_id += c.block_layers[b];
  for (int b = hi; b >= lo; --b) {
    const double cnt = (double)N * p.Vb[b];
    const long xns = (long)p.ctot_b[b] * p.Vb[b];
    const long tns = (long)p.mid * p.Vb[b];
    const int grp = p.wg_group[b];
    for (int l = c.block_layers[b] - 1; l >= 0; --l) {
      --layer_id;
      const LayerOff& lo = p.layers[b][l];
      const BnFwd bn1 = bnfwd(p, statptr(ws, p.o_st_x[b], p.ctot_b[b], 0, p.nrep_b[b]), params, run, lo.n1w, lo.n1b, lo.r1m, lo.r1v, cnt, 1);
      const BnFwd bn2 = bnfwd(p, statptr(ws, p.o_st_t1[b][l], p.mid, 0, p.nrep_b[b]), params, run, lo.n2w, lo.n2b, lo.r2m, lo.r2v, cnt, 1);
      const StatPtr dg2 = statptr(ws, p.o_dg_n2[b][l], p.mid, 0, p.nrep_b[b]);
      const StatPtr dg1 = statptr(ws, p.o_dg_n1[b][l], lo.cin, 0, p.nrep_b[b]);
      const DropCfg drop = dropcfg(p, seed, layer_id, 1);
      // conv2 data gradient -> dZ2 (ReLU mask of norm2 applied) + dgamma2/dbeta2
      FpropArgs a;
      memset(&a, 0, sizeof(a));
      set_kz(a, p, ws);
      a.N = N; a.D = p.Db[b]; a.H = p.Hb[b]; a.W = p.Wb[b];
      a.Cin = c.growth; a.M = p.mid; a.nrep = p.nrep_b[b];
      a.in0 = fptr(ws, p.o_g[b]); a.in0_ns = xns; a.in0_coff = lo.cin;
      a.in1 = fptr(ws, p.o_x[b]); a.in1_ns = xns; a.in1_coff = lo.cin;
      a.gr_in = concat_grad(b, lo.cin);
      a.drop_in = drop;
      a.w = fptr(ws, p.o_pk_c2b[b][l]); a.w_ld = p.mid;
      float* dz2 = fptr(ws, p.o_dz2[b]) + (long)l * N * tns;
      a.out = dz2; a.out_ns = tns; a.out_coff = 0;
      a.ex = fptr(ws, p.o_t1[b][l]); a.ex_ns = tns; a.ex_coff = 0;
      a.ebn = bn2;
      a.dbeta = dg2.sum; a.dgamma = dg2.sq;
      a.pf_ptr = params + lo.c1; a.pf_bytes = (unsigned)(sizeof(float) * p.mid * lo.cin);
      { ScopedTimer t(p, T_CONV2_DGRAD, b, stream); rc = launch_fprop(a, 27, PRO_GRAD, EPI_MASK_STORE, stream); }
      if (rc) return rc;
      WgradArgs w2, w1;
      layer_wgrad_args(p, params, run, ws, b, l, layer_id, seed, w2, w1);
      w2.trace = (p.trace_base && p.trace_seq < p.trace_slots) ? p.trace_base + (size_t)(p.trace_seq++) * 64 * 16 : nullptr;   // developer aid
      w1.trace = (p.trace_base && p.trace_seq < p.trace_slots) ? p.trace_base + (size_t)(p.trace_seq++) * 64 * 16 : nullptr;
      // BN-backward of T1 (single consumer norm2): S1 = dbeta2, S2 = dgamma2, scaled by gamma2
      BnBwd g1;
      g1.st = statptr(ws, p.o_st_t1[b][l], p.mid, 0, p.nrep_b[b]);
      g1.s = dg2;
      g1.gamma = params + lo.n2w;
      g1.inv_count = 1.0 / cnt;
      g1.eps = c.eps;
      // conv1 data gradient -> G[0:cin) += gamma1 * mask * (...), dgamma1/dbeta1, S1/S2
      FpropArgs d;
      memset(&d, 0, sizeof(d));
      set_kz(d, p, ws);
      d.N = N; d.D = p.Db[b]; d.H = p.Hb[b]; d.W = p.Wb[b];
      d.Cin = p.mid; d.M = lo.cin; d.nrep = p.nrep_b[b];
      d.in0 = dz2; d.in0_ns = tns; d.in0_coff = 0;
      d.in1 = fptr(ws, p.o_t1[b][l]); d.in1_ns = tns; d.in1_coff = 0;
      d.gr_in = g1;
      d.w = params + lo.c1; d.w_ld = lo.cin;
      d.out = fptr(ws, p.o_g[b]); d.out_ns = xns; d.out_coff = 0;
      d.ex = fptr(ws, p.o_x[b]); d.ex_ns = xns; d.ex_coff = 0;
      d.ebn = bn1;
      d.dbeta = dg1.sum; d.dgamma = dg1.sq;
      d.s_acc = sptr(b, 0);
      if (l > 0) { d.pf_ptr = fptr(ws, p.o_pk_c2b[b][l - 1]); d.pf_bytes = (unsigned)(sizeof(float) * c.growth * p.mid * 27); }
      { ScopedTimer t(p, T_CONV1_DGRAD, b, stream); rc = launch_fprop(d, 1, PRO_GRAD, EPI_MASK_ACCUM, stream); }
      if (rc) return rc;
      // both weight gradients of this layer can run from here on (its G slice was final before conv2 dgrad, dZ2 and
      // dgamma2/dbeta2 since conv2 dgrad): queue them for the side streams
      PendingW pw;
      pw.w2 = w2; pw.w1 = w1; pw.b = b; pw.id = layer_id;
      pend.push_back(pw);
      if ((int)pend.size() >= grp || l == 0) {
        if ((rc = flush())) return rc;
      }
    }
    if (two) {
      // Every gradient of this block (its layers, and the transition / norm5 that consumed it) is final once the side streams
      // drain: reduce its slabs there (HBM-bound) while the chain moves on to the next block / the stem (MFMA-bound).
      if ((rc = order(stream, side))) return rc;
      if (side2 != side && (rc = order(side2, side))) return rc;
      if ((rc = launch_finalize(reinterpret_cast<const GradJob*>(ws + p.o_jobs_grad) + p.gj_begin[b], p.gj_begin[b + 1] - p.gj_begin[b],
                                p.gj_max[b], grad_params, accumulate, side))) return rc;
    }
    if (b > 0) {
      const int pb = b - 1;
      const TransOff& t = p.trans[pb];
      const long pxns = (long)p.ctot_b[pb] * p.Vb[pb];
      // transition conv weight gradient (input = pooled activations)
      WgradArgs w;
      memset(&w, 0, sizeof(w));
      w.N = N; w.D = p.Db[b]; w.H = p.Hb[b]; w.W = p.Wb[b];
      w.M = t.cout; w.Cin = t.cin;
      w.g0 = fptr(ws, p.o_g[b]); w.g0_ns = xns; w.g0_coff = 0;
      w.g1 = fptr(ws, p.o_x[b]); w.g1_ns = xns; w.g1_coff = 0;
      w.gr = concat_grad(b, 0);
      w.x = fptr(ws, p.o_ap[pb]); w.x_ns = (long)t.cin * p.Vb[b]; w.x_coff = 0;
      w.slab = fptr(ws, p.o_sl_tr[pb]); w.slab_stride = (long)t.cout * t.cin; w.nsplit = p.ns_tr[pb];
      if ((rc = launch_wgrad(w, 1, PRO_NONE, stream))) return rc;
      // transition conv data gradient -> gradient wrt the pooled activations
      FpropArgs d;
      memset(&d, 0, sizeof(d));
      set_kz(d, p, ws);
      d.N = N; d.D = p.Db[b]; d.H = p.Hb[b]; d.W = p.Wb[b];
      d.Cin = t.cout; d.M = t.cin;
      d.in0 = w.g0; d.in0_ns = xns; d.in0_coff = 0;
      d.in1 = w.g1; d.in1_ns = xns; d.in1_coff = 0;
      d.gr_in = w.gr;
      d.w = params + t.cw; d.w_ld = t.cin;
      d.out = fptr(ws, p.o_dap); d.out_ns = (long)t.cin * p.Vb[b]; d.out_coff = 0;
      if ((rc = launch_fprop(d, 1, PRO_GRAD, EPI_STORE, stream))) return rc;
      // un-pool + ReLU + BN of the transition: first contribution to the previous block's G
      ConsumerBwdArgs q;
      q.N = N; q.C = t.cin; q.D = p.Db[pb]; q.H = p.Hb[pb]; q.W = p.Wb[pb]; q.mode = 1; q.nrep = p.nrep_b[pb];
      q.dy = fptr(ws, p.o_dap);
      q.x = fptr(ws, p.o_x[pb]); q.x_ns = pxns;
      q.bn = bnfwd(p, statptr(ws, p.o_st_x[pb], p.ctot_b[pb], 0, p.nrep_b[pb]), params, run, t.nw, t.nb, t.rm, t.rv, (double)N * p.Vb[pb], 1);
      q.g = fptr(ws, p.o_g[pb]); q.g_ns = pxns;
      StatPtr dg = statptr(ws, p.o_dg_tr[pb], t.cin, 0, p.nrep_b[pb]);
      q.dbeta = dg.sum; q.dgamma = dg.sq;
      q.s_acc = sptr(pb, 0);
      if ((rc = launch_consumer_bwd(q, stream))) return rc;
    } else {
      const double cnt0 = (double)N * p.D0 * p.H0 * p.W0;
      StemPoolBwdArgs q;
      q.N = N; q.C = c.init_features; q.Di = p.D0; q.Hi = p.H0; q.Wi = p.W0; q.Do = p.Db[0]; q.Ho = p.Hb[0]; q.Wo = p.Wb[0];
      q.x = fptr(ws, p.o_conv0);
      q.bn = bnfwd(p, statptr(ws, p.o_st_conv0, c.init_features, 0), params, run, p.p_n0w, p.p_n0b, p.r_n0m, p.r_n0v, cnt0, 1);
      q.g = fptr(ws, p.o_g[0]); q.g_ns = xns;
      q.xp = fptr(ws, p.o_x[0]); q.xp_ns = xns;
      q.gr = concat_grad(0, 0);
      q.idx = reinterpret_cast<const unsigned char*>(ws + p.o_idx);
      q.dz = fptr(ws, p.o_dz0);
      StatPtr dg = statptr(ws, p.o_dg_n0, c.init_features, 0);
      q.dbeta = dg.sum; q.dgamma = dg.sq;
      if ((rc = launch_stem_pool_bwd(q, stream))) return rc;
      StemWgradArgs w;
      w.N = N; w.Cin = c.in_channels; w.D = p.D; w.H = p.H; w.W = p.W; w.Do = p.D0; w.Ho = p.H0; w.Wo = p.W0; w.M = c.init_features;
      w.x = x; w.dz = fptr(ws, p.o_dz0); w.y = fptr(ws, p.o_conv0);
      w.gr.st = statptr(ws, p.o_st_conv0, c.init_features, 0);
      w.gr.s = dg;
      w.gr.gamma = params + p.p_n0w;
      w.gr.inv_count = 1.0 / cnt0;
      w.gr.eps = c.eps;
      w.slab = fptr(ws, p.o_sl_conv0); w.slab_stride = (long)c.in_channels * c.init_features * 352; w.nsplit = p.ns_conv0;
      { ScopedTimer t(p, T_STEM_WGRAD, -1, stream); rc = launch_stem_wgrad(w, stream); }
      if (rc) return rc;
    }
  }
  if ((rc = order(side, stream))) return rc;                // join: every weight-gradient slab is written / reduced
  if (side2 != side && (rc = order(side2, stream))) return rc;
  const long stem_count = (long)c.init_features * c.in_channels * 343;
  const GradJob* jobs = reinterpret_cast<const GradJob*>(ws + p.o_jobs_grad);
  if (two) {                                                // the blocks' jobs were reduced on the side stream as they became final
    if (lo > 0) return 0;
    return launch_finalize(jobs, 3, stem_count, grad_params, accumulate, stream);
  }
  // one launch for every job of this call: blocks [lo, hi] (their layers + the transition / norm5 behind them), + the stem when lo == 0
  const int j0 = lo == 0 ? 0 : p.gj_begin[lo], j1 = p.gj_begin[hi + 1];
  long jmax = lo == 0 ? stem_count : 0;
  for (int b = lo; b <= hi; ++b) jmax = std::max(jmax, p.gj_max[b]);
  return launch_finalize(jobs + j0, j1 - j0, jmax, grad_params, accumulate, stream);
}

int plan_relu_mask(Plan& p, const float* params, char* ws, int kind, int b, int l, unsigned char* out, hipStream_t stream) {
  MMNN_REQUIRE(p.tab_ws == ws && p.tab_params == params, "relu_mask: must follow a training forward on the same buffers");
  const NetCfg& c = p.cfg;
  float* run = p.tab_run;
  MaskArgs a;
  a.out = out; a.N = p.N;
  if (kind == 0) {
    a.C = c.init_features; a.V = p.D0 * p.H0 * p.W0;
    a.x = fptr(ws, p.o_conv0); a.x_ns = (long)a.C * a.V;
    a.bn = bnfwd(p, statptr(ws, p.o_st_conv0, c.init_features, 0), params, run, p.p_n0w, p.p_n0b, p.r_n0m, p.r_n0v, (double)p.N * a.V, 1);
    return launch_relu_mask(a, stream);
  }
  MMNN_REQUIRE(b >= 0 && b < c.nblocks, "relu_mask: bad block %d", b);
  const double cnt = (double)p.N * p.Vb[b];
  a.V = p.Vb[b];
  if (kind == 3) {
    MMNN_REQUIRE(b < c.nblocks - 1, "relu_mask: block %d has no transition", b);
    const TransOff& t = p.trans[b];
    a.C = t.cin; a.x = fptr(ws, p.o_x[b]); a.x_ns = (long)p.ctot_b[b] * p.Vb[b];
    a.bn = bnfwd(p, statptr(ws, p.o_st_x[b], p.ctot_b[b], 0, p.nrep_b[b]), params, run, t.nw, t.nb, t.rm, t.rv, cnt, 1);
    return launch_relu_mask(a, stream);
  }
  MMNN_REQUIRE(l >= 0 && l < c.block_layers[b] && (kind == 1 || kind == 2), "relu_mask: bad site (%d,%d,%d)", kind, b, l);
  const LayerOff& lo = p.layers[b][l];
  if (kind == 1) {
    a.C = lo.cin; a.x = fptr(ws, p.o_x[b]); a.x_ns = (long)p.ctot_b[b] * p.Vb[b];
    a.bn = bnfwd(p, statptr(ws, p.o_st_x[b], p.ctot_b[b], 0, p.nrep_b[b]), params, run, lo.n1w, lo.n1b, lo.r1m, lo.r1v, cnt, 1);
  } else {
    a.C = p.mid; a.x = fptr(ws, p.o_t1[b][l]); a.x_ns = (long)p.mid * p.Vb[b];
    a.bn = bnfwd(p, statptr(ws, p.o_st_t1[b][l], p.mid, 0, p.nrep_b[b]), params, run, lo.n2w, lo.n2b, lo.r2m, lo.r2v, cnt, 1);
  }
  return launch_relu_mask(a, stream);
}

long plan_ws_offset(const Plan& p, const char* name, int i, int j) {
  const std::string s(name);
  const int nb = p.cfg.nblocks;
  auto okb = [&](int b) { return b >= 0 && b < nb; };
  auto okl = [&](int b, int l) { return okb(b) && l >= 0 && l < p.cfg.block_layers[b]; };
  if (s == "conv0") return (long)p.o_conv0;
  if (s == "idx") return (long)p.o_idx;
  if (s == "dz0") return (long)p.o_dz0;
  if (s == "dz2" && okl(i, j)) return (long)(p.o_dz2[i] + (size_t)j * p.N * p.mid * p.Vb[i] * sizeof(float));
  if (s == "dap") return (long)p.o_dap;
  if (s == "x" && okb(i)) return (long)p.o_x[i];
  if (s == "g" && okb(i)) return (long)p.o_g[i];
  if (s == "ap" && okb(i) && i < nb - 1) return (long)p.o_ap[i];
  if (s == "t1" && okl(i, j)) return (long)p.o_t1[i][j];
  if (s == "st_conv0") return (long)p.o_st_conv0;
  if (s == "st_x" && okb(i)) return (long)p.o_st_x[i];
  if (s == "st_t1" && okl(i, j)) return (long)p.o_st_t1[i][j];
  if (s == "s_x" && okb(i)) return (long)p.o_s_x[i];
  if (s == "pk_c1" && okl(i, j)) return (long)p.o_pk_c1[i][j];
  if (s == "pk_c2f" && okl(i, j)) return (long)p.o_pk_c2f[i][j];
  if (s == "pk_c2b" && okl(i, j)) return (long)p.o_pk_c2b[i][j];
  if (s == "pk_conv0") return (long)p.o_pk_conv0;
  if (s == "blk_sync") return (long)p.o_blk_sync;
  if (s == "#pack_launches") return p.pack_launches;      // counter, not an offset (tests)
  return -1;
}

}  // namespace mmnn
