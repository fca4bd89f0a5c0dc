// Implicit-GEMM "forward-shaped" convolution on fp32 MFMA for NCDHW tensors (stride 1, "same" padding).
//
//     out[n][m][v] = sum_{c < Cin} sum_{tap < TAPS}  w[(c*TAPS + tap)][m] * f(in[n][c][v + off(tap)])
//
// TAPS = 1 (1x1x1) or 27 (3x3x3).  One kernel template serves four reference ops (models/densenet.py:76-82 and
// their autograd adjoints, main.py:469):
//   conv1 forward   : f = ReLU(BN(x))        (PRO_BNRELU)  epilogue: store + batch statistics      (EPI_STORE_STATS)
//   conv2 forward   : same, 27 taps, zero padding applied AFTER BN+ReLU
//   conv2 data-grad : f = BN-backward(G, Y)   (PRO_GRAD)    epilogue: ReLU mask, dgamma/dbeta sums  (EPI_MASK_STORE)
//   conv1 data-grad : f = BN-backward(dZ, T)  (PRO_GRAD)    epilogue: ReLU mask, G += gamma*z, sums (EPI_MASK_ACCUM)
//
// MFMA mapping (v_mfma_f32_32x32x2_f32): A = weights (i = output row m, k = input channel parity), B = activations
// (k = channel parity, j = voxel), so the 32 lanes of a half read 32 CONSECUTIVE voxels of one channel from LDS
// (conflict-free ds_read_b32) and the accumulator tile comes out as rows = m, lanes = voxels: every store
// instruction writes 128-byte runs of one output channel.  Lanes 0-31 take channel 2k, lanes 32-63 channel 2k+1.
#pragma once
#include "common.hpp"

namespace mmnn {

// Cross-workgroup K-split hand-off (fprop_kernel, "cross-block K-split").  0: the fence-free form measured valid on gfx950 / ROCm 7.2
// (MI355X_MICROARCH.md, inter-workgroup visibility, table row "ONE lane of each storing workgroup ... an agent-scope atomic add / the
// workgroup whose add came last"): partial tiles leave with write-through `sc1` stores, every storing wave drains `vmcnt(0)`, a workgroup
// barrier, ONE lane's agent-scope ticket; the last arriver reads them with `sc1` loads behind a workgroup barrier.  That is a property of
// this chip's cache policy, not of the HIP memory model, so it is the default ONLY for gfx950; any other target -- or -DMMNN_KZ_FENCED=1
// (`MMNN_KZ_FENCED=1 python -m mmnn_sts_amd.build` builds libmmnn_sts_fenced.so) -- gets the portable form: plain stores, an agent-scope
// RELEASE fence before the ticket, an agent-scope ACQUIRE fence in every reading wave after it.  tests/test_kz_handoff_gpu.py stresses
// whichever library is loaded; tests/test_host_cpu.py::test_kz_handoff_isa checks that the emitted gfx950 code really carries `sc1`.
#if !defined(MMNN_KZ_FENCED)
#if defined(__gfx950__) || !defined(__HIP_DEVICE_COMPILE__)
#define MMNN_KZ_FENCED 0
#else
#define MMNN_KZ_FENCED 1
#endif
#endif

enum Pro { PRO_NONE = 0, PRO_BNRELU = 1, PRO_GRAD = 2 };
enum Epi { EPI_STORE = 0, EPI_STORE_STATS = 1, EPI_MASK_STORE = 2, EPI_MASK_ACCUM = 3 };

struct FpropArgs {
  int N, D, H, W;
  int Cin, M;
  const float* in0; long in0_ns; int in0_coff;   // in0[n*in0_ns + (in0_coff + c)*V + v]
  const float* in1; long in1_ns; int in1_coff;   // PRO_GRAD: the normalised tensor (in0 is its upstream G)
  BnFwd bn_in;
  BnBwd gr_in;
  DropCfg drop_in;
  const float* w; int w_ld;                      // w[(c*TAPS + tap)*w_ld + m]
  float* out; long out_ns; int out_coff;
  DropCfg drop_out;
  StatPtr st_out;
  const float* ex; long ex_ns; int ex_coff;
  BnFwd ebn;
  double* dgamma; double* dbeta;                 // [NREP][M]
  StatPtr s_acc;
  int nrep;                                      // replicas this launch's epilogue spreads its fp64 atomics over (0: NREP); see StatPtr::nrep
  // cross-block K-split (gridDim.z slices of the channel axis): partial tiles + per-tile arrival counters (zero on entry)
  float* kz_part; unsigned* kz_cnt;
  size_t kz_part_bytes; unsigned kz_cnt_entries;   // capacity of kz_part / kz_cnt as provided by the caller (the split is skipped when it would not fit)
  // Weights of the NEXT launch on this stream (may be null): every layer's weights are read exactly once per pass, i.e. always
  // cold, and the first chunk's weight load is the longest single wait of the small-extent kernels.  Each XCD's workgroups touch
  // one dword per 128-byte line of [pf_ptr, pf_ptr + pf_bytes) while their own first chunk is in flight, which pulls the range
  // into that XCD's L2 before the next kernel starts.
  const float* pf_ptr; unsigned pf_bytes;
  // Filled in by launch_cfg (callers leave them zero): ceil(2^32 / d) for the divisors of the workgroup -> tile decomposition
  // (3x3x3: tiles along W, H, D; 1x1x1: tiles per sample), 0 when d == 1.  q = mulhi(b, magic) is exact for b * d < 2^32 and is two
  // scalar instructions; the compiler's sequence for a run-time divisor is ~25 dependent ones, three times over, ahead of the first load.
  unsigned mg_w, mg_h, mg_d;
  // developer aid (tools/phase_trace.py): when non-null, thread 0 of the first 64 blocks of the launch stores shader-clock stamps
  // of its phases to trace[block * 16 + k]; null in normal operation
  unsigned long long* trace;
};

int launch_fprop(const FpropArgs& a, int taps, int pro, int epi, hipStream_t stream);
int current_device_slot();   // index of the calling thread's current HIP device, for per-device caches of kernel attributes
constexpr int MAX_DEVICES = 32;

#if defined(__HIPCC__)

template <int TAPS, int PRO, int EPI, int WM, int WN, int KS, int MT, int NT, int KC, int TD, int TH, int TW, bool SPEC = false>
struct FpropCfg {
  static constexpr int NWAVES = WM * WN * KS;          // KS wave groups split the channel (reduction) axis of every chunk
  // SPEC: wave specialisation -- a second set of NWAVES "loader" waves stages chunk k+1 into the other LDS buffer while
  // the compute waves run the MFMAs of chunk k (one loader + one compute wave per SIMD: VALU/LDS-write beside the matrix pipe)
  static constexpr int NCOMPUTE = NWAVES * 64;
  static constexpr int NTHREADS = NCOMPUTE * (SPEC ? 2 : 1);
  static constexpr int M_B = WM * MT * 32;
  static constexpr int V_B = WN * NT * 32;
  static constexpr int RS = (TAPS == 27) ? TW + 8 : TW;
  static constexpr int HS = (TAPS == 27) ? TH + 2 : 1;
  static constexpr int DS = (TAPS == 27) ? TD + 2 : 1;
  static constexpr int XS = DS * HS * RS;
  static constexpr int NCOEF = (PRO == PRO_BNRELU) ? 2 : (PRO == PRO_GRAD ? 3 : 0);
  static constexpr int ECOEF = 6 + 2 * WN;   // per output row: a, b, mean, rstd, gamma, dropscale, then per-wave partial sums
  static_assert(TD * TH * TW == V_B, "tile volume must equal the block's voxel count");
  static_assert(KC % 2 == 0 && (KC / 2) % KS == 0, "channel-pair count of a chunk must be a multiple of the K-split");
  static constexpr int STAGE = KC * XS + KC * TAPS * M_B;                       // floats: activations + weights of a chunk
  // small tiles (one accumulator per wave, K split over wave groups): every group keeps 16 / KS rows of the tile from the in-block
  // reduction to the stores (fprop_kernel, "distributed tail"); all KS groups' tiles pass through LDS
  static constexpr bool DIST = (MT * NT == 1) && KS > 1 && !SPEC;
  static constexpr int REDN = (DIST ? KS : KS - 1) * WM * WN * MT * NT * 1024;   // floats: cross-group accumulator reduction
  static constexpr int BUF0 = ((SPEC ? 2 : 1) * STAGE) > REDN ? ((SPEC ? 2 : 1) * STAGE) : REDN;
  // wide mask epilogue (see fprop_kernel): one 32 x 36 float transposition tile per wave, in the staging area after the K loop
  // (forward 3x3x3 tiles with two epilogue waves -- the 16^3 layers -- measured 1.8 us slower in the wide form: direct form kept)
  static constexpr bool WIDE = (EPI == EPI_MASK_STORE || EPI == EPI_MASK_ACCUM || (EPI == EPI_STORE_STATS && !(TAPS == 27 && WM * WN > 1))) && !SPEC;
  static constexpr int TSTRIDE = 36;
  static constexpr int TRANS = WIDE ? WM * WN * 32 * TSTRIDE : 0;
  static constexpr int BUF = BUF0 > TRANS ? BUF0 : TRANS;
  static size_t smem_bytes(int Cin) {
    int cpad = ((Cin + KC - 1) / KC) * KC;
    size_t ncoef = ((size_t)NCOEF * cpad + 3) & ~(size_t)3;   // keep the staging buffers 16-byte aligned
    return sizeof(float) * (ncoef + (size_t)BUF + (size_t)ECOEF * M_B + 4);   // + arrival ticket of the cross-block K-split
  }
};

template <int PRO>
__device__ __forceinline__ float pro_apply(const float* coef, int cpad, int c, float x0, float x1) {
  if (PRO == PRO_BNRELU) return fmaxf(fmaf(coef[c], x0, coef[cpad + c]), 0.f);
  if (PRO == PRO_GRAD) return fmaf(coef[c], x0, fmaf(coef[cpad + c], x1, coef[2 * cpad + c]));
  return x0;
}

// Sum each of R (2, 4 or 8) per-lane values over the 32 lanes of a wave half (half_reduce16 for fewer registers): log2(R) exchange-and-halve
// steps, then plain butterflies.  On return every lane of a half holds the total of register index (l31 >> (5 - log2 R)).
template <int R>
__device__ __forceinline__ float half_reduce_n(const float (&v)[R], int lane) {
  static_assert(R == 2 || R == 4 || R == 8, "half_reduce_n: 2, 4 or 8 registers");
  float w[R];
#pragma unroll
  for (int r = 0; r < R; ++r) w[r] = v[r];
  if (R >= 8) {
    const bool b = lane & 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float send = b ? w[r] : w[r + 4], keep = b ? w[r + 4] : w[r]; w[r] = keep + swz_xor<16>(send); }
  }
  if (R >= 4) {
    constexpr int X = (R == 8) ? 8 : 16;
    const bool b = lane & X;
#pragma unroll
    for (int r = 0; r < 2; ++r) { const float send = b ? w[r] : w[r + 2], keep = b ? w[r + 2] : w[r]; w[r] = keep + swz_xor<X>(send); }
  }
  {
    constexpr int X = (R == 8) ? 4 : (R == 4 ? 8 : 16);
    const bool b = lane & X;
    const float send = b ? w[0] : w[1], keep = b ? w[1] : w[0];
    w[0] = keep + swz_xor<X>(send);
  }
  float t = w[0];
  if (R <= 2) t += swz_xor<8>(t);
  if (R <= 4) t += swz_xor<4>(t);
  t += swz_xor<2>(t);
  t += swz_xor<1>(t);
  return t;
}

template <int TAPS, int PRO, int EPI, int WM, int WN, int KS, int MT, int NT, int KC, int TD, int TH, int TW, bool SPEC = false>
__global__ void __launch_bounds__(WM* WN* KS * 64 * (SPEC ? 2 : 1)) fprop_kernel(const FpropArgs a) {
  using C = FpropCfg<TAPS, PRO, EPI, WM, WN, KS, MT, NT, KC, TD, TH, TW, SPEC>;
  constexpr int NL = C::NCOMPUTE;                         // threads that take part in staging (fast path)
  constexpr int NTHREADS = C::NTHREADS, M_B = C::M_B, V_B = C::V_B, RS = C::RS, HS = C::HS, DS = C::DS, XS = C::XS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
#if defined(MMNN_PHASE_TRACE)   // developer builds only (MMNN_PHASE_TRACE=1 python -m mmnn_sts_amd.build): costs ~11 registers
  const bool tracing = a.trace != nullptr && tid == 0 && blockIdx.x < 64 && blockIdx.y == 0;
#else
  constexpr bool tracing = false;
#endif
  auto stamp = [&](int k) {
    if (tracing) a.trace[(blockIdx.x * (gridDim.z > 1 ? 2 : 1) + (blockIdx.z ? 1 : 0)) % 64 * 16 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  kernarg_warm<sizeof(FpropArgs)>();
  stamp(13);
  float pf_sink = 0.f, pf_sink2 = 0.f;                   // keep the prefetch loads alive (see FpropArgs::pf_ptr)
  const bool loader = SPEC && tid >= NL;                 // wave-uniform role
  const int ltid = loader ? tid - NL : tid;              // index within the role's thread set
  const int wave = ltid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int kg = wave / (WM * WN);                      // K-split group of this wave
  const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;
  const int V = a.D * a.H * a.W;
  const int cpad = ((a.Cin + KC - 1) / KC) * KC;

  float* coef = smem;
  float* Xs = coef + ((C::NCOEF * cpad + 3) & ~3);
  float* Ws = Xs + KC * XS;                              // buffer 0; buffer 1 (SPEC) follows at + C::STAGE
  float* ecoef = Xs + C::BUF;

  // ---- which tile ----
  int n, d0 = 0, h0 = 0, w0 = 0, v0_ = 0;
  {
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (blockIdx.x & 7 = XCD), each with an L2 of its own.
    // Giving XCD x the x-th CONTIGUOUS eighth of the tile sequence makes neighbouring tiles -- which share halo rows and, for
    // the 1x1x1 kernels, nothing -- meet in the same L2 instead of each pulling the halo from HBM / the Infinity Cache again.
    int b = blockIdx.x;
    // (Applying the same order to the 1x1x1 kernels, so that producer and consumer tiles share an XCD, measured no gain: r02.)
    if (TAPS == 27 && (gridDim.x & 7) == 0) b = (b & 7) * (int)(gridDim.x >> 3) + (b >> 3);
    auto divmod = [](int& x, int d, unsigned magic) {      // x <- x / d, returns x % d
      const int q = magic ? (int)__umulhi((unsigned)x, magic) : x;
      const int r = x - q * d;
      x = q;
      return r;
    };
    if (TAPS == 27) {
      const int nw = (a.W + TW - 1) / TW, nh = (a.H + TH - 1) / TH, nd = (a.D + TD - 1) / TD;
      w0 = divmod(b, nw, a.mg_w) * TW;
      h0 = divmod(b, nh, a.mg_h) * TH;
      d0 = divmod(b, nd, a.mg_d) * TD;
      n = b;
    } else {
      const int nt = (V + V_B - 1) / V_B;
      v0_ = divmod(b, nt, a.mg_w) * V_B;
      n = b;
    }
  }
  const int m0 = blockIdx.y * M_B;
  const int rep = blockIdx.x & ((a.nrep > 0 ? a.nrep : NREP) - 1);
  // channel slice of this block (cross-block K-split): whole chunks [c_begin, c_end)
  // (kz is a power of two -- launch_cfg doubles it -- so the slice bounds are shifts: the 64-bit run-time division this used to be
  // is ~100 dependent instructions, twice, ahead of the first load of every launch)
  const int kz = gridDim.z, nch_all = (a.Cin + KC - 1) / KC;
  const int kz_sh = __builtin_ctz((unsigned)kz);
  const int c_begin = ((nch_all * (int)blockIdx.z) >> kz_sh) * KC;
  const int c_end = min(a.Cin, ((nch_all * ((int)blockIdx.z + 1)) >> kz_sh) * KC);

  // ---- small tiles: the loads the coefficients need go out FIRST (a handful per thread), the first chunks' operand loads follow,
  // and the coefficient arithmetic (prologue_fast below) runs when the statistics are back -- memory returns in order, so it no
  // longer waits behind the 50-110 KB of operands, and the operands no longer wait for it: one memory round trip where the phase
  // trace of r03 showed three (tools/phase_trace.py: "prologue" 3.6k + "1st store" 3.3k cycles of a 27k-cycle 8^3 launch). ----
  constexpr bool EARLY_K = (MT * NT == 1) || (TAPS == 27 && TW <= 16);          // the latency-bound tile shapes (see EARLY below)
  constexpr bool MASK_E = (EPI == EPI_MASK_STORE || EPI == EPI_MASK_ACCUM);
  static_assert(!EARLY_K || M_B <= NTHREADS, "one epilogue row per thread");
  const int c_hi = min(cpad, c_begin + ((c_end - c_begin + KC - 1) / KC) * KC);   // end of this slice's coefficient range
  bool fastp = EARLY_K && (c_hi - c_begin) <= NTHREADS;
  if (PRO == PRO_BNRELU) fastp = fastp && a.bn_in.training && a.bn_in.st.nrep <= 2;
  if (PRO == PRO_GRAD) fastp = fastp && a.gr_in.st.nrep <= 2 && a.gr_in.s.nrep <= 2;
  if (MASK_E) fastp = fastp && a.ebn.training && a.ebn.st.nrep <= 2;
  BnFwdRaw raw_in, raw_e;
  BnBwdRaw raw_gr;
  if (EARLY_K && fastp) {
    // Only the waves that own a channel / a row load (a slice is 16-128 channels of a 512-thread block): memory returns in order, and
    // eight waves' worth of redundant statistic loads ahead of the operand loads cost more than they hide.  Indices clamped inside a
    // wave, values of padding channels discarded in prologue_fast.
    const int pc = max(0, min(c_begin + tid, a.Cin - 1)), pm = max(0, min(m0 + tid, a.M - 1));
    if (PRO != PRO_NONE && c_begin + (tid & ~63) < c_hi) {
      if (PRO == PRO_BNRELU) bn_fwd_issue(a.bn_in, pc, raw_in);
      if (PRO == PRO_GRAD) bn_bwd_issue(a.gr_in, pc, raw_gr);
    }
    if (MASK_E && (tid & ~63) < M_B) bn_fwd_issue(a.ebn, pm, raw_e);
  }
  stamp(14);
  auto prologue_fast = [&]() {
    const int c = c_begin + tid;
    if (c < c_hi) {
      if (PRO == PRO_BNRELU) {
        float ca = 0.f, cb = 0.f, mu, rs;
        bn_fwd_finish(a.bn_in, raw_in, ca, cb, mu, rs);
        const bool ok = c < a.Cin;
        coef[c] = ok ? ca : 0.f; coef[cpad + c] = ok ? cb : 0.f;
      } else if (PRO == PRO_GRAD) {
        float p = 0.f, q = 0.f, r = 0.f;
        bn_bwd_finish(a.gr_in, raw_gr, p, q, r);
        const bool ok = c < a.Cin;
        const float sc = drop_scale(a.drop_in, n, c);
        coef[c] = ok ? p * sc : 0.f; coef[cpad + c] = ok ? q * sc : 0.f; coef[2 * cpad + c] = ok ? r * sc : 0.f;
      }
    }
    if (tid < M_B) {
      const int m = tid;
      float ea = 0.f, eb = 0.f, mu = 0.f, rs = 0.f, g = 0.f, ds = 1.f;
      if (m0 + m < a.M) {
        if (MASK_E) {
          bn_fwd_finish(a.ebn, raw_e, ea, eb, mu, rs);
          g = raw_e.g;
        }
        if (EPI == EPI_STORE_STATS) ds = drop_scale(a.drop_out, n, m0 + m);
      }
      ecoef[m] = ea; ecoef[M_B + m] = eb; ecoef[2 * M_B + m] = mu; ecoef[3 * M_B + m] = rs;
      ecoef[4 * M_B + m] = g; ecoef[5 * M_B + m] = ds;
    }
    __syncthreads();
  };

  // ---- per-channel prologue coefficients, per-row epilogue coefficients.  Called AFTER the first chunk's global loads have
  // been issued (fast path): the coefficients cost a memory round trip of their own (fp64 statistics) and are only needed when
  // the chunk is written to LDS, so the two latencies overlap instead of adding up -- the small-extent layers are a chain of
  // such latencies and little else. ----
  auto prologue = [&]() {
    if (PRO == PRO_BNRELU) {
      for (int c = c_begin + tid; c < min(cpad, c_begin + ((c_end - c_begin + KC - 1) / KC) * KC); c += NTHREADS) {
        float ca = 0.f, cb = 0.f, mu, rs;
        if (c < a.Cin) bn_fwd_coef(a.bn_in, c, ca, cb, mu, rs);
        coef[c] = ca; coef[cpad + c] = cb;
      }
    } else if (PRO == PRO_GRAD) {
      for (int c = c_begin + tid; c < min(cpad, c_begin + ((c_end - c_begin + KC - 1) / KC) * KC); c += NTHREADS) {
        float p = 0.f, q = 0.f, r = 0.f;
        if (c < a.Cin) {
          bn_bwd_coef(a.gr_in, c, p, q, r);
          const float s = drop_scale(a.drop_in, n, c);
          p *= s; q *= s; r *= s;
        }
        coef[c] = p; coef[cpad + c] = q; coef[2 * cpad + c] = r;
      }
    }
    for (int m = tid; m < M_B; m += NTHREADS) {
      float ea = 0.f, eb = 0.f, mu = 0.f, rs = 0.f, g = 0.f, ds = 1.f;
      if (m0 + m < a.M) {
        if (EPI == EPI_MASK_STORE || EPI == EPI_MASK_ACCUM) {
          bn_fwd_coef(a.ebn, m0 + m, ea, eb, mu, rs);
          g = a.ebn.gamma[m0 + m];
        }
        if (EPI == EPI_STORE_STATS) ds = drop_scale(a.drop_out, n, m0 + m);
      }
      ecoef[m] = ea; ecoef[M_B + m] = eb; ecoef[2 * M_B + m] = mu; ecoef[3 * M_B + m] = rs;
      ecoef[4 * M_B + m] = g; ecoef[5 * M_B + m] = ds;
    }
    __syncthreads();
  };

  // ---- accumulators and per-lane voxel positions ----
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int pos[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = (wn * NT + j) * 32 + l31;
    if (TAPS == 27) {
      const int wx = t % TW, hy = (t / TW) % TH, dz = t / (TW * TH);
      pos[j] = (dz * HS + hy) * RS + wx + 3;
    } else {
      pos[j] = t;
    }
  }

  const float* in0n = a.in0 + (long)n * a.in0_ns + (long)a.in0_coff * V;
  const float* in1n = (PRO == PRO_GRAD) ? a.in1 + (long)n * a.in1_ns + (long)a.in1_coff * V : nullptr;
  const bool al_x = (((uintptr_t)in0n | (uintptr_t)in1n) & 15) == 0;
  const bool vecx = al_x && ((TAPS == 27) ? ((a.W & 3) == 0) : ((V & 3) == 0));
  const bool vecw = ((a.w_ld & 3) == 0) && ((a.M & 3) == 0) && (((uintptr_t)a.w & 15) == 0);

  // ---- MFMA over one staged chunk.  The operand reads of step s+1 are issued before the MFMAs of step s (two register
  // sets), so that the LDS latency hides behind the matrix pipe even with a single wave per SIMD. ----
  auto mfma_chunk = [&](int boff = 0) {
    const float* xb = Xs + boff + (2 * kg + half) * XS;
    const float* wb = Ws + boff + (2 * kg + half) * TAPS * M_B + wm * MT * 32 + l31;
    constexpr int PAIR = 2 * KS;                 // channel distance between consecutive pairs of one wave group
    constexpr int NSTEP = (KC / 2 / KS) * TAPS;
    auto rd = [&](int st, float (&av)[MT], float (&bv)[NT]) {
      const int jj = st / TAPS, tap = st % TAPS;
      const int toff = (TAPS == 27) ? (((tap / 9) * HS + (tap / 3) % 3) * RS + tap % 3) : 0;
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = wb[(jj * PAIR * TAPS + tap) * M_B + i * 32];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = xb[jj * PAIR * XS + pos[j] + toff];
    };
    auto mm = [&](const float (&av)[MT], const float (&bv)[NT]) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    };
    float a0[MT], b0[NT], a1[MT], b1[NT];
    rd(0, a0, b0);
#pragma unroll
    for (int st = 0; st < NSTEP; st += 2) {
      if (st + 1 < NSTEP) rd(st + 1, a1, b1);
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);   // DS reads of the next step first ...
      mm(a0, b0);
      __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);   // ... then this step's MFMAs
      if (st + 2 < NSTEP) rd(st + 2, a0, b0);
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
      if (st + 1 < NSTEP) mm(a1, b1);
      __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
    }
  };

  // ---- output positions of this lane, and (MASK_STORE) the epilogue's operand: the pre-activation tensor `ex`, whole tile.  In the
  // single-slice fast path its loads CAN be issued before the MFMAs of the last chunk (XPRE below): every block of a launch
  // reaches its epilogue at the same time, and the chip-wide burst of these reads is 40k of a block's 290k cycles in block 1's
  // data gradient (phase trace r02), matrix pipe idle. ----
  float* outn = a.out + (long)n * a.out_ns + (long)a.out_coff * V;
  const float* exn = (EPI == EPI_MASK_STORE || EPI == EPI_MASK_ACCUM) ? a.ex + (long)n * a.ex_ns + (long)a.ex_coff * V : nullptr;
  constexpr bool ALL_ROWS = (EPI == EPI_MASK_STORE);
  float xall[ALL_ROWS ? MT : 1][ALL_ROWS ? 16 : 1][NT];
  bool xall_loaded = false;
  auto load_xall = [&]() {
    if (ALL_ROWS) {
      int vox[NT];      // (recomputed in the epilogue: kept live across the K loop these cost the small-tile kernels their last registers)
      bool vok[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int t = (wn * NT + j) * 32 + l31;
        if (TAPS == 27) {
          const int wx = t % TW, hy = (t / TW) % TH, dz = t / (TW * TH);
          const int d = d0 + dz, h = h0 + hy, w = w0 + wx;
          vok[j] = d < a.D && h < a.H && w < a.W;
          vox[j] = (d * a.H + h) * a.W + w;
        } else {
          vox[j] = v0_ + t;
          vok[j] = vox[j] < V;
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ml = wm * MT * 32 + i * 32 + acc_row(q, half);
          const bool mok = (m0 + ml) < a.M;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const unsigned o = (mok && vok[j]) ? (unsigned)(m0 + ml) * (unsigned)V + (unsigned)vox[j] : 0u;   // unconditional loads (clamped)
            xall[ALL_ROWS ? i : 0][ALL_ROWS ? q : 0][j] = exn[o];
          }
        }
    }
    xall_loaded = true;
  };
  // r02 RESULT: measured SLOWER on the one kernel it targets (block 1's conv2 data gradient 145 -> 167 us: the 64 loads live
  // across the last chunk push the kernel from 156 to 256 registers), so it is switched off; the smaller tiles have no registers
  // to spare either.  Kept as a compile-time switch for a version that holds the operand in LDS instead.
  constexpr bool XPRE = false && ALL_ROWS && TAPS == 27 && TW > 16 && MT * NT > 1 && !SPEC;
  const bool xall_early = XPRE && gridDim.z == 1 && kg == 0 && !loader;     // wave-uniform

  if (vecx && vecw && (long)KC * V < (1l << 30)) {
    // ===== fast path: 16-byte staging through registers, software-pipelined.  The global loads of chunk k+1 are issued
    // (all at once) before the MFMA loop of chunk k and only waited for when they are written to LDS afterwards. =====
    constexpr int ROWS = (TAPS == 27) ? KC * DS * HS : KC;                   // staged rows (channel x halo row)
    constexpr int VPR = (TAPS == 27) ? TW / 4 : V_B / 4;                      // 16-byte items per row
    constexpr int XV_ITEMS = ROWS * VPR;
    constexpr int XH_ITEMS = (TAPS == 27) ? ROWS * 2 : 0;                     // halo columns: one float each
    constexpr int W_ITEMS = KC * TAPS * (M_B / 4);
    constexpr int XV_IT = (XV_ITEMS + NL - 1) / NL, XH_IT = (XH_ITEMS + NL - 1) / NL;
    constexpr int W_IT = (W_ITEMS + NL - 1) / NL;
    static_assert(XV_IT <= 32 && XH_IT <= 32 && W_IT <= 32, "validity bits of the staged items are kept in 32-bit masks");
    constexpr bool GR = (PRO == PRO_GRAD);
    // small tiles (one accumulator per wave) are latency-bound in this loop: keep TWO chunks of loads in flight
    constexpr int PF = (MT * NT == 1) ? 2 : 1;
    struct Stage {
      f32x4 xv0[XV_IT], xv1[GR ? XV_IT : 1], wr[W_IT];
      float xh0[XH_IT > 0 ? XH_IT : 1], xh1[(GR && XH_IT > 0) ? XH_IT : 1];
      unsigned okv, okh, okw;
    };
    Stage stA, stB;

    // Per-item descriptors, computed ONCE: element offset relative to the chunk's first channel (-1: never valid), LDS
    // destination and local channel.  The per-chunk staging code is then one add + one select per item -- with one wave per
    // SIMD every integer instruction spent here is a cycle the matrix pipe idles.
    int xv_off[XV_IT], xv_dst[XV_IT], xv_cl[XV_IT];
    int xh_off[XH_IT > 0 ? XH_IT : 1], xh_dst[XH_IT > 0 ? XH_IT : 1], xh_cl[XH_IT > 0 ? XH_IT : 1];
    int w_off[W_IT], w_cl[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int it = ltid + i * NL;
      const int q = it % (M_B / 4), kr = it / (M_B / 4);
      const bool ok = (it < W_ITEMS) && (m0 + 4 * q < a.M);
      w_off[i] = ok ? kr * a.w_ld + m0 + 4 * q : -1;
      w_cl[i] = kr / TAPS;
    }
    const bool w_nomask = (a.Cin % KC == 0) && (m0 + M_B <= a.M);   // block-uniform: no padded channel or row in any weight chunk
    auto load_w = [&](int c0, Stage& st) {
      const int crem = a.Cin - c0;
      const float* bw = a.w + (long)c0 * TAPS * a.w_ld;
      unsigned okw = 0;
#pragma unroll
      for (int i = 0; i < W_IT; ++i) {
        const bool ok = w_off[i] >= 0 && w_cl[i] < crem;
        okw |= (ok ? 1u : 0u) << i;
        st.wr[i] = *reinterpret_cast<const f32x4*>(ok ? bw + w_off[i] : a.w);   // masked when written to LDS: a select on the
      }                                                                          // loaded value here would wait for the load
      st.okw = okw;
    };
    // Small tiles: the first chunk's weight loads go out NOW, before the activation descriptors are worked out -- that set-up
    // arithmetic is ~2 us of the small-extent kernels and now overlaps the (always cold) weight fetch.
    constexpr bool EARLY = (MT * NT == 1) || (TAPS == 27 && TW <= 16);
    const bool first_chunk_mine = (c_begin < c_end) && (!SPEC || loader);
    if (EARLY && first_chunk_mine) load_w(c_begin, stA);
    stamp(15);
    {
      // 32-bit offsets: this path is only taken when KC * V < 2^30 (64-bit multiplies cost four instructions each, and this
      // set-up runs before the first load of every launch: 2.5 us of the 8^3 / 4^3 kernels in the phase trace)
      auto row_info = [&](int r, int& cl, int& lrow, int& gro, bool& rowok) {
        if (TAPS == 27) {
          const int hy = r % HS, dz = (r / HS) % DS;
          cl = r / (HS * DS);
          const int d = d0 + dz - 1, h = h0 + hy - 1;
          rowok = (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H;
          gro = cl * V + (d * a.H + h) * a.W;
          lrow = cl * XS + (dz * HS + hy) * RS;
        } else {
          cl = r;
          rowok = true;
          gro = cl * V;
          lrow = cl * XS;
        }
      };
#pragma unroll
      for (int i = 0; i < XV_IT; ++i) {
        const int it = ltid + i * NL;
        int cl, lrow, gro; bool rowok;
        row_info(it / VPR, cl, lrow, gro, rowok);
        const int col = ((TAPS == 27) ? w0 : v0_) + 4 * (it % VPR);
        const bool ok = (it < XV_ITEMS) && rowok && col < ((TAPS == 27) ? a.W : V);
        xv_off[i] = ok ? gro + col : -1;
        xv_dst[i] = lrow + ((TAPS == 27) ? 4 : 0) + 4 * (it % VPR);
        xv_cl[i] = cl;
      }
#pragma unroll
      for (int i = 0; i < XH_IT; ++i) {
        const int it = ltid + i * NL;
        int cl, lrow, gro; bool rowok;
        row_info(it >> 1, cl, lrow, gro, rowok);
        const int w = (it & 1) ? w0 + TW : w0 - 1;
        const bool ok = (it < XH_ITEMS) && rowok && (unsigned)w < (unsigned)a.W;
        xh_off[i] = ok ? gro + w : -1;
        xh_dst[i] = lrow + ((it & 1) ? TW + 4 : 3);
        xh_cl[i] = cl;
      }
    }

    // NOTE: every load below is UNCONDITIONAL (out-of-range items read element 0 of their tensor and are zeroed when
    // they are written to LDS).  A load under a divergent `if` makes hipcc branch around it and wait vmcnt(0) at the join,
    // which serialises the whole batch (one memory round trip per item instead of one per chunk).
    auto load_x = [&](int c0, Stage& st) {
      unsigned okv = 0, okh = 0;
      const float* b0 = in0n + (long)c0 * V;
      const float* b1 = GR ? in1n + (long)c0 * V : nullptr;
      const int crem = a.Cin - c0;                    // channels left: items of local channel >= crem are padding
#pragma unroll
      for (int i = 0; i < XV_IT; ++i) {
        const bool ok = xv_off[i] >= 0 && xv_cl[i] < crem;
        const int off = ok ? xv_off[i] : 0;
        okv |= (ok ? 1u : 0u) << i;
        st.xv0[i] = *reinterpret_cast<const f32x4*>(ok ? b0 + off : in0n);
        if (GR) st.xv1[i] = *reinterpret_cast<const f32x4*>(ok ? b1 + off : in1n);
      }
#pragma unroll
      for (int i = 0; i < XH_IT; ++i) {
        const bool ok = xh_off[i] >= 0 && xh_cl[i] < crem;
        const int off = ok ? xh_off[i] : 0;
        okh |= (ok ? 1u : 0u) << i;
        st.xh0[i] = *(ok ? b0 + off : in0n);
        if (GR) st.xh1[i] = *(ok ? b1 + off : in1n);
      }
      st.okv = okv; st.okh = okh;
    };

    auto load_chunk = [&](int c0, Stage& st) { load_x(c0, st); load_w(c0, st); };
    auto load_first = [&](Stage& st) { if (EARLY) load_x(c_begin, st); else load_chunk(c_begin, st); };   // EARLY: weights already in flight

    auto store_chunk = [&](int c0, const Stage& st, int boff = 0) {
      const unsigned okv = st.okv, okh = st.okh;
#pragma unroll
      for (int i = 0; i < XV_IT; ++i) {
        if (ltid + i * NL < XV_ITEMS) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pro_apply<PRO>(coef, cpad, c0 + xv_cl[i], st.xv0[i][e], GR ? st.xv1[i][e] : 0.f);
          if (!((okv >> i) & 1u)) o = f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(Xs + boff + xv_dst[i]) = o;
        }
      }
#pragma unroll
      for (int i = 0; i < XH_IT; ++i) {
        if (ltid + i * NL < XH_ITEMS) {
          const float o = pro_apply<PRO>(coef, cpad, c0 + xh_cl[i], st.xh0[i], GR ? st.xh1[i] : 0.f);
          Xs[boff + xh_dst[i]] = ((okh >> i) & 1u) ? o : 0.f;
        }
      }
      if (w_nomask) {   // uniform branch: the selects below are vector instructions, and those come out of the fp32 matrix rate
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
          const int it = ltid + i * NL;
          if (it < W_ITEMS) *reinterpret_cast<f32x4*>(Ws + boff + 4 * it) = st.wr[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
          const int it = ltid + i * NL;
          if (it < W_ITEMS) *reinterpret_cast<f32x4*>(Ws + boff + 4 * it) = ((st.okw >> i) & 1u) ? st.wr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    };

    // Small tiles (the 16^3 .. 4^3 layers) are latency chains: issue the first chunk's loads, THEN compute the coefficients.  The
    // big tiles of block 1 amortise the prologue over a long K loop and cannot afford the registers (staging registers live
    // across the fp64 coefficient math: 116 -> 188 VGPRs, one register short of losing the second wave per SIMD).
    if (!EARLY) prologue();
    // One dword per 128-byte line of the NEXT launch's weights, at most two lines per thread, addresses clamped instead of guarded
    // and the values consumed only at the very end of the kernel: nothing ever waits for these loads.  Issued after the first
    // chunk's loads and the coefficient loads (memory returns in order: a cold weight line ahead of them would hold them back).
    auto prefetch_next_weights = [&]() {
      const unsigned lines = a.pf_bytes >> 7;
      if (a.pf_ptr == nullptr || lines == 0) return;
      const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);      // dispatch order: XCD = lin & 7
      const unsigned per_xcd = (gridDim.x * gridDim.y * gridDim.z + 7u) >> 3;
      const unsigned ln0 = (lin >> 3) * NTHREADS + tid, ln1 = ln0 + per_xcd * NTHREADS;
      pf_sink = a.pf_ptr[(size_t)min(ln0, lines - 1) * 32];
      pf_sink2 = a.pf_ptr[(size_t)min(ln1, lines - 1) * 32];
    };
    if (SPEC) {
      // loader waves fill buffer (k+1)&1 while compute waves consume buffer k&1; one barrier per chunk
      if (loader && c_begin < c_end) load_first(stA);
      stamp(1);
      if (EARLY) { if (fastp) prologue_fast(); else prologue(); }
      prefetch_next_weights();
      stamp(2);
      if (loader && c_begin < c_end) store_chunk(c_begin, stA, 0);
      __syncthreads();
      stamp(9);
      int k = 0;
      for (int c0 = c_begin; c0 < c_end; c0 += KC, ++k) {
        if (loader) {
          if (c0 + KC < c_end) { load_chunk(c0 + KC, stA); store_chunk(c0 + KC, stA, ((k + 1) & 1) * C::STAGE); }
        } else {
          mfma_chunk((k & 1) * C::STAGE);
        }
        __syncthreads();
      }
    } else if (PF == 1) {
      if (c_begin < c_end) load_first(stA);
      stamp(1);
      if (EARLY) { if (fastp) prologue_fast(); else prologue(); }
      prefetch_next_weights();
      stamp(2);
      for (int c0 = c_begin; c0 < c_end; c0 += KC) {
        store_chunk(c0, stA);
        __syncthreads();
        if (c0 == c_begin) stamp(9);
        if (c0 + KC < c_end) load_chunk(c0 + KC, stA);
        else if (XPRE && xall_early) load_xall();
        mfma_chunk();
        __syncthreads();
      }
    } else {
      if (c_begin < c_end) load_first(stA);
      if (c_begin + KC < c_end) load_chunk(c_begin + KC, stB);
      stamp(1);
      if (EARLY) { if (fastp) prologue_fast(); else prologue(); }
      prefetch_next_weights();
      stamp(2);
      for (int c0 = c_begin; c0 < c_end; c0 += 2 * KC) {
        store_chunk(c0, stA);
        __syncthreads();
        if (c0 == c_begin) stamp(9);
        if (c0 + 2 * KC < c_end) load_chunk(c0 + 2 * KC, stA);
        mfma_chunk();
        __syncthreads();
        if (c0 + KC < c_end) {
          store_chunk(c0 + KC, stB);
          __syncthreads();
          if (c0 + 3 * KC < c_end) load_chunk(c0 + 3 * KC, stB);
          mfma_chunk();
          __syncthreads();
        }
      }
    }
  } else {
  prologue();
  for (int c0 = c_begin; c0 < c_end; c0 += KC) {
    // ================= stage activations =================
    if (TAPS == 27) {
      if (vecx) {
        constexpr int IPR = TW / 4 + 2;
        constexpr int ITEMS = KC * DS * HS * IPR;
        for (int it = tid; it < ITEMS; it += NTHREADS) {
          const int q = it % IPR;
          int row = it / IPR;
          const int hy = row % HS; row /= HS;
          const int dz = row % DS;
          const int cl = row / DS;
          const int c = c0 + cl;
          const int d = d0 + dz - 1, h = h0 + hy - 1;
          const bool rowok = (c < a.Cin) && (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H;
          float* dst = Xs + cl * XS + (dz * HS + hy) * RS;
          const long gro = (long)c * V + ((long)d * a.H + h) * a.W;
          if (q < TW / 4) {
            const int w = w0 + 4 * q;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (rowok && w < a.W) {
              f32x4 x0 = *reinterpret_cast<const f32x4*>(in0n + gro + w);
              f32x4 x1 = x0;
              if (PRO == PRO_GRAD) x1 = *reinterpret_cast<const f32x4*>(in1n + gro + w);
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = pro_apply<PRO>(coef, cpad, c, x0[e], x1[e]);
            }
            *reinterpret_cast<f32x4*>(dst + 4 + 4 * q) = o;
          } else {
            const int w = (q == TW / 4) ? w0 - 1 : w0 + TW;
            float o = 0.f;
            if (rowok && (unsigned)w < (unsigned)a.W) {
              const float x0 = in0n[gro + w];
              const float x1 = (PRO == PRO_GRAD) ? in1n[gro + w] : x0;
              o = pro_apply<PRO>(coef, cpad, c, x0, x1);
            }
            dst[(q == TW / 4) ? 3 : TW + 4] = o;
          }
        }
      } else {
        constexpr int IPR = TW + 2;
        constexpr int ITEMS = KC * DS * HS * IPR;
        for (int it = tid; it < ITEMS; it += NTHREADS) {
          const int q = it % IPR;
          int row = it / IPR;
          const int hy = row % HS; row /= HS;
          const int dz = row % DS;
          const int cl = row / DS;
          const int c = c0 + cl;
          const int d = d0 + dz - 1, h = h0 + hy - 1, w = w0 + q - 1;
          float o = 0.f;
          if ((c < a.Cin) && (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) {
            const long g = (long)c * V + ((long)d * a.H + h) * a.W + w;
            const float x0 = in0n[g];
            const float x1 = (PRO == PRO_GRAD) ? in1n[g] : x0;
            o = pro_apply<PRO>(coef, cpad, c, x0, x1);
          }
          Xs[cl * XS + (dz * HS + hy) * RS + 3 + q] = o;
        }
      }
    } else {
      if (vecx) {
        constexpr int ITEMS = KC * V_B / 4;
        for (int it = tid; it < ITEMS; it += NTHREADS) {
          const int q = it % (V_B / 4);
          const int cl = it / (V_B / 4);
          const int c = c0 + cl;
          const int v = v0_ + 4 * q;
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          if (c < a.Cin && v < V) {
            const long g = (long)c * V + v;
            f32x4 x0 = *reinterpret_cast<const f32x4*>(in0n + g);
            f32x4 x1 = x0;
            if (PRO == PRO_GRAD) x1 = *reinterpret_cast<const f32x4*>(in1n + g);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pro_apply<PRO>(coef, cpad, c, x0[e], x1[e]);
          }
          *reinterpret_cast<f32x4*>(Xs + cl * XS + 4 * q) = o;
        }
      } else {
        constexpr int ITEMS = KC * V_B;
        for (int it = tid; it < ITEMS; it += NTHREADS) {
          const int q = it % V_B;
          const int cl = it / V_B;
          const int c = c0 + cl;
          const int v = v0_ + q;
          float o = 0.f;
          if (c < a.Cin && v < V) {
            const long g = (long)c * V + v;
            const float x0 = in0n[g];
            const float x1 = (PRO == PRO_GRAD) ? in1n[g] : x0;
            o = pro_apply<PRO>(coef, cpad, c, x0, x1);
          }
          Xs[cl * XS + q] = o;
        }
      }
    }
    // ================= stage weights =================
    {
      const int krows = KC * TAPS;
      const long kbase = (long)c0 * TAPS;
      const long klim = (long)a.Cin * TAPS;
      if (vecw) {
        const int items = krows * (M_B / 4);
        for (int it = tid; it < items; it += NTHREADS) {
          const int q = it % (M_B / 4);
          const int kr = it / (M_B / 4);
          const int m = m0 + 4 * q;
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          if (kbase + kr < klim && m < a.M) o = *reinterpret_cast<const f32x4*>(a.w + (kbase + kr) * a.w_ld + m);
          *reinterpret_cast<f32x4*>(Ws + kr * M_B + 4 * q) = o;
        }
      } else {
        const int items = krows * M_B;
        for (int it = tid; it < items; it += NTHREADS) {
          const int q = it % M_B;
          const int kr = it / M_B;
          float o = 0.f;
          if (kbase + kr < klim && m0 + q < a.M) o = a.w[(kbase + kr) * a.w_ld + m0 + q];
          Ws[kr * M_B + q] = o;
        }
      }
    }
    __syncthreads();
    if (!loader) mfma_chunk();
    __syncthreads();
  }
  }

  stamp(3);
  float* red0 = ecoef + 6 * M_B;            // [WN][M_B] partial sums, one writer per slot (no LDS atomics: reproducible)
  float* red1 = red0 + WN * M_B;
  const bool want_sums = (EPI == EPI_STORE_STATS) ? (a.st_out.sum != nullptr) : (EPI != EPI_STORE);
  if constexpr (C::DIST) {
    // ================= distributed tail of the small tiles (r03).  One accumulator tile per wave, the K axis split over KS wave
    // groups.  The r02 tail funnelled everything through the group-0 waves: they summed (KS - 1) x 16 rows from LDS, published 16 rows,
    // summed the K-split slices and ran the whole epilogue while seven of eight waves waited (phase trace: 1.3k + 3.4k + 2.2k + 3.7k
    // cycles of a 23k-cycle 8^3 launch).  Here every group keeps 16 / KS rows of its tile from the in-block reduction through the
    // cross-workgroup hand-off to the stores; sums are taken in the same order as before (group order, then slice order). =======
    constexpr int RP = 16 / KS, NSLOT = WM * WN;
    constexpr bool MASK = (EPI == EPI_MASK_STORE || EPI == EPI_MASK_ACCUM);
    float* rbuf = Xs;                                   // staging buffers are free after the last barrier
    const int slot = wm * WN + wn;
#pragma unroll
    for (int r = 0; r < 16; ++r) rbuf[((kg * NSLOT + slot) * 16 + r) * 64 + lane] = acc[0][0][r];
    __syncthreads();
    float my[RP];
#pragma unroll
    for (int q = 0; q < RP; ++q) {
      float t = rbuf[((0 * NSLOT + slot) * 16 + kg * RP + q) * 64 + lane];
#pragma unroll
      for (int g = 1; g < KS; ++g) t += rbuf[((g * NSLOT + slot) * 16 + kg * RP + q) * 64 + lane];
      my[q] = t;
    }
    stamp(4);
    if (kz > 1) {
      // cross-workgroup K-split, as in the general tail below (write-through partials, drained, ticket; the last arriver reads
      // with sc1 loads) -- but every wave publishes and sums its own rows
      const long tile_id = blockIdx.x + (long)gridDim.x * blockIdx.y;
      float* part = a.kz_part + (tile_id * kz) * NSLOT * 1024;
#pragma unroll
      for (int q = 0; q < RP; ++q) {
        float* dst = part + (((long)blockIdx.z * NSLOT + slot) * 16 + kg * RP + q) * 64 + lane;
#if MMNN_KZ_FENCED
        *dst = my[q];
#else
        __hip_atomic_store(dst, my[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_store_dword ... sc1
#endif
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores before the barrier
      __syncthreads();
      unsigned* ticket = reinterpret_cast<unsigned*>(ecoef + C::ECOEF * M_B);
      if (tid == 0) {
#if MMNN_KZ_FENCED
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // cumulative over the barrier: publishes the whole workgroup's stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        *ticket = __hip_atomic_fetch_add(a.kz_cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      stamp(5);
      const unsigned arrived = *ticket;
      if (arrived != (unsigned)(kz - 1)) return;            // not the last slice of this tile: done
      if (tid == 0) __hip_atomic_store(a.kz_cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
#if MMNN_KZ_FENCED
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
      constexpr int ZB = (RP <= 4) ? 8 : 4;               // slices in flight: at most 32 loads per lane
#pragma unroll
      for (int q = 0; q < RP; ++q) my[q] = 0.f;
      for (int z0 = 0; z0 < kz; z0 += ZB) {
        float pz[ZB][RP];
#pragma unroll
        for (int z = 0; z < ZB; ++z) {
          const int zc = min(z0 + z, kz - 1);
#pragma unroll
          for (int q = 0; q < RP; ++q) {
            const float* sp = part + (((long)zc * NSLOT + slot) * 16 + kg * RP + q) * 64 + lane;
#if MMNN_KZ_FENCED
            pz[z][q] = *sp;
#else
            pz[z][q] = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // global_load_dword ... sc1
#endif
          }
        }
#pragma unroll
        for (int z = 0; z < ZB; ++z)
#pragma unroll
          for (int q = 0; q < RP; ++q) my[q] += (z0 + z < kz) ? pz[z][q] : 0.f;
      }
    }
    stamp(6);
    // ---- epilogue of this wave's rows: register q holds row acc_row(kg * RP + q, half) of tile (wm, wn), voxel l31 ----
    int vox; bool vok;
    {
      const int t = wn * 32 + l31;
      if (TAPS == 27) {
        const int wx = t % TW, hy = (t / TW) % TH, dz = t / (TW * TH);
        const int d = d0 + dz, h = h0 + hy, w = w0 + wx;
        vok = d < a.D && h < a.H && w < a.W;
        vox = (d * a.H + h) * a.W + w;
      } else {
        vox = v0_ + t;
        vok = vox < V;
      }
    }
    int ml[RP]; bool ok[RP]; unsigned off[RP];
    float xe[MASK ? RP : 1], go[(EPI == EPI_MASK_ACCUM) ? RP : 1];
#pragma unroll
    for (int q = 0; q < RP; ++q) {
      ml[q] = wm * 32 + acc_row(kg * RP + q, half);
      ok[q] = vok && (m0 + ml[q]) < a.M;
      off[q] = ok[q] ? (unsigned)(m0 + ml[q]) * (unsigned)V + (unsigned)vox : 0u;     // unconditional loads (clamped), before any store (aliasing)
      if (MASK) xe[MASK ? q : 0] = exn[off[q]];
      if (EPI == EPI_MASK_ACCUM) go[(EPI == EPI_MASK_ACCUM) ? q : 0] = outn[off[q]];
    }
    float s0[RP], s1[RP];
#pragma unroll
    for (int q = 0; q < RP; ++q) {
      float val = my[q], t0 = 0.f, t1 = 0.f;
      if (EPI == EPI_STORE) {
        if (ok[q]) outn[off[q]] = val;
      } else if (EPI == EPI_STORE_STATS) {
        val *= ecoef[5 * M_B + ml[q]];
        if (ok[q]) { outn[off[q]] = val; t0 = val; t1 = val * val; }
      } else {
        if (ok[q]) {
          const float x = xe[MASK ? q : 0];
          const float pre = fmaf(ecoef[ml[q]], x, ecoef[M_B + ml[q]]);
          const float z = pre > 0.f ? val : 0.f;
          const float xh = (x - ecoef[2 * M_B + ml[q]]) * ecoef[3 * M_B + ml[q]];
          t0 = z; t1 = z * xh;
          if (EPI == EPI_MASK_STORE) outn[off[q]] = z;
          else outn[off[q]] = go[(EPI == EPI_MASK_ACCUM) ? q : 0] + ecoef[4 * M_B + ml[q]] * z;
        }
      }
      s0[q] = t0; s1[q] = t1;
    }
    if (want_sums) {
      const float r0 = half_reduce_n<RP>(s0, lane), r1 = half_reduce_n<RP>(s1, lane);
      constexpr int SH = (RP == 8) ? 2 : (RP == 4 ? 3 : 4);      // lanes l31 >> SH share a total: the first of each group writes it
      if ((l31 & ((1 << SH) - 1)) == 0) {
        const int mrow = wm * 32 + acc_row(kg * RP + (l31 >> SH), half);
        red0[wn * M_B + mrow] = r0;
        red1[wn * M_B + mrow] = r1;
      }
    }
  } else {
  // ================= cross-group reduction of the accumulators (K-split) =================
  if (KS > 1) {
    float* rbuf = Xs;   // staging buffers are free after the last barrier
    const int slot = (wm * WN + wn) * MT * NT;
    if (kg > 0 && !loader) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) rbuf[(((kg - 1) * WM * WN * MT * NT + slot + i * NT + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 0 && !loader) {
#pragma unroll
      for (int g = 1; g < KS; ++g)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += rbuf[(((g - 1) * WM * WN * MT * NT + slot + i * NT + j) * 16 + r) * 64 + lane];
    }
  }

  stamp(4);
  // ================= cross-block K-split: publish the partial tile; the LAST arriving slice sums all slices (in slice
  // order: bit-reproducible) and runs the epilogue.  No spinning.  Hand-off without fences (MI355X_MICROARCH.md, inter-workgroup
  // visibility, "valid forms"): every partial is written with write-through (`sc1`) stores, drained with vmcnt(0) before the
  // workgroup's agent-scope ticket; the last arriver reads the partials with `sc1` loads, which bypass its own XCD's L2.  An
  // agent-scope release fence instead writes back the XCD's whole dirty L2 -- 3-6 us per kernel in the phase trace, as long as
  // the K loop of the 8^3 / 4^3 layers itself. =======
  if (kz > 1) {
    constexpr int NSLOT = WM * WN * MT * NT;
    const long tile_id = blockIdx.x + (long)gridDim.x * blockIdx.y;
    float* part = a.kz_part + (tile_id * kz) * NSLOT * 1024;
    const int slot = (wm * WN + wn) * MT * NT;
    if (kg == 0 && !loader) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* dst = part + (((long)blockIdx.z * NSLOT + slot + i * NT + j) * 16 + r) * 64 + lane;
#if MMNN_KZ_FENCED
            *dst = acc[i][j][r];
#else
            __hip_atomic_store(dst, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_store_dword ... sc1
#endif
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores before the barrier
    __syncthreads();
    unsigned* ticket = reinterpret_cast<unsigned*>(ecoef + C::ECOEF * M_B);
    if (tid == 0) {
#if MMNN_KZ_FENCED
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // cumulative over the barrier: publishes the whole workgroup's stores
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the compiler may drop the wait after buffer_wbl2: MI355X_MICROARCH.md, compiler hazard)
#endif
      *ticket = __hip_atomic_fetch_add(a.kz_cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    stamp(5);
    const unsigned arrived = *ticket;
    if (arrived != (unsigned)(kz - 1)) return;            // not the last slice of this tile: done
    // Ready for the next launch: nobody else touches this tile's counter any more in THIS launch (all kz slices have arrived), and the
    // kernel boundary orders the store before the next launch's first add.
    if (tid == 0) __hip_atomic_store(a.kz_cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if MMNN_KZ_FENCED
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // every reading wave: drops this CU's stale L1 lines of the partial tiles
#endif
    if constexpr (KS > 1) {
      // Every wave group takes 16 / KS accumulator rows of each tile and has ALL slices' loads in flight at once (slices past kz
      // re-read the last one and add an exact zero): one memory round trip instead of kz / 2 dependent ones by the group-0 waves
      // alone (phase "kz sum": 3.7k of a 27k-cycle launch at kz = 8).  Summed in slice order as before -- same bits -- and handed to
      // the epilogue waves through the staging area, which is free since the barriers above.
      constexpr int RP = 16 / KS;                       // rows per wave group
      constexpr int ZB = (RP <= 4) ? 8 : 4;             // slices in flight: at most 32 loads per lane
      float* xbuf = Xs;
      if (!loader) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            float sacc[RP];
#pragma unroll
            for (int q = 0; q < RP; ++q) sacc[q] = 0.f;
            for (int z0 = 0; z0 < kz; z0 += ZB) {
              float pz[ZB][RP];
#pragma unroll
              for (int z = 0; z < ZB; ++z) {
                const int zc = min(z0 + z, kz - 1);
#pragma unroll
                for (int q = 0; q < RP; ++q) {
                  const float* sp = part + (((long)zc * NSLOT + slot + i * NT + j) * 16 + kg * RP + q) * 64 + lane;
#if MMNN_KZ_FENCED
                  pz[z][q] = *sp;
#else
                  pz[z][q] = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // global_load_dword ... sc1
#endif
                }
              }
#pragma unroll
              for (int z = 0; z < ZB; ++z)
#pragma unroll
                for (int q = 0; q < RP; ++q) sacc[q] += (z0 + z < kz) ? pz[z][q] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < RP; ++q) xbuf[((slot + i * NT + j) * 16 + kg * RP + q) * 64 + lane] = sacc[q];
          }
      }
      __syncthreads();
      if (kg == 0 && !loader) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = xbuf[((slot + i * NT + j) * 16 + r) * 64 + lane];
      }
    } else {
    if (kg == 0 && !loader) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          // kz is 2, 4 or 8: two slices' loads are in flight together (one memory round trip per pair), summed in slice order
          for (int z = 0; z < kz; z += 2) {
            float p0[16], p1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float* s0 = part + (((long)z * NSLOT + slot + i * NT + j) * 16 + r) * 64 + lane;
              const float* s1 = part + (((long)(z + 1) * NSLOT + slot + i * NT + j) * 16 + r) * 64 + lane;
#if MMNN_KZ_FENCED
              p0[r] = *s0; p1[r] = *s1;
#else
              p0[r] = __hip_atomic_load(s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);              // global_load_dword ... sc1
              p1[r] = __hip_atomic_load(s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] + p0[r]) + p1[r];
          }
        }
    }
    }
  }

  stamp(6);
  // the wide epilogue reuses the staging area: with an in-block K-split the group-0 waves were still reading their partners'
  // accumulators (kz == 1) or the summed slices (kz > 1) from it
  if (C::WIDE && KS > 1) __syncthreads();
  // ================= epilogue =================

  int vox[NT];
  bool vok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = (wn * NT + j) * 32 + l31;
    if (TAPS == 27) {
      const int wx = t % TW, hy = (t / TW) % TH, dz = t / (TW * TH);
      const int d = d0 + dz, h = h0 + hy, w = w0 + wx;
      vok[j] = d < a.D && h < a.H && w < a.W;
      vox[j] = (d * a.H + h) * a.W + w;
    } else {
      vox[j] = v0_ + t;
      vok[j] = vox[j] < V;
    }
  }

  if (kg == 0 && !loader) {
  // The mask epilogues read one (ex) or two (ex, out) tensors per element.  Their loads are issued BEFORE the stores: `out` may
  // alias `ex` as far as the compiler knows, so a load placed after a store is never hoisted above it and every accumulator row
  // would pay a full memory round trip of its own (phase trace r02: 34 us of a 126 us block in block 1's data gradient).  One
  // tensor (MASK_STORE): the whole tile's loads go out together; two tensors (MASK_ACCUM): one 32-row slab at a time (register
  // budget: the big tiles of block 1 must stay at two waves per SIMD).  Element offsets are 32-bit (host check: M * V < 2^31),
  // so an address costs one register beside the uniform base pointer.
  constexpr bool MASK = (EPI == EPI_MASK_STORE || EPI == EPI_MASK_ACCUM);
  // Wide form (big tiles, 16-byte aligned tensors): an MFMA accumulator holds ONE voxel per lane and row, so the direct form below
  // moves 4 bytes per lane and instruction -- 128..192 memory instructions per thread, and the address coalescer, not HBM, bounds
  // the epilogue (phase trace r02: 33k of a block's 85k cycles in block 1's conv1 data gradient, 40k of 290k in conv2's).  Each
  // 32 x 32 accumulator tile is therefore transposed through LDS (wave-private, the staging buffers are free) so that a lane holds
  // 4 consecutive voxels of 4 rows: every global access becomes a dwordx4 -- a quarter of the memory instructions.
  const bool wide_ok = C::WIDE && (V & 3) == 0 && ((TAPS == 27) ? ((a.W & 3) == 0) : true) && ((((uintptr_t)outn | (uintptr_t)exn) & 15) == 0);
  if (C::WIDE && wide_ok) {
    float* tb = Xs + (wm * WN + wn) * (32 * C::TSTRIDE);
    const int q4 = 4 * (lane & 7), rl = lane >> 3;            // this lane's voxel quad and first row within a tile
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x4 vt[NT][4], xq[MASK ? NT : 1][4], gq[(EPI == EPI_MASK_ACCUM) ? NT : 1][4];
      unsigned off[NT][4];
      bool okq[NT][4];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        // voxel quad of this lane in tile j
        const int t = (wn * NT + j) * 32 + q4;
        int vq; bool vk;
        if (TAPS == 27) {
          const int wx = t % TW, hy = (t / TW) % TH, dz = t / (TW * TH);
          const int d = d0 + dz, h = h0 + hy, w = w0 + wx;
          vk = d < a.D && h < a.H && w < a.W;
          vq = (d * a.H + h) * a.W + w;
        } else {
          vq = v0_ + t;
          vk = vq < V;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();        // the previous tile's reads precede these writes (LDS executes a wave's accesses in order)
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[acc_row(r, half) * C::TSTRIDE + l31] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ml = wm * MT * 32 + i * 32 + rl + 8 * k;
          vt[j][k] = *reinterpret_cast<const f32x4*>(tb + (rl + 8 * k) * C::TSTRIDE + q4);
          okq[j][k] = vk && (m0 + ml) < a.M;
          off[j][k] = okq[j][k] ? (unsigned)(m0 + ml) * (unsigned)V + (unsigned)vq : 0u;      // unconditional loads (clamped)
        }
      }
      // all loads of the slab before its stores (see the note on aliasing below)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (MASK) xq[MASK ? j : 0][k] = *reinterpret_cast<const f32x4*>(exn + off[j][k]);
          if (EPI == EPI_MASK_ACCUM) gq[(EPI == EPI_MASK_ACCUM) ? j : 0][k] = *reinterpret_cast<const f32x4*>(outn + off[j][k]);
        }
      float s0[4], s1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ml = wm * MT * 32 + i * 32 + rl + 8 * k;
        const float ea = ecoef[ml], eb = ecoef[M_B + ml], mu = ecoef[2 * M_B + ml], rs = ecoef[3 * M_B + ml], gm = ecoef[4 * M_B + ml];
        const float ds = ecoef[5 * M_B + ml];
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (okq[j][k]) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (MASK) {
                const float x = xq[MASK ? j : 0][k][e];
                const float pre = fmaf(ea, x, eb);
                const float z = pre > 0.f ? vt[j][k][e] : 0.f;
                const float xh = (x - mu) * rs;
                t0 += z;
                t1 += z * xh;
                o[e] = (EPI == EPI_MASK_STORE) ? z : gq[(EPI == EPI_MASK_ACCUM) ? j : 0][k][e] + gm * z;
              } else {     // EPI_STORE_STATS: channel dropout scale, sums for the consumer's batch statistics
                const float val = vt[j][k][e] * ds;
                t0 += val;
                t1 += val * val;
                o[e] = val;
              }
            }
            *reinterpret_cast<f32x4*>(outn + off[j][k]) = o;
          }
        }
        s0[k] = t0; s1[k] = t1;
      }
      if (want_sums) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float r0 = s0[k], r1 = s1[k];
          r0 += swz_xor<1>(r0); r1 += swz_xor<1>(r1);
          r0 += swz_xor<2>(r0); r1 += swz_xor<2>(r1);
          r0 += swz_xor<4>(r0); r1 += swz_xor<4>(r1);
          if ((lane & 7) == 0) {
            const int ml = wm * MT * 32 + i * 32 + rl + 8 * k;
            red0[wn * M_B + ml] = r0;
            red1[wn * M_B + ml] = r1;
          }
        }
      }
    }
  } else {
  if (ALL_ROWS && !(XPRE && xall_loaded)) load_xall();
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float s0[16], s1[16];
    float xe[(MASK && !ALL_ROWS) ? 16 : 1][NT], go[(EPI == EPI_MASK_ACCUM) ? 16 : 1][NT];
    if (MASK && !ALL_ROWS) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ml = wm * MT * 32 + i * 32 + acc_row(q, half);
        const bool mok = (m0 + ml) < a.M;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const unsigned o = (mok && vok[j]) ? (unsigned)(m0 + ml) * (unsigned)V + (unsigned)vox[j] : 0u;
          xe[(MASK && !ALL_ROWS) ? q : 0][j] = exn[o];
          if (EPI == EPI_MASK_ACCUM) go[(EPI == EPI_MASK_ACCUM) ? q : 0][j] = outn[o];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = wm * MT * 32 + i * 32 + acc_row(r, half);
      const bool mok = (m0 + ml) < a.M;
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bool ok = mok && vok[j];
        const unsigned o = (unsigned)(m0 + ml) * (unsigned)V + (unsigned)vox[j];
        float val = acc[i][j][r];
        if (EPI == EPI_STORE) {
          if (ok) outn[o] = val;
        } else if (EPI == EPI_STORE_STATS) {
          val *= ecoef[5 * M_B + ml];
          if (ok) {
            outn[o] = val;
            t0 += val;
            t1 += val * val;
          }
        } else {
          if (ok) {
            const float x = ALL_ROWS ? xall[ALL_ROWS ? i : 0][ALL_ROWS ? r : 0][j] : xe[(MASK && !ALL_ROWS) ? r : 0][j];
            const float pre = fmaf(ecoef[ml], x, ecoef[M_B + ml]);
            const float z = pre > 0.f ? val : 0.f;
            const float xh = (x - ecoef[2 * M_B + ml]) * ecoef[3 * M_B + ml];
            t0 += z;
            t1 += z * xh;
            if (EPI == EPI_MASK_STORE) outn[o] = z;
            else outn[o] = go[(EPI == EPI_MASK_ACCUM) ? r : 0][j] + ecoef[4 * M_B + ml] * z;
          }
        }
      }
      s0[r] = t0;
      s1[r] = t1;
    }
    if (want_sums) {
      const float r0 = half_reduce16(s0, lane);
      const float r1 = half_reduce16(s1, lane);
      if ((lane & 1) == 0) {
        const int ml = wm * MT * 32 + i * 32 + acc_row((l31 >> 1) & 15, half);
        red0[wn * M_B + ml] = r0;
        red1[wn * M_B + ml] = r1;
      }
    }
  }
  }
  }
  }
  stamp(7);
  if (want_sums) {
    __syncthreads();
    for (int m = tid; m < M_B; m += NTHREADS) {
      if (m0 + m >= a.M) continue;
      double v0d = 0.0, v1d = 0.0;
#pragma unroll
      for (int j = 0; j < WN; ++j) { v0d += (double)red0[j * M_B + m]; v1d += (double)red1[j * M_B + m]; }
      if (EPI == EPI_STORE_STATS) {
        atomicAdd(a.st_out.sum + (long)rep * a.st_out.stride + a.st_out.off + m0 + m, v0d);
        atomicAdd(a.st_out.sq + (long)rep * a.st_out.stride + a.st_out.off + m0 + m, v1d);
      } else {
        atomicAdd(a.dbeta + (long)rep * a.M + m0 + m, v0d);
        atomicAdd(a.dgamma + (long)rep * a.M + m0 + m, v1d);
        if (EPI == EPI_MASK_ACCUM) {
          const double g = (double)ecoef[4 * M_B + m];
          atomicAdd(a.s_acc.sum + (long)rep * a.s_acc.stride + a.s_acc.off + m0 + m, g * v0d);
          atomicAdd(a.s_acc.sq + (long)rep * a.s_acc.stride + a.s_acc.off + m0 + m, g * v1d);
        }
      }
    }
  }
  if (pf_sink + pf_sink2 == 1.2345678e-33f) a.out[0] = pf_sink;     // never true in practice: the prefetched values are not used
  stamp(8);
  if (tracing) {
    a.trace[10] = ((unsigned long long)TAPS << 48) | ((unsigned long long)PRO << 40) | ((unsigned long long)EPI << 32) | ((unsigned long long)a.M << 16) | (unsigned long long)a.Cin;
    a.trace[11] = ((unsigned long long)gridDim.x << 32) | ((unsigned long long)gridDim.y << 16) | gridDim.z;
    a.trace[12] = ((unsigned long long)(WM * WN * KS) << 32) | ((unsigned long long)KC << 16) | (unsigned long long)(MT * NT);
  }
}

#endif  // __HIPCC__
}  // namespace mmnn
