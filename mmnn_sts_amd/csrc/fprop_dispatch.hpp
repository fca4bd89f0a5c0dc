// Tile selection + launch of the implicit-GEMM convolution template (fprop.hpp).  The template definitions live here so that
// every (taps, prologue, epilogue) combination is compiled in its own translation unit (csrc/fprop_inst_*.hip): the kernels
// are heavily unrolled and one TU holding all ~25 instantiations takes a quarter of an hour to compile.
#pragma once
#include <stdlib.h>

#include <algorithm>

#include "fprop.hpp"

namespace mmnn {

template <int TAPS, int PRO, int EPI, int WM, int WN, int KS, int MT, int NT, int KC, int TD, int TH, int TW, bool SPEC = false>
static int launch_cfg(const FpropArgs& a_in, hipStream_t stream) {
  FpropArgs a = a_in;
  constexpr bool KZ_OK = (MT * NT == 1) || (TAPS == 27 && TW <= 16);   // only the small-extent tiles are ever short of blocks
  using C = FpropCfg<TAPS, PRO, EPI, WM, WN, KS, MT, NT, KC, TD, TH, TW, SPEC>;
  auto kern = fprop_kernel<TAPS, PRO, EPI, WM, WN, KS, MT, NT, KC, TD, TH, TW, SPEC>;
  size_t smem = C::smem_bytes(a.Cin);
  MMNN_REQUIRE(smem <= 160 * 1024, "fprop: %zu bytes of LDS needed (Cin=%d) exceeds 160 KiB", smem, a.Cin);
  {
    static const char* env = getenv("MMNN_FPROP_MIN_SMEM");   // experiment knob: force fewer blocks per CU
    if (env) { size_t v = (size_t)atol(env); if (v > smem && v <= 160 * 1024) smem = v; }
  }
  static size_t configured[MAX_DEVICES] = {0};   // hipFuncSetAttribute is per device: remember the largest request of each
  size_t& conf = configured[current_device_slot()];
  if (smem > conf) {
    MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    conf = smem;
  }
  long tiles;
  if (TAPS == 27) tiles = (long)a.N * cdiv(a.D, TD) * cdiv(a.H, TH) * cdiv(a.W, TW);
  else tiles = (long)a.N * cdiv((long)a.D * a.H * a.W, C::V_B);
  const int mtiles = cdiv(a.M, C::M_B);
  MMNN_REQUIRE(tiles > 0 && tiles < (1l << 31) && mtiles <= 65535, "fprop: grid out of range");
  {
    // divisors of the kernel's workgroup -> tile decomposition as multiply-high constants (FpropArgs::mg_*)
    auto magic = [](long d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned long long)d - 1ull) / (unsigned long long)d); };
    long dw, dh = 1, dd = 1;
    if (TAPS == 27) { dw = cdiv(a.W, TW); dh = cdiv(a.H, TH); dd = cdiv(a.D, TD); }
    else dw = cdiv((long)a.D * a.H * a.W, C::V_B);
    MMNN_REQUIRE(tiles * std::max(dw, std::max(dh, dd)) < (1l << 32), "fprop: tile count out of range for the multiply-high decomposition");
    a.mg_w = magic(dw); a.mg_h = magic(dh); a.mg_d = magic(dd);
  }
  // cross-block K-split: when the (voxel, row) tiles alone cannot fill the chip, slices of the channel axis become blocks too
  int kz = 1;
  static const bool kz_off = []() { const char* e = getenv("MMNN_NO_KZ"); return e && e[0] == '1'; }();   // debugging aid
  if (a.kz_part && a.kz_cnt && KZ_OK && !kz_off) {
    const int nch = cdiv(a.Cin, KC);
    while (kz * 2 <= nch && tiles * mtiles * kz * 2 <= 256 && kz < 8) kz *= 2;
    if (kz > 1) {
      const size_t need = (size_t)tiles * mtiles * kz * (WM * WN * MT * NT) * 1024 * sizeof(float);
      if (need > a.kz_part_bytes || (size_t)tiles * mtiles > a.kz_cnt_entries) kz = 1;
    }
  }
  MMNN_LAUNCH(kern, dim3((unsigned)tiles, (unsigned)mtiles, (unsigned)kz), dim3(C::NTHREADS), smem, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

// Tile selection.  Template arguments: <TAPS, PRO, EPI, WM, WN, KS, MT, NT, KC, TD, TH, TW>; a block computes
// (WM*MT*32) output rows x (WN*NT*32) voxels with WM*WN*KS waves, KS wave groups splitting each channel chunk.
// Small extents get small voxel tiles + a deep K-split so that the late dense blocks (8^3, 4^3 voxels) still put
// hundreds of waves on the chip instead of a dozen.
template <int TAPS, int PRO, int EPI>
int dispatch(const FpropArgs& a, hipStream_t s) {
  const long V = (long)a.D * a.H * a.W;
  if constexpr (TAPS == 1) {   // constexpr: a 1x1 instantiation must not drag the nine 3x3x3 kernels into its translation unit
    const long blocks_a = (long)a.N * cdiv(V, 128) * cdiv(a.M, 128);
    const long blocks_b = (long)a.N * cdiv(V, 64) * cdiv(a.M, 64);
    // (r02: forcing the 64-wide tile at 32^3 is far slower -- conv1 forward 39 -> 56 us, data gradient 63 -> 109 us.)
    // 128-wide tile: KC = 8 / 16 / 32 measure the same (37 us at 32^3: a fixed ~17 us of prologue, output write and statistics,
    // then 0.15 us per input channel = 115 TFLOP/s).  64-wide tile: deep chunks, 4x fewer barriers than KC = 16 (25 -> 17 us at 16^3).
    if (blocks_a >= 192) return launch_cfg<1, PRO, EPI, 2, 2, 1, 2, 2, 16, 1, 1, 128>(a, s);
    if (blocks_b >= 192) return launch_cfg<1, PRO, EPI, 2, 2, 2, 1, 1, 64, 1, 1, 64>(a, s);
    return launch_cfg<1, PRO, EPI, 1, 1, 8, 1, 1, 128, 1, 1, 32>(a, s);   // few voxels: deep K chunks (the K loop is latency-bound)
  } else {
  if (a.M <= 32) {
    // Tried and measured no better at 32^3 (r02, tools/exp_classes.py): 8 compute waves with an in-block K-split instead of
    // loader waves (141 vs 140 us), 128-voxel tiles for two blocks per CU (172 us).
    if (a.W > 16) return launch_cfg<27, PRO, EPI, 1, 4, 1, 1, 2, 8, 2, 4, 32, true>(a, s);
    // (r02: loader waves + two LDS buffers for the 16^3 tile as well -- <1, 2, 2, 1, 1, 8, 1, 4, 16, true> -- measured slower: 36 vs 31 us.)
    if (a.W > 8) return launch_cfg<27, PRO, EPI, 1, 2, 4, 1, 1, 16, 1, 4, 16>(a, s);
    if (a.W > 4) return launch_cfg<27, PRO, EPI, 1, 1, 8, 1, 1, 16, 1, 4, 8>(a, s);
    return launch_cfg<27, PRO, EPI, 1, 1, 8, 1, 1, 16, 2, 4, 4>(a, s);
  }
  if (a.W > 16) {
    // KC = 2: the smallest chunk (one MFMA k-pair per tap) keeps the LDS footprint low enough for 3-4 blocks per CU, which
    // hides the staging latency better than loader waves or KC = 4 do here (194 -> 173 us at block 1).
    // Tried and measured slower for the data gradient at 32^3 (r02): 64-voxel tiles / 4 blocks per CU (180 us, register
    // allocation still caps the CU at 3 waves per SIMD), 8 waves with an in-block K-split and KC = 4 (231 us), loader waves (213 us).
    // r03: ONE 8-wave block of 128 rows x 256 voxels (2 x 4 x 32) per CU instead of two 4-wave blocks of 128 x 128: the 442 KB of weights
    // are staged once per 256 voxels instead of once per 128, the halo shrinks from 4.8x to 3.2x of the tile, a thread stages 4 + 1 items
    // per chunk instead of 7 + 4, and the kernel needs 163 registers instead of 228 + 64 accumulation registers.  Measured on the conv2
    // data gradient of block 1 (A/B in one process, profiles/r03_ab_experiments.txt): 141.6 -> 137.7 us.  MMNN_DGRAD_TILE=0: the r02 tile.
    static const int big = [] { const char* e = getenv("MMNN_DGRAD_TILE"); return e ? atoi(e) : 1; }();
    if (big == 1) return launch_cfg<27, PRO, EPI, 2, 4, 1, 2, 2, 2, 2, 4, 32>(a, s);
    return launch_cfg<27, PRO, EPI, 2, 2, 1, 2, 2, 2, 1, 4, 32>(a, s);
  }
  if (a.W > 8) {
    // r03: the 128 x 64 tile as eight one-tile waves (4 x 2) instead of 2 x 2 x 2 waves with two tiles each and an in-block K-split: no
    // in-block reduction, and slice sum and epilogue run on all eight waves instead of four (9.5k + 4.7k of the launch's 66k cycles).
    // conv2 data gradient at 2 x 16^3: 0.407 -> 0.370 ms per step (twelve launches).  MMNN_DGRAD16=0: the r02 shape.
    static const int alt = [] { const char* e = getenv("MMNN_DGRAD16"); return e ? atoi(e) : 1; }();
    if (alt == 1) return launch_cfg<27, PRO, EPI, 4, 2, 1, 1, 1, 4, 1, 4, 16>(a, s);
    return launch_cfg<27, PRO, EPI, 2, 2, 2, 2, 1, 4, 1, 4, 16>(a, s);
  }
  if (a.W > 4) return launch_cfg<27, PRO, EPI, 4, 1, 2, 1, 1, 4, 1, 4, 8>(a, s);
  return launch_cfg<27, PRO, EPI, 4, 1, 2, 1, 1, 4, 2, 4, 4>(a, s);
  }
}

}  // namespace mmnn
