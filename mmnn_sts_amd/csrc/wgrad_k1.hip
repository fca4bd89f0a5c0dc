// 1x1x1 weight-gradient kernels: instantiations + launches (see wgrad_k3.hip).
#include <stdlib.h>

#include <algorithm>

#include "wgrad.hpp"

namespace mmnn {

template <int PRO_X, int WC>
static int launch1(const WgradArgs& a, hipStream_t stream) {
  using C = Wg1Cfg<WC>;
  auto kern = wgrad1_kernel<PRO_X, WC>;
  const size_t smem = C::smem_bytes();
  static bool configured[MAX_DEVICES] = {false};   // per device
  bool& conf = configured[current_device_slot()];
  if (!conf) {
    MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    conf = true;
  }
  MMNN_LAUNCH(kern, dim3(a.nsplit, cdiv(a.Cin, 32 * WC), cdiv(a.M, 128)), dim3(C::NTHREADS), smem, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

template <int PRO_X, int WC>
static int launch1_batched(const WgradArgs* host, const WgradArgs* dev, int count, uint64_t seed, hipStream_t stream) {
  using C = Wg1Cfg<WC>;
  auto kern = wgrad1_batched_kernel<PRO_X, WC>;
  const size_t smem = C::smem_bytes();
  static bool configured[MAX_DEVICES] = {false};   // per device
  bool& conf = configured[current_device_slot()];
  if (!conf) {
    MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    conf = true;
  }
  int gx = 1, gy = 1, members = 0;
  bool uniform = true;
  for (int i = 0; i < count; ++i) {
    gx = std::max(gx, host[i].nsplit); gy = std::max(gy, cdiv(host[i].Cin, 32 * WC));
    members += cdiv(host[i].Cin, 32 * WC);
    uniform = uniform && host[i].nsplit == host[0].nsplit;
  }
  // r03 A/B (profiles/r03_ab_experiments.txt): the XCD-aware order measured 2-3 % SLOWER here (block 1: 285 vs 278 us) -- unlike the 3x3x3
  // kernel the blocks of one split do not stage the same tiles in lockstep (their channel counts differ), so it stays off: MMNN_WG1_XCD=1
  static const bool remap = [] { const char* e = getenv("MMNN_WG1_XCD"); return e && e[0] == '1'; }();
  if (uniform && members > 1 && remap) {
    const long blocks = 8l * members * ((gx + 7) / 8);
    MMNN_REQUIRE(blocks < (1l << 31), "wgrad batch: grid out of range");
    MMNN_LAUNCH(kern, dim3((unsigned)blocks), dim3(C::NTHREADS), smem, stream, dev, seed, count, gx, members);
  } else {
    MMNN_LAUNCH(kern, dim3(gx, gy, count), dim3(C::NTHREADS), smem, stream, dev, seed, count, gx, 0);
  }
  MMNN_HIP(hipGetLastError());
  return 0;
}

int wgrad1_launch(const WgradArgs& a, int pro_x, int wc, hipStream_t s) {
  if (pro_x == PRO_BNRELU) return wc == 8 ? launch1<PRO_BNRELU, 8>(a, s) : launch1<PRO_BNRELU, 4>(a, s);
  return wc == 8 ? launch1<PRO_NONE, 8>(a, s) : launch1<PRO_NONE, 4>(a, s);
}

int wgrad1_launch_batched(const WgradArgs* host, const WgradArgs* dev, int count, uint64_t seed, int wc, hipStream_t stream) {
  return wc == 8 ? launch1_batched<PRO_BNRELU, 8>(host, dev, count, seed, stream) : launch1_batched<PRO_BNRELU, 4>(host, dev, count, seed, stream);
}

}  // namespace mmnn
