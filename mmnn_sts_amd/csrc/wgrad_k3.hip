// 3x3x3 weight-gradient kernels: instantiations + launches (its own translation unit: the kernels are heavily unrolled and this
// file and wgrad_k1.hip are the two longest compiles of the build).  Host-side validation and split selection: wgrad.hip.
#include <stdlib.h>

#include <algorithm>

#include "wgrad.hpp"

namespace mmnn {

template <int PRO_X, int TD, int TH, int TW>
static int launch3(const WgradArgs& a, hipStream_t stream) {
  using C = Wg3Cfg<TD, TH, TW>;
  auto kern = wgrad3_kernel<PRO_X, TD, TH, TW>;
  const size_t smem = C::smem_bytes();
  static bool configured[MAX_DEVICES] = {false};   // per device
  bool& conf = configured[current_device_slot()];
  if (!conf) {
    MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    conf = true;
  }
  MMNN_LAUNCH(kern, dim3(a.nsplit, cdiv(a.Cin, 32)), dim3(C::NTHREADS), smem, stream, a);
  MMNN_HIP(hipGetLastError());
  return 0;
}

template <int PRO_X, int TD, int TH, int TW>
static int launch3_batched(const WgradArgs* host, const WgradArgs* dev, int count, uint64_t seed, hipStream_t stream) {
  using C = Wg3Cfg<TD, TH, TW>;
  auto kern = wgrad3_batched_kernel<PRO_X, TD, TH, TW>;
  const size_t smem = C::smem_bytes();
  static bool configured[MAX_DEVICES] = {false};   // per device
  bool& conf = configured[current_device_slot()];
  if (!conf) {
    MMNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    conf = true;
  }
  int gx = 1, gy = 1;
  bool uniform = true;
  for (int i = 0; i < count; ++i) {
    gx = std::max(gx, host[i].nsplit); gy = std::max(gy, cdiv(host[i].Cin, 32));
    uniform = uniform && host[i].nsplit == host[0].nsplit && cdiv(host[i].Cin, 32) == cdiv(host[0].Cin, 32);
  }
  static const bool no_remap = [] { const char* e = getenv("MMNN_WG3_NO_XCD"); return e && e[0] == '1'; }();   // A/B knob
  if (uniform && gy > 1 && !no_remap) {
    const long ngroups = (long)count * gx;                       // (layer, split) pairs; each owns gy blocks on one XCD
    const long blocks = 8l * gy * ((ngroups + 7) / 8);
    MMNN_REQUIRE(blocks < (1l << 31), "wgrad batch: grid out of range");
    MMNN_LAUNCH(kern, dim3((unsigned)blocks), dim3(C::NTHREADS), smem, stream, dev, seed, gx, gy, (int)ngroups);
  } else {
    MMNN_LAUNCH(kern, dim3(gx, gy, count), dim3(C::NTHREADS), smem, stream, dev, seed, gx, gy, 0);
  }
  MMNN_HIP(hipGetLastError());
  return 0;
}

int wgrad3_launch(const WgradArgs& a, int pro_x, hipStream_t s) {
  if (pro_x == PRO_BNRELU) {
    if (a.W > 16) return launch3<PRO_BNRELU, 1, 2, 32>(a, s);
    if (a.W > 8) return launch3<PRO_BNRELU, 1, 4, 16>(a, s);
    if (a.W > 4) return launch3<PRO_BNRELU, 2, 4, 8>(a, s);
    return launch3<PRO_BNRELU, 4, 4, 4>(a, s);
  }
  if (a.W > 16) return launch3<PRO_NONE, 1, 2, 32>(a, s);
  if (a.W > 8) return launch3<PRO_NONE, 1, 4, 16>(a, s);
  if (a.W > 4) return launch3<PRO_NONE, 2, 4, 8>(a, s);
  return launch3<PRO_NONE, 4, 4, 4>(a, s);
}

int wgrad3_launch_batched(const WgradArgs* host, const WgradArgs* dev, int count, uint64_t seed, hipStream_t stream) {
  const WgradArgs& f = host[0];
  if (f.W > 16) return launch3_batched<PRO_BNRELU, 1, 2, 32>(host, dev, count, seed, stream);
  if (f.W > 8) return launch3_batched<PRO_BNRELU, 1, 4, 16>(host, dev, count, seed, stream);
  if (f.W > 4) return launch3_batched<PRO_BNRELU, 2, 4, 8>(host, dev, count, seed, stream);
  return launch3_batched<PRO_BNRELU, 4, 4, 4>(host, dev, count, seed, stream);
}

}  // namespace mmnn
