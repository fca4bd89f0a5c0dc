// Host-side entry of the implicit-GEMM convolution (fprop.hpp): validates every shape the kernel indexes with and forwards to
// the (taps, prologue, epilogue) instantiation, which picks a tile shape from the extent and launches (fprop_dispatch.hpp).
#include <stdlib.h>

#include "fprop.hpp"

namespace mmnn {

int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev % MAX_DEVICES;
}

// defined (explicitly instantiated) in fprop_inst_*.hip, one translation unit per combination
template <int TAPS, int PRO, int EPI>
int dispatch(const FpropArgs& a, hipStream_t s);
extern template int dispatch<1, PRO_BNRELU, EPI_STORE_STATS>(const FpropArgs&, hipStream_t);
extern template int dispatch<1, PRO_NONE, EPI_STORE_STATS>(const FpropArgs&, hipStream_t);
extern template int dispatch<1, PRO_GRAD, EPI_MASK_ACCUM>(const FpropArgs&, hipStream_t);
extern template int dispatch<1, PRO_GRAD, EPI_STORE>(const FpropArgs&, hipStream_t);
extern template int dispatch<27, PRO_BNRELU, EPI_STORE_STATS>(const FpropArgs&, hipStream_t);
extern template int dispatch<27, PRO_GRAD, EPI_MASK_STORE>(const FpropArgs&, hipStream_t);

// conv3_bf16x3.hip: the dense-layer conv2 forward on three-piece bf16 MFMAs (wide extents, 32 output channels)
bool conv3_fwd_bf16x3_eligible(const FpropArgs& a);
int launch_conv3_fwd_bf16x3(const FpropArgs& a, hipStream_t stream);
bool conv3_dgrad_bf16x3_eligible(const FpropArgs& a);
int launch_conv3_dgrad_bf16x3(const FpropArgs& a, hipStream_t stream);

int launch_fprop(const FpropArgs& a, int taps, int pro, int epi, hipStream_t stream) {
  MMNN_REQUIRE(a.N > 0 && a.D > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.M > 0, "fprop: non-positive extent");
  MMNN_REQUIRE((long)a.D * a.H * a.W < (1l << 30), "fprop: volume too large for 32-bit voxel indices");
  MMNN_REQUIRE(a.in0 && a.w && a.out && a.w_ld >= a.M, "fprop: null operand or w_ld < M");
  MMNN_REQUIRE((long)(a.M + 128) * a.D * a.H * a.W < (1l << 31), "fprop: output rows x volume too large for 32-bit element offsets");
  MMNN_REQUIRE(pro != PRO_GRAD || a.in1, "fprop: PRO_GRAD needs the normalised tensor (in1)");
  MMNN_REQUIRE((epi != EPI_MASK_STORE && epi != EPI_MASK_ACCUM) || (a.ex && a.dgamma && a.dbeta), "fprop: mask epilogue operands missing");
  if (taps == 27 && pro == PRO_BNRELU && epi == EPI_STORE_STATS && conv3_fwd_bf16x3_eligible(a)) return launch_conv3_fwd_bf16x3(a, stream);
  if (taps == 27 && pro == PRO_GRAD && epi == EPI_MASK_STORE && conv3_dgrad_bf16x3_eligible(a)) return launch_conv3_dgrad_bf16x3(a, stream);
#define MMNN_CASE(T, P, E) \
  if (taps == T && pro == P && epi == E) return dispatch<T, P, E>(a, stream);
  MMNN_CASE(1, PRO_BNRELU, EPI_STORE_STATS)    // dense-layer conv1 forward
  MMNN_CASE(1, PRO_NONE, EPI_STORE_STATS)      // transition conv forward (input already pooled)
  MMNN_CASE(1, PRO_GRAD, EPI_MASK_ACCUM)       // conv1 data gradient -> G of the block buffer
  MMNN_CASE(1, PRO_GRAD, EPI_STORE)            // transition conv data gradient
  MMNN_CASE(27, PRO_BNRELU, EPI_STORE_STATS)   // dense-layer conv2 forward
  MMNN_CASE(27, PRO_GRAD, EPI_MASK_STORE)      // conv2 data gradient
#undef MMNN_CASE
  set_error("fprop: unsupported (taps=%d, pro=%d, epi=%d)", taps, pro, epi);
  return 1;
}

}  // namespace mmnn
