// Stem of the 3-D DenseNet (models/densenet.py:199-202): conv 7x7x7 stride 2 pad 3 (no bias) -> BN -> ReLU ->
// max-pool 3 stride 2 pad 1, and its backward (weight gradient only: the network input needs no gradient).
#pragma once
#include "common.hpp"

namespace mmnn {

struct StemConvArgs {
  int N, Cin, D, H, W;        // input volume
  int Do, Ho, Wo;             // conv output extent
  int M;                      // output channels (init_features, <= 64)
  const float* x;             // [N][Cin][D*H*W]
  const float* wp;            // packed weights, see pack.hip:  [7 kd][KROWS][64]
  float* out;                 // [N][M][Do*Ho*Wo]
  StatPtr st_out;             // batch statistics of the conv output (null => skip)
};
int launch_stem_conv(const StemConvArgs& a, hipStream_t stream);
inline int stem_krows(int Cin) { return (Cin % 2 == 0) ? Cin * 49 : Cin * 56; }   // rows per kd slab of the packed weights

struct StemPoolArgs {
  int N, C, Di, Hi, Wi, Do, Ho, Wo;
  const float* x;             // conv output [N][C][Vi]
  BnFwd bn;                   // norm0
  float* out; long out_ns;    // block-1 buffer, channels [0, C)
  unsigned char* idx;         // [N][C][Vo] winning tap (kd*9 + kh*3 + kw)
  StatPtr st_out;             // statistics of the pooled output (concat channels 0..C-1)
};
int launch_stem_pool(const StemPoolArgs& a, hipStream_t stream);

struct StemPoolBwdArgs {
  int N, C, Di, Hi, Wi, Do, Ho, Wo;
  const float* x;             // conv output
  BnFwd bn;                   // norm0 (mask + xhat)
  const float* g; long g_ns;  // G of block 1, channels [0, C)
  const float* xp; long xp_ns;  // pooled activations (block-1 buffer channels [0, C))
  BnBwd gr;                   // BN-backward of the pooled tensor
  const unsigned char* idx;
  float* dz;                  // [N][C][Vi] gradient wrt the BN output, ReLU mask applied
  double* dgamma; double* dbeta;   // [NREP][C]
};
int launch_stem_pool_bwd(const StemPoolBwdArgs& a, hipStream_t stream);

struct StemWgradArgs {
  int N, Cin, D, H, W, Do, Ho, Wo, M;
  const float* x;             // network input
  const float* dz;            // [N][M][Vo]
  const float* y;             // conv output [N][M][Vo]
  BnBwd gr;                   // BN-backward of norm0: dY = p*dz + q*y + r
  float* slab; long slab_stride; int nsplit;   // slab[split][c][m][352]  (tap = kd*49 + kh*7 + kw < 343)
};
int launch_stem_wgrad(const StemWgradArgs& a, hipStream_t stream);
int stem_wgrad_pick_splits(int N, int Do, int Ho, int Wo, int Cin);

}  // namespace mmnn
