// fp32 GEMM tiles on the bf16 matrix pipe: x = hi + mid + lo (three bf16 pieces, 24 mantissa bits), six v_mfma_f32_32x32x16_bf16 per
// 16 k-steps (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid; fp32 accumulation) against v_mfma_f32_32x32x2_f32.
// (developer microbenchmark, hipcc --offload-arch=gfx950 -- measures what a rewrite of the MFMA-bound convolution kernels would buy:
// mfma_valu_overlap.hip / mfma_acc_file.hip show that the fp32 matrix instruction shares its issue slot with every vector instruction,
// 64 cycles per 4096 FLOP, while the bf16 one takes 32 cycles per 32768 FLOP and lets ~4 vector instructions through for free.)
// One workgroup of four waves (one per SIMD); a wave owns a 64 x 64 tile (2 x 2 accumulators) of C = A[256 x K] * B[K x 64], K = 3456 (= 128 channels x 27 taps,
// the conv2 forward reduction), operands staged through LDS in chunks of 64 k.  Prints the error of both forms against an fp64 host
// product and the shader cycles per k-step spent in the LDS-read + MFMA section (staging excluded: it is written for clarity, not speed).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int MR = 256, NC = 64, KT = 3456, KC = 64;
constexpr int BS = KC + 8;                       // bf16 row stride (k contiguous): 144 bytes, 16-byte aligned, spreads the banks

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// C layout of the 32x32 MFMAs: register r of lane l holds row 8 * (r / 4) + 4 * (l / 32) + r % 4, column l % 32
__device__ __forceinline__ void store_tile(float* C, const f32x16& acc, int row0, int col0, int lane) {
  for (int r = 0; r < 16; ++r) C[(row0 + 8 * (r / 4) + 4 * (lane / 32) + r % 4) * NC + col0 + lane % 32] = acc[r];
}

__global__ void __launch_bounds__(256) gemm_f32(const float* A, const float* B, float* C, unsigned long long* cyc, int restage) {
  extern __shared__ float sm[];
  float* sa = sm;                 // [KC][MR]
  float* sb = sm + KC * MR;       // [KC][NC]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  unsigned long long spent = 0;
  for (int k0 = 0; k0 < KT; k0 += KC) {
    __syncthreads();
    if (restage || k0 == 0) {
      for (int e = tid; e < KC * MR; e += 256) { const int k = e / MR, m = e % MR; sa[e] = A[(long)m * KT + k0 + k]; }
      for (int e = tid; e < KC * NC; e += 256) { const int k = e / NC, n = e % NC; sb[e] = B[(long)(k0 + k) * NC + n]; }
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
    for (int k = 0; k < KC; k += 2) {             // A operand: lane holds A[row = lane % 32][k + lane / 32]; B likewise by column
      const int kk = k + lane / 32;
      const float a0 = sa[kk * MR + 64 * w + lane % 32], a1 = sa[kk * MR + 64 * w + 32 + lane % 32];
      const float b0 = sb[kk * NC + lane % 32], b1 = sb[kk * NC + 32 + lane % 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    spent += __builtin_amdgcn_s_memtime() - t0;
  }
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) store_tile(C, acc[i][j], 64 * w + 32 * i, 32 * j, lane);
  if (lane == 0 && blockIdx.x == 0) cyc[w] = spent;
}

// three bf16 pieces of an fp32 value (round to nearest even at every step)
__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

template <int NPROD>   // 6: the full set above; 3: two pieces (hi.hi, hi.mid, mid.hi)
__global__ void __launch_bounds__(256) gemm_bf16x3(const float* A, const float* B, float* C, unsigned long long* cyc, int restage) {
  extern __shared__ float sm[];
  __bf16* sa = reinterpret_cast<__bf16*>(sm);    // [3][MR][BS]   k contiguous
  __bf16* sb = sa + 3 * MR * BS;                 // [3][NC][BS]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  unsigned long long spent = 0;
  for (int k0 = 0; k0 < KT; k0 += KC) {
    __syncthreads();
    if (restage || k0 == 0) {
    for (int e = tid; e < KC * MR; e += 256) {
      const int m = e / KC, k = e % KC;
      __bf16 h, mi, l; split3(A[(long)m * KT + k0 + k], h, mi, l);
      sa[(0 * MR + m) * BS + k] = h; sa[(1 * MR + m) * BS + k] = mi; sa[(2 * MR + m) * BS + k] = l;
    }
    for (int e = tid; e < KC * NC; e += 256) {
      const int k = e / NC, n = e % NC;
      __bf16 h, mi, l; split3(B[(long)(k0 + k) * NC + n], h, mi, l);
      sb[(0 * NC + n) * BS + k] = h; sb[(1 * NC + n) * BS + k] = mi; sb[(2 * NC + n) * BS + k] = l;
    }
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int k = 0; k < KC; k += 16) {            // A operand: lane holds A[row = lane % 32][k + 8 * (lane / 32) .. + 7]; B likewise by column
      const int kk = k + 8 * (lane / 32);
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (NPROD == 3 && p == 2) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i][p] = *reinterpret_cast<const bf16x8*>(sa + (p * MR + 64 * w + 32 * i + lane % 32) * BS + kk);
          b[i][p] = *reinterpret_cast<const bf16x8*>(sb + (p * NC + 32 * i + lane % 32) * BS + kk);
        }
      }
      // smallest products first, so that their sum is formed before it meets the large one
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (NPROD == 6) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
    spent += __builtin_amdgcn_s_memtime() - t0;
  }
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) store_tile(C, acc[i][j], 64 * w + 32 * i, 32 * j, lane);
  if (lane == 0 && blockIdx.x == 0) cyc[w] = spent;
}

static void report(const char* name, const std::vector<float>& c, const std::vector<double>& ref, const unsigned long long* cyc, double flop_per_mfma_cycle) {
  double mx = 0, se = 0, sr = 0, scale = 0;
  for (size_t i = 0; i < ref.size(); ++i) scale = fmax(scale, fabs(ref[i]));
  for (size_t i = 0; i < ref.size(); ++i) { const double d = c[i] - ref[i]; mx = fmax(mx, fabs(d)); se += d * d; sr += ref[i] * ref[i]; }
  unsigned long long cm = 0; for (int i = 0; i < 4; ++i) cm = cyc[i] > cm ? cyc[i] : cm;
  const double per_k = (double)cm / KT;
  printf("%-44s max |err| / max |C| %.2e   rms err / rms C %.2e   %6.2f shader-clock ticks per k-step of a 64 x 64 wave tile (LDS reads + MFMAs)\n",
         name, mx / scale, sqrt(se / sr), per_k);
  (void)flop_per_mfma_cycle;
}

int main() {
  std::vector<float> A((size_t)MR * KT), B((size_t)KT * NC), c((size_t)MR * NC);
  srand(7);
  auto nrm = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); };
  for (int variant = 0; variant < 2; ++variant) {      // 0: normal operands; 1: ReLU-like activations (half of them zero), as the convolutions see them
    for (auto& x : A) x = (float)(0.05 * nrm());
    for (auto& x : B) { const double v = nrm(); x = (float)(variant ? (v > 0 ? v : 0.0) : v); }
    std::vector<double> ref((size_t)MR * NC, 0.0);
    for (int m = 0; m < MR; ++m) for (int k = 0; k < KT; ++k) { const double a = A[(size_t)m * KT + k]; for (int n = 0; n < NC; ++n) ref[(size_t)m * NC + n] += a * B[(size_t)k * NC + n]; }
    float *dA, *dB, *dC; unsigned long long* dcyc; unsigned long long cyc[4];
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, c.size() * 4)); CK(hipMalloc(&dcyc, 32));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    const size_t sm32 = (size_t)(KC * MR + KC * NC) * 4, sm16 = (size_t)3 * (MR + NC) * BS * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm16));
    printf("operands: %s\n", variant ? "A normal, B = max(normal, 0)" : "A, B normal");
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(gemm_f32, dim3(1), dim3(256), sm32, 0, dA, dB, dC, dcyc, 1);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cyc, dcyc, 32, hipMemcpyDeviceToHost));
    report("v_mfma_f32_32x32x2_f32", c, ref, cyc, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(gemm_bf16x3<6>, dim3(1), dim3(256), sm16, 0, dA, dB, dC, dcyc, 1);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cyc, dcyc, 32, hipMemcpyDeviceToHost));
    report("3 bf16 pieces, 6 x v_mfma_f32_32x32x16_bf16", c, ref, cyc, 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(gemm_bf16x3<3>, dim3(1), dim3(256), sm16, 0, dA, dB, dC, dcyc, 1);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cyc, dcyc, 32, hipMemcpyDeviceToHost));
    report("2 bf16 pieces, 3 x v_mfma_f32_32x32x16_bf16", c, ref, cyc, 1024);
    if (variant == 0) {   // whole chip, operands staged once (the LDS-read + MFMA pipeline alone, under the chip's power limit): fp32-equivalent rate
      const int grid = 1024;
      const double flop = 2.0 * MR * NC * KT * grid;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      auto timed = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %d workgroups of 4 waves, operands resident in LDS: %7.1f us per launch = %6.1f fp32-equivalent TFLOP/s on the chip\n", name, grid, ms * 100.0, flop / (ms * 1e-4) / 1e12);
      };
      timed("v_mfma_f32_32x32x2_f32", [&] { hipLaunchKernelGGL(gemm_f32, dim3(grid), dim3(256), sm32, 0, dA, dB, dC, dcyc, 0); });
      timed("3 bf16 pieces, 6 x v_mfma_f32_32x32x16_bf16", [&] { hipLaunchKernelGGL(gemm_bf16x3<6>, dim3(grid), dim3(256), sm16, 0, dA, dB, dC, dcyc, 0); });
      timed("2 bf16 pieces, 3 x v_mfma_f32_32x32x16_bf16", [&] { hipLaunchKernelGGL(gemm_bf16x3<3>, dim3(grid), dim3(256), sm16, 0, dA, dB, dC, dcyc, 0); });
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dcyc));
  }
  return 0;
}
