// Does the cost of vector-ALU work beside fp32 matrix instructions depend on WHERE the accumulators live (architectural VGPRs vs AccVGPRs),
// on the MFMA shape, or on the kind of vector instruction?  (developer microbenchmark; hipcc --offload-arch=gfx950)
// mfma_valu_overlap.hip found 4 matrix-pipe cycles lost per v_fma_f32 next to v_mfma_f32_32x32x2_f32 with the accumulators in VGPRs (what the
// compiler picks for the convolution kernels: `v_mfma_f32_32x32x2_f32 v[2:17], v34, v36, v[2:17]`).  This one repeats the measurement with
//   MF 0: v_mfma_f32_32x32x2_f32, accumulators in VGPRs        MF 1: the same, accumulators in AccVGPRs (inline asm, "a" constraint)
//   MF 2: v_mfma_f32_16x16x4_f32, accumulators in VGPRs        MF 3: the same, accumulators in AccVGPRs
// and the fillers v_fma_f32 / v_add_u32 / v_max_f32 / v_cndmask_b32.  One or two waves per SIMD, REPS x { 4 independent MFMAs, K fillers after each }.
// Reported: shader cycles per MFMA in a wave's own stream and per SIMD (span of the waves on a SIMD / MFMAs issued there).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int MF, int FILL>
__global__ void __launch_bounds__(512) kern(float* out, unsigned long long* cyc, int reps) {   // cyc: [wave][3] = hw_id, t0, t1
  f32x16 acc[4];
  f32x4 acs[4];
  for (int t = 0; t < 4; ++t) { for (int r = 0; r < 16; ++r) acc[t][r] = 0.f; for (int r = 0; r < 4; ++r) acs[t][r] = 0.f; }
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float f[16];
  unsigned u[16];
  for (int i = 0; i < 16; ++i) { f[i] = a + i; u[i] = threadIdx.x + i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (MF == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
      if (MF == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
      if (MF == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acs[t]) : "v"(a), "v"(b));
      if (MF == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acs[t]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[k]) : "v"(b));
        if (FILL == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[15]));
        if (FILL == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[k]) : "v"(b));
        if (FILL == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[k]) : "v"(b));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) { for (int r = 0; r < 16; ++r) s += acc[t][r]; for (int r = 0; r < 4; ++r) s += acs[t][r]; }
  for (int i = 0; i < 16; ++i) s += f[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int w = threadIdx.x >> 6;
    cyc[w * 3 + 0] = hw; cyc[w * 3 + 1] = t0; cyc[w * 3 + 2] = t1;
  }
}

static const char* MFN[4] = {"f32 32x32x2 acc in VGPRs", "f32 32x32x2 acc in AGPRs", "f32 16x16x4 acc in VGPRs", "f32 16x16x4 acc in AGPRs"};
static const char* FN[4] = {"v_fma_f32", "v_add_u32", "v_max_f32", "v_cndmask_b32"};

template <int K, int MF, int FILL = 0>
void run(float* out, unsigned long long* cyc, int waves_per_simd) {
  const int reps = 2000;
  const int nw = 4 * waves_per_simd;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kern<K, MF, FILL>), dim3(1), dim3(64 * nw), 0, 0, out, cyc, reps);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
  unsigned long long c[8 * 3];
  if (hipMemcpy(c, cyc, sizeof(unsigned long long) * 3 * nw, hipMemcpyDeviceToHost) != hipSuccess) return;
  int cnt[4] = {0, 0, 0, 0};
  unsigned long long lo[4], hi[4];
  double own = 0.0;
  for (int w = 0; w < nw; ++w) {
    const int simd = (int)((c[w * 3] >> 4) & 3);          // HW_ID[5:4] = SIMD_ID on gfx9
    if (cnt[simd] == 0) { lo[simd] = c[w * 3 + 1]; hi[simd] = c[w * 3 + 2]; }
    else { if (c[w * 3 + 1] < lo[simd]) lo[simd] = c[w * 3 + 1]; if (c[w * 3 + 2] > hi[simd]) hi[simd] = c[w * 3 + 2]; }
    cnt[simd]++;
    own += (double)(c[w * 3 + 2] - c[w * 3 + 1]) / (reps * 4.0) / nw;
  }
  bool paired = true;
  double pipe = 0.0;
  for (int s = 0; s < 4; ++s) {
    if (cnt[s] != waves_per_simd) paired = false;
    if (cnt[s]) { const double v = (double)(hi[s] - lo[s]) / (reps * 4.0 * cnt[s]); pipe = pipe > v ? pipe : v; }
  }
  printf("%-26s waves/SIMD %d  K=%2d %-14s %7.1f cycles per MFMA in a wave's own stream, %6.1f matrix-pipe cycles per MFMA on the busiest SIMD%s\n",
         MFN[MF], waves_per_simd, K, FN[FILL], own, pipe, paired ? "" : "  NOT PAIRED");
}

template <int MF>
void sweep(float* out, unsigned long long* cyc, int w) {
  run<0, MF, 0>(out, cyc, w); run<1, MF, 0>(out, cyc, w); run<2, MF, 0>(out, cyc, w); run<4, MF, 0>(out, cyc, w); run<8, MF, 0>(out, cyc, w);
  run<2, MF, 1>(out, cyc, w); run<4, MF, 1>(out, cyc, w); run<8, MF, 1>(out, cyc, w);
  run<4, MF, 2>(out, cyc, w); run<4, MF, 3>(out, cyc, w);
}

int main() {
  float* out; unsigned long long* cyc;
  if (hipMalloc(&out, 4096 * 4) != hipSuccess || hipMalloc(&cyc, 8 * 3 * 8) != hipSuccess) return 1;
  for (int w = 1; w <= 2; ++w) { sweep<0>(out, cyc, w); sweep<1>(out, cyc, w); sweep<2>(out, cyc, w); sweep<3>(out, cyc, w); }
  return 0;
}
