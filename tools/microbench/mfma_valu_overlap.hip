// Does vector-ALU work overlap with fp32 matrix instructions on gfx950?  (developer microbenchmark; hipcc --offload-arch=gfx950)
// One wave per SIMD runs REPS x { 4 independent v_mfma_f32_32x32x2_f32 + K independent v_fma_f32 } and reports shader cycles
// per MFMA, for K = 0, 2, 4, 8, 16; the same with v_mfma_f32_32x32x16_bf16 as the matrix instruction for comparison.
// r03 repair: every wave records its HW_ID (SIMD it runs on) and its own start / end stamps.  The two-waves-per-SIMD
// rows are reported PER SIMD: matrix-pipe cycles per MFMA = (last end - first start of the waves on that SIMD) / MFMAs issued on it, which
// cannot be below 64 for v_mfma_f32_32x32x2_f32 if the waves really share the SIMD; rows whose waves did not pair up say so.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, bool BF16, int FILL>
__global__ void __launch_bounds__(512) kern(float* out, unsigned long long* cyc, int reps) {   // cyc: [wave][3] = hw_id, t0, t1
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const float* lp = lds + (threadIdx.x & 63) * 33;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  bf16x8 ab, bb;
  for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)a; bb[e] = (__bf16)b; }
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = a + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (BF16) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
      else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[k]) : "v"(b));
        else if (FILL == 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[k]) : "v"((unsigned)(size_t)lp), "n"(k * 4));
        else asm volatile("ds_write_b32 %1, %0 offset:%2" :: "v"(f[k]), "v"((unsigned)(size_t)lp), "n"(k * 4));
      }
      if (FILL != 0 && t == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int w = threadIdx.x >> 6;
    cyc[w * 3 + 0] = hw; cyc[w * 3 + 1] = t0; cyc[w * 3 + 2] = t1;
  }
}

template <int K, bool BF16, int FILL = 0>
void run(float* out, unsigned long long* cyc, int waves_per_simd) {
  const int reps = 2000;
  const int nw = 4 * waves_per_simd;
  hipLaunchKernelGGL((kern<K, BF16, FILL>), dim3(1), dim3(64 * nw), 0, 0, out, cyc, reps);
  hipLaunchKernelGGL((kern<K, BF16, FILL>), dim3(1), dim3(64 * nw), 0, 0, out, cyc, reps);
  hipDeviceSynchronize();
  unsigned long long c[8 * 3];
  hipMemcpy(c, cyc, sizeof(unsigned long long) * 3 * nw, hipMemcpyDeviceToHost);
  // per SIMD: waves on it, span of their activity, MFMAs issued there
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
    if (cnt[s]) pipe = pipe > (double)(hi[s] - lo[s]) / (reps * 4.0 * cnt[s]) ? pipe : (double)(hi[s] - lo[s]) / (reps * 4.0 * cnt[s]);
  }
  printf("%s  waves/SIMD %d  K=%2d %-12s per MFMA: %7.1f cycles per MFMA in a wave's own stream, %6.1f matrix-pipe cycles per MFMA on the busiest SIMD"
         "  [waves per SIMD %d %d %d %d%s]\n", BF16 ? "bf16 32x32x16" : "f32  32x32x2 ", waves_per_simd, K,
         FILL == 0 ? "v_fma_f32" : FILL == 1 ? "ds_read_b32" : "ds_write_b32", own, pipe, cnt[0], cnt[1], cnt[2], cnt[3], paired ? "" : "  NOT PAIRED");
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8 * 3 * 8);
  for (int w = 1; w <= 2; ++w) {
    run<0, false>(out, cyc, w); run<2, false>(out, cyc, w); run<4, false>(out, cyc, w); run<8, false>(out, cyc, w); run<16, false>(out, cyc, w);
    run<0, true>(out, cyc, w); run<4, true>(out, cyc, w); run<8, true>(out, cyc, w); run<16, true>(out, cyc, w);
    run<1, false, 1>(out, cyc, w); run<2, false, 1>(out, cyc, w); run<4, false, 1>(out, cyc, w); run<8, false, 1>(out, cyc, w);
    run<1, false, 2>(out, cyc, w); run<2, false, 2>(out, cyc, w); run<4, false, 2>(out, cyc, w);
    run<2, true, 1>(out, cyc, w); run<4, true, 1>(out, cyc, w);
  }
  return 0;
}
