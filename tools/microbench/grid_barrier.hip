// What would a persistent per-dense-block kernel pay per layer on this chip?  (developer microbenchmark; hipcc --offload-arch=gfx950)
// Measures, with HIP events around the launch and R rounds inside it:
//   A  one monotonic-counter barrier among 256 workgroups, one per CU (whole chip)
//   B  the same among the 32 workgroups of ONE XCD (launch 256, keep blockIdx % 8 == 0; XCC_ID verified)
//   C  the same among 32 workgroups spread over all XCDs (blockIdx < 32)
//   D  B / C with a 4 KB tile per workgroup published with write-through stores before the barrier and a neighbour's tile read
//      with sc1 loads after it (the hand-off a layer boundary needs: activations of the previous phase)
//   E  the boundary between dependent empty kernels of the same stream, for comparison
// Every spin loop has an iteration bound: a barrier that cannot complete sets an error flag and the kernel drains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

struct Args {
  unsigned* counter;      // zeroed before the launch
  unsigned* error;        // set when a spin loop gave up
  unsigned* xcc;          // [grid] XCC id of every participating workgroup
  float* tiles;           // [grid][1024] hand-off payload
  float* sink;
  int rounds, mode;       // mode 0: all blocks take part; 1: blockIdx % 8 == 0; 2: blockIdx < 32
  int payload;            // 0 / 1
};

__global__ void __launch_bounds__(256) barrier_kernel(const Args a) {
  const int b = blockIdx.x;
  bool in = a.mode == 0 || (a.mode == 1 && (b & 7) == 0) || (a.mode == 2 && b < 32);
  if (!in) return;
  int nwg = a.mode == 0 ? gridDim.x : 32;
  int me = a.mode == 1 ? b >> 3 : b;
  if (threadIdx.x == 0) a.xcc[b] = xcc_id();
  float acc = 0.f;
  __shared__ int give_up;
  if (threadIdx.x == 0) give_up = 0;
  __syncthreads();
  for (int r = 0; r < a.rounds; ++r) {
    if (a.payload) {
      float* mine = a.tiles + (size_t)b * 1024;
      for (int i = threadIdx.x; i < 1024; i += 256) __hip_atomic_store(mine + i, (float)(r + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * (unsigned)nwg;
      int spins = 0;
      while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000) { give_up = 1; atomicExch(a.error, 1u); break; }
      }
    }
    __syncthreads();
    if (give_up) return;
    if (a.payload) {
      const int nb = a.mode == 1 ? (((me + 1) % nwg) << 3) : ((me + 1) % nwg);
      const float* theirs = a.tiles + (size_t)nb * 1024;
      for (int i = threadIdx.x; i < 1024; i += 256) acc += __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 1.2345f) a.sink[0] = acc;
}

__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 1.f; }

static double run(Args a, int grid, const char* what) {
  hipMemset(a.counter, 0, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  Args warm = a; warm.rounds = 4;
  hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), 0, 0, warm);
  hipDeviceSynchronize();
  hipMemset(a.counter, 0, 4);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned err = 0;
  hipMemcpy(&err, a.error, 4, hipMemcpyDeviceToHost);
  std::vector<unsigned> x(grid);
  hipMemcpy(x.data(), a.xcc, 4 * grid, hipMemcpyDeviceToHost);
  int hist[16] = {0};
  for (int b = 0; b < grid; ++b) {
    bool in = a.mode == 0 || (a.mode == 1 && (b & 7) == 0) || (a.mode == 2 && b < 32);
    if (in) hist[x[b] & 15]++;
  }
  printf("%-62s %7.2f us per round%s   workgroups per XCC:", what, ms * 1e3 / a.rounds, err ? "  [A SPIN LOOP GAVE UP]" : "");
  for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
  printf("\n");
  return ms * 1e3 / a.rounds;
}

int main() {
  Args a;
  hipMalloc(&a.counter, 4); hipMalloc(&a.error, 4); hipMalloc(&a.xcc, 4 * 256); hipMalloc(&a.tiles, 256 * 1024 * 4); hipMalloc(&a.sink, 4);
  hipMemset(a.error, 0, 4); hipMemset(a.xcc, 0xFF, 4 * 256); hipMemset(a.tiles, 0, 256 * 1024 * 4);
  a.rounds = 200;
  a.payload = 0;
  a.mode = 0; run(a, 256, "A  counter barrier, 256 workgroups (one per CU)");
  a.mode = 1; run(a, 256, "B  counter barrier, 32 workgroups of one XCD");
  a.mode = 2; run(a, 256, "C  counter barrier, 32 workgroups over all XCDs");
  a.payload = 1;
  a.mode = 0; run(a, 256, "D0 barrier + 4 KB sc1 hand-off per workgroup, 256 workgroups");
  a.mode = 1; run(a, 256, "D1 barrier + 4 KB sc1 hand-off per workgroup, one XCD");
  a.mode = 2; run(a, 256, "D2 barrier + 4 KB sc1 hand-off per workgroup, 32 over all XCDs");
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, a.sink);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, a.sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-62s %7.2f us per launch\n", "E  dependent empty kernels (256 x 256 threads), same stream", ms * 1e3 / 400);
  return 0;
}
