// What does the boundary between two dependent kernels of one stream cost on this chip, and does a hipGraph change it?
// (developer microbenchmark; hipcc --offload-arch=gfx950 -O3 -o launch_chain.bin launch_chain.hip)
//   S  plain stream launches           G  the same chain captured once into a hipGraph and replayed
// for  (a) an empty kernel, 256 x 256 threads          (host- or command-processor-bound)
//      (b) a kernel that spins ~5 us in every workgroup (GPU-bound: per-launch time minus the spin = the boundary)
//      (c) (b) with a 448-byte argument struct whose last field is read (the size of the convolution kernels' arguments)
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { float* p; long spin; long pad[53]; long last; };     // 448 bytes
static_assert(sizeof(Big) == 448, "argument struct");

__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 1.f; }

__global__ void spin_kernel(float* p, long spin) {      // spin: ticks of the 100 MHz constant clock
  const unsigned long long t0 = wall_clock64();
  while ((long)(wall_clock64() - t0) < spin) __builtin_amdgcn_s_sleep(1);
  if (p == nullptr) p[0] = 1.f;
}

__global__ void spin_big_kernel(const Big a) {
  const unsigned long long t0 = wall_clock64();
  while ((long)(wall_clock64() - t0) < a.spin + a.last) __builtin_amdgcn_s_sleep(1);
  if (a.p == nullptr) a.p[0] = 1.f;
}

template <class F>
static int chain(const char* what, int n, F launch, double spin_us) {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 50; ++i) launch(s);
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < n; ++i) launch(s);
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms_s = 0.f;
  CK(hipEventElapsedTime(&ms_s, e0, e1));

  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) launch(s);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms_g = 0.f;
  CK(hipEventElapsedTime(&ms_g, e0, e1));
  printf("%-58s S %6.2f us/launch (boundary %5.2f)   G %6.2f us/launch (boundary %5.2f)\n", what, ms_s * 1e3 / n, ms_s * 1e3 / n - spin_us,
         ms_g * 1e3 / n, ms_g * 1e3 / n - spin_us);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
  return 0;
}

int main() {
  float* sink;
  CK(hipMalloc(&sink, 4));
  const int n = 1000;
  const long spin = 500;                                // 5 us at 100 MHz
  int wc = 0;
  CK(hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0));   // kHz
  const double spin_us = spin * 1e3 / (double)wc;
  printf("wall clock %d kHz: spin of %ld ticks = %.2f us\n", wc, spin, spin_us);
  Big big{}; big.p = sink; big.spin = spin; big.last = 0;
  if (chain("a  empty, 256 x 256 threads", n, [&](hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, sink); }, 0.0)) return 1;
  if (chain("a' empty, 1 x 64 threads", n, [&](hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, sink); }, 0.0)) return 1;
  if (chain("b  5 us spin, 256 x 512 threads", n, [&](hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), 0, s, sink, spin); }, spin_us)) return 1;
  if (chain("b' 5 us spin, 256 x 512 threads, 64 KB LDS", n, [&](hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), 65536, s, sink, spin); }, spin_us)) return 1;
  if (chain("c  5 us spin, 448-byte arguments", n, [&](hipStream_t s) { hipLaunchKernelGGL(spin_big_kernel, dim3(256), dim3(512), 0, s, big); }, spin_us)) return 1;
  if (chain("d  5 us spin, 32 x 512 threads", n, [&](hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(512), 0, s, sink, spin); }, spin_us)) return 1;
  return 0;
}
