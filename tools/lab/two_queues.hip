// Kernel lab (round 4): do kernels of two HIP streams (two HSA queues) overlap on this chip, and what stops them?
//   hipcc -O3 --offload-arch=gfx950 -o two_queues two_queues.hip
// A chain of `n` dependent kernels per stream; each kernel = `wgs` workgroups of 256 threads spinning ~`us` microseconds, holding
// `lds` bytes of LDS.  Reported: both chains on ONE stream (serial) against one chain per stream (two queues).  If the two-queue
// time is about half the serial one, the queues overlap and what serialised the forward-level experiments
// (subbatch_two_streams.py, the forked remainder in gemm_lab_plan.hip) is resources — one chain's launches holding every CU's
// LDS / wave slots — not the command processor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_spin(long long ticks, int lds_floats, float* sink) {
  extern __shared__ float smem[];
  if (lds_floats > 0) smem[threadIdx.x % lds_floats] = (float)threadIdx.x;
  const long long t0 = wall_clock64();
  float acc = 0.f;
  while (wall_clock64() - t0 < ticks) acc += 1.0f;
  if (acc < 0.f) sink[0] = acc + (lds_floats > 0 ? smem[0] : 0.f);
}

int main() {
  float* sink; CK(hipMalloc(&sink, 4));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  struct Case { int wgs, lds; float us; } cases[] = {{64, 0, 10.f}, {256, 0, 10.f}, {256, 32 * 1024, 10.f}, {256, 96 * 1024, 10.f}, {256, 128 * 1024, 10.f},
                                                     {512, 64 * 1024, 10.f}, {1024, 16 * 1024, 10.f}, {128, 96 * 1024, 10.f}};
  const int n = 50;
  for (auto& c : cases) {
    const long long ticks = (long long)(c.us * 100.0f);  // wall_clock64: 100 MHz
    auto chain = [&](hipStream_t st) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(c.wgs), dim3(256), c.lds, st, ticks, c.lds / 4, sink); };
    auto timed = [&](bool two) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, s0));
        chain(s0);
        if (two) chain(s1); else chain(s0);
        CK(hipStreamSynchronize(s1)); CK(hipEventRecord(b, s0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        // (two queues: b is recorded on s0 after the host has waited for s1, so it bounds both chains)
        if (ms < best) best = ms;
      }
      return best * 1e3f / n;
    };
    const float ser = timed(false), par = timed(true);
    printf("%4d workgroups, %3d KB LDS, %.0f us spin: 2 chains on one stream %.1f us per pair, one chain per stream %.1f us per pair (ratio %.2f)\n",
           c.wgs, c.lds / 1024, c.us, ser, par, par / ser);
  }
  return 0;
}
