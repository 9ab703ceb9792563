// What does a stage boundary cost on MI355X: a launch boundary vs an in-kernel grid barrier?
//
//   hipcc -O3 --offload-arch=gfx950 -o grid_barrier grid_barrier.hip && ./grid_barrier
//
// Stage: every workgroup reads 16 KB that ANOTHER workgroup (a different XCD) wrote in the previous stage, adds 1,
// writes its own 16 KB.  (a) one launch per stage on one stream; (b) one persistent launch, stages separated by a
// monotonic-counter grid barrier, data published with sc1 (write-through) stores and read with sc1 loads — the protocol
// the ticketed row epilogues use; (c) the barrier alone; (d) an empty kernel per stage.
// Every spin is bounded: a barrier that does not complete within ~20 ms sets a flag and the kernel exits.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;           // threads per workgroup
constexpr int FPW = 4096;         // floats per workgroup and stage (16 KB)

__device__ __forceinline__ bool grid_barrier(int* counter, int target, int* abort_flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 2000000ll) { *abort_flag = 1; break; }  // 100 MHz clock: 20 ms
    }
  }
  __syncthreads();
  return true;
}

__global__ __launch_bounds__(NT) void k_stage(const float* in, float* out, int shift) {
  const int nb = gridDim.x, src = (blockIdx.x + shift) % nb;
  const f32x4* s = reinterpret_cast<const f32x4*>(in + (size_t)src * FPW);
  f32x4* d = reinterpret_cast<f32x4*>(out + (size_t)blockIdx.x * FPW);
#pragma unroll
  for (int i = 0; i < FPW / 4 / NT; ++i) {
    f32x4 v = s[threadIdx.x + i * NT];
    v += 1.0f;
    d[threadIdx.x + i * NT] = v;
  }
}

__global__ __launch_bounds__(NT) void k_empty(int* p) {
  if (p && threadIdx.x == 9999) *p = 1;
}

template <bool DATA>
__global__ __launch_bounds__(NT) void k_persistent(float* a, float* b, int nstage, int shift, int* counter, int* abort_flag) {
  const int nb = gridDim.x;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a, (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)b, (short)0, 0x7FFFFFFF, 0x00020000);
  for (int st = 0; st < nstage; ++st) {
    if (DATA) {
      const __amdgpu_buffer_rsrc_t rin = (st & 1) ? rb : ra, rout = (st & 1) ? ra : rb;
      const int src = (blockIdx.x + shift) % nb;
      f32x4 v[FPW / 4 / NT];
#pragma unroll
      for (int i = 0; i < FPW / 4 / NT; ++i)
        v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (src * FPW + (threadIdx.x + i * NT) * 4) * 4, 0, 16 /* sc1 */));
#pragma unroll
      for (int i = 0; i < FPW / 4 / NT; ++i) {
        v[i] += 1.0f;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), rout, (blockIdx.x * FPW + (threadIdx.x + i * NT) * 4) * 4, 0, 16 /* sc1 */);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    grid_barrier(counter, (st + 1) * nb, abort_flag);
    if (*(volatile int*)abort_flag) return;
  }
}

int main() {
  const int NB = 256, NSTAGE = 200;
  float *a, *b;
  int *counter, *abort_flag;
  CK(hipMalloc(&a, (size_t)NB * FPW * 4));
  CK(hipMalloc(&b, (size_t)NB * FPW * 4));
  CK(hipMalloc(&counter, 8));
  abort_flag = counter + 1;
  CK(hipMemset(a, 0, (size_t)NB * FPW * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timed = [&](const char* name, auto fn) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemsetAsync(counter, 0, 8, st));
      CK(hipEventRecord(e0, st));
      fn();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-58s %8.2f us per stage\n", name, best * 1e3f / NSTAGE);
  };
  for (int shift : {0, 1, 3}) {
    printf("shift %d (source workgroup = me + shift; XCD = workgroup %% 8)\n", shift);
    timed("(a) one launch per stage (16 KB in, 16 KB out per WG)", [&] {
      for (int s = 0; s < NSTAGE; ++s) hipLaunchKernelGGL(k_stage, dim3(NB), dim3(NT), 0, st, (s & 1) ? b : a, (s & 1) ? a : b, shift);
    });
    timed("(b) persistent, grid barrier + sc1 data", [&] {
      hipLaunchKernelGGL(k_persistent<true>, dim3(NB), dim3(NT), 0, st, a, b, NSTAGE, shift, counter, abort_flag);
    });
  }
  timed("(c) persistent, grid barrier alone", [&] {
    hipLaunchKernelGGL(k_persistent<false>, dim3(NB), dim3(NT), 0, st, a, b, NSTAGE, 0, counter, abort_flag);
  });
  timed("(d) empty kernel per stage", [&] {
    for (int s = 0; s < NSTAGE; ++s) hipLaunchKernelGGL(k_empty, dim3(NB), dim3(NT), 0, st, (int*)nullptr);
  });
  // check the data path of (b): after NSTAGE stages every float is NSTAGE (even count: result sits in a)
  CK(hipMemsetAsync(counter, 0, 8, st));
  CK(hipMemsetAsync(a, 0, (size_t)NB * FPW * 4, st));
  hipLaunchKernelGGL(k_persistent<true>, dim3(NB), dim3(NT), 0, st, a, b, NSTAGE, 3, counter, abort_flag);
  CK(hipStreamSynchronize(st));
  std::vector<float> h((size_t)NB * FPW);
  int hab[2];
  CK(hipMemcpy(h.data(), a, h.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hab, counter, 8, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (float v : h) bad += v != (float)NSTAGE;
  printf("persistent data check: %zu wrong of %zu, abort flag %d\n", bad, h.size(), hab[1]);
  // graph of the per-stage launches
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int s = 0; s < NSTAGE; ++s) hipLaunchKernelGGL(k_stage, dim3(NB), dim3(NT), 0, st, (s & 1) ? b : a, (s & 1) ? a : b, 3);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  timed("(e) hipGraph of (a)", [&] { CK(hipGraphLaunch(ge, st)); });
  return 0;
}
