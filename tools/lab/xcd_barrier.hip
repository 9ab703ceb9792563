// Kernel lab (round 4): what does a stage boundary cost when every participant sits on ONE XCD?
//
//   hipcc -O3 --offload-arch=gfx950 -o xcd_barrier xcd_barrier.hip && ./xcd_barrier
//
// grid_barrier.hip (round 3) measured the chip-wide form: 256 workgroups on 8 XCDs, hand-over through memory (sc1), barrier
// 3.9 us, stage 6.4-8.9 us against 2.6-4.4 us for a launch boundary.  The question left open: a phase with only a few row
// tiles of work (the single-utterance encoder: 4 row tiles per GEMM) could run on the 32 CUs of ONE XCD, hand data over
// through that XCD's L2 (plain write-through-L1 stores, sc0 = L1-bypassing loads) and synchronise with atomics that resolve
// in that L2 (workgroup-scope RMWs carry no sc bits) — no trip to the memory side at all.
// Placement: workgroup b of a launch runs on XCD b % 8 (observed); the launch has 8 * NWG workgroups, those with b % 8 != 0
// exit at once, and every live workgroup records the XCC_ID it really ran on, so the run shows whether the assumption held.
// Stage: read 16 KB another live workgroup wrote in the previous stage, add 1, write 16 KB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256, FPW = 4096;

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// SCOPE 0: L2-local (workgroup-scope RMWs: no sc bits, resolved in this XCD's L2).  Polled with a REAL read-modify-write,
// fetch_max(counter, 0): `fetch_add(counter, 0)` is an idempotent RMW that LLVM turns into a plain atomic LOAD, and a
// workgroup-scope load may be served by the CU's L1 forever (the first version of this harness timed out that way with >= 16
// workgroups and passed with 8 by luck).  1: agent scope
template <int SCOPE>
__device__ __forceinline__ void bar(int* counter, int target, int* abort_flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t0 = wall_clock64();
    if (SCOPE == 0) {
      __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_fetch_max(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        if (wall_clock64() - t0 > 2000000ll) { *abort_flag = 1; break; }
      }
    } else {
      __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 2000000ll) { *abort_flag = 1; break; }
      }
    }
  }
  __syncthreads();
}

// AUX: cache bits of the data path — 1 = sc0 (bypass L1, served by this XCD's L2), 16 = sc1 (write-through / memory side)
template <bool DATA, int SCOPE, int AUX>
__global__ __launch_bounds__(NT) void k_persistent(float* a, float* b, int nstage, int shift, int stride, int* counter, int* abort_flag, int* where) {
  if (blockIdx.x % stride != 0) return;
  const int me = blockIdx.x / stride, nb = gridDim.x / stride;
  if (threadIdx.x == 0) where[me] = xcc_id();
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a, (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)b, (short)0, 0x7FFFFFFF, 0x00020000);
  for (int st = 0; st < nstage; ++st) {
    if (DATA) {
      const __amdgpu_buffer_rsrc_t rin = (st & 1) ? rb : ra, rout = (st & 1) ? ra : rb;
      const int src = (me + shift) % nb;
      f32x4 v[FPW / 4 / NT];
#pragma unroll
      for (int i = 0; i < FPW / 4 / NT; ++i)
        v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (src * FPW + (threadIdx.x + i * NT) * 4) * 4, 0, AUX == 17 ? 16 : AUX));
#pragma unroll
      for (int i = 0; i < FPW / 4 / NT; ++i) {
        v[i] += 1.0f;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), rout, (me * FPW + (threadIdx.x + i * NT) * 4) * 4, 0, AUX == 16 ? 16 : 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar<SCOPE>(counter, (st + 1) * nb, abort_flag);
    if (*(volatile int*)abort_flag) return;
  }
}

__global__ __launch_bounds__(NT) void k_stage(const float* in, float* out, int shift, int stride) {
  if (blockIdx.x % stride != 0) return;
  const int me = blockIdx.x / stride, nb = gridDim.x / stride, src = (me + shift) % nb;
  const f32x4* s = reinterpret_cast<const f32x4*>(in + (size_t)src * FPW);
  f32x4* d = reinterpret_cast<f32x4*>(out + (size_t)me * FPW);
#pragma unroll
  for (int i = 0; i < FPW / 4 / NT; ++i) {
    f32x4 v = s[threadIdx.x + i * NT];
    v += 1.0f;
    d[threadIdx.x + i * NT] = v;
  }
}

int main() {
  const int NSTAGE = 200;
  float *a, *b;
  int *counter, *abort_flag, *where;
  CK(hipMalloc(&a, (size_t)256 * FPW * 4));
  CK(hipMalloc(&b, (size_t)256 * FPW * 4));
  CK(hipMalloc(&counter, 8 + 256 * 4));
  abort_flag = counter + 1;
  where = counter + 2;
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
    printf("  %-72s %8.2f us per stage\n", name, best * 1e3f / NSTAGE);
  };
  auto check = [&](int nwg, int stride, auto launch) {
    CK(hipMemsetAsync(counter, 0, 8 + 256 * 4, st));
    CK(hipMemsetAsync(a, 0, (size_t)256 * FPW * 4, st));
    launch();
    CK(hipStreamSynchronize(st));
    std::vector<float> h((size_t)nwg * FPW);
    std::vector<int> hc(2 + 256);
    CK(hipMemcpy(h.data(), a, h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), counter, hc.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (float v : h) bad += v != (float)NSTAGE;
    int hist[16] = {0};
    for (int i = 0; i < nwg; ++i) hist[hc[2 + i] & 15]++;
    printf("    data check: %zu wrong of %zu, abort flag %d, live workgroups per XCC_ID:", bad, h.size(), hc[1]);
    for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
    printf("\n");
  };
  for (int nwg : {8, 16, 32, 64}) {
    printf("%d live workgroups, all meant for one XCD (stride 8)\n", nwg);
    const int grid = nwg * 8;
    for (int shift : {1, 3}) {
      printf(" shift %d\n", shift);
      timed("(a) one launch per stage", [&] {
        for (int s = 0; s < NSTAGE; ++s) hipLaunchKernelGGL(k_stage, dim3(grid), dim3(NT), 0, st, (s & 1) ? b : a, (s & 1) ? a : b, shift, 8);
      });
      timed("(b) persistent: L2-local barrier, data through this XCD's L2 (sc0 loads)", [&] {
        hipLaunchKernelGGL((k_persistent<true, 0, 1>), dim3(grid), dim3(NT), 0, st, a, b, NSTAGE, shift, 8, counter, abort_flag, where);
      });
      timed("(b2) persistent: L2-local barrier, sc1 loads + plain stores", [&] {
        hipLaunchKernelGGL((k_persistent<true, 0, 17>), dim3(grid), dim3(NT), 0, st, a, b, NSTAGE, shift, 8, counter, abort_flag, where);
      });
      timed("(c) persistent: agent-scope barrier, data through memory (sc1)", [&] {
        hipLaunchKernelGGL((k_persistent<true, 1, 16>), dim3(grid), dim3(NT), 0, st, a, b, NSTAGE, shift, 8, counter, abort_flag, where);
      });
    }
    timed("(d) persistent: L2-local barrier alone", [&] {
      hipLaunchKernelGGL((k_persistent<false, 0, 1>), dim3(grid), dim3(NT), 0, st, a, b, NSTAGE, 0, 8, counter, abort_flag, where);
    });
    timed("(e) persistent: agent-scope barrier alone", [&] {
      hipLaunchKernelGGL((k_persistent<false, 1, 16>), dim3(grid), dim3(NT), 0, st, a, b, NSTAGE, 0, 8, counter, abort_flag, where);
    });
    check(nwg, 8, [&] { hipLaunchKernelGGL((k_persistent<true, 0, 1>), dim3(grid), dim3(NT), 0, st, a, b, NSTAGE, 3, 8, counter, abort_flag, where); });
    check(nwg, 8, [&] { hipLaunchKernelGGL((k_persistent<true, 0, 17>), dim3(grid), dim3(NT), 0, st, a, b, NSTAGE, 3, 8, counter, abort_flag, where); });
  }
  printf("reference: the same 32 live workgroups spread over all XCDs (stride 1), L2-local protocol (expected to FAIL the data check)\n");
  timed("(b') persistent: L2-local barrier + sc0 data, 32 workgroups on 8 XCDs", [&] {
    hipLaunchKernelGGL((k_persistent<true, 0, 1>), dim3(32), dim3(NT), 0, st, a, b, NSTAGE, 3, 1, counter, abort_flag, where);
  });
  check(32, 1, [&] { hipLaunchKernelGGL((k_persistent<true, 0, 1>), dim3(32), dim3(NT), 0, st, a, b, NSTAGE, 3, 1, counter, abort_flag, where); });
  return 0;
}
