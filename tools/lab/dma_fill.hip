// Kernel lab: what one CU can pull through LDS-DMA (buffer_load_dwordx4 ... lds), as a function of the number of
// waves, the DMA instructions in flight per wave and the source's residency (L2 / Infinity Cache / HBM).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 dma_fill.hip -o dma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// every wave owns DEPTH KB of LDS and refills it `iters` times from its own slice of `src` (bytes wraps inside `span`)
template <int DEPTH>
__global__ __launch_bounds__(1024) void k_fill(const float* src, size_t span_bytes, int iters, float* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, 0x7FFFFFFF, 0x00020000);
  float* mine = lds + wid * DEPTH * 256;
  // consecutive (block, wave, iteration) slices of DEPTH KB, wrapped into the span
  size_t off = ((size_t)blockIdx.x * nw + wid) * DEPTH * 1024;
  const size_t stride = (size_t)gridDim.x * nw * DEPTH * 1024;
  for (int it = 0; it < iters; ++it) {
    const unsigned o = (unsigned)(off % span_bytes);
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(mine + i * 256), 16, lane * 16 + i * 1024, o, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    off += stride;
  }
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = lds[0];
#endif
}

template <int DEPTH>
void run(const float* src, size_t span, int blocks, int waves, float* sink, const char* tag) {
  const int iters = 2000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t lds_bytes = (size_t)waves * DEPTH * 1024;
  CK(hipFuncSetAttribute((const void*)k_fill<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((k_fill<DEPTH>), dim3(blocks), dim3(waves * 64), lds_bytes, 0, src, span, 50, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL((k_fill<DEPTH>), dim3(blocks), dim3(waves * 64), lds_bytes, 0, src, span, iters, sink);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)blocks * waves * DEPTH * 1024 * iters;
  printf("  %-14s blocks %3d waves %2d depth %2d KB/wave in flight %3d KB/CU: %7.1f GB/s per CU  %8.1f GB/s total  round trip %5.2f us\n",
         tag, blocks, waves, DEPTH, waves * DEPTH, bytes / ms / 1e6 / blocks, bytes / ms / 1e6, ms * 1e3 / iters);
}

int main() {
  const size_t big = (size_t)2 << 30;
  float *src, *sink; CK(hipMalloc(&src, big)); CK(hipMalloc(&sink, 4096)); CK(hipMemset(src, 0, big));
  struct { size_t span; const char* tag; } spans[] = {{(size_t)2 << 20, "L2 (2 MB)"}, {(size_t)64 << 20, "MALL (64 MB)"}, {big, "HBM (2 GB)"}};
  for (auto& sp : spans) {
    printf("source span: %s\n", sp.tag);
    for (int blocks : {32, 256}) {
      run<1>(src, sp.span, blocks, 4, sink, sp.tag);
      run<4>(src, sp.span, blocks, 4, sink, sp.tag);
      run<8>(src, sp.span, blocks, 4, sink, sp.tag);
      run<2>(src, sp.span, blocks, 8, sink, sp.tag);
      run<4>(src, sp.span, blocks, 8, sink, sp.tag);
      run<8>(src, sp.span, blocks, 8, sink, sp.tag);
      run<4>(src, sp.span, blocks, 16, sink, sp.tag);
      run<8>(src, sp.span, blocks, 16, sink, sp.tag);
    }
  }
  return 0;
}
