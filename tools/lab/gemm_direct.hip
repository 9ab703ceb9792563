// Kernel lab: an LDS-free, barrier-free K loop for the small-grid GEMMs (tiles one MFMA tile tall: WGM == 1).
//
// The shipped small tiles (32 x 32..256, K split over wave groups) stage A and B through LDS-DMA and run at ~0.7 of a CU's
// matrix rate: ~20 KB of DMA per 64 MFMAs, 80 % of it the weight panel, which no other wave of the tile re-uses (every wave
// owns its own 32 output columns).  Here each wave loads its MFMA operands STRAIGHT from L2 / L1 into registers in operand
// layout — lane (row | column = lane & 31, k-half h = lane >> 5) reads 16 bytes (k = 8g + 4h + 0..3) per k-group g — with a
// PD-step register prefetch; no LDS, no DMA issue, no barrier until the K-split reduction.  A is re-read by the WGN waves
// of a group (L1 hits).  Same chunk order, same k order inside a chunk, same group-order reduction as k_conv_gemm.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_direct.hip -o gemm_direct
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "gemm_conv.hip"

using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int WGN, int KS, int PD>
__global__ __launch_bounds__(64 * WGN * KS) void k_gemm_direct(ConvGemm p, int ntn) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 32, BN = 32 * WGN, BK = 32, NG = BK / 8;
  const int tile_m = blockIdx.x / ntn, tile_n = blockIdx.x % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wall / WGN, wn = wall % WGN;
  const int r = lane & 31, h = lane >> 5;
  const int Kt = p.KW * p.Cin, cpj = p.Cin / BK, nch = p.KW * cpj;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + ((ptrdiff_t)m0 - p.pad) * p.ldx), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)(n0 + wn * 32) * Kt), (short)0, 0x7FFFFFFF, 0x00020000);
  // A: row m0 + r shifted by tap j; valid taps [jlo, jlo + jn)
  const int m = m0 + r;
  const int t = (m < p.M) ? (m % p.S) : -1;
  const int jlo = max(0, p.pad - t), jhi = min(p.KW, p.S + p.pad - t);
  const unsigned a_jlo = (unsigned)jlo, a_jn = (t >= 0 && jhi > jlo) ? (unsigned)(jhi - jlo) : 0u;
  const int a_base = (r * p.ldx + 4 * h) * 4;
  const int b_base = (n0 + wn * 32 + r < p.N) ? (r * Kt + 4 * h) * 4 : OOR;

  f32x4 a[PD][NG], b[PD][NG];
  auto load = [&](int u, int ch) {
    const int cc = ch / p.KW, j = ch - cc * p.KW;
    const int soA = (cc * BK + j * p.ldx) * 4, soB = (j * p.Cin + cc * BK) * 4;
    const int va = ((unsigned)j - a_jlo < a_jn) ? a_base : OOR;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      a[u][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, va == OOR ? OOR : va + 32 * g, soA, 0));
      b[u][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, b_base == OOR ? OOR : b_base + 32 * g, soB, 0));
    }
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nsteps = (nch + KS - 1) / KS;
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (u * KS + grp < nch) load(u, u * KS + grp);
  for (int st = 0; st < nsteps; st += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int ch = (st + u) * KS + grp;
      if (ch < nch) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][g][e], b[u][g][e], acc, 0, 0, 0);
      }
      if (ch + PD * KS < nch) load(u, ch + PD * KS);
    }
  }
  // K-split reduction + bias / activation through LDS, all waves
  constexpr int RS = BN + 8;
  __shared__ __attribute__((aligned(16))) float part[KS * BM * RS];
  const int ecol = lane & 31, erow = (lane >> 5) * 4;
#pragma unroll
  for (int i = 0; i < 16; ++i) part[grp * BM * RS + ((i & 3) + 8 * (i >> 2) + erow) * RS + wn * 32 + ecol] = acc[i];
  __syncthreads();
  constexpr int NT = 64 * WGN * KS, U = BM * BN / 4, CPRW = BN / 4;
  for (int u = tid; u < U; u += NT) {
    const int row = u / CPRW, c4 = u % CPRW, mm = m0 + row, n = n0 + c4 * 4;
    if (mm >= p.M || n >= p.N) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + row * RS + c4 * 4);
#pragma unroll
    for (int g2 = 1; g2 < KS; ++g2) v += *reinterpret_cast<const f32x4*>(part + g2 * BM * RS + row * RS + c4 * 4);
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.act == ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(p.Y + (size_t)mm * p.ldy + n) = v;
  }
#endif
}

template <int WGN, int KS, int PD>
static void run_direct(const ConvGemm& p, double gf, const std::vector<float>& ref, float* hy_dev) {
  const int ntm = (p.M + 31) / 32, ntn = (p.N + 32 * WGN - 1) / (32 * WGN);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 20;
  CK(hipMemset(p.Y, 0, (size_t)p.M * p.N * 4));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gemm_direct<WGN, KS, PD>), dim3(ntm * ntn), dim3(64 * WGN * KS), 0, 0, p, ntn);
  CK(hipDeviceSynchronize());
  std::vector<float> hy((size_t)p.M * p.N);
  CK(hipMemcpy(hy.data(), p.Y, hy.size() * 4, hipMemcpyDeviceToHost));
  double md = 0; size_t nbits = 0;
  for (size_t i = 0; i < hy.size(); ++i) { md = fmax(md, fabs((double)hy[i] - ref[i])); nbits += memcmp(&hy[i], &ref[i], 4) != 0; }
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_gemm_direct<WGN, KS, PD>), dim3(ntm * ntn), dim3(64 * WGN * KS), 0, 0, p, ntn);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
  printf("   direct 32x%3d KS=%d PD=%d  %4d wgs %8.1f us  %6.1f TF/s   max|diff| %.2e  differing words %zu\n", 32 * WGN, KS, PD, ntm * ntn, ms * 1e3, gf / ms, md, nbits);
  (void)hy_dev;
}

int main() {
  struct Shape { const char* name; int M, S, Cin, KW, N; } shapes[] = {
    {"dec conv9  M788  k9 256->1024 ", 788, 788, 256, 9, 1024},
    {"postnet    M788  k5 512->512  ", 788, 788, 512, 5, 512},
    {"dec w2     M788  k1 1024->256 ", 788, 788, 1024, 1, 256},
    {"dec qkv    M788  k1 256->768  ", 788, 788, 256, 1, 768},
    {"dec fc     M788  k1 256->256  ", 788, 788, 256, 1, 256},
    {"enc conv9  M100  k9 256->1024 ", 100, 100, 256, 9, 1024},
    {"enc qkv    M100  k1 256->768  ", 100, 100, 256, 1, 768},
    {"b16 conv9  M2048 k9 256->1024 ", 2048, 128, 256, 9, 1024},
    {"dec fc     M16160 k1 256->256 ", 16160, 1010, 256, 1, 256},
    {"dec w2     M16160 k1 1024->256", 16160, 1010, 1024, 1, 256},
    {"pred k3    M16160 k3 256->256 ", 16160, 1010, 256, 3, 256},
  };
  for (auto& s : shapes) {
    size_t nx = (size_t)s.M * s.Cin, nw = (size_t)s.N * s.KW * s.Cin, ny = (size_t)s.M * s.N;
    std::vector<float> hx(nx), hw(nw), hb(s.N);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    float *dx, *dw, *db, *dy;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice));
    ConvGemm p; memset(&p, 0, sizeof(p)); p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = ACT_RELU;
    const double gf = 2.0 * s.M * s.Cin * s.KW * s.N / 1e9;
    printf("%s %6.2f GFLOP\n", s.name, gf);
    // shipped
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(launch_conv_gemm(p, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> ref(ny);
    CK(hipMemcpy(ref.data(), dy, ny * 4, hipMemcpyDeviceToHost));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(p, 0));
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
    printf("   shipped launch_conv_gemm            %8.1f us  %6.1f TF/s\n", ms * 1e3, gf / ms);
    run_direct<4, 2, 2>(p, gf, ref, dy);
    run_direct<4, 2, 3>(p, gf, ref, dy);
    run_direct<4, 4, 2>(p, gf, ref, dy);
    run_direct<4, 4, 3>(p, gf, ref, dy);
    run_direct<2, 4, 3>(p, gf, ref, dy);
    run_direct<2, 8, 2>(p, gf, ref, dy);
    run_direct<1, 8, 3>(p, gf, ref, dy);
    run_direct<1, 16, 2>(p, gf, ref, dy);
    run_direct<8, 1, 3>(p, gf, ref, dy);
    run_direct<8, 2, 2>(p, gf, ref, dy);
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
