// Kernel lab (build with -DNS_LAB_XGN=1|2|4): XCD grouping along N for the workgroup remap.  Was: 64x64 register tiles per wave (2 LDS fragment reads per 8 MFMAs instead of 3) on the dominant shapes — VALU / LDS
// instruction issue is additive to MFMA time on a SIMD (mfma_mix.hip), so fewer fragment reads per MFMA should raise the ceiling.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_xgn.hip -o gemm_lab_xgn
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static float* g_ref = nullptr;
template <int BM, int BN, int BK, int KS, int WGM, int WGN>
void run(const ConvGemm& p, double gf, size_t ny) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipMemset(p.Y, 0xff, ny * 4));
  for (int i = 0; i < 3; ++i) CK((launch_t<BM, BN, BK, KS, WGM, WGN>(p, 0)));
  CK(hipDeviceSynchronize());
  std::vector<float> y(ny); CK(hipMemcpy(y.data(), p.Y, ny * 4, hipMemcpyDeviceToHost));
  double md = 0; for (size_t i = 0; i < ny; ++i) { double d = fabs((double)y[i] - g_ref[i]); if (!(d <= md)) md = d; }
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(a, 0)); for (int i = 0; i < 20; ++i) CK((launch_t<BM, BN, BK, KS, WGM, WGN>(p, 0)));
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms / 20 < best ? ms / 20 : best;
  }
  printf("   %3dx%3dx%2d KS=%d %dx%d waves %5d wgs %8.1f us %6.1f TF/s  maxdiff vs shipped %.1e\n", BM, BN, BK, KS, WGM, WGN,
         ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), best * 1e3, gf / best, md);
}
int main() {
  struct Shape { const char* name; int M, S, Cin, KW, N; } shapes[] = {
    {"dec w_1 cfg2 (k9 256->1024) M16160", 16160, 1010, 256, 9, 1024}, {"postnet (k5 512->512) M16160", 16160, 1010, 512, 5, 512},
    {"dec w_1 cfg5 (k9 256->1024) M31248", 31248, 3906, 256, 9, 1024}, {"dec QKV cfg2 (k1 256->768) M16160", 16160, 1010, 256, 1, 768},
    {"dec w_1 cfg4 (k9 512->1024) M68928", 68928, 1077, 512, 9, 1024},
  };
  for (auto& s : shapes) {
    size_t nx = (size_t)s.M * s.Cin, nw = (size_t)s.N * s.KW * s.Cin, ny = (size_t)s.M * s.N;
    std::vector<float> hx(nx), hw(nw), hb(s.N);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX;
    float *dx, *dw, *db, *dy;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice));
    ConvGemm p; memset(&p, 0, sizeof(p)); p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = ACT_RELU;
    double gf = 2.0 * s.M * s.Cin * s.KW * s.N / 1e9;
    for (int i = 0; i < 3; ++i) CK(launch_conv_gemm(p, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> ref(ny); CK(hipMemcpy(ref.data(), dy, ny * 4, hipMemcpyDeviceToHost)); g_ref = ref.data();
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 3; ++r) { CK(hipEventRecord(a, 0)); for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(p, 0)); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms / 20 < best ? ms / 20 : best; }
    printf("%s %.2f GFLOP\n   shipped choice %37.1f us %6.1f TF/s\n", s.name, gf, best * 1e3, gf / best);
    run<64, 256, 32, 1, 2, 4>(p, gf, ny);
    run<64, 256, 32, 1, 2, 4>(p, gf, ny);
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
