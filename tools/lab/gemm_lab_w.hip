// Kernel lab: times k_conv_gemm tile variants on the dominant shape (FFN w_1: M=16160, Cin=256, KW=9, N=1024)
// and a few others, standalone (no torch).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab.hip -o gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define NS_LAB 1
#include "gemm_conv.hip"

using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int BM, int BN, int BK, int WGM = 2, int WGN = 2>
float time_variant(ConvGemm p, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK((launch_t<BM, BN, BK, 1, WGM, WGN>(p, 0)));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) CK((launch_t<BM, BN, BK, 1, WGM, WGN>(p, 0)));
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main(int argc, char** argv) {
  struct Shape { const char* name; int M, S, Cin, KW, N; } shapes[] = {
    {"ffn_w1 dec  (k9 256->1024)", 16160, 1010, 256, 9, 1024},
    {"postnet mid (k5 512->512) ", 16160, 1010, 512, 5, 512},
    {"ffn_w2 dec  (k1 1024->256)", 16160, 1010, 1024, 1, 256},
    {"qkv dec     (k1 256->768) ", 16160, 1010, 256, 1, 768},
    {"pred conv   (k3 256->256) ", 16160, 1010, 256, 3, 256},
    {"ffn_w1 enc  (k9 256->1024)", 2048, 128, 256, 9, 1024},
    {"fc dec      (k1 256->256) ", 16160, 1010, 256, 1, 256},
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
    ConvGemm p; memset(&p, 0, sizeof(p)); p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.resid = nullptr; p.ldr = 0; p.Y = dy; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = ACT_RELU;
    double gf = 2.0 * s.M * s.Cin * s.KW * s.N / 1e9;
    printf("%s  %.1f GFLOP\n", s.name, gf);
#define RUN(BM, BN, BK) { float ms = time_variant<BM, BN, BK>(p, 10); printf("   %3dx%3dx%2d  %8.1f us  %6.1f TF/s\n", BM, BN, BK, ms * 1e3, gf / ms); }
#define RUNW(BM, BN, BK, WGM, WGN) { float ms = time_variant<BM, BN, BK, WGM, WGN>(p, 10); printf("   %3dx%3dx%2d waves %dx%d %8.1f us  %6.1f TF/s\n", BM, BN, BK, WGM, WGN, ms * 1e3, gf / ms); }
    RUNW(64, 128, 32, 2, 4) RUNW(64, 128, 32, 2, 2) RUNW(128, 128, 32, 4, 4) RUNW(64, 256, 32, 2, 8) RUNW(64, 256, 32, 2, 4) RUNW(128, 64, 32, 4, 2) RUNW(64, 64, 32, 2, 2) RUNW(32, 256, 32, 1, 8) RUNW(128, 256, 32, 4, 4) RUNW(32, 128, 32, 1, 4) RUNW(32, 64, 32, 1, 2) RUNW(64, 64, 32, 2, 2)
#ifdef NS_LAB_EXTRA
    NS_LAB_EXTRA
#endif
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
