// Kernel lab: times k_conv_gemm tile variants on the small-grid shapes (single utterance, encoder side) against
// launch_conv_gemm's own choice, standalone (no torch).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab.hip -o gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define NS_LAB 1
#include "gemm_conv.hip"

using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int BM, int BN, int BK, int KS, int WGM, int WGN>
void run(const ConvGemm& p, double gf) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 20;
  for (int i = 0; i < 3; ++i) CK((launch_t<BM, BN, BK, KS, WGM, WGN>(p, 0)));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) CK((launch_t<BM, BN, BK, KS, WGM, WGN>(p, 0)));
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  ms /= iters;
  const int wgs = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  printf("   %3dx%3dx%2d KS=%d %dx%d waves  %4d wgs %8.1f us  %6.1f TF/s\n", BM, BN, BK, KS, WGM, WGN, wgs, ms * 1e3, gf / ms);
}

int main(int argc, char** argv) {
  struct Shape { const char* name; int M, S, Cin, KW, N; } shapes[] = {
    {"enc qkv    M100  k1 256->768  ", 100, 100, 256, 1, 768},
    {"enc fc     M100  k1 256->256  ", 100, 100, 256, 1, 256},
    {"enc conv9  M100  k9 256->1024 ", 100, 100, 256, 9, 1024},
    {"enc w2     M100  k1 1024->256 ", 100, 100, 1024, 1, 256},
    {"enc vp k3  M100  k3 256->256  ", 100, 100, 256, 3, 256},
    {"dec qkv    M788  k1 256->768  ", 788, 788, 256, 1, 768},
    {"dec fc     M788  k1 256->256  ", 788, 788, 256, 1, 256},
    {"dec conv9  M788  k9 256->1024 ", 788, 788, 256, 9, 1024},
    {"dec w2     M788  k1 1024->256 ", 788, 788, 1024, 1, 256},
    {"dec vp k3  M788  k3 256->256  ", 788, 788, 256, 3, 256},
    {"postnet    M788  k5 512->512  ", 788, 788, 512, 5, 512},
    {"postnet L  M788  k5 512->80   ", 788, 788, 512, 5, 80},
    {"mel_linear M788  k1 256->80   ", 788, 788, 256, 1, 80},
    {"b8  qkv    M1024 k1 256->768  ", 1024, 128, 256, 1, 768},
    {"b8  fc     M1024 k1 256->256  ", 1024, 128, 256, 1, 256},
    {"b8  conv9  M1024 k9 256->1024 ", 1024, 128, 256, 9, 1024},
    {"b8  w2     M1024 k1 1024->256 ", 1024, 128, 1024, 1, 256},
    {"b8  vp k3  M1024 k3 256->256  ", 1024, 128, 256, 3, 256},
    {"b16 qkv    M2048 k1 256->768  ", 2048, 128, 256, 1, 768},
    {"b16 fc     M2048 k1 256->256  ", 2048, 128, 256, 1, 256},
    {"b16 conv9  M2048 k9 256->1024 ", 2048, 128, 256, 9, 1024},
    {"b16 w2     M2048 k1 1024->256 ", 2048, 128, 1024, 1, 256},
    {"b16 vp k3  M2048 k3 256->256  ", 2048, 128, 256, 3, 256},
    {"d512 qkv   M8192 k1 512->1536 ", 8192, 128, 512, 1, 1536},
    {"d512 fc    M8192 k1 512->512  ", 8192, 128, 512, 1, 512},
    {"d512 conv9 M8192 k9 512->1024 ", 8192, 128, 512, 9, 1024},
    {"d512 w2    M8192 k1 1024->512 ", 8192, 128, 1024, 1, 512},
    {"d512 vp k3 M8192 k3 512->256  ", 8192, 128, 512, 3, 256},
    {"postnet 1  M788  k5 80->512   ", 788, 788, 80, 5, 512},
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
    {
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      for (int i = 0; i < 3; ++i) CK(launch_conv_gemm(p, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(p, 0));
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
      printf("   launch_conv_gemm's choice                 %8.1f us  %6.1f TF/s\n", ms * 1e3, gf / ms);
    }
    if (s.Cin % 32) {
      run<64, 64, 16, 1, 2, 2>(p, gf);
      run<32, 32, 16, 4, 1, 1>(p, gf);
      run<32, 64, 16, 2, 1, 2>(p, gf);
      CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
      continue;
    }
    run<64, 64, 32, 1, 2, 2>(p, gf);
    run<64, 64, 32, 2, 2, 2>(p, gf);
    run<64, 64, 32, 4, 2, 2>(p, gf);
    run<64, 128, 32, 1, 2, 4>(p, gf);
    run<32, 32, 32, 4, 1, 1>(p, gf);
    run<32, 32, 32, 8, 1, 1>(p, gf);
    run<32, 64, 32, 2, 1, 2>(p, gf);
    run<32, 64, 32, 4, 1, 2>(p, gf);
    run<64, 32, 32, 4, 2, 1>(p, gf);
    run<32, 128, 32, 1, 1, 4>(p, gf);
    run<32, 128, 32, 2, 1, 4>(p, gf);
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
