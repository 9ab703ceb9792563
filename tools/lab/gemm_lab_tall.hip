// Kernel lab: which tile serves a ROW REMAINDER best — the rows left over when a packed (variable-length) batch's row count is
// cut into full rounds of the tall tiles (gemm_conv.hip split plan), or a packed row count that fills no round at all.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_rem.hip -o gemm_lab_rem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
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
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
  const int wgs = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  printf("   %3dx%3dx%2d KS=%d %dx%d waves  %5d wgs %8.1f us  %6.1f TF/s\n", BM, BN, BK, KS, WGM, WGN, wgs, ms * 1e3, gf / ms);
}

int main() {
  struct Shape { const char* name; int Cin, KW, N; } shapes[] = {{"k9 256->1024", 256, 9, 1024}, {"k5 512->512 ", 512, 5, 512}};
  const int Ms[] = {10240, 10350, 12288, 12800, 14000, 14336, 16160};
  for (auto& s : shapes)
    for (int M : Ms) {
      size_t nx = (size_t)M * s.Cin, nw = (size_t)s.N * s.KW * s.Cin, ny = (size_t)M * s.N;
      std::vector<float> hx(nx), hw(nw), hb(s.N);
      for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
      for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
      for (auto& v : hb) v = (float)rand() / RAND_MAX;
      float *dx, *dw, *db, *dy;
      CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy, ny * 4));
      CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice));
      ConvGemm p; memset(&p, 0, sizeof(p)); p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy; p.ldy = s.N;
      p.M = M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = M; p.act = ACT_RELU;  // (one utterance: S must divide M, a tap never reads past row M-1)
      const double gf = 2.0 * M * s.Cin * s.KW * s.N / 1e9;
      printf("%s M=%5d  %6.2f GFLOP  (at 0.90 of peak: %6.1f us)\n", s.name, M, gf, gf / (0.9 * 157.3) * 1e3);
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      for (int i = 0; i < 3; ++i) CK(launch_conv_gemm(p, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(p, 0));
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
      printf("   launch_conv_gemm's own plan              %8.1f us  %6.1f TF/s\n", ms * 1e3, gf / ms);
      run<160, 256, 32, 1, 5, 2>(p, gf);
      run<192, 256, 32, 1, 6, 2>(p, gf);
      run<224, 256, 32, 1, 7, 2>(p, gf);
      run<64, 128, 32, 1, 2, 4>(p, gf);
      run<64, 256, 32, 1, 2, 4>(p, gf);
      run<128, 256, 32, 1, 4, 4>(p, gf);
      run<256, 256, 32, 1, 8, 2>(p, gf);
      CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
    }
  return 0;
}
