// Kernel lab: the opt-in bf16x3 GEMM (gemm_bf16x3.hip) vs the fp32-MFMA GEMM on the big shapes: sampled outputs of both
// against an fp64 host reference, then interleaved timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_b3_lab.hip -o gemm_b3_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"
#include "gemm_bf16x3.hip"
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  struct Shape { const char* name; int M, S, Cin, KW, N, act; } shapes[] = {
    {"ffn_w1 dec  (k9 256->1024) M16160", 16160, 1010, 256, 9, 1024, ACT_RELU},
    {"postnet mid (k5 512->512)  M16160", 16160, 1010, 512, 5, 512, ACT_TANH},
    {"ffn_w1 long (k9 256->1024) M31200", 31200, 3900, 256, 9, 1024, ACT_RELU},
    {"ffn_w1 d512 (k9 512->1024) M64640", 64640, 1010, 512, 9, 1024, ACT_RELU},
    {"ffn_w2 d512 (k1 1024->512) M64640", 64640, 1010, 1024, 1, 512, ACT_NONE},
    {"qkv d512    (k1 512->1536) M64640", 64640, 1010, 512, 1, 1536, ACT_NONE},
  };
  srand(3);
  for (auto& s : shapes) {
    const int Kt = s.KW * s.Cin;
    size_t nx = (size_t)s.M * s.Cin, nw = (size_t)s.N * Kt, ny = (size_t)s.M * s.N;
    std::vector<float> hx(nx), hw(nw), hb(s.N);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    std::vector<unsigned short> hp(3 * nw);
    split_weights_b3(hw.data(), nw, hp.data(), hp.data() + nw, hp.data() + 2 * nw);
    float *dx, *dw, *db, *dy0, *dy1; unsigned short* dp;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy0, ny * 4)); CK(hipMalloc(&dy1, ny * 4));
    CK(hipMalloc(&dp, 3 * nw * 2));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, hp.data(), 3 * nw * 2, hipMemcpyHostToDevice));
    ConvGemm p; memset(&p, 0, sizeof(p));
    p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy0; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = s.act;
    ConvGemm q = p; q.Wb3 = dp; q.Y = dy1;
    CK(hipMemset(dy0, 0xff, ny * 4)); CK(hipMemset(dy1, 0xff, ny * 4));
    CK(launch_conv_gemm(p, 0)); CK(launch_conv_gemm_b3(q, 0)); CK(hipDeviceSynchronize());
    std::vector<float> y0(ny), y1(ny);
    CK(hipMemcpy(y0.data(), dy0, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dy1, ny * 4, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, d01 = 0;
    for (int t = 0; t < 4000; ++t) {
      const int m = (t < 64) ? (t % 2 ? s.M - 1 - t : t) : rand() % s.M, n = (t < 64) ? (t * 37) % s.N : rand() % s.N;
      const int ti = m % s.S;
      double acc = hb[n];
      for (int j = 0; j < s.KW; ++j) {
        const int ts = ti + j - p.pad;
        if (ts < 0 || ts >= s.S) continue;
        const float* xr = &hx[(size_t)(m + j - p.pad) * s.Cin];
        const float* wr = &hw[(size_t)n * Kt + (size_t)j * s.Cin];
        for (int c = 0; c < s.Cin; ++c) acc += (double)xr[c] * wr[c];
      }
      if (s.act == ACT_RELU) acc = acc > 0 ? acc : 0; else if (s.act == ACT_TANH) acc = tanh(acc);
      e0 = fmax(e0, fabs(acc - y0[(size_t)m * s.N + n])); e1 = fmax(e1, fabs(acc - y1[(size_t)m * s.N + n]));
    }
    for (size_t i = 0; i < ny; ++i) { double d = fabs((double)y0[i] - y1[i]); if (!(d <= d01)) d01 = d; }
    double gf = 2.0 * s.M * Kt * s.N / 1e9;
    printf("%s %6.1f GFLOP  err vs fp64 (4000 samples): fp32-MFMA %.2e  bf16x3 %.2e   max |fp32-MFMA - bf16x3| over all %.2e\n   ", s.name, gf, e0, e1, d01);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < rounds; ++r) {
      const int iters = 10; float m0, m1;
      CK(hipEventRecord(a, 0)); for (int i = 0; i < iters; ++i) CK(launch_conv_gemm(p, 0)); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&m0, a, b)); m0 /= iters;
      CK(hipEventRecord(a, 0)); for (int i = 0; i < iters; ++i) CK(launch_conv_gemm_b3(q, 0)); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&m1, a, b)); m1 /= iters;
      printf("  fp32 %7.1f us (%5.1f TF/s) | bf16x3 %7.1f us (%5.1f fp32-equivalent TF/s, x%.2f)", m0 * 1e3, gf / m0, m1 * 1e3, gf / m1, m0 / m1);
    }
    printf("\n");
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy0)); CK(hipFree(dy1)); CK(hipFree(dp));
  }
  return 0;
}
