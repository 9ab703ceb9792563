// Kernel lab: stream-K GEMM (gemm_streamk.hip) vs launch_conv_gemm's choice on the single-utterance shapes: sampled outputs
// against fp64, run-to-run bit identity over many launches (fresh epochs, scratch poisoned between), interleaved timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_sk_lab.hip -o gemm_sk_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"
#include "gemm_streamk.hip"
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main() {
  struct Shape { const char* name; int M, S, Cin, KW, N, act; } shapes[] = {
    {"dec conv9  M788  k9 256->1024", 788, 788, 256, 9, 1024, ACT_RELU},
    {"postnet    M788  k5 512->512 ", 788, 788, 512, 5, 512, ACT_TANH},
    {"dec conv9  M1576 k9 256->1024 (B=2)", 1576, 788, 256, 9, 1024, ACT_RELU},
    {"dec conv9  M400  k9 256->1024", 400, 400, 256, 9, 1024, ACT_RELU},
    {"d512 conv9 M788  k9 512->1024", 788, 788, 512, 9, 1024, ACT_RELU},
  };
  srand(5);
  float* part; unsigned* flag;
  CK(hipMalloc(&part, conv_gemm_sk_scratch_bytes()));
  flag = (unsigned*)(part + (size_t)2 * SK_GRID * 32 * 128);
  for (auto& s : shapes) {
    const int Kt = s.KW * s.Cin;
    size_t nx = (size_t)s.M * s.Cin, nw = (size_t)s.N * Kt, ny = (size_t)s.M * s.N;
    std::vector<float> hx(nx), hw(nw), hb(s.N), hr(ny);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hr) v = (float)rand() / RAND_MAX - 0.5f;
    float *dx, *dw, *db, *dr, *dy0, *dy1;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dr, ny * 4)); CK(hipMalloc(&dy0, ny * 4)); CK(hipMalloc(&dy1, ny * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), ny * 4, hipMemcpyHostToDevice));
    ConvGemm p; memset(&p, 0, sizeof(p));
    p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.resid = dr; p.ldr = s.N; p.Y = dy0; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = s.act;
    ConvGemm q = p; q.Y = dy1;
    printf("%s  sk_ok=%d\n", s.name, (int)conv_gemm_sk_ok(s.M, s.N, s.Cin, s.KW));
    if (!conv_gemm_sk_ok(s.M, s.N, s.Cin, s.KW)) continue;
    unsigned epoch = 0;
    CK(hipMemsetAsync(flag, 0, 2 * SK_GRID * 4, 0));
    auto sk = [&]() { CK(launch_conv_gemm_sk(q, part, flag, ++epoch, 0)); };
    CK(launch_conv_gemm(p, 0)); sk(); CK(hipDeviceSynchronize());
    std::vector<float> y0(ny), y1(ny), y2(ny);
    CK(hipMemcpy(y0.data(), dy0, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dy1, ny * 4, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, d01 = 0;
    for (int t = 0; t < 3000; ++t) {
      const int m = rand() % s.M, n = rand() % s.N, ti = m % s.S;
      double acc = hb[n];
      for (int j = 0; j < s.KW; ++j) {
        const int ts = ti + j - p.pad; if (ts < 0 || ts >= s.S) continue;
        const float* xr = &hx[(size_t)(m + j - p.pad) * s.Cin]; const float* wr = &hw[(size_t)n * Kt + (size_t)j * s.Cin];
        for (int c = 0; c < s.Cin; ++c) acc += (double)xr[c] * wr[c];
      }
      if (s.act == ACT_RELU) acc = acc > 0 ? acc : 0; else if (s.act == ACT_TANH) acc = tanh(acc);
      acc += hr[(size_t)m * s.N + n];
      e0 = fmax(e0, fabs(acc - y0[(size_t)m * s.N + n])); e1 = fmax(e1, fabs(acc - y1[(size_t)m * s.N + n]));
    }
    for (size_t i = 0; i < ny; ++i) { double d = fabs((double)y0[i] - y1[i]); if (!(d <= d01)) d01 = d; }
    // determinism / race screen: 200 launches with scratch poisoned in between, every output must repeat bit for bit
    size_t bad = 0;
    for (int it = 0; it < 200; ++it) {
      if (it % 20 == 0) CK(hipMemsetAsync(part, 0xff, (size_t)2 * SK_GRID * 32 * 128 * 4, 0));
      CK(hipMemsetAsync(dy1, 0x7f, ny * 4, 0));
      sk();
      if (it % 10 == 9) { CK(hipMemcpy(y2.data(), dy1, ny * 4, hipMemcpyDeviceToHost)); bad += memcmp(y2.data(), y1.data(), ny * 4) != 0; }
    }
    double gf = 2.0 * s.M * Kt * s.N / 1e9;
    printf("   err vs fp64 (3000 samples): shipped %.2e  stream-K %.2e   max |shipped - stream-K| %.2e   repeat mismatches %zu/20\n   ", e0, e1, d01, bad);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < 3; ++r) {
      float m0, m1;
      CK(hipEventRecord(a, 0)); for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(p, 0)); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&m0, a, b)); m0 /= 20;
      CK(hipEventRecord(a, 0)); for (int i = 0; i < 20; ++i) sk(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&m1, a, b)); m1 /= 20;
      printf("  shipped %6.1f us | stream-K %6.1f us (%5.1f TF/s)", m0 * 1e3, m1 * 1e3, gf / m1);
    }
    printf("\n");
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dr)); CK(hipFree(dy0)); CK(hipFree(dy1));
  }
  return 0;
}
