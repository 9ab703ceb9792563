// Kernel lab: full-row GEMM with the LayerNorm (+mask) row epilogue vs the two-launch form (GEMM, then k_layernorm)
// on the path's N == d shapes.  Checks the fused output bit for bit against the two-launch output, then times both
// in interleaved rounds (same process, same buffers).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_ln.hip -o gemm_lab_ln
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "gemm_conv.hip"
#include "rowops.hip"

using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  struct Shape { const char* name; int M, S, Cin, KW, N; bool resid; int act; } shapes[] = {
    {"fc dec   (k1 256->256)  M16160", 16160, 1010, 256, 1, 256, true, ACT_NONE},
    {"w_2 dec  (k1 1024->256) M16160", 16160, 1010, 1024, 1, 256, true, ACT_NONE},
    {"pred c1  (k3 256->256)  M16160", 16160, 1010, 256, 3, 256, false, ACT_RELU},
    {"fc enc   (k1 256->256)  M2048 ", 2048, 128, 256, 1, 256, true, ACT_NONE},
    {"w_2 enc  (k1 1024->256) M2048 ", 2048, 128, 1024, 1, 256, true, ACT_NONE},
    {"pred c1  (k3 256->256)  M2048 ", 2048, 128, 256, 3, 256, false, ACT_RELU},
    {"fc B1    (k1 256->256)  M788  ", 788, 788, 256, 1, 256, true, ACT_NONE},
    {"w_2 B1   (k1 1024->256) M788  ", 788, 788, 1024, 1, 256, true, ACT_NONE},
    {"pred B1  (k3 256->256)  M788  ", 788, 788, 256, 3, 256, false, ACT_RELU},
    {"fc B1e   (k1 256->256)  M100  ", 100, 100, 256, 1, 256, true, ACT_NONE},
    {"w_2 B1e  (k1 1024->256) M100  ", 100, 100, 1024, 1, 256, true, ACT_NONE},
    {"fc d512  (k1 512->512)  M64640", 64640, 1010, 512, 1, 512, true, ACT_NONE},
    {"w_2 d512 (k1 1024->512) M64640", 64640, 1010, 1024, 1, 512, true, ACT_NONE},
    {"fc long  (k1 256->256)  M31200", 31200, 3900, 256, 1, 256, true, ACT_NONE},
    {"w_2 long (k1 1024->256) M31200", 31200, 3900, 1024, 1, 256, true, ACT_NONE},
  };
  for (auto& s : shapes) {
    const int B = s.M / s.S;
    size_t nx = (size_t)s.M * s.Cin, nw = (size_t)s.N * s.KW * s.Cin, ny = (size_t)s.M * s.N;
    std::vector<float> hx(nx), hw(nw), hb(s.N), hr(ny), hg(s.N), hbt(s.N);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX;
    for (auto& v : hr) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : hg) v = 1.f + 0.1f * ((float)rand() / RAND_MAX - 0.5f);
    for (auto& v : hbt) v = 0.1f * ((float)rand() / RAND_MAX - 0.5f);
    std::vector<long long> hl(B);
    for (int b = 0; b < B; ++b) hl[b] = s.S - (b * 37) % (s.S / 4 + 1);  // ragged lengths: some rows masked
    float *dx, *dw, *db, *dr, *dg, *dbt, *dt1, *dy0, *dy1;
    long long* dl;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dr, ny * 4));
    CK(hipMalloc(&dg, s.N * 4)); CK(hipMalloc(&dbt, s.N * 4)); CK(hipMalloc(&dt1, ny * 4)); CK(hipMalloc(&dy0, ny * 4));
    CK(hipMalloc(&dy1, ny * 4)); CK(hipMalloc(&dl, B * 8));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), ny * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), s.N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbt, hbt.data(), s.N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dl, hl.data(), B * 8, hipMemcpyHostToDevice));
    ConvGemm p; memset(&p, 0, sizeof(p));
    p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.resid = s.resid ? dr : nullptr; p.ldr = s.N; p.Y = dt1; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = s.act; p.epi = EPI_NONE;
    ConvGemm q = p; q.Y = dy1; q.epi = EPI_LN; q.e.ln_g = dg; q.e.ln_b = dbt; q.e.lens = dl;
    auto two = [&]() { CK(launch_conv_gemm(p, 0)); CK(launch_layernorm(dt1, dg, dbt, dy0, s.M, s.N, s.S, dl, 0)); };
    auto one = [&]() { CK(launch_conv_gemm(q, 0)); };
    CK(hipMemset(dy0, 0xff, ny * 4)); CK(hipMemset(dy1, 0xff, ny * 4));
    two(); one(); CK(hipDeviceSynchronize());
    std::vector<float> y0(ny), y1(ny);
    CK(hipMemcpy(y0.data(), dy0, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dy1, ny * 4, hipMemcpyDeviceToHost));
    size_t nbad = 0; double maxd = 0;
    for (size_t i = 0; i < ny; ++i) { nbad += memcmp(&y0[i], &y1[i], 4) != 0; double d = fabs((double)y0[i] - y1[i]); if (!(d <= maxd)) maxd = d; }
    double gf = 2.0 * s.M * s.Cin * s.KW * s.N / 1e9;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) { two(); one(); }
    CK(hipDeviceSynchronize());
    printf("%s %6.2f GFLOP bit-mismatches %zu maxdiff %.2e:", s.name, gf, nbad, maxd);
    for (int r = 0; r < rounds; ++r) {
      const int iters = 20;
      float ms2, ms1;
      CK(hipEventRecord(a, 0)); for (int i = 0; i < iters; ++i) two(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms2, a, b)); ms2 /= iters;
      CK(hipEventRecord(a, 0)); for (int i = 0; i < iters; ++i) one(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms1, a, b)); ms1 /= iters;
      printf("  two %6.1f us | fused %6.1f us (%5.1f TF/s)", ms2 * 1e3, ms1 * 1e3, gf / ms1);
    }
    printf("\n");
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dr)); CK(hipFree(dg)); CK(hipFree(dbt)); CK(hipFree(dt1));
    CK(hipFree(dy0)); CK(hipFree(dy1)); CK(hipFree(dl));
  }
  return 0;
}
