// Kernel lab: three staging buffers + hand-counted vmcnt (k_conv_gemm's STAGES parameter) against the two-buffer loop, on the
// small-grid and mid-size shapes.  Checks each variant bit-for-bit against the first one, then times it (20 launches back to back).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_stages.hip -o gemm_lab_stages
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static std::vector<float> g_ref;
template <int BM, int BN, int BK, int KS, int WGM, int WGN, int STG>
void run(const ConvGemm& p, double gf, bool is_ref) {
  const long wgs = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipMemset(p.Y, 0, (size_t)p.M * p.N * 4));
  for (int i = 0; i < 3; ++i) CK((launch_t<BM, BN, BK, KS, WGM, WGN, false, 0, STG>(p, 0)));
  CK(hipDeviceSynchronize());
  std::vector<float> h((size_t)p.M * p.N);
  CK(hipMemcpy(h.data(), p.Y, h.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  if (is_ref) g_ref = h; else for (size_t i = 0; i < h.size(); ++i) bad += memcmp(&h[i], &g_ref[i], 4) != 0;
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < 20; ++i) CK((launch_t<BM, BN, BK, KS, WGM, WGN, false, 0, STG>(p, 0)));
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
  printf("   %3dx%3dx%2d KS=%d %dx%d waves, %d stages  %5ld wgs %8.1f us  %6.1f TF/s  %s\n", BM, BN, BK, KS, WGM, WGN, STG, wgs, ms * 1e3, gf / ms,
         is_ref ? "(reference)" : bad ? "MISMATCH" : "bit-identical");
}

int main() {
  struct Shape { const char* name; int M, S, Cin, KW, N; int kind; } shapes[] = {
    {"dec conv9  M788  k9 256->1024 ", 788, 788, 256, 9, 1024, 0}, {"postnet    M788  k5 512->512  ", 788, 788, 512, 5, 512, 1},
    {"dec w2     M788  k1 1024->256 ", 788, 788, 1024, 1, 256, 2}, {"enc conv9  M100  k9 256->1024 ", 100, 100, 256, 9, 1024, 2},
    {"dec qkv    M788  k1 256->768  ", 788, 788, 256, 1, 768, 0},  {"b16 conv9  M2048 k9 256->1024 ", 2048, 128, 256, 9, 1024, 0},
    {"cfg2 qkv   M16160 k1 256->768 ", 16160, 1010, 256, 1, 768, 3}, {"cfg2 post0 M16160 k5 80->512  ", 16160, 1010, 80, 5, 512, 5},
    {"cfg2 w1    M16160 k9 256->1024", 16160, 1010, 256, 9, 1024, 4}, {"cfg2 post  M16160 k5 512->512 ", 16160, 1010, 512, 5, 512, 4},
    {"cfg5 qkv   M31248 k1 256->768 ", 31248, 3906, 256, 1, 768, 3},
  };
  for (auto& s : shapes) {
    const int K = s.KW * s.Cin;
    std::vector<float> hx((size_t)s.M * s.Cin), hw((size_t)s.N * K), hb(s.N);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX;
    float *dx, *dw, *db, *dy;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy, (size_t)s.M * s.N * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice));
    ConvGemm p; memset(&p, 0, sizeof(p));
    p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = ACT_RELU;
    const double gf = 2.0 * s.M * K * s.N / 1e9;
    printf("%s  %.2f GFLOP\n", s.name, gf);
    if (s.kind == 0) {
      run<32, 128, 32, 2, 1, 4, 2>(p, gf, true); run<32, 128, 32, 2, 1, 4, 3>(p, gf, false);
      run<32, 64, 32, 4, 1, 2, 2>(p, gf, false); run<32, 64, 32, 4, 1, 2, 3>(p, gf, false);
    } else if (s.kind == 1) {
      run<32, 64, 32, 4, 1, 2, 2>(p, gf, true); run<32, 64, 32, 4, 1, 2, 3>(p, gf, false);
      run<32, 128, 32, 2, 1, 4, 2>(p, gf, false); run<32, 128, 32, 2, 1, 4, 3>(p, gf, false);
    } else if (s.kind == 2) {
      run<32, 32, 32, 8, 1, 1, 2>(p, gf, true); run<32, 32, 32, 4, 1, 1, 2>(p, gf, false); run<32, 32, 32, 4, 1, 1, 3>(p, gf, false);
      run<32, 64, 32, 4, 1, 2, 3>(p, gf, false);
    } else if (s.kind == 3) {
      run<64, 128, 32, 1, 2, 4, 2>(p, gf, true); run<64, 128, 32, 1, 2, 4, 3>(p, gf, false);
      run<64, 256, 32, 1, 2, 4, 2>(p, gf, false); run<64, 256, 32, 1, 2, 4, 3>(p, gf, false);
    } else if (s.kind == 4) {
      run<64, 256, 32, 1, 2, 4, 2>(p, gf, true); run<64, 256, 32, 1, 2, 4, 3>(p, gf, false);
      run<128, 256, 32, 1, 4, 4, 2>(p, gf, false); run<128, 256, 32, 1, 4, 4, 3>(p, gf, false);
      run<256, 256, 32, 1, 8, 2, 2>(p, gf, false);
    } else {
      run<64, 256, 16, 1, 2, 4, 2>(p, gf, true);
    }
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
