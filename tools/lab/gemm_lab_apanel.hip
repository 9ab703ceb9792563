// Kernel lab: resident activation panel (k_conv_gemm's APANEL parameter: the rows of all taps of a channel block staged once)
// against the tile-per-tap form, on the small-grid convolution shapes.  The K order differs (a K group owns whole channel
// blocks), so results are compared with a tolerance against the tile-per-tap result and against an fp64 host reference on a
// sample of outputs.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_apanel.hip -o gemm_lab_apanel
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static std::vector<float> g_ref, g_x, g_w, g_b;
static double host_ref(const ConvGemm& p, int m, int n) {
  const int b = m / p.S, t = m % p.S;
  double acc = g_b[n];
  for (int j = 0; j < p.KW; ++j) {
    const int tt = t + j - p.pad;
    if (tt < 0 || tt >= p.S) continue;
    for (int c = 0; c < p.Cin; ++c) acc += (double)g_x[((size_t)b * p.S + tt) * p.Cin + c] * g_w[((size_t)n * p.KW + j) * p.Cin + c];
  }
  return acc > 0 ? acc : 0;
}
template <int BM, int BN, int BK, int KS, int WGM, int WGN, bool AP>
void run(const ConvGemm& p, double gf, bool is_ref) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipMemset(p.Y, 0xff, (size_t)p.M * p.N * 4));
  for (int i = 0; i < 3; ++i) CK((launch_t<BM, BN, BK, KS, WGM, WGN, false, 0, AP>(p, 0)));
  CK(hipDeviceSynchronize());
  std::vector<float> h((size_t)p.M * p.N);
  CK(hipMemcpy(h.data(), p.Y, h.size() * 4, hipMemcpyDeviceToHost));
  double dmax = 0, rmax = 0;
  if (is_ref) g_ref = h; else for (size_t i = 0; i < h.size(); ++i) dmax = fmax(dmax, fabs((double)h[i] - g_ref[i]));
  for (int k = 0; k < 400; ++k) {
    const int m = (k * 7919) % p.M, n = (k * 104729) % p.N;
    rmax = fmax(rmax, fabs(h[(size_t)m * p.N + n] - host_ref(p, m, n)));
  }
  for (int k = 0; k < 64; ++k) {  // utterance edges
    const int m = (k % 2 ? (k / 2 % (p.M / p.S)) * p.S + (k % 5) : (k / 2 % (p.M / p.S) + 1) * p.S - 1 - (k % 5)), n = (k * 31) % p.N;
    rmax = fmax(rmax, fabs(h[(size_t)m * p.N + n] - host_ref(p, m, n)));
  }
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < 20; ++i) CK((launch_t<BM, BN, BK, KS, WGM, WGN, false, 0, AP>(p, 0)));
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
  printf("   %3dx%3dx%2d KS=%d %dx%d waves %s  %8.1f us  %6.1f TF/s  max|x - tile-per-tap| %.2e  max|x - fp64| %.2e\n", BM, BN, BK, KS, WGM, WGN,
         AP ? "PANEL       " : "tile per tap", ms * 1e3, gf / ms, dmax, rmax);
}

int main() {
  struct Shape { const char* name; int M, S, Cin, KW, N; int kind; } shapes[] = {
    {"dec conv9  B1 S788  k9 256->1024", 788, 788, 256, 9, 1024, 0},  {"enc conv9  B1 S100  k9 256->1024", 100, 100, 256, 9, 1024, 2},
    {"postnet    B1 S788  k5 512->512 ", 788, 788, 512, 5, 512, 1},   {"postnet L  B1 S788  k5 512->80  ", 788, 788, 512, 5, 80, 3},
    {"vp conv    B1 S788  k3 256->256 ", 788, 788, 256, 3, 256, 2},   {"enc conv9  B16 S128 k9 256->1024", 2048, 128, 256, 9, 1024, 0},
    {"ragged     B5 S37   k9 256->1024", 185, 37, 256, 9, 1024, 2},   {"ragged     B3 S70   k5 512->512 ", 210, 70, 512, 5, 512, 1},
  };
  for (auto& s : shapes) {
    const int K = s.KW * s.Cin;
    g_x.resize((size_t)s.M * s.Cin); g_w.resize((size_t)s.N * K); g_b.resize(s.N);
    for (auto& v : g_x) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : g_w) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    for (auto& v : g_b) v = (float)rand() / RAND_MAX - 0.5f;
    float *dx, *dw, *db, *dy;
    CK(hipMalloc(&dx, g_x.size() * 4)); CK(hipMalloc(&dw, g_w.size() * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy, (size_t)s.M * s.N * 4));
    CK(hipMemcpy(dx, g_x.data(), g_x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, g_w.data(), g_w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, g_b.data(), s.N * 4, hipMemcpyHostToDevice));
    ConvGemm p; memset(&p, 0, sizeof(p));
    p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = ACT_RELU;
    const double gf = 2.0 * s.M * K * s.N / 1e9;
    printf("%s  %.2f GFLOP\n", s.name, gf);
    if (s.kind == 0) { run<32, 128, 32, 2, 1, 4, false>(p, gf, true); run<32, 128, 32, 2, 1, 4, true>(p, gf, false); run<32, 64, 32, 4, 1, 2, true>(p, gf, false); }
    else if (s.kind == 1) { run<32, 64, 32, 4, 1, 2, false>(p, gf, true); run<32, 64, 32, 4, 1, 2, true>(p, gf, false); run<32, 128, 32, 2, 1, 4, true>(p, gf, false); run<32, 32, 32, 8, 1, 1, true>(p, gf, false); }
    else if (s.kind == 2) { run<32, 32, 32, 8, 1, 1, false>(p, gf, true); run<32, 32, 32, 8, 1, 1, true>(p, gf, false); run<32, 32, 32, 4, 1, 1, true>(p, gf, false); run<32, 64, 32, 4, 1, 2, true>(p, gf, false); }
    else { run<32, 96, 32, 4, 1, 3, false>(p, gf, true); run<32, 96, 32, 4, 1, 3, true>(p, gf, false); }
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
