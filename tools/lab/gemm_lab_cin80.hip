// Kernel lab (round 5): the PostNet's first layer (Conv1d k=5, 80 -> 512: Cin % 32 != 0, so BK = 16) on tall tiles of the 16-row
// family (one step of tiles as tall as the rows ask for, 32 MFMAs per barrier interval at 128 rows) against the 64-row BK = 16 tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_cin80.hip -o gemm_lab_cin80
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "gemm_conv.hip"
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef hipError_t (*LaunchFn)(const ConvGemm&, hipStream_t, const LaunchTiming*);
struct Variant { const char* name; LaunchFn fn; int bm, bn; };
#define V16(BM) Variant{#BM "x256 bk16 1x16 mf16", &launch_t<BM, 256, 16, 1, 1, 16, false, 0, 16>, BM, 256}
int main() {
  std::vector<Variant> vars = {
      Variant{"64x128 bk16 2x4 mf32", &launch_t<64, 128, 16, 1, 2, 4>, 64, 128}, Variant{"64x256 bk16 2x4 mf32", &launch_t<64, 256, 16, 1, 2, 4>, 64, 256},
      Variant{"128x256 bk16 4x4 mf32", &launch_t<128, 256, 16, 1, 4, 4>, 128, 256},
      V16(48), V16(64), V16(80), V16(96), V16(112), V16(128), V16(144), V16(160), V16(192), V16(256)};
  const int Ms[] = {8080, 9090, 10490, 11110, 16160, 17170, 20200, 31248};
  const int MAXM = 31248, Cin = 80, KW = 5, N = 512;
  std::vector<float> hx((size_t)MAXM * Cin), hw((size_t)N * KW * Cin), hb(N);
  srand(4);
  for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
  for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
  for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
  float *dx, *dw, *db, *dy0, *dy1;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, N * 4));
  CK(hipMalloc(&dy0, (size_t)MAXM * N * 4)); CK(hipMalloc(&dy1, (size_t)MAXM * N * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  std::vector<float> y0, y1;
  for (int M : Ms) {
    ConvGemm p; memset(&p, 0, sizeof(p));
    p.X = dx; p.ldx = Cin; p.W = dw; p.bias = db; p.Y = dy0; p.ldy = N;
    p.M = M; p.N = N; p.Cin = Cin; p.KW = KW; p.pad = 2; p.S = M > 20000 && M % 1010 ? 3906 : 1010; p.act = ACT_TANH;
    const double gf = 2.0 * M * Cin * KW * N / 1e9;
    auto time_fn = [&](auto&& f) {
      for (int i = 0; i < 3; ++i) f();
      CK(hipDeviceSynchronize()); CK(hipEventRecord(ea, 0));
      for (int i = 0; i < 20; ++i) f();
      CK(hipEventRecord(eb, 0)); CK(hipEventSynchronize(eb));
      float ms; CK(hipEventElapsedTime(&ms, ea, eb));
      return ms / 20 * 1e3f;
    };
    const size_t ny = (size_t)M * N;
    time_fn([&] { CK(launch_conv_gemm(p, 0)); });
    const float t_plan = time_fn([&] { CK(launch_conv_gemm(p, 0)); });
    y0.resize(ny); CK(hipMemcpy(y0.data(), dy0, ny * 4, hipMemcpyDeviceToHost));
    printf("post0 k5 80->512 M %5d (%.1f GFLOP) current %6.1f us %5.1f TF/s\n", M, gf, t_plan, gf / t_plan * 1e3);
    ConvGemm q = p; q.Y = dy1;
    for (auto& v : vars) {
      const long wgs = (long)((M + v.bm - 1) / v.bm) * (N / v.bn);
      if (wgs < 100) continue;
      CK(hipMemset(dy1, 0xff, ny * 4));
      const float t = time_fn([&] { CK(v.fn(q, 0, nullptr)); });
      y1.resize(ny); CK(hipMemcpy(y1.data(), dy1, ny * 4, hipMemcpyDeviceToHost));
      double maxd = 0; for (size_t i = 0; i < ny; ++i) { const double d = fabs((double)y0[i] - y1[i]); if (!(d <= maxd)) maxd = d; }
      printf("    %-26s wgs %5ld %7.1f us %5.1f TF/s %+6.1f %%  maxdiff %.1e\n", v.name, wgs, t, gf / t * 1e3, 100.0 * (t / t_plan - 1.0), maxd);
    }
    fflush(stdout);
  }
  return 0;
}
