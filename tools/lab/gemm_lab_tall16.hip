// Kernel lab (round 5): 16-row-family tiles TALLER than 256 rows (272 ... 320: 17 ... 20 slabs per wave, 68 ... 80 accumulator registers)
// for batches just above one full step of the 256 x 256 tile (B = 17 ... 20 x 1010 rows), against the plan's 256 x 256 + remainder.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_tall16.hip -o gemm_lab_tall16
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
struct Variant { const char* name; LaunchFn fn; int bm; };
#define V(BM) Variant{#BM "x256 1x16 mf16", &launch_t<BM, 256, 32, 1, 1, 16, false, 0, 16>, BM}
int main() {
  std::vector<Variant> vars = {V(144), V(160), V(256), V(272), V(288), V(304), V(320)};
  struct Shape { const char* name; int Cin, KW, N, act; } shapes[] = {{"w_1  k9 256->1024", 256, 9, 1024, ACT_RELU}, {"post k5 512->512 ", 512, 5, 512, ACT_TANH}};
  const int Ms[] = {16160, 17170, 18180, 19190, 20200};
  const int MAXM = 20200;
  std::vector<float> hx((size_t)MAXM * 512), hw((size_t)1024 * 9 * 256 + 512 * 5 * 512), hb(1024);
  srand(3);
  for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
  for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.03f;
  for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
  float *dx, *dw, *db, *dy0, *dy1;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, 1024 * 4));
  CK(hipMalloc(&dy0, (size_t)MAXM * 1024 * 4)); CK(hipMalloc(&dy1, (size_t)MAXM * 1024 * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), 1024 * 4, hipMemcpyHostToDevice));
  hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  std::vector<float> y0, y1;
  for (auto& s : shapes)
    for (int M : Ms) {
      ConvGemm p; memset(&p, 0, sizeof(p));
      p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy0; p.ldy = s.N;
      p.M = M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = 1010; p.act = s.act;
      const double gf = 2.0 * M * s.Cin * s.KW * s.N / 1e9;
      auto time_fn = [&](auto&& f) {
        for (int i = 0; i < 3; ++i) f();
        CK(hipDeviceSynchronize()); CK(hipEventRecord(ea, 0));
        for (int i = 0; i < 10; ++i) f();
        CK(hipEventRecord(eb, 0)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        return ms / 10 * 1e3f;
      };
      const size_t ny = (size_t)M * s.N;
      time_fn([&] { CK(launch_conv_gemm(p, 0)); });
      const float t_plan = time_fn([&] { CK(launch_conv_gemm(p, 0)); });
      y0.resize(ny); CK(hipMemcpy(y0.data(), dy0, ny * 4, hipMemcpyDeviceToHost));
      int pl[8]; conv_gemm_plan(M, s.N, s.Cin, s.KW, pl);
      printf("%s M %5d  plan %6.1f us %5.1f TF/s [%dx%d mf%d on %d + %dx%d on %d]\n", s.name, M, t_plan, gf / t_plan * 1e3, pl[0], pl[1], pl[6], pl[2], pl[3], pl[4], pl[5]);
      ConvGemm q = p; q.Y = dy1;
      for (auto& v : vars) {
        const long wgs = (long)((M + v.bm - 1) / v.bm) * (s.N / 256);
        if (wgs > 520) continue;
        CK(hipMemset(dy1, 0xff, ny * 4));
        const float t = time_fn([&] { CK(v.fn(q, 0, nullptr)); });
        y1.resize(ny); CK(hipMemcpy(y1.data(), dy1, ny * 4, hipMemcpyDeviceToHost));
        double maxd = 0; for (size_t i = 0; i < ny; ++i) { const double d = fabs((double)y0[i] - y1[i]); if (!(d <= maxd)) maxd = d; }
        printf("    %-20s wgs %4ld %7.1f us %5.1f TF/s %+6.1f %%  maxdiff %.1e\n", v.name, wgs, t, gf / t * 1e3, 100.0 * (t / t_plan - 1.0), maxd);
      }
      fflush(stdout);
    }
  return 0;
}
