// Kernel lab (round 5, second sweep): (1) the ragged config-2 batch's 10 490 packed rows — 128-column tiles of the 16-row family with
// BK = 16 (half the staging LDS: three or more workgroups per CU instead of two) against the BK = 32 forms the plan takes;
// (2) single-utterance decoder shapes (M = 788): squarer K-split tiles (64x64 KS4 on 16 waves, 48-row rungs) against the ladder's
// choice — operand bytes per CU are (BM + BN) * K * 4, a 64x64 tile pulls 20 % less than 32x128 for the same outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_r5b.hip -o gemm_lab_r5b
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
struct Variant { const char* name; LaunchFn fn; int bm, bn; int set; };
#define V(SET, BM, BN, BK, KS, WGM, WGN, MF) Variant{#BM "x" #BN " bk" #BK " ks" #KS " " #WGM "x" #WGN " mf" #MF, &launch_t<BM, BN, BK, KS, WGM, WGN, false, 0, MF>, BM, BN, SET}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10;
  std::vector<Variant> vars = {
      // set 0: packed rows (M = 10490)
      V(0, 112, 128, 32, 1, 1, 8, 16), V(0, 112, 128, 16, 1, 1, 8, 16), V(0, 80, 128, 32, 1, 1, 8, 16), V(0, 80, 128, 16, 1, 1, 8, 16),
      V(0, 176, 256, 32, 1, 1, 16, 16), V(0, 176, 256, 16, 1, 1, 16, 16), V(0, 96, 256, 32, 1, 1, 16, 16), V(0, 96, 128, 16, 1, 1, 8, 16),
      V(0, 96, 128, 32, 1, 1, 8, 16), V(0, 64, 128, 32, 1, 2, 4, 32), V(0, 64, 256, 32, 1, 1, 16, 16), V(0, 128, 256, 32, 1, 1, 16, 16),
      V(0, 160, 128, 16, 1, 1, 8, 16), V(0, 144, 128, 16, 1, 1, 8, 16),
      // set 1: single utterance (M = 788)
      V(1, 32, 128, 32, 2, 1, 4, 32), V(1, 32, 64, 32, 4, 1, 2, 32), V(1, 32, 32, 32, 8, 1, 1, 32), V(1, 64, 64, 32, 4, 2, 2, 32),
      V(1, 48, 64, 32, 4, 1, 2, 16), V(1, 48, 128, 32, 2, 1, 4, 16), V(1, 16, 128, 32, 2, 1, 4, 16), V(1, 16, 64, 32, 4, 1, 2, 16),
      V(1, 16, 256, 32, 2, 1, 8, 16), V(1, 64, 32, 32, 4, 2, 1, 32),
  };
  struct Shape { const char* name; int Cin, KW, N, act; int set, M, S; } shapes[] = {
      {"w_1  k9 256->1024 M 10490", 256, 9, 1024, ACT_RELU, 0, 10490, 1002}, {"post k5 512->512  M 10490", 512, 5, 512, ACT_TANH, 0, 10490, 1002},
      {"qkv  k1 256->768  M 10490", 256, 1, 768, ACT_NONE, 0, 10490, 1002},
      {"w_1  k9 256->1024 M 788", 256, 9, 1024, ACT_RELU, 1, 788, 788}, {"post k5 512->512  M 788", 512, 5, 512, ACT_TANH, 1, 788, 788},
      {"w_2  k1 1024->256 M 788", 1024, 1, 256, ACT_NONE, 1, 788, 788}, {"qkv  k1 256->768  M 788", 256, 1, 768, ACT_NONE, 1, 788, 788},
      {"w_1  k9 256->1024 M 100", 256, 9, 1024, ACT_RELU, 1, 100, 100},
  };
  const int MAXM = 10490;
  std::vector<float> hx((size_t)MAXM * 1024), hw((size_t)1024 * 9 * 256 + 512 * 5 * 512), hb(1024);
  srand(2);
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
  for (auto& s : shapes) {
    ConvGemm p; memset(&p, 0, sizeof(p));
    p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy0; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = s.act;
    const double gf = 2.0 * s.M * s.Cin * s.KW * s.N / 1e9;
    auto time_fn = [&](auto&& f) {
      for (int i = 0; i < 3; ++i) f();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(ea, 0));
      for (int i = 0; i < iters; ++i) f();
      CK(hipEventRecord(eb, 0)); CK(hipEventSynchronize(eb));
      float ms; CK(hipEventElapsedTime(&ms, ea, eb));
      return ms / iters * 1e3f;
    };
    const size_t ny = (size_t)s.M * s.N;
    time_fn([&] { CK(launch_conv_gemm(p, 0)); });
    const float t_plan = time_fn([&] { CK(launch_conv_gemm(p, 0)); });
    y0.resize(ny); CK(hipMemcpy(y0.data(), dy0, ny * 4, hipMemcpyDeviceToHost));
    printf("%s (%.1f GFLOP)  launch plan %6.1f us %5.1f TF/s\n", s.name, gf, t_plan, gf / t_plan * 1e3);
    ConvGemm q = p; q.Y = dy1;
    for (const auto& v : vars) {
      if (v.set != s.set || s.N % v.bn != 0) continue;
      const long wgs = (long)((s.M + v.bm - 1) / v.bm) * (s.N / v.bn);
      CK(hipMemset(dy1, 0xff, ny * 4));
      const float t = time_fn([&] { CK(v.fn(q, 0, nullptr)); });
      y1.resize(ny); CK(hipMemcpy(y1.data(), dy1, ny * 4, hipMemcpyDeviceToHost));
      double maxd = 0; size_t nan = 0;
      for (size_t i = 0; i < ny; ++i) { const double d = fabs((double)y0[i] - y1[i]); if (d > maxd) maxd = d; nan += !(d == d); }
      printf("    %-30s wgs %5ld %7.1f us %5.1f TF/s  %+6.1f %%  maxdiff %.1e%s\n", v.name, wgs, t, gf / t * 1e3, 100.0 * (t / t_plan - 1.0), maxd,
             nan ? "  NaN!" : (maxd > 2e-4 ? "  MISMATCH" : ""));
    }
    fflush(stdout);
  }
  return 0;
}
