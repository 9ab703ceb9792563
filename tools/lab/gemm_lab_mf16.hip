// Kernel lab (round 5): the 16-row tile family (k_conv_gemm<..., MF = 16>, v_mfma_f32_16x16x4_f32) against the 32-row tiles the
// launch plan uses today, at row counts BETWEEN the steps of 256 workgroups (B = 9, 11, 17 utterances of ~1010 frames, the
// ragged config-2 batch's 10 490 packed rows, and the phase-1 grids B x 128).  Every variant is first compared with the
// launch plan's own result on the same operands (max abs difference: fp32 summation order only), then timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_mf16.hip -o gemm_lab_mf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include "gemm_conv.hip"

using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef hipError_t (*LaunchFn)(const ConvGemm&, hipStream_t, const LaunchTiming*);
struct Variant { const char* name; LaunchFn fn; int bm, bn; bool rowepi; int wgs_per_cu; };

#define V(BM, BN, KS, WGM, WGN, EPI, MF, PERCU) Variant{#BM "x" #BN " ks" #KS " " #WGM "x" #WGN " mf" #MF, &launch_t<BM, BN, 32, KS, WGM, WGN, EPI, 0, MF>, BM, BN, EPI, PERCU}

static std::vector<Variant> variants() {
  return {
      // today's tiles (MF 32)
      V(256, 256, 1, 8, 2, false, 32, 1), V(128, 256, 1, 4, 4, false, 32, 1), V(64, 256, 1, 2, 4, false, 32, 2), V(64, 128, 1, 2, 4, false, 32, 3),
      V(64, 64, 1, 2, 2, false, 32, 4), V(32, 128, 1, 1, 4, false, 32, 4),
      V(32, 256, 1, 1, 8, true, 32, 2), V(64, 256, 1, 1, 8, true, 32, 1),
      // 16-row family, full-row LayerNorm tiles: 8 waves x 32 columns, 16 waves x 16 columns
      V(16, 256, 1, 1, 8, true, 16, 2), V(32, 256, 1, 1, 8, true, 16, 2), V(48, 256, 1, 1, 8, true, 16, 2), V(64, 256, 1, 1, 8, true, 16, 1),
      V(80, 256, 1, 1, 8, true, 16, 1), V(96, 256, 1, 1, 8, true, 16, 1), V(112, 256, 1, 1, 8, true, 16, 1), V(128, 256, 1, 1, 8, true, 16, 1),
      V(48, 256, 1, 1, 16, true, 16, 2), V(64, 256, 1, 1, 16, true, 16, 1), V(80, 256, 1, 1, 16, true, 16, 1), V(96, 256, 1, 1, 16, true, 16, 1),
      // 16-row family, plain tiles 256 columns wide: column-split (16 waves x 16 columns, TM = BM / 16 slabs per wave)
      V(48, 256, 1, 1, 16, false, 16, 2), V(64, 256, 1, 1, 16, false, 16, 1), V(80, 256, 1, 1, 16, false, 16, 1), V(96, 256, 1, 1, 16, false, 16, 1),
      V(112, 256, 1, 1, 16, false, 16, 1), V(128, 256, 1, 1, 16, false, 16, 1), V(144, 256, 1, 1, 16, false, 16, 1), V(160, 256, 1, 1, 16, false, 16, 1),
      V(176, 256, 1, 1, 16, false, 16, 1), V(192, 256, 1, 1, 16, false, 16, 1), V(208, 256, 1, 1, 16, false, 16, 1), V(224, 256, 1, 1, 16, false, 16, 1),
      V(240, 256, 1, 1, 16, false, 16, 1), V(256, 256, 1, 1, 16, false, 16, 1),
      // ... 8 waves x 32 columns (two workgroups per CU up to 64 rows)
      V(48, 256, 1, 1, 8, false, 16, 2), V(80, 256, 1, 1, 8, false, 16, 1), V(96, 256, 1, 1, 8, false, 16, 1), V(112, 256, 1, 1, 8, false, 16, 1),
      // ... waves stacked 2 x 8 and 3 x 4 (fewer A-fragment reads per wave)
      V(96, 256, 1, 2, 8, false, 16, 1), V(160, 256, 1, 2, 8, false, 16, 1), V(192, 256, 1, 2, 8, false, 16, 1), V(224, 256, 1, 2, 8, false, 16, 1),
      V(144, 256, 1, 3, 4, false, 16, 1), V(96, 256, 1, 3, 4, false, 16, 1), V(192, 256, 1, 3, 4, false, 16, 1),
      // 16-row family, 128 columns wide (2 x 4 / 1 x 8 waves), for the PostNet's N = 512 and QKV's N = 768
      V(48, 128, 1, 1, 8, false, 16, 3), V(80, 128, 1, 1, 8, false, 16, 2), V(96, 128, 1, 1, 8, false, 16, 2), V(112, 128, 1, 1, 8, false, 16, 2),
      V(144, 128, 1, 1, 8, false, 16, 1), V(160, 128, 1, 1, 8, false, 16, 1),
      // small grids (phase 1): in-workgroup K split, today's 32-row rungs and their 48-row forms
      V(32, 128, 2, 1, 4, false, 32, 1), V(48, 128, 2, 1, 4, false, 16, 1), V(32, 64, 4, 1, 2, false, 32, 1), V(48, 64, 4, 1, 2, false, 16, 1),
      V(16, 128, 2, 1, 4, false, 16, 2), V(16, 64, 4, 1, 2, false, 16, 2), V(48, 256, 2, 1, 8, false, 16, 1),
  };
}

struct Shape { const char* name; int Cin, KW, N; bool resid; int act; bool ln; int S; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10;
  Shape shapes[] = {
      {"w_1  k9 256->1024 relu", 256, 9, 1024, false, ACT_RELU, false, 1010},
      {"post k5 512->512 tanh ", 512, 5, 512, false, ACT_TANH, false, 1010},
      {"qkv  k1 256->768      ", 256, 1, 768, false, ACT_NONE, false, 1010},
      {"w_2  k1 1024->256 +LN ", 1024, 1, 256, true, ACT_NONE, true, 1010},
      {"fc   k1 256->256  +LN ", 256, 1, 256, true, ACT_NONE, true, 1010},
      {"pred k3 256->256  +LN ", 256, 3, 256, false, ACT_RELU, true, 1010},
      {"enc w_1 k9 256->1024  ", 256, 9, 1024, false, ACT_RELU, false, 128},
      {"enc w_2 k1 1024->256  ", 1024, 1, 256, true, ACT_NONE, false, 128},
  };
  const int Ms_dec[] = {8080, 9090, 10490, 11110, 17170, 16160}, Ms_enc[] = {1024, 1152, 1408, 2176};
  const int MAXM = 17170;
  std::vector<float> hx((size_t)MAXM * 1024), hw((size_t)1024 * 9 * 256 + 512 * 5 * 512), hb(1024), hr((size_t)MAXM * 1024), hg(512), hbt(512);
  srand(1);
  for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
  for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.03f;
  for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hr) v = (float)rand() / RAND_MAX * 2 - 1;
  for (auto& v : hg) v = 1.f + 0.1f * ((float)rand() / RAND_MAX - 0.5f);
  for (auto& v : hbt) v = 0.1f * ((float)rand() / RAND_MAX - 0.5f);
  std::vector<long long> hl(64);
  float *dx, *dw, *db, *dr, *dg, *dbt, *dy0, *dy1;
  long long* dl;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, 1024 * 4)); CK(hipMalloc(&dr, hr.size() * 4));
  CK(hipMalloc(&dg, 512 * 4)); CK(hipMalloc(&dbt, 512 * 4)); CK(hipMalloc(&dy0, (size_t)MAXM * 1024 * 4)); CK(hipMalloc(&dy1, (size_t)MAXM * 1024 * 4));
  CK(hipMalloc(&dl, 64 * 8));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), 1024 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, hg.data(), 512 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbt, hbt.data(), 512 * 4, hipMemcpyHostToDevice));
  hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  const auto vars = variants();
  std::vector<float> y0, y1;
  for (auto& s : shapes) {
    const bool enc = s.S == 128;
    for (int M : (enc ? std::vector<int>(Ms_enc, Ms_enc + 4) : std::vector<int>(Ms_dec, Ms_dec + 6))) {
      const int B = (M + s.S - 1) / s.S;
      for (int b = 0; b < B; ++b) hl[b] = s.S - (b * 37) % (s.S / 4 + 1);
      CK(hipMemcpy(dl, hl.data(), B * 8, hipMemcpyHostToDevice));
      ConvGemm p; memset(&p, 0, sizeof(p));
      p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.resid = s.resid ? dr : nullptr; p.ldr = s.N; p.Y = dy0; p.ldy = s.N;
      p.M = M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = s.act;
      p.epi = s.ln ? EPI_LN : EPI_NONE; p.e.ln_g = dg; p.e.ln_b = dbt; p.e.lens = dl;
      const double gf = 2.0 * M * s.Cin * s.KW * s.N / 1e9;
      auto time_fn = [&](auto&& f) {
        for (int i = 0; i < 2; ++i) f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(ea, 0));
        for (int i = 0; i < iters; ++i) f();
        CK(hipEventRecord(eb, 0)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        return ms / iters * 1e3f;
      };
      // reference: the launch plan's own choice
      const size_t ny = (size_t)M * s.N;
      CK(hipMemset(dy0, 0xff, ny * 4));
      const float t_plan = time_fn([&] { CK(launch_conv_gemm(p, 0)); });
      y0.resize(ny); CK(hipMemcpy(y0.data(), dy0, ny * 4, hipMemcpyDeviceToHost));
      int plan[6] = {0, 0, 0, 0, 0, 0};
      const bool planned = !s.ln && conv_gemm_plan(M, s.N, s.Cin, s.KW, plan);
      printf("%s M %5d (%.1f GFLOP)  plan %6.1f us %5.1f TF/s", s.name, M, gf, t_plan, gf / t_plan * 1e3);
      if (planned) printf("  [%dx%d on %d rows + %dx%d on %d]", plan[0], plan[1], plan[2], plan[3], plan[4], plan[5]);
      printf("\n");
      ConvGemm q = p; q.Y = dy1;
      for (const auto& v : vars) {
        if (v.rowepi != s.ln) continue;
        if (s.ln ? v.bn != s.N : (s.N % v.bn != 0)) continue;
        const long wgs = (long)((M + v.bm - 1) / v.bm) * (s.N / v.bn);
        if (wgs > 4096 || wgs < 100) continue;
        const bool ks = strstr(v.name, "ks1") == nullptr;
        if (ks != enc) continue;  // K-split rungs on the phase-1 grids only, and nothing else there
        CK(hipMemset(dy1, 0xff, ny * 4));
        const float t = time_fn([&] { CK(v.fn(q, 0, nullptr)); });
        y1.resize(ny); CK(hipMemcpy(y1.data(), dy1, ny * 4, hipMemcpyDeviceToHost));
        double maxd = 0; size_t nan = 0;
        for (size_t i = 0; i < ny; ++i) { const double d = fabs((double)y0[i] - y1[i]); if (d > maxd) maxd = d; nan += !(d == d); }
        printf("    %-26s wgs %5ld (%4.2f steps) %7.1f us %5.1f TF/s  %+6.1f %%  maxdiff %.1e%s\n", v.name, wgs, (double)wgs / (256.0 * v.wgs_per_cu), t,
               gf / t * 1e3, 100.0 * (t / t_plan - 1.0), maxd, nan ? "  NaN!" : (maxd > 2e-4 ? "  MISMATCH" : ""));
      }
      fflush(stdout);
    }
  }
  return 0;
}
