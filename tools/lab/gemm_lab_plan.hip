// Kernel lab (round 4): inputs for the step-aware launch planner of gemm_conv.hip.
//   1. the staircase of every tile shape that sums a row's contraction in ONE order (no in-workgroup K split): time against
//      the number of workgroups, on the two long-K shapes (FFN k=9 256->1024, PostNet k=5 512->512) and the mid-size ones
//   2. a row REMAINDER behind full rounds of a tall tile: serialized launch, `hipExtAnyOrderLaunch` (no barrier bit on the
//      remainder's AQL packet: it may start while the main launch drains) and a forked second stream
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_plan.hip -o gemm_lab_plan
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"

using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int BM, int BN, int BK, int KS, int WGM, int WGN>
static hipError_t launch_any(const ConvGemm& p, hipStream_t st, unsigned flags) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  hipExtLaunchKernelGGL((k_conv_gemm<BM, BN, BK, KS, WGM, WGN, false, 0>), dim3(ntm * ntn), dim3(64 * WGM * WGN * KS), 0, st, nullptr, nullptr, flags, p, ntn);
  return hipGetLastError();
}

template <typename F>
static float time_us(F&& f, int iters = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / iters * 1e3f;
}

template <int BM, int BN, int BK, int KS, int WGM, int WGN>
static void stair(const ConvGemm& base, const char* name) {
  const int ntn = (base.N + BN - 1) / BN;
  printf("  %-12s %3dx%3d %dx%d waves:", name, BM, BN, WGM, WGN);
  const int wgs_list[] = {64, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 1280, 1536, 2048};
  for (int w : wgs_list) {
    const int rt = w / ntn;
    if (rt < 1) continue;
    ConvGemm p = base;
    p.M = rt * BM;
    if (p.M > base.M) break;
    p.S = p.M;
    const float us = time_us([&] { CK((launch_t<BM, BN, BK, KS, WGM, WGN>(p, 0))); });
    const double gf = 2.0 * p.M * p.Cin * p.KW * p.N / 1e9;
    printf("  %d:%.1f(%.0f)", rt * ntn, us, gf / us * 1e3);
  }
  printf("\n");
}

int main() {
  struct Shape { const char* name; int Cin, KW, N; } shapes[] = {
      {"k9 256->1024", 256, 9, 1024}, {"k5 512->512", 512, 5, 512}};
  const int MAXM = 36000;
  for (auto& s : shapes) {
    size_t nx = (size_t)MAXM * s.Cin, nw = (size_t)s.N * s.KW * s.Cin, ny = (size_t)MAXM * s.N;
    std::vector<float> hx(nx), hw(nw), hb(s.N);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX;
    float *dx, *dw, *db, *dy;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice));
    ConvGemm p; memset(&p, 0, sizeof(p)); p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy; p.ldy = s.N;
    p.M = MAXM; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = MAXM; p.act = ACT_RELU;
    printf("== %s: workgroups:us(TFLOP/s)\n", s.name);
    if (getenv("NS_LAB_STAIRS")) {
    if (s.N >= 512) stair<256, 256, 32, 1, 8, 2>(p, s.name);
    if (s.N >= 256) stair<128, 256, 32, 1, 4, 4>(p, s.name);
    if (s.N >= 256) stair<64, 256, 32, 1, 2, 4>(p, s.name);
    stair<64, 128, 32, 1, 2, 4>(p, s.name);
    stair<64, 64, 32, 1, 2, 2>(p, s.name);
    stair<32, 256, 32, 1, 1, 8>(p, s.name);
    stair<32, 128, 32, 1, 1, 4>(p, s.name);
    stair<32, 64, 32, 1, 1, 2>(p, s.name);
    stair<32, 128, 64, 1, 1, 4>(p, s.name);
    }

    if (s.KW > 1) {
      // main + remainder: B = 9 rows (9090) as 8192 rows of 64x256 tiles + 898 rows on a finer tile
      const int ntn256 = (s.N + 255) / 256;
      const int main_rows = (512 / ntn256) * 64 > 8192 ? 8192 : (512 / ntn256) * 64;
      for (int rem : {450, 898, 1800}) {
        ConvGemm pm = row_range(p, 0, main_rows), pr = row_range(p, main_rows, rem);
        ConvGemm pall = p; pall.M = main_rows + rem;
        const float t_main = time_us([&] { CK((launch_t<64, 256, 32, 1, 2, 4>(pm, 0))); });
        const float t_all = time_us([&] { CK(launch_conv_gemm(pall, 0)); });
        const float t_r64 = time_us([&] { CK((launch_t<64, 128, 32, 1, 2, 4>(pr, 0))); });
        const float t_r32 = time_us([&] { CK((launch_t<32, 128, 32, 1, 1, 4>(pr, 0))); });
        const float t_r32w = time_us([&] { CK((launch_t<32, 256, 32, 1, 1, 8>(pr, 0))); });
        const float t_r3264 = time_us([&] { CK((launch_t<32, 64, 32, 1, 1, 2>(pr, 0))); });
        const float s64 = time_us([&] { CK((launch_t<64, 256, 32, 1, 2, 4>(pm, 0))); CK((launch_t<64, 128, 32, 1, 2, 4>(pr, 0))); });
        const float s32 = time_us([&] { CK((launch_t<64, 256, 32, 1, 2, 4>(pm, 0))); CK((launch_t<32, 128, 32, 1, 1, 4>(pr, 0))); });
        const float a64 = time_us([&] { CK((launch_t<64, 256, 32, 1, 2, 4>(pm, 0))); CK((launch_any<64, 128, 32, 1, 2, 4>(pr, 0, hipExtAnyOrderLaunch))); });
        const float a32 = time_us([&] { CK((launch_t<64, 256, 32, 1, 2, 4>(pm, 0))); CK((launch_any<32, 128, 32, 1, 1, 4>(pr, 0, hipExtAnyOrderLaunch))); });
        // remainder FIRST (small tiles spread over the chip), main behind it without a barrier
        const float b32 = time_us([&] { CK((launch_t<32, 128, 32, 1, 1, 4>(pr, 0))); CK((launch_any<64, 256, 32, 1, 2, 4>(pm, 0, hipExtAnyOrderLaunch))); });
        // the remainder on a forked second stream, so that it co-resides with the main launch (fork: event on st0 -> wait on st1;
        // join: event on st1 -> wait on st0).  Main = the 128x256 tile (one workgroup per CU, 96 KB LDS: a 32 KB 64x64 workgroup fits beside it)
        static hipStream_t s1 = nullptr; static hipEvent_t ef, ej;
        if (!s1) { CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming)); }
        ConvGemm pm128 = row_range(p, 0, 8192 / ((s.N + 255) / 256) * ((s.N + 255) / 256) == 0 ? 8192 : (256 / ((s.N + 255) / 256)) * 128);
        ConvGemm pr128 = row_range(p, pm128.M, rem);
        const float m128 = time_us([&] { CK((launch_t<128, 256, 32, 1, 4, 4>(pm128, 0))); });
        const float ser128 = time_us([&] { CK((launch_t<128, 256, 32, 1, 4, 4>(pm128, 0))); CK((launch_t<64, 64, 32, 1, 2, 2>(pr128, 0))); });
        const float fj128 = time_us([&] {
          CK(hipEventRecord(ef, 0)); CK(hipStreamWaitEvent(s1, ef, 0));
          CK((launch_t<64, 64, 32, 1, 2, 2>(pr128, s1))); CK(hipEventRecord(ej, s1));
          CK((launch_t<128, 256, 32, 1, 4, 4>(pm128, 0))); CK(hipStreamWaitEvent(0, ej, 0)); });
        const float fj64 = time_us([&] {
          CK(hipEventRecord(ef, 0)); CK(hipStreamWaitEvent(s1, ef, 0));
          CK((launch_t<64, 64, 32, 1, 2, 2>(pr, s1))); CK(hipEventRecord(ej, s1));
          CK((launch_t<64, 256, 32, 1, 2, 4>(pm, 0))); CK(hipStreamWaitEvent(0, ej, 0)); });
        printf("  fork/join: main 128x256 (%d rows) alone %.1f, + 64x64 rem serial %.1f, forked %.1f | main 64x256 + 64x64 rem forked %.1f\n", pm128.M, m128, ser128, fj128, fj64);
        printf("  main %d rows (64x256) + rem %d: plan-now %.1f | main %.1f, rem alone 64x128 %.1f 32x128 %.1f 32x256 %.1f 32x64 %.1f | serial +64x128 %.1f +32x128 %.1f | any-order +64x128 %.1f +32x128 %.1f | rem-first any-order %.1f\n",
               main_rows, rem, t_all, t_main, t_r64, t_r32, t_r32w, t_r3264, s64, s32, a64, a32, b32);
      }
    }
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
