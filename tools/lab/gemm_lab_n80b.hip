// Kernel lab (round 4): the narrow-output GEMMs (PostNet k=5 512->80, mel_linear 256->80) at arbitrary row counts, and a
// check that hipExtLaunchKernel's start / stop events (kernels.h LaunchTiming) read the kernel's own duration.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_n80b.hip -o gemm_lab_n80b
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"

using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

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

int main() {
  struct Shape { const char* name; int Cin, KW, N; } shapes[] = {{"k5 512->80", 512, 5, 80}, {"k1 256->80", 256, 1, 80}};
  const int Ms[] = {2020, 4040, 5050, 7070, 9090, 12120, 16160, 20200, 24240, 32480};
  const int MAXM = 33000;
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
    for (int M : Ms) {
      ConvGemm p; memset(&p, 0, sizeof(p)); p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy; p.ldy = s.N;
      p.M = M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = M; p.act = ACT_NONE;
      printf("%s M=%5d: own %.1f | 64x96 %.1f | 32x96 KS4 %.1f KS2 %.1f KS1 %.1f | 64x128 %.1f | 32x128 KS2 %.1f KS1 %.1f\n", s.name, M,
             time_us([&] { CK(launch_conv_gemm(p, 0)); }),
             time_us([&] { CK((launch_t<64, 96, 32, 1, 2, 3>(p, 0))); }),
             time_us([&] { CK((launch_t<32, 96, 32, 4, 1, 3>(p, 0))); }),
             time_us([&] { CK((launch_t<32, 96, 32, 2, 1, 3>(p, 0))); }),
             time_us([&] { CK((launch_t<32, 96, 32, 1, 1, 3>(p, 0))); }),
             time_us([&] { CK((launch_t<64, 128, 32, 1, 2, 4>(p, 0))); }),
             time_us([&] { CK((launch_t<32, 128, 32, 2, 1, 4>(p, 0))); }),
             time_us([&] { CK((launch_t<32, 128, 32, 1, 1, 4>(p, 0))); }));
    }
    // timing check: the kernel's own events against a marker pair around it and against the back-to-back average
    {
      ConvGemm p; memset(&p, 0, sizeof(p)); p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy; p.ldy = s.N;
      p.M = 16160; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = p.M; p.act = ACT_NONE;
      hipEvent_t e0, e1, m0, m1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1));
      const float avg = time_us([&] { CK(launch_conv_gemm(p, 0)); });
      LaunchTiming tm{e0, e1};
      CK(launch_conv_gemm(p, 0)); CK(launch_conv_gemm(p, 0, &tm)); CK(launch_conv_gemm(p, 0));
      CK(hipDeviceSynchronize());
      float ext; CK(hipEventElapsedTime(&ext, e0, e1));
      CK(hipEventRecord(m0, 0)); CK(launch_conv_gemm(p, 0)); CK(hipEventRecord(m1, 0)); CK(hipDeviceSynchronize());
      float mk; CK(hipEventElapsedTime(&mk, m0, m1));
      // the cost of the markers themselves: 20 launches with a marker pair around each against 20 bare ones
      const float bare = time_us([&] { for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(p, 0)); }, 3);
      const float marked = time_us([&] { for (int i = 0; i < 20; ++i) { CK(hipEventRecord(m0, 0)); CK(launch_conv_gemm(p, 0)); CK(hipEventRecord(m1, 0)); } }, 3);
      const float exted = time_us([&] { for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(p, 0, &tm)); }, 3);
      printf("%s M=16160 timing: back-to-back avg %.1f us, dispatch events %.1f us, marker pair %.1f us; 20 launches bare %.1f, with marker pairs %.1f, with dispatch events %.1f us\n",
             s.name, avg, ext * 1e3f, mk * 1e3f, bare, marked, exted);
    }
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
