// Kernel lab: where a K-loop step of the small-grid GEMM spends its cycles (DMA issue / fragment reads + MFMA / vmcnt wait /
// barrier), from cycle stamps written by wave 0 of two workgroups.  Needs step_timestamps.patch applied to csrc/gemm_conv.hip
// (git apply tools/lab/step_timestamps.patch; build; git checkout the file).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_steps.hip -o gemm_lab_steps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"
using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main() {
  struct Shape { const char* name; int M, S, Cin, KW, N; } shapes[] = {
    {"dec conv9 M788 ", 788, 788, 256, 9, 1024}, {"dec conv9 M394 ", 394, 394, 256, 9, 1024}, {"dec w2 M788    ", 788, 788, 1024, 1, 256},
    {"postnet M788   ", 788, 788, 512, 5, 512},
  };
  long long* dbg; CK(hipMalloc(&dbg, 8192 * 8));
  for (auto& s : shapes) {
    const int K = s.KW * s.Cin;
    std::vector<float> hx((size_t)s.M * s.Cin, 0.5f), hw((size_t)s.N * K, 0.01f);
    float *dx, *dw, *db, *dy;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy, (size_t)s.M * s.N * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(db, 0, s.N * 4));
    ConvGemm p; memset(&p, 0, sizeof(p));
    p.X = dx; p.ldx = s.Cin; p.W = dw; p.bias = db; p.Y = dy; p.ldy = s.N;
    p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = ACT_RELU;
    for (int i = 0; i < 3; ++i) CK(launch_conv_gemm(p, 0));
    CK(hipMemset(dbg, 0, 8192 * 8));
    p.e.x_out = (float*)dbg;
    CK(launch_conv_gemm(p, 0));
    CK(hipDeviceSynchronize());
    std::vector<long long> h(8192);
    CK(hipMemcpy(h.data(), dbg, 8192 * 8, hipMemcpyDeviceToHost));
    for (int blk = 0; blk < 2; ++blk) {
      long long* d = h.data() + blk * 4096;
      printf("%s block %d: prologue->first step %lld cyc\n", s.name, blk ? 77 : 0, d[1] - d[0]);
      int n = 0; while (n < 200 && d[1 + n * 4] != 0 && d[1 + n*4 + 3] != 0) ++n;
      long long tot_issue = 0, tot_mfma = 0, tot_wait = 0, tot_bar = 0;
      for (int st = 0; st < n; ++st) {
        long long t0 = d[1 + st * 4], t1 = d[2 + st * 4], t2 = d[3 + st * 4], t3 = d[4 + st * 4];
        long long tn = d[1 + (st + 1) * 4];
        tot_issue += t1 - t0; tot_mfma += t2 - t1; tot_wait += t3 - t2; if (st + 1 < n) tot_bar += tn - t3;
        if (st < 6 || st >= n - 2) printf("   step %2d: issue %5lld  frag+mfma %5lld  vmcnt wait %5lld  barrier->next %5lld\n", st, t1 - t0, t2 - t1, t3 - t2, st + 1 < n ? tn - t3 : -1);
      }
      printf("   %d steps; avg issue %.0f  frag+mfma %.0f  vmcnt wait %.0f  barrier %.0f ; loop total %lld cyc; epilogue start at %lld\n", n, (double)tot_issue / n, (double)tot_mfma / n,
             (double)tot_wait / n, (double)tot_bar / (n > 1 ? n - 1 : 1), d[1 + (n - 1) * 4 + 3] - d[1], d[1 + n * 4] ? d[1 + n * 4] - d[0] : -1);
    }
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
  }
  return 0;
}
