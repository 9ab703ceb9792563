// Kernel lab: does the ROW STRIDE of the operands matter on the small-grid shapes?  Every row of a K chunk is one 128-byte
// line; with dense weights [N][K] consecutive rows are K*4 bytes apart — 9216 (k=9, 256 ch), 4096 (k=1, 1024 ch), 10240 (k=5,
// 512 ch): multiples of 2 KB or 4 KB, so if the L2 / fabric channel is picked from low address bits all rows of a chunk land on
// one or two channels.  Times launch_conv_gemm's choice with dense and padded (+32 floats = one line) strides for W (ldw) and X (ldx).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../smart-nar_fast_tts_amd/csrc gemm_lab_stride.hip -o gemm_lab_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_conv.hip"

using namespace ns;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static float time_it(const ConvGemm& p) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK(launch_conv_gemm(p, 0));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(p, 0));
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / 20 * 1e3f;
}

int main() {
  struct Shape { const char* name; int M, S, Cin, KW, N; } shapes[] = {
    {"enc qkv    M100  k1 256->768  ", 100, 100, 256, 1, 768},   {"enc fc     M100  k1 256->256  ", 100, 100, 256, 1, 256},
    {"enc conv9  M100  k9 256->1024 ", 100, 100, 256, 9, 1024},  {"enc w2     M100  k1 1024->256 ", 100, 100, 1024, 1, 256},
    {"dec qkv    M788  k1 256->768  ", 788, 788, 256, 1, 768},   {"dec fc     M788  k1 256->256  ", 788, 788, 256, 1, 256},
    {"dec conv9  M788  k9 256->1024 ", 788, 788, 256, 9, 1024},  {"dec w2     M788  k1 1024->256 ", 788, 788, 1024, 1, 256},
    {"dec vp k3  M788  k3 256->256  ", 788, 788, 256, 3, 256},   {"postnet    M788  k5 512->512  ", 788, 788, 512, 5, 512},
    {"postnet L  M788  k5 512->80   ", 788, 788, 512, 5, 80},    {"b16 conv9  M2048 k9 256->1024 ", 2048, 128, 256, 9, 1024},
    {"b16 w2     M2048 k1 1024->256 ", 2048, 128, 1024, 1, 256}, {"cfg2 conv9 M16160 k9 256->1024", 16160, 1010, 256, 9, 1024},
    {"cfg2 w2    M16160 k1 1024->256", 16160, 1010, 1024, 1, 256}, {"cfg2 qkv   M16160 k1 256->768 ", 16160, 1010, 256, 1, 768},
    {"cfg2 post  M16160 k5 512->512 ", 16160, 1010, 512, 5, 512},
  };
  for (auto& s : shapes) {
    const int K = s.KW * s.Cin;
    printf("%s", s.name);
    for (int padx : {0, 32}) for (int padw : {0, 32, 64, 96}) {
      const int ldx = s.Cin + padx, ldw = K + padw;
      std::vector<float> hx((size_t)s.M * ldx), hw((size_t)s.N * ldw), hb(s.N);
      for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
      for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
      float *dx, *dw, *db, *dy;
      CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, s.N * 4)); CK(hipMalloc(&dy, (size_t)s.M * s.N * 4));
      CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemset(db, 0, s.N * 4));
      ConvGemm p; memset(&p, 0, sizeof(p));
      p.X = dx; p.ldx = ldx; p.W = dw; p.ldw = ldw; p.bias = db; p.Y = dy; p.ldy = s.N;
      p.M = s.M; p.N = s.N; p.Cin = s.Cin; p.KW = s.KW; p.pad = (s.KW - 1) / 2; p.S = s.S; p.act = ACT_RELU;
      printf("  x+%d w+%d: %6.1f", padx, padw, time_it(p));
      CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy));
    }
    printf("  us\n");
  }
  return 0;
}
