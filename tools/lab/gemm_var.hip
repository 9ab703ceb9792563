// Conv1D-as-GEMM / Linear on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Every dense contraction of the path except attention goes through this kernel:
//   Linear (KW=1): QKV / fc projections (transformer/SubLayers.py:18-25), mel_linear
//   Conv1d k=9 / k=1 of the FFT block's feed-forward (transformer/SubLayers.py:70-82)
//   Conv1d k=3 of the variance predictors (model/modules.py:245-276)
//   Conv1d k=5 of the PostNet with eval-BatchNorm folded in (transformer/Layers.py:107-167)
//
// The convolution is an implicit GEMM: activations stay [B*S, Cin] row-major in HBM and tap j of the
// kernel window is just the same matrix shifted by (j - pad) rows, zero outside the utterance's [0,S)
// window.  K runs tap-major (k = j*Cin + c), BK divides Cin, so one K-chunk touches one tap.
//
// Tiling (wave64, 4 waves = 2x2 per workgroup): block tile BMxBN, wave tile (BM/2)x(BN/2) as a grid of
// 32x32 MFMA tiles, K-chunk BK staged through LDS (double buffered, register-staged prefetch so the HBM/L2
// latency of chunk t+1 hides under the MFMAs of chunk t).  One K-chunk of a 128x128x32 tile is 64 MFMAs
// x 64 cycles per wave, far longer than a global load, so one barrier per chunk is enough.
//
// Operand reads use the freedom to permute k identically on both operands: lane-half h of MFMA step e in
// group g consumes k = 8g + 4h + e, so each lane reads its 4 steps' operands with ONE ds_read_b128.
// Row stride BK+4 floats makes those b128 reads bank-conflict free (MI355X_MICROARCH.md §LDS).
#include "kernels.h"

namespace ns {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int BK, int ABL = 0>
__global__ __launch_bounds__(256) void k_conv_gemm(ConvGemm p, int ntn) {
  __shared__ float dummy_lds[(ABL & 4) ? 12000 : 1];
  if (p.M < 0) dummy_lds[threadIdx.x] = 1.f;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LS = BK + 4;
  constexpr int TPR = BK / 4;        // float4 lanes per tile row
  constexpr int RPP = 256 / TPR;     // tile rows per pass
  constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(PA >= 1 && PB >= 1, "tile too small for 256 threads");

  __shared__ __attribute__((aligned(16))) float As[2][BM * LS];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LS];

  // XCD-aware bijective remap: consecutive tile ids (same M-tile, all N-tiles) land on one XCD / one L2
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int id2 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (id2 / ntn) * BM, n0 = (id2 % ntn) * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid >> 1) * WM, wn0 = (wid & 1) * WN;
  const int lrow = tid / TPR, lcol = (tid % TPR) * 4;

  const int Kt = p.KW * p.Cin;
  const int cpj = p.Cin / BK;        // chunks per tap
  const int nch = p.KW * cpj;

  // per-thread A rows: global row and position inside the utterance (fixed across chunks)
  int a_t[PA];
  bool a_ok[PA];
  const float* a_ptr[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + lrow + i * RPP;
    a_ok[i] = m < p.M;
    a_t[i] = a_ok[i] ? (m % p.S) : 0;
    a_ptr[i] = p.X + (size_t)(a_ok[i] ? m : 0) * p.ldx + lcol;
  }
  const float* b_ptr[PB];
  bool b_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = n0 + lrow + i * RPP;
    b_ok[i] = n < p.N;
    b_ptr[i] = p.W + (size_t)(b_ok[i] ? n : 0) * Kt + lcol;
  }

  f32x4 ra[PA], rb[PB];
  auto load_chunk = [&](int ch) {
    const int j = ch / cpj, c0 = (ch - j * cpj) * BK;
    const int sh = j - p.pad;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int ts = a_t[i] + sh;
      if (a_ok[i] && ts >= 0 && ts < p.S)
        ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + (ptrdiff_t)sh * p.ldx + c0);
      else
        ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      if (b_ok[i])
        rb[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + (size_t)ch * BK);
      else
        rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(&As[buf][(lrow + i * RPP) * LS + lcol]) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + i * RPP) * LS + lcol]) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
#define TS() ((ABL & 8) ? (long long)__builtin_amdgcn_s_memtime() : 0ll)

  const int frag_off = (lane & 31) * LS + (lane >> 5) * 4;
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    const long long t0 = TS();
    if (!(ABL & 1)) { if (ch + 1 < nch) load_chunk(ch + 1); }
    const long long t1 = TS();
    const float* as = &As[buf][wm0 * LS + frag_off];
    const float* bs = &Bs[buf][wn0 * LS + frag_off];
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(as + mi * 32 * LS + g * 8);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(bs + ni * 32 * LS + g * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], acc[mi][ni], 0, 0, 0);
    }
    const long long t2 = TS();
    if (ABL & 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    const long long t3 = TS();
    if (!(ABL & 1)) { if (ch + 1 < nch) store_chunk(buf ^ 1); }
    if (ABL & 8) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    const long long t4 = TS();
    if (!(ABL & 2)) __syncthreads();
    const long long t5 = TS();
    tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2; tacc[3] += t4 - t3; tacc[4] += t5 - t4;
  }

  if ((ABL & 8) && lane == 0 && p.lens) {
    long long* o = (long long*)p.lens + ((size_t)blockIdx.x * 4 + wid) * 8;
    for (int i = 0; i < 5; ++i) o[i] = tacc[i];
    o[5] = nch;
  }
  // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int ecol = lane & 31, erow = (lane >> 5) * 4;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int n = n0 + wn0 + ni * 32 + ecol;
    if (n >= p.N) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + erow;
        if (m >= p.M) continue;
        float v = acc[mi][ni][r] + bv;
        if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == ACT_TANH) v = tanhf(v);
        if (p.resid) v += p.resid[(size_t)m * p.ldr + n];
        p.Y[(size_t)m * p.ldy + n] = v;
      }
    }
  }
}

template <int BM, int BN, int BK, int ABL = 0>
static hipError_t launch_t(const ConvGemm& p, hipStream_t st) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((k_conv_gemm<BM, BN, BK, ABL>), dim3(ntm * ntn), dim3(256), 0, st, p, ntn);
  return hipGetLastError();
}

hipError_t launch_conv_gemm(const ConvGemm& p, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if (p.Cin % 16 != 0 || (p.ldx & 3) != 0) return hipErrorInvalidValue;
  const bool bk32 = (p.Cin % 32) == 0;
  // 128x128 tiles when they still give every CU (256) about two workgroups; 64x64 otherwise
  const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const bool big = tiles128 >= 384 && p.N >= 96;
  if (big) return bk32 ? launch_t<128, 128, 32>(p, st) : launch_t<128, 128, 16>(p, st);
  return bk32 ? launch_t<64, 64, 32>(p, st) : launch_t<64, 64, 16>(p, st);
}

}  // namespace ns
