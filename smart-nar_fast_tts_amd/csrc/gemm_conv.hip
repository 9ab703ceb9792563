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
// 32x32 MFMA tiles, K-chunk BK double-buffered in LDS.
//
// Staging is LDS-DMA (`buffer_load_dwordx4 ... lds`): operands go HBM/L2 -> LDS without touching VGPRs, the
// zero padding of the convolution and the M/N tile tails come for free from the buffer descriptor's
// out-of-range rule (a lane whose voffset is the OOR marker writes zeros to LDS), and the per-chunk cost on the
// issuing wave is 8 DMA instructions with a scalar offset bump — no address VALU, no ds_write pass, nothing
// between a chunk's MFMAs but 16 ds_read_b128.  (Measured on the dominant shape: register-staged version
// 106 TFLOP/s, its compute-only ablation 137; see tools/lab.)
//
// The DMA destination is lane-linear (wave-uniform base + lane*16 B), so rows cannot be padded; bank conflicts
// of the fragment reads are removed by an XOR swizzle applied on the SOURCE side: 16-byte slot s of tile row r
// holds column chunk c = s ^ f(r), f(r) = (r>>1)&7 for 128-B rows (BK=32), (r>>2)&3 for 64-B rows (BK=16);
// the reads apply the same XOR.  With it every ds_read_b128 lane group touches 16 distinct 16-B bank slots.
//
// Operand reads use the freedom to permute k identically on both operands: lane-half h of MFMA step e in
// group g consumes k = 8g + 4h + e, so each lane reads its 4 steps' operands with ONE ds_read_b128.
#include "kernels.h"

namespace ns {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int OOR = (int)0x80000000;  // voffset marker: beyond num_records -> the DMA writes zeros

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void k_conv_gemm(ConvGemm p, int ntn) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-descriptor type does not exist in the host pass; it only needs the stub
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int CPR = BK / 4;         // 16-B chunks per tile row
  constexpr int RPI = 64 / CPR;       // tile rows one wave-wide DMA instruction fills
  constexpr int IA = BM / (4 * RPI);  // DMA instructions per wave per chunk, A and B
  constexpr int IB = BN / (4 * RPI);
  constexpr int FSH = (BK == 64) ? 0 : (BK == 32) ? 1 : 2;
  constexpr int FMSK = CPR - 1;
  static_assert(BK == 64 || BK == 32 || BK == 16, "BK");
  static_assert(IA >= 1 && IB >= 1, "tile too small for 4 waves");

  // Four DISTINCT LDS objects (not [2][...] arrays) and a 2x unrolled K loop with a static buffer index: hipcc tracks
  // in-flight LDS-DMA per LDS object, so a ds_read from As0 does not wait for a DMA that is filling As1.  With one
  // object per operand it inserted `s_waitcnt vmcnt(..)` in front of the first fragment read of every chunk, exposing
  // the whole L2/HBM latency of the prefetch it had just issued.
  __shared__ __attribute__((aligned(16))) float As0[BM * BK];
  __shared__ __attribute__((aligned(16))) float As1[BM * BK];
  __shared__ __attribute__((aligned(16))) float Bs0[BN * BK];
  __shared__ __attribute__((aligned(16))) float Bs1[BN * BK];

  // XCD-aware bijective remap.  Workgroup b runs on XCD b%8 (observed, used for speed only).  Tiles are ordered
  // "super-row by super-row": the M-tiles are split into 8 contiguous groups, and inside a group the order is
  // N-tile major / M-tile minor.  XCD x takes the x-th contiguous slice of that order, i.e. (up to a few tiles)
  // one group: its activation rows (a few MB) stay resident in that XCD's 4 MiB L2 while one weight slice
  // (BN x K floats) at a time streams through, instead of every workgroup re-fetching the whole weight matrix.
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  int pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntm = nblk / ntn, mq = ntm >> 3, mr = ntm & 7;
  int tile_m = 0, tile_n = 0, mstart = 0;
  for (int x = 0; x < 8; ++x) {
    const int gm = mq + (x < mr ? 1 : 0), gsz = gm * ntn;
    if (pos < gsz) {
      tile_n = pos / gm;
      tile_m = mstart + pos % gm;
      break;
    }
    pos -= gsz;
    mstart += gm;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wid >> 1) * WM, wn0 = (wid & 1) * WN;

  const int Kt = p.KW * p.Cin;
  const int cpj = p.Cin / BK;  // chunks per tap
  const int nch = p.KW * cpj;

  // block-relative descriptors: A rows are addressed from row (m0 - pad), B rows from row n0
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.X + ((ptrdiff_t)m0 - p.pad) * p.ldx), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)n0 * Kt), (short)0, 0x7FFFFFFF, 0x00020000);

  // per-lane DMA geometry: instruction i of this wave fills tile rows (wid*I + i)*RPI + lane/CPR, slot lane%CPR
  const int lr = lane / CPR, ls = lane % CPR;
  int a_row[IA], a_t[IA], a_col[IA];
  bool a_ok[IA];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int r = (wid * IA + i) * RPI + lr;
    const int m = m0 + r;
    a_row[i] = r;
    a_ok[i] = m < p.M;
    a_t[i] = a_ok[i] ? (m % p.S) : 0;
    a_col[i] = (ls ^ ((r >> FSH) & FMSK)) * 4;
  }
  int vb[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int r = (wid * IB + i) * RPI + lr;
    vb[i] = (n0 + r < p.N) ? (r * Kt + (ls ^ ((r >> FSH) & FMSK)) * 4) * 4 : OOR;
  }
  int va[IA];
  auto set_tap = [&](int j) {
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int ts = a_t[i] + j - p.pad;
      va[i] = (a_ok[i] && ts >= 0 && ts < p.S) ? ((a_row[i] + j) * p.ldx + a_col[i]) * 4 : OOR;
    }
  };
  auto dma_chunk = [&](float* As, float* Bs, int cc, int ch) {
    const int soA = cc * BK * 4, soB = ch * BK * 4;
#pragma unroll
    for (int i = 0; i < IA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)&As[(wid * IA + i) * RPI * BK], 16, va[i], soA, 0, 0);
#pragma unroll
    for (int i = 0; i < IB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)&Bs[(wid * IB + i) * RPI * BK], 16, vb[i], soB, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // fragment read offsets (floats): row (lane&31), slot ((2g + h) ^ f(row)); f is the same for every 32-row tile
  const int frow = lane & 31, fh = lane >> 5;
  int foff[BK / 8];
#pragma unroll
  for (int g = 0; g < BK / 8; ++g) foff[g] = frow * BK + (((2 * g + fh) ^ ((frow >> FSH) & FMSK)) * 4);

  int j = 0, cc = 0;
  set_tap(0);
  dma_chunk(As0, Bs0, 0, 0);
  __syncthreads();

  // one K-chunk: prefetch chunk ch+1 into the OTHER buffer pair, then 4*TM*TN*(BK/8) MFMAs on this one
  auto step = [&](int ch, const float* Ac, const float* Bc, float* An, float* Bn) {
    if (ch + 1 < nch) {
      if (++cc == cpj) {
        cc = 0;
        ++j;
        set_tap(j);
      }
      dma_chunk(An, Bn, cc, ch + 1);
    }
    const float* as = Ac + wm0 * BK;
    const float* bs = Bc + wn0 * BK;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(as + mi * 32 * BK + foff[g]);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(bs + ni * 32 * BK + foff[g]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();  // drains this chunk's DMA (vmcnt) and fences the buffer swap
  };
  for (int ch = 0; ch < nch; ch += 2) {
    step(ch, As0, Bs0, As1, Bs1);
    if (ch + 1 < nch) step(ch + 1, As1, Bs1, As0, Bs0);
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
#endif
}

template <int BM, int BN, int BK>
static hipError_t launch_t(const ConvGemm& p, hipStream_t st) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((k_conv_gemm<BM, BN, BK>), dim3(ntm * ntn), dim3(256), 0, st, p, ntn);
  return hipGetLastError();
}

hipError_t launch_conv_gemm(const ConvGemm& p, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if (p.Cin % 16 != 0 || (p.ldx & 3) != 0) return hipErrorInvalidValue;
  // descriptor offsets are 31-bit: a tile's rows (BM + KW) * ldx and BN * K floats must stay below 2^29 floats
  if ((long long)(128 + p.KW) * p.ldx >= (1ll << 29) || (long long)128 * p.KW * p.Cin >= (1ll << 29)) return hipErrorInvalidValue;
  const bool bk32 = (p.Cin % 32) == 0;
  // Tile choice (tools/lab/gemm_lab.hip sweep on the path's shapes, MI355X): the kernel is fastest with MANY small
  // independent workgroups per CU (their barrier/DMA phases interleave and keep the matrix pipe fed), so the
  // 64-row tile wins over 128x128 everywhere; 64x128 halves the B-operand traffic when N and the grid allow it.
  const long tiles_wide = (long)((p.M + 63) / 64) * ((p.N + 127) / 128);
  const bool wide = p.N >= 128 && tiles_wide >= 1024;
  if (wide) return bk32 ? launch_t<64, 128, 32>(p, st) : launch_t<64, 128, 16>(p, st);
  return bk32 ? launch_t<64, 64, 32>(p, st) : launch_t<64, 64, 16>(p, st);
}

}  // namespace ns
