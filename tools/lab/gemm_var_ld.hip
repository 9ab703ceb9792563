#include "gemm_conv.hip"
namespace ns {
template <int BM, int BN, int BK>
__global__ __launch_bounds__(320) void k_conv_gemm_ld(ConvGemm p, int ntn) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-descriptor type does not exist in the host pass; it only needs the stub
  constexpr int KS = 1; constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int CPR = BK / 4;         // 16-B chunks per tile row
  constexpr int RPI = 64 / CPR;       // tile rows one wave-wide DMA instruction fills
  constexpr int IA = BM / (4 * RPI);  // DMA instructions per wave per chunk, A and B
  constexpr int IB = BN / (4 * RPI);
  constexpr int FSH = (BK == 64) ? 0 : (BK == 32) ? 1 : 2;
  constexpr int FMSK = CPR - 1;
  static_assert(BK == 64 || BK == 32 || BK == 16, "BK");
  static_assert(IA >= 1 && IB >= 1, "tile too small for 4 waves");
  static_assert(true, "split-K partial tiles must fit the A staging buffers");

  // Four DISTINCT LDS objects (not [2][...] arrays) and a 2x unrolled K loop with a static buffer index: hipcc tracks
  // in-flight LDS-DMA per LDS object, so a ds_read from As0 does not wait for a DMA that is filling As1.  With one
  // object per operand it inserted `s_waitcnt vmcnt(..)` in front of the first fragment read of every chunk, exposing
  // the whole L2/HBM latency of the prefetch it had just issued.  (The KS groups index INSIDE each object.)
  __shared__ __attribute__((aligned(16))) float As0[KS * BM * BK];
  __shared__ __attribute__((aligned(16))) float As1[KS * BM * BK];
  __shared__ __attribute__((aligned(16))) float Bs0[KS * BN * BK];
  __shared__ __attribute__((aligned(16))) float Bs1[KS * BN * BK];

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
  const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wall == 4;
  const int grp = 0, wid = wall & 3;
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
  constexpr int LA = IA * 4, LB = IB * 4;  // the loader wave covers the whole tile
  int a_row[LA], a_t[LA], a_col[LA];
  bool a_ok[LA];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int r = i * RPI + lr;
    const int m = m0 + r;
    a_row[i] = r;
    a_ok[i] = m < p.M;
    a_t[i] = a_ok[i] ? (m % p.S) : 0;
    a_col[i] = (ls ^ ((r >> FSH) & FMSK)) * 4;
  }
  int vb[LB];
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int r = i * RPI + lr;
    vb[i] = (n0 + r < p.N) ? (r * Kt + (ls ^ ((r >> FSH) & FMSK)) * 4) * 4 : OOR;
  }
  // stage chunk ch (tap j = ch / cpj, channel block cc = ch % cpj) of this group into (As, Bs)
  auto dma_chunk = [&](float* As, float* Bs, int ch) {
    const int j = ch / cpj, cc = ch - j * cpj;
    const int soA = cc * BK * 4, soB = ch * BK * 4;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int ts = a_t[i] + j - p.pad;
      const int va = (a_ok[i] && ts >= 0 && ts < p.S) ? ((a_row[i] + j) * p.ldx + a_col[i]) * 4 : OOR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)&As[i * RPI * BK], 16, va, soA, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)&Bs[i * RPI * BK], 16, vb[i], soB, 0, 0);
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

  // this group's slices of the four staging objects
  float* const A0 = As0 + grp * BM * BK;
  float* const A1 = As1 + grp * BM * BK;
  float* const B0 = Bs0 + grp * BN * BK;
  float* const B1 = Bs1 + grp * BN * BK;

  // group g owns chunks g, g + KS, g + 2 KS, ...; all groups run the same number of steps (barriers are block-wide)
  if (loader) {
    dma_chunk(As0, Bs0, 0);
    __syncthreads();
    for (int ch = 0; ch < nch; ch += 2) {
      if (ch + 1 < nch) dma_chunk(As1, Bs1, ch + 1);
      __syncthreads();
      if (ch + 1 < nch) {
        if (ch + 2 < nch) dma_chunk(As0, Bs0, ch + 2);
        __syncthreads();
      }
    }
    return;
  }
  __syncthreads();
  auto step = [&](const float* Ac, const float* Bc) {
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
    __syncthreads();
  };
  for (int ch = 0; ch < nch; ch += 2) {
    step(As0, Bs0);
    if (ch + 1 < nch) step(As1, Bs1);
  }

  if (KS > 1) {
    // sum the K-split partial tiles: groups 1..KS-1 park their accumulators in the (now idle) A staging buffers,
    // lane-linear, group 0 adds them in group order and runs the epilogue
    constexpr int TILE = BM * BN;
    constexpr int PER_OBJ = (KS * BM * BK) / TILE > 0 ? (KS * BM * BK) / TILE : 1;
    if (grp > 0) {
      float* red = ((grp - 1) / PER_OBJ ? As1 : As0) + ((grp - 1) % PER_OBJ) * TILE;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((wid * TM + mi) * TN + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g2 = 1; g2 < KS; ++g2) {
      const float* red = ((g2 - 1) / PER_OBJ ? As1 : As0) + ((g2 - 1) % PER_OBJ) * TILE;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] += red[(((wid * TM + mi) * TN + ni) * 16 + r) * 64 + lane];
    }
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
static hipError_t launch_ld(const ConvGemm& p, hipStream_t st) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((k_conv_gemm_ld<BM, BN, BK>), dim3(ntm * ntn), dim3(320), 0, st, p, ntn);
  return hipGetLastError();
}
}
