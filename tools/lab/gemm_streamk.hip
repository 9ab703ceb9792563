// LAB ONLY (measured, not adopted — see README.md): stream-K variant of the Conv1D-as-GEMM for SMALL grids (single-utterance latency: M = 788 rows gives 200 tiles of
// 32 x 128 — fewer tiles than CUs, one workgroup per CU, each bound by its own CU's MFMA work and by barrier phases that
// nothing else on the CU covers).  Here the launch always has G = 2 x 256 workgroups, and the work is the flat list of
// (tile, K-step) units cut into G equal contiguous shares, so every CU holds two independent workgroups with the same
// amount of MFMA work whatever the tile count.  A workgroup's share touches at most two tiles; a share that does not
// cover its tile completely leaves an fp32 partial tile in scratch, and the workgroup that holds the tile's LAST K-steps
// (the owner) folds the earlier partials in — in K order, so the result does not depend on timing — and runs the epilogue.
//
// Hand-off (cdna_hip_programming.md Guideline 16, counter-free form): contributor = plain partial stores -> every wave
// `s_waitcnt vmcnt(0)` -> __syncthreads -> one lane agent-scope release -> asm `s_waitcnt vmcnt(0)` -> relaxed agent store of
// the launch's epoch into the partial's flag word; owner = one lane polls that word relaxed (bounded; a give-up traps
// instead of hanging) -> agent-scope acquire -> __syncthreads -> plain loads.  Owners only ever wait for workgroups with a
// LOWER id (earlier K-steps of the same tile), and every workgroup produces its own contribution before it waits, so the
// waits cannot form a cycle.  Flag words are zeroed once per forward phase (hipMemsetAsync) and every launch of the phase
// uses a fresh epoch.
//
// Same staging as gemm_conv.hip (LDS-DMA, source-side XOR swizzle, conv halo / zero padding by the buffer descriptor's
// out-of-range rule, in-workgroup K-split groups); BM = 32, BK = 32.
#include "kernels.h"

namespace ns {

constexpr int SK_GRID = 512;

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_s_t;
typedef __attribute__((address_space(1))) unsigned gu32_t;

[[maybe_unused]] constexpr int OORS = (int)0x80000000;

template <int BN, int KS, int WGN>
__global__ __launch_bounds__(64 * WGN * KS) void k_conv_gemm_sk(ConvGemm p, int ntm, int ntn, float* sk_part, unsigned* sk_flag,
                                                                  unsigned sk_epoch) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 32, BK = 32, NW = WGN;
  constexpr int WN = BN / WGN, TN = WN / 32;
  constexpr int TOTA = BM / 8, TOTB = BN / 8;
  static_assert(TN >= 1 && TOTB % NW == 0 && TOTA <= NW, "tile / wave-grid geometry");
  constexpr int IB = TOTB / NW;
  constexpr int TILE = BM * BN;
  constexpr int PER_OBJ = (KS * BN * BK) / TILE;  // K-split partial tiles per B staging object (= KS)
  static_assert(KS == 1 || (PER_OBJ >= 1 && (KS - 1) <= 2 * PER_OBJ), "K-split partial tiles must fit the B staging buffers");

  __shared__ __attribute__((aligned(16))) float As0[KS * BM * BK];
  __shared__ __attribute__((aligned(16))) float As1[KS * BM * BK];
  __shared__ __attribute__((aligned(16))) float Bs0[KS * BN * BK];
  __shared__ __attribute__((aligned(16))) float Bs1[KS * BN * BK];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wall / NW, wid = wall % NW;
  const int wn0 = wid * WN;
  const int Kt = p.KW * p.Cin, cpj = p.Cin / BK, nch = p.KW * cpj;
  const int nsteps = (nch + KS - 1) / KS;
  const int G = gridDim.x, bid = blockIdx.x;
  const long long U = (long long)ntm * ntn * nsteps;
  auto ubeg = [&](int c) { return (int)((long long)c * U / G); };
  const int u0 = ubeg(bid), u1 = ubeg(bid + 1);
  if (u1 <= u0) return;

  const int lr = lane >> 3, ls = lane & 7;
  const int frow = lane & 31, fh = lane >> 5;
  int foff[BK / 8];
#pragma unroll
  for (int g = 0; g < BK / 8; ++g) foff[g] = frow * BK + (((2 * g + fh) ^ ((frow >> 1) & 7)) * 4);
  float* const A0 = As0 + grp * BM * BK;
  float* const A1 = As1 + grp * BM * BK;
  float* const B0 = Bs0 + grp * BN * BK;
  float* const B1 = Bs1 + grp * BN * BK;
  gu32_t* const flags = (gu32_t*)sk_flag;

  // one share of one tile: K-steps [s0, s1) of tile t; `ord` = which of this workgroup's (at most two) shares it is
  auto segment = [&](int t, int s0, int s1, int ord) {
    const int tile_n = t / ntm, tile_m = t - tile_n * ntm;  // consecutive tiles share a weight slice
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.X + ((ptrdiff_t)m0 - p.pad) * p.ldx), (short)0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)n0 * Kt), (short)0, 0x7FFFFFFF, 0x00020000);
    // A: waves 0..TOTA-1 of a group stage 8 rows each; B: IB instructions per wave
    const int a_r = wid * 8 + lr;
    const int a_m = m0 + a_r;
    const int a_t = (wid < TOTA && a_m < p.M) ? (a_m % p.S) : -1;
    const int a_col = (ls ^ ((a_r >> 1) & 7)) * 4;
    int vb[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int r = (wid * IB + i) * 8 + lr;
      vb[i] = (n0 + r < p.N) ? (r * Kt + (ls ^ ((r >> 1) & 7)) * 4) * 4 : OORS;
    }
    auto dma_chunk = [&](float* As, float* Bs, int ch) {
      const int cc = ch / p.KW, j = ch - cc * p.KW;
      const int soA = cc * BK * 4, soB = (j * p.Cin + cc * BK) * 4;
      const int ts = a_t + j - p.pad;
      const int va = (a_t >= 0 && ts >= 0 && ts < p.S) ? ((a_r + j) * p.ldx + a_col) * 4 : OORS;
      if (wid < TOTA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_s_t)&As[wid * 8 * BK], 16, va, soA, 0, 0);
#pragma unroll
      for (int i = 0; i < IB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_s_t)&Bs[(wid * IB + i) * 8 * BK], 16, vb[i], soB, 0, 0);
    };
    f32x16s acc[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

    const int ch_end = s1 * KS < nch ? s1 * KS : nch;  // chunks of this share: [s0*KS, ch_end)
    if (s0 * KS + grp < ch_end) dma_chunk(A0, B0, s0 * KS + grp);
    __syncthreads();
    auto step = [&](int st, const float* Ac, const float* Bc, float* An, float* Bn) {
      const int ch = st * KS + grp;
      if (ch + KS < ch_end) dma_chunk(An, Bn, ch + KS);
      if (ch < ch_end) {
        const float* bs = Bc + wn0 * BK;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
          const f32x4s a = *reinterpret_cast<const f32x4s*>(Ac + foff[g]);
          f32x4s b[TN];
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const f32x4s*>(bs + ni * 32 * BK + foff[g]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[ni][e], acc[ni], 0, 0, 0);
        }
      }
      __syncthreads();
    };
    for (int st = s0; st < s1; st += 2) {
      step(st, A0, B0, A1, B1);
      if (st + 1 < s1) step(st + 1, A1, B1, A0, B0);
    }
    if (KS > 1) {  // in-workgroup K-split groups -> group 0, through the idle B staging buffers
      if (grp > 0) {
        float* red = ((grp - 1) / PER_OBJ ? Bs1 : Bs0) + ((grp - 1) % PER_OBJ) * TILE;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((wid * TN + ni) * 16 + r) * 64 + lane] = acc[ni][r];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int g2 = 1; g2 < KS; ++g2) {
          const float* red = ((g2 - 1) / PER_OBJ ? Bs1 : Bs0) + ((g2 - 1) % PER_OBJ) * TILE;
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] += red[((wid * TN + ni) * 16 + r) * 64 + lane];
        }
      }
    }
    const bool owner = s1 == nsteps;
#if defined(NS_LAB_SK_NOFIX)  // lab ablation: no hand-off at all (wrong results, pure compute time)
    if (!owner) { __syncthreads(); return; }
#endif
    if (!owner) {
      // contributor: publish the partial tile of K-steps [s0, s1)
      float* dst = sk_part + (size_t)(bid * 2 + ord) * TILE;
      if (grp == 0) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((wid * TN + ni) * 16 + r) * 64 + lane] = acc[ni][r];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flags + bid * 2 + ord, sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();  // LDS (staging / K-split scratch) is reused by the next share
      return;
    }
#if defined(NS_LAB_SK_NOFIX) || defined(NS_LAB_SK_NOWAIT)  // lab ablation: owners neither wait nor read
    if (false) {
#else
    if (s0 > 0) {
#endif
      // owner: fold in the earlier shares of this tile, lowest K-steps first.  They belong to the workgroups c < bid whose
      // range reaches into this tile.
      const int T0 = t * nsteps;
      int c_lo = bid;
      while (c_lo > 0 && ubeg(c_lo) > T0) --c_lo;  // first workgroup holding a unit of tile t
      // acc so far = the LAST K-steps; the sum must run in K order: total = p(c_lo) + ... + p(bid-1) + own
      f32x16s own[TN];
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) own[ni] = acc[ni];
      bool first = true;
      for (int c = c_lo; c < bid; ++c) {
        const int c_ord = (ubeg(c) / nsteps == t) ? 0 : 1;
        if (tid == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(flags + c * 2 + c_ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk_epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) __builtin_trap();  // give up loudly instead of hanging the queue
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (grp == 0) {
          const float* src = sk_part + (size_t)(c * 2 + c_ord) * TILE;
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = src[((wid * TN + ni) * 16 + r) * 64 + lane];
              acc[ni][r] = first ? v : acc[ni][r] + v;
            }
        }
        first = false;
      }
      if (grp == 0) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ni][r] += own[ni][r];
      }
    }
    if (grp == 0) {
      const int ecol = lane & 31, erow = (lane >> 5) * 4;
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int n = n0 + wn0 + ni * 32 + ecol;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (r & 3) + 8 * (r >> 2) + erow;
          if (m >= p.M) continue;
          float v = acc[ni][r] + bv;
          if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
          else if (p.act == ACT_TANH) v = tanhf(v);
          if (p.resid) v += p.resid[(size_t)m * p.ldr + n];
          p.Y[(size_t)m * p.ldy + n] = v;
        }
      }
    }
    __syncthreads();
  };

  // this workgroup's units [u0, u1): at most two tiles (the host guarantees U / G <= nsteps).  The share that STARTS a tile
  // (a contribution, unless it also ends it) goes first, so that owners downstream find it ready; the share that may
  // have to wait for others goes last.
  const int tA = u0 / nsteps, sA0 = u0 - tA * nsteps;
  const int endA = (tA + 1) * nsteps < u1 ? (tA + 1) * nsteps : u1;
  const int sA1 = endA - tA * nsteps;
  if (endA < u1) segment(tA + 1, 0, u1 - endA, 1);
  segment(tA, sA0, sA1, 0);
#endif
}

// scratch the stream-K launches of one forward phase need: 2*G partial tiles + 2*G flag words
size_t conv_gemm_sk_scratch_bytes() { return (size_t)2 * SK_GRID * 32 * 128 * sizeof(float) + 2 * SK_GRID * sizeof(unsigned); }

bool conv_gemm_sk_ok(int M, int N, int Cin, int KW) {
  if (Cin % 32 != 0) return false;
  const long tiles = (long)((M + 31) / 32) * ((N + 127) / 128);
  const int nsteps = (KW * (Cin / 32) + 1) / 2;
  // fewer tiles than CU slots, and a K loop long enough that a share is still several steps (the fix-up costs ~4 us)
  return N >= 128 && tiles >= 32 && tiles < 400 && nsteps >= 16 && tiles * nsteps / SK_GRID >= 6 && tiles <= SK_GRID;
}

hipError_t launch_conv_gemm_sk(const ConvGemm& p, float* sk_part, unsigned* sk_flag, unsigned sk_epoch, hipStream_t st) {
  if (!sk_part || !sk_flag || !conv_gemm_sk_ok(p.M, p.N, p.Cin, p.KW) || (p.ldx & 3) || p.epi != EPI_NONE) return hipErrorInvalidValue;
  if ((long long)(32 + p.KW) * p.ldx >= (1ll << 29) || (long long)128 * p.KW * p.Cin >= (1ll << 29)) return hipErrorInvalidValue;
  const int ntm = (p.M + 31) / 32, ntn = (p.N + 127) / 128;
  hipLaunchKernelGGL((k_conv_gemm_sk<128, 2, 4>), dim3(SK_GRID), dim3(512), 0, st, p, ntm, ntn, sk_part, sk_flag, sk_epoch);
  return hipGetLastError();
}

}  // namespace ns
