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
// window.  K is indexed tap-major in memory (k = j*Cin + c), BK divides Cin, so one K-chunk touches one tap.
//
// Tiling (wave64): block tile BMxBN computed by WGM x WGN waves, wave tile (BM/WGM)x(BN/WGN) as a grid of 32x32 MFMA
// tiles (MF = 32) or 16x16 ones (MF = 16: block tiles of any multiple of 16 rows, see k_conv_gemm), K-chunk BK double-buffered in LDS.  KS > 1 adds an in-workgroup split of K: KS groups of waves each take
// every KS-th chunk into their own accumulators and the partial tiles are summed through LDS at the end.  It is
// used when the output has too few tiles to occupy the chip (the encoder's [B*L, *] GEMMs, single-utterance
// latency): the serial K loop of a tile, not the matrix pipe, bounds those launches.
//
// Staging is LDS-DMA (`buffer_load_dwordx4 ... lds`): operands go HBM/L2 -> LDS without touching VGPRs, the
// zero padding of the convolution and the M/N tile tails come for free from the buffer descriptor's
// out-of-range rule (a lane whose voffset is the OOR marker writes zeros to LDS), and the per-chunk cost on the
// issuing wave is a handful of DMA instructions with a scalar offset bump — no address VALU, no ds_write pass.
// (Measured on the dominant shape: register-staged version 106 TFLOP/s, its compute-only ablation 137; see tools/lab.)
//
// The DMA destination is lane-linear (wave-uniform base + lane*16 B), so rows cannot be padded; bank conflicts
// of the fragment reads are removed by an XOR swizzle applied on the SOURCE side: 16-byte slot s of tile row r
// holds column chunk c = s ^ f(r), f(r) = (r>>1)&7 for 128-B rows (BK=32), (r>>2)&3 for 64-B rows (BK=16);
// the reads apply the same XOR.  With it every ds_read_b128 lane group touches 16 distinct 16-B bank slots
// (SQ_LDS_BANK_CONFLICT = 0 in profiles/r01_pmc.md).
//
// Operand reads use the freedom to permute k identically on both operands: lane-half h of MFMA step e in
// group g consumes k = 8g + 4h + e (MF = 16: lane-quarter q, k = 16g + 4q + e), so each lane reads its 4 steps' operands with
// ONE ds_read_b128.
#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>
#include "rowln.h"

namespace ns {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

[[maybe_unused]] constexpr int OOR = (int)0x80000000;  // voffset marker: beyond num_records -> the DMA writes zeros

// TICKET (small-grid launches whose N columns are one whole activation row, tiled BM x BN with BN < N): the row epilogue
// of kernels.h RowEpilogue without a full-row tile and without a second launch.  Every workgroup stores its raw tile
// (act(contraction + bias) + resid) write-through into p.Y, then draws a ticket on its row block's counter; the workgroup
// that draws the last one (all ntn column tiles of the BM rows are then in memory) runs the row functions of rowln.h on
// those rows — the same code, on the same values, as the separate k_layernorm / k_ln_linear_embed launch would: bit-identical.
// Visibility across the 8 XCDs' private L2s / the CUs' L1s (cdna_hip_programming.md Guideline 16, counter form, with the
// fences traded for cache-bypassing accesses): the tile goes out as sc1 (write-through) stores, every storing wave drains
// vmcnt — the stores have then reached memory-side coherence — then a workgroup barrier and ONE relaxed agent-scope
// fetch_add; the last arriver reads the rows back with sc1 LOADS, which cannot hit a stale line of this CU's L1 or this
// XCD's L2, so it needs no acquire fence / cache invalidate (an agent-scope acquire here cost 900-1800 cycles per row
// block).  The ordering rests on: (1) vmcnt(0) = the write-through stores are complete at the device-coherent level,
// (2) the barrier orders every wave's drain before the ticket, (3) the RMW is performed at L2/memory in ticket order.
// It is exercised under load by tests/test_gpu_stress.py (thousands of ticketed launches on four streams against the
// two-launch form, bit for bit).  The counters are zeroed by the first kernel of the forward phase.
// MF: edge of the MFMA tile a wave's output is built from.  32 = v_mfma_f32_32x32x2_f32 (every tile of rounds 1-4); 16 =
// v_mfma_f32_16x16x4_f32, the 16-ROW family (round 5): the same matrix rate (64 flop / clk / SIMD, tools/lab/mfma_16x16.hip),
// the same LDS traffic per flop (a ds_read_b128 round feeds 16 rows x 16 k instead of 32 rows x 8 k), but block tiles whose
// height is any multiple of 16 — so a launch can give every CU ceil(rows / 16 / CUs) x 16 rows instead of a multiple of 32 / 64,
// which is what makes a forward's time follow B*T between the steps of 256 workgroups.  A launch uses ONE
// MF for all of its rows (the two instructions walk k in different orders: lane group q of step e holds k = 16g + 4q + e
// against 8g + 4h + e), so the rows of a launch — replicas of an utterance inside a batch — always carry the same bits.
template <int BM, int BN, int BK, int KS, int WGM = 2, int WGN = 2, bool ROWEPI = false, int TICKET = 0, int MF = 32>  // TICKET: row width / 256, 0 = off
__global__ __launch_bounds__(64 * WGM * WGN * KS) void k_conv_gemm(ConvGemm p, int ntn, int fl) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-descriptor type does not exist in the host pass; it only needs the stub
  constexpr int NW = WGM * WGN;       // waves per K-split group, arranged WGM x WGN over the block tile
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / MF, TN = WN / MF;
  constexpr int NR = MF == 32 ? 16 : 4;  // accumulator registers of one MFMA tile
  constexpr int LG = 64 / MF;            // lane groups of a fragment read: each holds 4 consecutive k of its row
  constexpr int KG = 4 * LG;             // k values one ds_read_b128 round feeds: 8 (MF 32) or 16 (MF 16)
  constexpr int CPR = BK / 4;         // 16-B chunks per tile row
  constexpr int RPI = 64 / CPR;       // tile rows one wave-wide DMA instruction fills
  static_assert(MF == 32 || MF == 16, "MFMA tile edge");
  static_assert(TM >= 1 && TN >= 1 && WM % MF == 0 && WN % MF == 0 && BM % RPI == 0 && BN % RPI == 0 && BK % KG == 0, "tile / wave-grid geometry");
  constexpr int FSH = (BK == 64) ? 0 : (BK == 32) ? 1 : 2;
  constexpr int FMSK = CPR - 1;
  static_assert(BK == 64 || BK == 32 || BK == 16, "BK");
  // row epilogue: the tile is parked 2 * BK rows at a time in the two B staging buffers (one pass up to 64 rows at BK = 32)
  constexpr int EPR = 2 * BK;                      // rows per parking pass
  constexpr int NPASS = (BM + EPR - 1) / EPR;
  static_assert(!ROWEPI || (KS == 1 && BN % 256 == 0 && (BM % EPR) % NW == 0 && (BM < EPR || EPR % NW == 0)), "row epilogue: every pass's rows divide over the waves");
  static_assert(!(ROWEPI && TICKET), "a full-row tile needs no ticket");
  using acc_t = typename std::conditional<MF == 32, f32x16, f32x4>::type;
  // CHUNKED ACCUMULATION (round 6).  One accumulator that takes all K / 2 (MF 32) MFMA steps of a long contraction in sequence
  // rounds every step against the whole running sum: the error of a row grows like sqrt(K) — measured against a float64
  // evaluation the k=9 convolution (K = 2304) came out 3.8x, at K = 4608 5.1x, as far from the truth as torch's blocked CPU
  // sums, while the K = 256 projections are level with them (profiles/r06_accuracy_vs_f64.md).  So a tile keeps TWO accumulator
  // sets: `acc` takes `fl` K steps (fl * BK k values) starting from zero, then is added into `tot` and cleared; the partial sums
  // the MFMAs round against stay short.  Error model (variance of a sequential sum ~ n^2, of c chunks of m steps ~ n (m + c)):
  // K = 2304, chunks of 128: 3.7x smaller (measured: 3.9x at 128, 4.7x at 64 — the default, conv_gemm_acc_chunk).  Cost: TM*TN*NR more registers — tiles of up to 32 accumulator registers per lane
  // have room (ACC2), the taller ones (144 ... 256 rows x 256) do not and are kept for K <= 256 only (plan_rows) — and NR * TM * TN
  // adds per chunk next to fl * 4 * (BK / KG) * TM * TN MFMAs.  fl == 0 (NS_ACC_CHUNK=0, A/B runs): one sequential sum, as before.
  constexpr bool ACC2 = TM * TN * NR <= 32;

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
  const int grp = wall / NW, wid = wall % NW;  // K-split group, wave inside the WGM x WGN arrangement
  const int wm0 = (wid / WGN) * WM, wn0 = (wid % WGN) * WN;

  const int Kt = p.ldw ? p.ldw : p.KW * p.Cin;  // weight row stride: dense, or padded (see ConvGemm::ldw)
  const int cpj = p.Cin / BK;  // chunks per tap
  const int nch = p.KW * cpj;

  // block-relative descriptors: A rows are addressed from row (m0 - pad), B rows from row n0
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.X + ((ptrdiff_t)m0 - p.pad) * p.ldx), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)n0 * Kt), (short)0, 0x7FFFFFFF, 0x00020000);

  // DMA work list of one chunk: TOTA = BM/RPI wave-wide instructions fill the A tile (RPI rows each), TOTB = BN/RPI the
  // B tile.  Round k of wave `wid` takes entry k*NW + wid, so the issue cost is spread over all waves of the group.
  // When NW divides the counts (every shipped configuration's hot path) there is no branch around any DMA: control
  // flow between the DMAs and the fragment reads makes hipcc's LDS-DMA scoreboard conservative again.
  constexpr int TOTA = BM / RPI, TOTB = BN / RPI;
  constexpr int IA = (TOTA + NW - 1) / NW, IB = (TOTB + NW - 1) / NW;
  constexpr bool A_EVEN = TOTA % NW == 0, B_EVEN = TOTB % NW == 0;
  const int lr = lane / CPR, ls = lane % CPR;
  // entry taken by round i of this wave: a contiguous run per wave when the list divides evenly, strided otherwise
  auto ea = [&](int i) { return A_EVEN ? wid * IA + i : i * NW + wid; };
  auto eb = [&](int i) { return B_EVEN ? wid * IB + i : i * NW + wid; };
  // Per A entry of this lane: byte offset of (tile row, swizzled column) WITHOUT the tap shift, and the range of taps
  // [a_jlo, a_jlo + a_jn) for which the shifted row is inside its utterance (empty for rows >= M).  The tap's row shift
  // j * ldx goes into the DMA's scalar offset, so a chunk costs three VALU ops per entry (sub, compare, select) instead of
  // the multiply-add chain — VALU issue next to the MFMAs is not free (tools/lab/mfma_mix.hip).
  int a_base[IA];
  unsigned a_jlo[IA], a_jn[IA];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int r = ea(i) * RPI + lr;
    const int m = m0 + r;
    int t = -1, sw = p.S;  // position of the row in its utterance and that utterance's window (packed rows: kernels.h RowMap)
    if (m < p.M) {
      const int mg = m + p.m_base;  // (row of the full problem when this launch covers a row range of it)
      if (p.rm.row_t) { t = p.rm.row_t[mg]; sw = p.rm.row_w[mg]; }
      else t = mg % p.S;
    }
    a_base[i] = (r * p.ldx + (ls ^ ((r >> FSH) & FMSK)) * 4) * 4;
    const int jlo = max(0, p.pad - t), jhi = min(p.KW, sw + p.pad - t);
    a_jlo[i] = (unsigned)jlo;
    a_jn[i] = (t >= 0 && jhi > jlo) ? (unsigned)(jhi - jlo) : 0u;
  }
  int vb[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int r = eb(i) * RPI + lr;
    vb[i] = (n0 + r < p.N) ? (r * Kt + (ls ^ ((r >> FSH) & FMSK)) * 4) * 4 : OOR;
  }
  // stage chunk ch of this group into (As, Bs).  The chunks run channel-block major, tap minor (cc = ch / KW, tap
  // j = ch % KW): the KW taps of one channel block re-read the same activation lines shifted by one row each, so they
  // are issued back to back and hit in L2.  (Tap-major order put Cin/BK chunks of the whole XCD's row group between two
  // uses of a line — a reuse distance of 2-5 MB against a 4 MB L2: 2-4x the fabric traffic, profiles/r01_pmc.md.)
  auto dma_chunk = [&](float* As, float* Bs, int ch) {
    const int cc = ch / p.KW, j = ch - cc * p.KW;
    const int soA = (cc * BK + j * p.ldx) * 4, soB = (j * p.Cin + cc * BK) * 4;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int va = ((unsigned)j - a_jlo[i] < a_jn[i]) ? a_base[i] : OOR;
      if (A_EVEN || ea(i) < TOTA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)&As[ea(i) * RPI * BK], 16, va, soA, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      if (B_EVEN || eb(i) < TOTB)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)&Bs[eb(i) * RPI * BK], 16, vb[i], soB, 0, 0);
    }
  };

  acc_t acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[mi][ni][r] = 0.f;
  [[maybe_unused]] acc_t tot[ACC2 ? TM : 1][ACC2 ? TN : 1];
  if constexpr (ACC2) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < NR; ++r) tot[mi][ni][r] = 0.f;
  }

  // this lane's bias values (KS == 1 epilogues): requested now, consumed after the K loop (in the epilogue the load's
  // latency was fully exposed)
  [[maybe_unused]] float bvp[TN];
  if constexpr (KS == 1) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int n = n0 + wn0 + ni * MF + (lane & (MF - 1));
      bvp[ni] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
  }

  // fragment read offsets (floats): row (lane & (MF-1)), slot ((LG g + h) ^ f(row)), h = lane / MF; f is the same for every MFMA tile
  // of the block tile (MF rows further down the swizzle repeats)
  const int frow = lane & (MF - 1), fh = lane / MF;
  int foff[BK / KG];
#pragma unroll
  for (int g = 0; g < BK / KG; ++g) foff[g] = frow * BK + (((LG * g + fh) ^ ((frow >> FSH) & FMSK)) * 4);

  // this group's slices of the four staging objects
  float* const A0 = As0 + grp * BM * BK;
  float* const A1 = As1 + grp * BM * BK;
  float* const B0 = Bs0 + grp * BN * BK;
  float* const B1 = Bs1 + grp * BN * BK;

  // group g owns chunks g, g + KS, g + 2 KS, ...; all groups run the same number of steps (barriers are block-wide)
  const int nsteps = (nch + KS - 1) / KS;
  if (grp < nch) dma_chunk(A0, B0, grp);
  __syncthreads();

  // one step: prefetch this group's next chunk into the OTHER buffer pair, then 4*TM*TN*(BK/8) MFMAs on this one
  // FRESH (a compile-time tag): the step opens an accumulation chunk — its first MFMA of every tile takes C = 0 (an inline
  // constant: no register clearing) instead of the previous chunk's sum, which the caller has just added into `tot`
  auto step = [&](auto fresh_c, int st, const float* Ac, const float* Bc, float* An, float* Bn) {
    constexpr bool FRESH = decltype(fresh_c)::value;
    const acc_t zero = {};
    const int ch = st * KS + grp;
    if (ch + KS < nch) dma_chunk(An, Bn, ch + KS);
    if (KS == 1 || ch < nch) {
      const float* as = Ac + wm0 * BK;
      const float* bs = Bc + wn0 * BK;
#pragma unroll
      for (int g = 0; g < BK / KG; ++g) {
        if constexpr (MF == 32) {
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
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], (FRESH && g == 0 && e == 0) ? zero : acc[mi][ni], 0, 0, 0);
        } else {
          // 16-row family: the B fragments of the wave's columns once, then one A slab at a time (a tall column-split tile has
          // up to 16 of them: all A fragments live at once would be 64 registers).  Every accumulator still sees its k values
          // in the order g-major, e-minor whatever the loop nest, so the nest is free to choose.
          f32x4 b[TN];
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(bs + ni * 16 * BK + foff[g]);
#pragma unroll
          for (int mi = 0; mi < TM; ++mi) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(as + mi * 16 * BK + foff[g]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[ni][e], (FRESH && g == 0 && e == 0) ? zero : acc[mi][ni], 0, 0, 0);
          }
        }
      }
    } else if constexpr (FRESH) {  // (a K group with no chunk left at this step: its accumulators were flushed, they restart from zero)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = zero;
    }
    __syncthreads();  // drains this step's DMA (vmcnt) and fences the buffer swap
  };
  // (fl is even or 0: a chunk ends behind the second step of a pair.  The last chunk is not flushed here: `acc` goes into the
  // epilogue as tot + acc — for a contraction of one chunk that is 0 + acc, the same bits as the sequential sum.)
  int due = fl;
  bool fresh = false;
  constexpr std::false_type CONT{};
  constexpr std::true_type OPEN{};
  for (int st = 0; st < nsteps; st += 2) {
    if (ACC2 && fresh) step(OPEN, st, A0, B0, A1, B1);
    else step(CONT, st, A0, B0, A1, B1);
    if (st + 1 < nsteps) step(CONT, st + 1, A1, B1, A0, B0);
    if constexpr (ACC2) {
      due -= 2;
      fresh = due == 0 && st + 2 < nsteps;
      if (fresh) {
        due = fl;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) tot[mi][ni] += acc[mi][ni];
      }
    }
  }
  if constexpr (ACC2) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = tot[mi][ni] + acc[mi][ni];
  }

  // C/D layout: 32x32 tile — col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), r < 16; 16x16 tile — col = lane & 15, row = r + 4 (lane >> 4), r < 4
  const int ecol = lane & (MF - 1);
  auto crow = [&](int r) { return MF == 32 ? (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) : r + 4 * (lane >> 4); };
  if constexpr (KS > 1) {
    // K-split epilogue, ALL waves of the workgroup: every group parks its partial tile ROW-MAJOR in the (now idle) staging
    // LDS — row stride BN + 8 floats, so the two lane halves of a store (rows 4 apart) land 32 banks apart — and after one
    // barrier each thread owns float4 units of the output tile: it sums the KS partials in group order (the same order as
    // before: bit-identical sums), adds bias / activation / residual and stores 16 bytes.  Bias and residual are float4 loads
    // issued BEFORE the parking stores and the barrier, so their latency is hidden.  (Before: group 0 alone ran the epilogue
    // element-wise — 16 scalar loads / stores per lane with the bias and residual latency exposed: ~3 us per launch, as long
    // as the K loop of the short GEMMs; ~6.5 us with write-through stores.  tools/lab/README.md.)
    constexpr int RS = BN + 8, TILE = BM * RS;
    constexpr int PB = (KS * BN * BK) / TILE, PA = (KS * BM * BK) / TILE;
    static_assert(2 * PB + 2 * PA >= KS, "the KS partial tiles must fit the four staging objects, whole tiles per object");
    auto part = [&](int g) -> float* {
      if (g < PB) return Bs0 + g * TILE;
      if (g < 2 * PB) return Bs1 + (g - PB) * TILE;
      if (g < 2 * PB + PA) return As0 + (g - 2 * PB) * TILE;
      return As1 + (g - 2 * PB - PA) * TILE;
    };
    constexpr int NT = 64 * NW * KS, U = BM * BN / 4, UPT = (U + NT - 1) / NT, CPRW = BN / 4;
    f32x4 bia[UPT], res[UPT];
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
      const int u = tid + k * NT, row = u / CPRW, m = m0 + row, n = n0 + (u % CPRW) * 4;
      const bool ok = u < U && m < p.M && n < p.N;
      bia[k] = (ok && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      res[k] = (ok && p.resid) ? *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    {
      float* mine = part(grp);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int r = 0; r < NR; ++r)
            mine[(wm0 + mi * MF + crow(r)) * RS + wn0 + ni * MF + ecol] = acc[mi][ni][r];
    }
    __syncthreads();
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsYs = __builtin_amdgcn_make_buffer_rsrc((void*)p.Y, (short)0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
      const int u = tid + k * NT, row = u / CPRW, c4 = u % CPRW, m = m0 + row, n = n0 + c4 * 4;
      if (u >= U || m >= p.M || n >= p.N) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(part(0) + row * RS + c4 * 4);
#pragma unroll
      for (int g2 = 1; g2 < KS; ++g2) v += *reinterpret_cast<const f32x4*>(part(g2) + row * RS + c4 * 4);
      v += bia[k];
      if (p.act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      } else if (p.act == ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
      }
      if (p.resid) v += res[k];
      if constexpr (TICKET != 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4g, v), rsYs, (m * p.ldy + n) * 4, 0, 16 /* sc1: write-through */);
      else *reinterpret_cast<f32x4*>(p.Y + (size_t)m * p.ldy + n) = v;
    }
  }

  if constexpr (ROWEPI) {
    // Full-row tile (BN == N): park act(acc + bias) as row-major [rows][BN] in the (idle) B staging buffers — 2 * BK rows per
    // pass, rows [0, BK) of a pass in Bs0, [BK, 2 BK) in Bs1 — then every wave takes whole rows — one row per wave64, float4
    // lanes — and runs the row kernel's own code on them (rowln.h): residual add, LayerNorm, mask / predictor tail, coalesced
    // 1-KB row stores.  Tiles of up to 64 rows are one pass; the taller ones of the 16-row family take ceil(BM / 64).
    static_assert(WGM == 1, "full-row tiles put their waves side by side");
    auto trow = [&](int ml) -> float* { return (ml < BK ? Bs0 : Bs1) + (ml % BK) * BN; };
    constexpr int NV = BN / 256;
    // what the row phase reads from memory besides the rows: the LayerNorm affine of this lane's columns (inside the row loop it
    // was a load -> wait per row)
    f32x4 lng[NV], lnb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      lng[i] = *reinterpret_cast<const f32x4*>(p.e.ln_g + lane * 4 + i * 256);
      lnb[i] = *reinterpret_cast<const f32x4*>(p.e.ln_b + lane * 4 + i * 256);
    }
    const RowEpilogue& e2 = p.e;  // (y_out == Y, ldy == BN: set / checked by the launcher — a modified copy of the struct kept all of its
                                  //  ~25 pointers live in SGPRs across the unrolled passes: 36-43 spilled SGPRs in the two-pass tiles, round 5)
    auto pass = [&](auto pc) {
      constexpr int P = decltype(pc)::value;
      constexpr int P0 = P * EPR, PR = (BM - P0 < EPR ? BM - P0 : EPR), RPW = PR / NW;
      if constexpr (P > 0) __syncthreads();  // (the previous pass's rows have been read)
      // the residual rows this wave will need: all loads issued now, so that their latency runs under the parking stores
      // and the barrier instead of once per row inside the row loop; the mask lengths of this wave's rows alongside
      const int mw = m0 + P0 + wid * RPW;
      f32x4 rv[RPW][NV];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int m = mw + rr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          rv[rr][i] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.resid && m < p.M) rv[rr][i] = *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.ldr + lane * 4 + i * 256);
        }
      }
      int tt[RPW];
      unsigned keep[RPW];
      // the rows of a wave are wave-uniform, so left to itself the compiler keeps every row's utterance index, position, 64-bit
      // length and mask bit in SGPRs — 4 rows x 6 scalars on top of the kernel's ~25 argument pointers: 36-43 spilled SGPRs in
      // the two-pass tiles (round 5).  Handing the first row over as a VGPR value makes the lookups vector loads and the row
      // addresses vector arithmetic; the mask is a keep word (rowln.h keep_or_zero): same values, no scalar pressure.
      int mv = mw;
      asm("" : "+v"(mv));
      row_batch_masks<RPW>(p.e, p.M, p.S, mv, 1, tt, keep);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int nl = wn0 + ni * MF + ecol;
        const float bv = bvp[ni];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
          if ((mi * MF) / EPR != P) continue;  // (compile-time per unrolled mi: MF divides EPR, WGM == 1)
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int ml = mi * MF + crow(r) - P0;
            float v = acc[mi][ni][r] + bv;
            if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
            else if (p.act == ACT_TANH) v = tanhf(v);
            trow(ml)[nl] = v;
          }
        }
      }
      __syncthreads();
      f32x4 v[RPW][NV];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          v[rr][i] = *reinterpret_cast<const f32x4*>(trow(wid * RPW + rr) + lane * 4 + i * 256);
          if (p.resid) v[rr][i] += rv[rr][i];
        }
      row_batch_finish<NV, RPW>(v, tt, keep, lane, p.epi, e2, p.M, mv, 1, lng, lnb);
    };
    pass(std::integral_constant<int, 0>{});
    if constexpr (NPASS > 1) pass(std::integral_constant<int, 1>{});
    if constexpr (NPASS > 2) pass(std::integral_constant<int, 2>{});
    if constexpr (NPASS > 3) pass(std::integral_constant<int, 3>{});
    static_assert(NPASS <= 4, "row epilogue: at most 4 parking passes (BM <= 8 BK)");
  } else {
    if constexpr (KS == 1) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int n = n0 + wn0 + ni * MF + ecol;
      if (n >= p.N) continue;
      const float bv = bvp[ni];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        // the tile's residual values first, all loads in flight together: interleaved with the stores (Y may alias
        // resid for all the compiler knows) every element paid a full load + store round trip, 16 in a row
        float rs[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int m = m0 + wm0 + mi * MF + crow(r);
          rs[r] = (p.resid && m < p.M) ? p.resid[(size_t)m * p.ldr + n] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int m = m0 + wm0 + mi * MF + crow(r);
          if (m >= p.M) continue;
          float v = acc[mi][ni][r] + bv;
          if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
          else if (p.act == ACT_TANH) v = tanhf(v);
          if (p.resid) v += rs[r];
          if constexpr (TICKET) __hip_atomic_store(p.Y + (size_t)m * p.ldy + n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1
          else p.Y[(size_t)m * p.ldy + n] = v;
        }
      }
    }
    }
    if constexpr (TICKET) {
      constexpr int NWALL = NW * KS;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have left
      __syncthreads();
      int* const last = reinterpret_cast<int*>(As0);      // the staging LDS is idle; no extra LDS object (see above)
      if (tid == 0) *last = __hip_atomic_fetch_add(p.e.ticket + tile_m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ntn - 1;
      __syncthreads();
      if (!*last) return;
      // (no acquire fence: the rows are read with sc1 loads, which do not hit stale lines of this CU's L1 / this XCD's L2)
      const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)p.Y, (short)0, p.M * p.ldy * 4, 0x00020000);
      // this wave's rows: wall, wall + NWALL, ... (BM / NWALL of them), at most 8 at a time
      constexpr int RPW = BM / NWALL, RB = RPW < 8 ? RPW : 8;
      static_assert(BM % NWALL == 0 && RPW % RB == 0, "rows divide evenly over the waves");
#pragma unroll
      for (int j0 = 0; j0 < RPW; j0 += RB)
        row_epilogue_batch<TICKET, RB>(rsY, p.ldy, lane, p.epi, p.e, p.M, p.S, m0 + wall + j0 * NWALL, NWALL);
    }
  }
#endif
}

// k values per accumulation chunk (see ACC2 in the kernel): 64 by default; NS_ACC_CHUNK=<multiple of 64> or 0 (off) for A/B runs.
// Measured (profiles/r06_accuracy_vs_f64*.md, profiles/r06_acc_chunk_ab.txt): distance from a float64 evaluation relative to the fp32
// reference's own, worst quantity (pitch / energy, config 2 and 4) 3.75x with one sequential sum, 1.71x with chunks of 128, 1.04x
// with chunks of 64; a forward costs +1.0 ... 1.8 % with chunks of 64 (of which 0.3 % over chunks of 128, the rest is the 256 x 256
// tile giving way to 128 x 256).
int conv_gemm_acc_chunk() {
  static const int v = [] {
    const char* e = getenv("NS_ACC_CHUNK");
    int k = e ? atoi(e) : 64;
    if (k < 0 || k % 64 != 0) k = 64;
    return k;
  }();
  return v;
}
static int acc_chunk_k() { return conv_gemm_acc_chunk(); }
// contraction length above which a launch must run on a tile with the second accumulator set: 512.  Up to there a tall tile's one
// sequential sum is at most 256 MFMA steps — K = 256 measured level with torch's CPU sums, K = 512 (the d = 512 model's QKV / fc) is
// sqrt(2) of that on two of a block's six contractions — and the tall tiles are worth 0.8 % of a config-4 forward (57.91 -> 57.45 ms,
// same box; float64 ratios of profiles/r06_accuracy_vs_f64.md unchanged).  Wherever such a launch lands on a tile with room it chunks anyway.
// NS_LONG_K (>= 256) for A/B runs.
static int long_k_threshold() {
  static const int v = [] { const char* e = getenv("NS_LONG_K"); const int k = e ? atoi(e) : 512; return k >= 256 ? k : 256; }();
  return v;
}
// a tile with room for the second accumulator set: <= 32 accumulator registers per lane
static constexpr bool tile_chunks(int bm, int bn, int waves) { return bm * bn / (64 * waves) <= 32; }

template <int BM, int BN, int BK, int KS = 1, int WGM = 2, int WGN = 2, bool ROWEPI = false, int TICKET = 0, int MF = 32>
static hipError_t launch_t(const ConvGemm& p, hipStream_t st, const LaunchTiming* tm = nullptr) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int fl = ((acc_chunk_k() / BK) + 1) & ~1;  // K steps per accumulation chunk, even (0: off)
  if (tm && (tm->start || tm->stop))  // the events ride on this kernel's own dispatch packet (kernels.h LaunchTiming)
    hipExtLaunchKernelGGL((k_conv_gemm<BM, BN, BK, KS, WGM, WGN, ROWEPI, TICKET, MF>), dim3(ntm * ntn), dim3(64 * WGM * WGN * KS), 0, st, tm->start, tm->stop, 0, p, ntn, fl);
  else
    hipLaunchKernelGGL((k_conv_gemm<BM, BN, BK, KS, WGM, WGN, ROWEPI, TICKET, MF>), dim3(ntm * ntn), dim3(64 * WGM * WGN * KS), 0, st, p, ntn, fl);
  return hipGetLastError();
}

bool conv_gemm_ticket_ok(int M, int N, int Cin) {
  // ... and few enough tiles for the ladder's K-split rungs (<= 512 workgroups of 32 x 128): beyond that the ladder's last rung was a
  // plain 64 x 64 tile whose ticketed instantiations spilled 85 / 106 SGPRs (round-4 review) for a range only a 512-wide model
  // with 4096 < M < 6400 rows ever reached — such launches take the two-launch form (GEMM, then the row kernel) instead
  return M > 0 && (N == 256 || N == 512) && Cin % 32 == 0 && (long long)M * N * 4 < (1ll << 31) && (long)((M + 31) / 32) * ((N + 127) / 128) <= 512;
}

bool conv_gemm_row_epilogue_ok(int M, int N, int Cin) {
  return M > 0 && (N == 256 || N == 512) && Cin % 32 == 0;
}

bool launch_planner_enabled() {
  static const bool on = [] { const char* e = getenv("NS_PLAN"); return !(e && e[0] == '0'); }();
  return on;
}

// rows [begin, begin + count) of p as a launch of its own (plain epilogue: the row-indexed operands are X, Y, resid)
static ConvGemm row_range(const ConvGemm& p, int begin, int count) {
  ConvGemm q = p;
  q.X += (size_t)begin * p.ldx;
  q.Y += (size_t)begin * p.ldy;
  if (q.resid) q.resid += (size_t)begin * p.ldr;
  q.M = count;
  q.m_base = p.m_base + begin;
  return q;
}

// ---------------------------------------------------------------------------------------------------------------------
// Step-aware launch plan.  A launch's time is a staircase in its workgroup count — one step per 256 workgroups, one per CU,
// whatever the tile (tools/lab/gemm_lab_plan.hip: flat inside a step, a full step more at 256 k + 1) — so ONE tile shape per
// launch makes the forward's time a staircase in the batch size: every launch class crosses a multiple of 256 tiles at the
// same row counts (round 3: B = 8 -> 9 cost +1.0 ms).  The planner prices every candidate with the measured model
//     t(tile, M) = a(tile, chunks) + b(tile, chunks) * ceil(workgroups / 256)      chunks = KW * Cin / 32
// and takes the cheapest of: one launch of any tile; FULL steps of a tall tile followed by the remaining rows on a finer one
// (rows are independent; a convolution's taps reach across the cut through the operand pointers, ConvGemm::m_base keeps the
// utterance positions).  Every candidate tile sums a row's contraction in the same order — no in-workgroup K split — so the
// rows behind a cut carry the same bits as the rows ahead of it, and replicas of one utterance inside a batch stay
// bit-identical wherever the cut falls.  (`hipExtAnyOrderLaunch` on the remainder — no barrier bit, so that it could start
// while the main launch drains — was measured and does nothing on gfx950; the two launches are plain stream neighbours.)
// a, b in us, fitted on one MI355X over chunks = 8 ... 80 (QKV, predictor k=3, FFN w_2, FFN w_1 k=9, PostNet k=5):
struct TileModel { int bm, bn; float a0, a1, b0, b1; };
enum TileId { T256, T128, T64W, T64, T64N, T32, N_TILES };
static constexpr TileModel kTile[N_TILES] = {
    {256, 256, 11.0f, 0.00f, 10.0f, 7.25f},  // 16 waves 8x2, one workgroup per CU: 0.91 of peak in full steps
    {128, 256, 5.0f, 0.20f, 8.0f, 3.60f},    // 16 waves 4x4
    {64, 256, 6.0f, 0.17f, 3.6f, 1.83f},     // 8 waves 2x4, two workgroups per CU
    {64, 128, 5.5f, 0.09f, 1.6f, 0.93f},     // 8 waves 2x4
    {64, 64, 5.0f, 0.20f, 1.2f, 0.46f},      // 4 waves 2x2
    {32, 128, 5.0f, 0.23f, 1.0f, 0.45f},     // 4 waves 1x4
};
static long tile_wgs(int t, long M, int N) { return ((M + kTile[t].bm - 1) / kTile[t].bm) * ((N + kTile[t].bn - 1) / kTile[t].bn); }
static float tile_time(int t, long M, int N, int chunks) {
  const long steps = (tile_wgs(t, M, N) + 255) / 256;
  const float a = kTile[t].a0 + kTile[t].a1 * chunks, b = kTile[t].b0 + kTile[t].b1 * chunks;
  // the 4-wave tiles re-read the weight matrix once per 32 / 64 rows: alone in a partial step they run at the rate above,
  // in launches of many steps the fabric slows every step after the first by ~15 % (forward traces: k=9 2280 workgroups
  // 365 us, k=5 512->512 2148 workgroups 383 us, QKV 3222 workgroups 74 us; the lab loop on a quiet chip showed 5 %)
  const float more = (t == T64N || t == T32) ? 1.15f : 1.0f;
  // between near-ties the taller tile (fewer passes over the weights, settled by forward A/B runs in round 3) keeps the launch
  return (a + b + b * more * (float)(steps - 1)) * (1.0f + 0.01f * (float)t);
}
// ---- the 16-row family (MF = 16, round 5) ------------------------------------------------------------------------------------------
// A launch whose time is a staircase in units of 256 workgroups gives every CU ceil(row tiles / 256) x BM rows: with 32 / 64 / 128 /
// 256-row tiles a batch between two steps pays for rows it does not have (B = 9 x 1010 rows: 35.5 rows per CU and column tile,
// 64 taken).  The 16-row family picks the tile HEIGHT for the launch — BM = 16 s, s = ceil(rows / 16 / row tiles per step) — so the
// launch is one (or k) full step(s) of tiles that are only as tall as the rows ask for, and it is ONE launch: no remainder, no cut.
//   F16W: 16 s x 256, 16 waves side by side (16 columns each, s slabs of 16 rows per wave), s = 3 ... 16; two workgroups per CU up to s = 4
//   F16N: 16 s x 128, 8 waves side by side, s = 3 ... 10; two or three workgroups per CU
// Lab (tools/lab/gemm_lab_mf16.hip, same-run, us): k=9 256->1024 M = 9090 378 (128x256 on 8192 rows + 64x64 on 898) -> 354-357 (144x256),
// M = 11 110 474 -> 432 (176x256), M = 10 490 471 -> 422-427; k=5 512->512 M = 9090 251 -> 211-216 (48x128 / 144x128), M = 17 170 414 -> 388
// (144x256); QKV M = 9090 46.2 -> 41.2 (112x256); at full steps of the tall 32-row tiles (M = 16 160) nothing changes.  Same tile height:
// 128x256 317 (MF 16) vs 324 (MF 32, 4x4 waves), 256x256 615 vs 597 (8x2 waves): the matrix rate and the LDS traffic per flop are the same.
enum TileFam { F32 = 0, F16W = 1, F16N = 2 };
struct TileSel { int fam, id; };  // F32: id = TileId; F16W / F16N: id = slabs (BM = 16 id)
static int sel_bm(TileSel t) { return t.fam == F32 ? kTile[t.id].bm : 16 * t.id; }
static int sel_bn(TileSel t) { return t.fam == F32 ? kTile[t.id].bn : t.fam == F16W ? 256 : 128; }
constexpr int F16W_MIN = 3, F16W_MAX = 16, F16N_MIN = 3, F16N_MAX = 10;
static bool tile16_enabled() {  // NS_TILE16=0: the planner without the 16-row family (A/B runs; read once)
  static const bool on = [] { const char* e = getenv("NS_TILE16"); return !(e && e[0] == '0'); }();
  return on;
}
bool conv_gemm_tile16_enabled() { return launch_planner_enabled() && tile16_enabled(); }
static bool tile16n_enabled() {  // NS_TILE16N=0: the family without its 128-column form (A/B runs; read once)
  static const bool on = [] { const char* e = getenv("NS_TILE16N"); return !(e && e[0] == '0'); }();
  return on;
}
// per-CU rows of a launch of `wgs` equal workgroups (the dispatcher hands a CU its next workgroup when a slot frees; co-resident
// workgroups share the CU's matrix pipe, so what counts is how many land on the fullest CU) times the per-row, per-chunk rate.
// The ramp + drain is charged per 256 workgroups although co-resident workgroups overlap theirs: many small workgroups per CU each
// stream their own weight panel ((BM + BN) operand rows per BM x BN outputs), and the charge stands in for that — with it the
// model ranks 48 x 256 x 3 per CU 3 % behind 144 x 256 x 1 as the lab does (369 vs 357 us); without it the plan drifts to the small tiles.
static float tile16_time(int fam, int slabs, long M, int N, int chunks) {
  const int bm = 16 * slabs, bn = fam == F16W ? 256 : 128;
  const long wgs = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  const float steps = (float)((wgs + 255) / 256);
  const float rows = (float)bm * (fam == F16W ? 1.0f : 0.5f);  // in rows of a 256-column tile
  // 0.0281 us per row and K chunk = the 128x256 tile's 3.6 us per chunk (forward traces); the 128-column form re-reads the activation
  // panel twice as often (+2 %); a step's ramp and drain, the launch itself
  const float b1 = 0.0281f * rows * (fam == F16N ? 1.02f : 1.0f);
  return 5.0f + 0.0016f * rows * (float)chunks + steps * (8.0f + b1 * (float)chunks);
}
template <int S>
static hipError_t launch_f16w(const ConvGemm& p, hipStream_t st, const LaunchTiming* tm) { return launch_t<16 * S, 256, 32, 1, 1, 16, false, 0, 16>(p, st, tm); }
template <int S>
static hipError_t launch_f16n(const ConvGemm& p, hipStream_t st, const LaunchTiming* tm) { return launch_t<16 * S, 128, 32, 1, 1, 8, false, 0, 16>(p, st, tm); }
static hipError_t launch_tile(TileSel t, const ConvGemm& p, hipStream_t st, const LaunchTiming* tm) {
  if (t.fam == F16W) {
    switch (t.id) {
      case 3: return launch_f16w<3>(p, st, tm); case 4: return launch_f16w<4>(p, st, tm); case 5: return launch_f16w<5>(p, st, tm);
      case 6: return launch_f16w<6>(p, st, tm); case 7: return launch_f16w<7>(p, st, tm); case 8: return launch_f16w<8>(p, st, tm);
      case 9: return launch_f16w<9>(p, st, tm); case 10: return launch_f16w<10>(p, st, tm); case 11: return launch_f16w<11>(p, st, tm);
      case 12: return launch_f16w<12>(p, st, tm); case 13: return launch_f16w<13>(p, st, tm); case 14: return launch_f16w<14>(p, st, tm);
      case 15: return launch_f16w<15>(p, st, tm); case 16: return launch_f16w<16>(p, st, tm);
      default: return hipErrorInvalidValue;
    }
  }
  if (t.fam == F16N) {
    switch (t.id) {
      case 3: return launch_f16n<3>(p, st, tm); case 4: return launch_f16n<4>(p, st, tm); case 5: return launch_f16n<5>(p, st, tm);
      case 6: return launch_f16n<6>(p, st, tm); case 7: return launch_f16n<7>(p, st, tm); case 8: return launch_f16n<8>(p, st, tm);
      case 9: return launch_f16n<9>(p, st, tm); case 10: return launch_f16n<10>(p, st, tm);
      default: return hipErrorInvalidValue;
    }
  }
  switch (t.id) {
    case T256: return launch_t<256, 256, 32, 1, 8, 2>(p, st, tm);
    case T128: return launch_t<128, 256, 32, 1, 4, 4>(p, st, tm);
    case T64W: return launch_t<64, 256, 32, 1, 2, 4>(p, st, tm);
    case T64: return launch_t<64, 128, 32, 1, 2, 4>(p, st, tm);
    case T64N: return launch_t<64, 64, 32, 1, 2, 2>(p, st, tm);
    default: return launch_t<32, 128, 32, 1, 1, 4>(p, st, tm);
  }
}
struct RowPlan { TileSel main; int main_rows; TileSel rem; float us; };  // main_rows == 0: one launch of rem
static RowPlan plan_rows(long M, int N, int chunks) {
  constexpr float CUT_US = 3.0f;  // a second launch: its ramp is in a(tile), this is the boundary itself
  RowPlan best{{F32, -1}, 0, {F32, T64}, 1e30f};
  // short contractions (QKV, fc: K = d, 8-16 chunks): prologue and epilogue weigh as much as the K loop, and three 64x128 workgroups
  // per CU interleave them better than one tall 16-wave tile — settled by forward A/B in round 3 (config 2 5.476 -> 5.456 ms) and
  // again by the model's own margin in round 4 (B = 20 QKV: 256x256 one step 83 us in the lab against 77 us on 64x128)
  // long contractions (K > 512, long_k_threshold) stay on the tiles that accumulate in chunks (k_conv_gemm ACC2): 128 x 256 and below — the 256 x 256
  // tile's 64 accumulator registers per lane leave no room for the second set.  (Round 5 ran the decoder's k=9 GEMM on it at
  // B = 16: 543 us per launch against 554 on two rounds of 128 x 256 — 0.8 % of a forward for 3.7x less rounding error.)
  const bool long_k = chunks * 32 > long_k_threshold() && acc_chunk_k() > 0;
  const int first = chunks <= 16 ? T64W : long_k ? T128 : T256;
  for (int t = first; t < N_TILES; ++t) {
    if (kTile[t].bn > N && t != T32 && t != T64N && t != T64) continue;  // (a 256-wide tile on a narrower output: never)
    const float c = tile_time(t, M, N, chunks);
    if (c < best.us) best = RowPlan{{F32, -1}, 0, {F32, t}, c};
  }
  for (int t = first; t <= T64; ++t) {
    if (kTile[t].bn > N) continue;
    const long ntn = (N + kTile[t].bn - 1) / kTile[t].bn;
    const long per = (256 / ntn) * kTile[t].bm;  // rows of one full step
    if (per <= 0) continue;
    for (long k = M / per; k >= 1 && k >= M / per - 1; --k) {
      const long rows = k * per, rest = M - rows;
      if (rest <= 0) continue;
      const float cm = tile_time(t, rows, N, chunks);
      for (int r = t + 1; r < N_TILES; ++r) {
        if (kTile[r].bn > N && r < T64) continue;
        const float c = cm + CUT_US + tile_time(r, rest, N, chunks);
        if (c < best.us) best = RowPlan{{F32, t}, (int)rows, {F32, r}, c};
      }
    }
  }
  // the 16-row family: one launch, the tile height chosen for the row count.  Where the best 32-row plan is ONE launch that loads the
  // CUs evenly (workgroups >= 90 % of a multiple of 256; B = 4: 512 tiles of 64x128, +1.7 % in the forward when the family took it)
  // the family has to beat it by 3 %: between near-ties the plans that forward A/B runs settled at the BASELINE
  // configurations stay (with the family allowed everywhere configs 2 / 4 / 5 measured 5.21-5.32 / 55.93-55.95 / 12.18-12.19 ms
  // against 5.23 / 55.91-55.93 / 12.16-12.17 without it).  Against a cut plan or badly filled steps: 1 %.
  if (tile16_enabled()) {
    float margin = 1.01f;
    if (best.main_rows == 0 && best.rem.fam == F32) {
      const long w = tile_wgs(best.rem.id, M, N);  // balance over the 256 CUs (co-resident workgroups share a CU's matrix pipe)
      if ((double)w / (double)(((w + 255) / 256) * 256) >= 0.9) margin = 1.03f;
    }
    // (16 s x 256 on 16 waves and 16 s x 128 on 8: 4 s accumulator registers per lane — s <= 8 has room for the second set)
    const int w_max = long_k ? 8 : F16W_MAX, n_max = long_k ? 8 : F16N_MAX;
    if (N % 256 == 0)
      for (int sl = F16W_MIN; sl <= w_max; ++sl) {
        const float c = tile16_time(F16W, sl, M, N, chunks) * margin;
        if (c < best.us) best = RowPlan{{F32, -1}, 0, {F16W, sl}, c};
      }
    if (N % 128 == 0 && tile16n_enabled())
      for (int sl = F16N_MIN; sl <= n_max; ++sl) {
        const float c = tile16_time(F16N, sl, M, N, chunks) * margin;
        if (c < best.us) best = RowPlan{{F32, -1}, 0, {F16N, sl}, c};
      }
  }
  return best;
}

// the plan of a plain (no row epilogue) GEMM of this shape, for introspection (nar_fs2.h ns_plan_gemm): false = the shape is below
// the planner's range (small-grid K-split ladder) or outside it (Cin % 32 != 0, N < 128), nothing is written
bool conv_gemm_plan(int M, int N, int Cin, int KW, int out[8]) {
  if (!launch_planner_enabled() || M <= 0 || N < 128 || Cin % 32 != 0) return false;
  const long rows64 = ((long)M + 63) / 64;
  if (rows64 * ((N + 127) / 128) <= 256) return false;
  const RowPlan pl = plan_rows(M, N, KW * (Cin / 32));
  const TileSel mt = pl.main_rows ? pl.main : pl.rem;
  out[0] = sel_bm(mt); out[1] = sel_bn(mt); out[2] = pl.main_rows ? pl.main_rows : M;
  out[3] = pl.main_rows ? sel_bm(pl.rem) : 0; out[4] = pl.main_rows ? sel_bn(pl.rem) : 0; out[5] = pl.main_rows ? M - pl.main_rows : 0;
  out[6] = mt.fam == F32 ? 32 : 16;
  out[7] = (int)(pl.us + 0.5f);
  return true;
}

// Height of the FULL-ROW tile (LayerNorm / predictor-tail epilogue; N = 256 or 512 columns = one activation row) for M rows.  The
// 32-row tile's launches are a staircase too — 285 tiles (B = 9) put two on 29 CUs and everyone waits for those: 64 rows per CU
// for 35.5 — so the height is the multiple of 16 that gives the fullest CU the fewest rows, BM x ceil(row tiles / 256); ties keep
// the 32-row tile (MF 32) the BASELINE configurations were settled on.  Lab (us): w_2 + LN M = 9090 75 -> 60 (48 rows), M = 17 170
// 117 -> 98 (80 rows); fc + LN 26.8 -> 22.4, 40.3 -> 35.1; predictor conv + LN 59 -> 47.5, 92 -> 76.5; M = 16 160 unchanged (64 = 2 x 32).
int conv_gemm_row_tile(int M, int N, int K) {
  if (!launch_planner_enabled() || !tile16_enabled()) return 32;
  // (256 and 512 columns take the same heights; at 512 the two B staging buffers alone are 128 KB of the CU's 160: 112 rows fit)
  // long contractions (K > 256: FFN w_2, the predictors' convolutions) keep to the heights with room for the second accumulator
  // set (k_conv_gemm ACC2: BM * N / 1024 <= 32 registers per lane): all of them at 256 columns, 32 and 48 rows at 512
  const int top = N > 256 ? 48 : 112;  // (N = 512: K >= 512 always — the attention output projection contracts over d = N)
  (void)K;
  long best = 32 * ((((long)M + 31) / 32 + 255) / 256);
  int bm = 32;
  for (int c = 48; c <= top; c += 32) {  // 48, 80, 112: the odd multiples of 16 — an even one only ever ties rounds of the 32-row tile
    const long rows = (long)c * ((((long)M + c - 1) / c + 255) / 256);
    if (rows < best) { best = rows; bm = c; }
  }
  return bm;
}

static hipError_t launch_conv_gemm_impl(const ConvGemm& p_in, hipStream_t st, bool allow_split, const LaunchTiming* tm);
// the tall tile of the one-tile-per-launch rules (NS_PLAN=0): 256 x 256, or — for a long contraction that accumulates in chunks —
// 128 x 256, the tallest tile with room for the second accumulator set (same rows, twice the rounds)
static hipError_t launch_tall(const ConvGemm& p, hipStream_t st, const LaunchTiming* tm) {
  if (p.KW * p.Cin > 256 && acc_chunk_k() > 0) return launch_t<128, 256, 32, 1, 4, 4>(p, st, tm);
  return launch_t<256, 256, 32, 1, 8, 2>(p, st, tm);
}
hipError_t launch_conv_gemm(const ConvGemm& p, hipStream_t st, const LaunchTiming* tm) { return launch_conv_gemm_impl(p, st, true, tm); }

static hipError_t launch_conv_gemm_impl(const ConvGemm& p_in, hipStream_t st, bool allow_split, const LaunchTiming* tm) {
  ConvGemm p = p_in;
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if (p.m_base != 0 && p.epi != EPI_NONE) return hipErrorInvalidValue;
  if (p.ldw == 0) p.ldw = p.KW * p.Cin;
  if (p.Cin % 16 != 0 || (p.ldx & 3) != 0 || (p.ldw & 3) != 0 || p.ldw < p.KW * p.Cin) return hipErrorInvalidValue;
  if ((p.N & 3) != 0 || (p.ldy & 3) != 0 || (p.resid && (p.ldr & 3) != 0)) return hipErrorInvalidValue;  // float4 epilogues
  // descriptor offsets are 31-bit: a tile's rows (BM + KW) * ldx and BN * K floats must stay below 2^29 floats
  if ((long long)(256 + p.KW) * p.ldx >= (1ll << 29) || (long long)512 * p.ldw >= (1ll << 29)) return hipErrorInvalidValue;
  if (p.epi != EPI_NONE && p.e.ticket) {
    // ticketed row epilogue on the small-grid ladder (same tile choices as below): the smallest tile that still gives about
    // one workgroup per CU, the rest of the CU spent on an in-workgroup K split.  Row widths 256 and 512.
    if (!conv_gemm_ticket_ok(p.M, p.N, p.Cin) || p.ldy != p.N || (p.resid && (p.ldr & 3)) || (p.epi == EPI_LN && !p.e.y_out)) return hipErrorInvalidValue;
    const long rows32 = (p.M + 31) / 32;
    auto wgs = [&](long rows, int bn) { return rows * ((p.N + bn - 1) / bn); };
    const int nch = p.KW * (p.Cin / 32);
#define NS_TICKET_LADDER(NVT)                                                                                                        \
    if (wgs(rows32, 32) <= 256) return nch >= 16 ? launch_t<32, 32, 32, 8, 1, 1, false, NVT>(p, st, tm) : launch_t<32, 32, 32, 4, 1, 1, false, NVT>(p, st, tm); \
    if (wgs(rows32, 64) <= 256) return launch_t<32, 64, 32, 4, 1, 2, false, NVT>(p, st, tm);                                             \
    /* between one and two rounds of that rung (B = 17 ... 24 encoder grids): 48 rows tall, one round (16-row family) */                 \
    if (launch_planner_enabled() && tile16_enabled() && wgs((p.M + 47) / 48, 64) <= 256)                                                 \
      return launch_t<48, 64, 32, 4, 1, 2, false, NVT, 16>(p, st, tm);                                                                   \
    if (wgs(rows32, 128) <= 512) return launch_t<32, 128, 32, 2, 1, 4, false, NVT>(p, st, tm);                                           \
    return hipErrorInvalidValue; /* (unreachable: conv_gemm_ticket_ok bounds the tile count) */
    if (p.N == 256) { NS_TICKET_LADDER(1) }
    NS_TICKET_LADDER(2)
#undef NS_TICKET_LADDER
  }
  if (p.epi != EPI_NONE) {
    // full-row tile: BM rows x N columns, the waves side by side (32 rows: N / 32 waves with one 32x32 MFMA tile each)
    if (!conv_gemm_row_epilogue_ok(p.M, p.N, p.Cin) || (p.epi == EPI_LN && p.ldy != p.N) || (p.resid && (p.ldr & 3))) return hipErrorInvalidValue;
    p.e.y_out = p.Y;  // the full-row tile's LayerNorm rows go straight to Y (the ticketed form above keeps raw rows in Y and y_out apart)
    // (the height follows the row count: conv_gemm_row_tile above; 16 waves side by side in the 16-row family)
    // (the heights the rule can pick: a 64- / 96- / 128-row tile only ever TIES two / three / four rounds of the 32-row one)
    const int bm = conv_gemm_row_tile(p.M, p.N, p.KW * p.Cin);
    if (p.N == 256) {
      switch (bm) {
        case 48: return launch_t<48, 256, 32, 1, 1, 16, true, 0, 16>(p, st, tm);
        case 80: return launch_t<80, 256, 32, 1, 1, 16, true, 0, 16>(p, st, tm);
        case 112: return launch_t<112, 256, 32, 1, 1, 16, true, 0, 16>(p, st, tm);
        default: return launch_t<32, 256, 32, 1, 1, 8, true>(p, st, tm);
      }
    }
    // (512 columns: 32 or 48 rows.  The 80- and 112-row forms of round 5 held 40 / 56 accumulator registers per lane — no room for
    //  the second accumulator set — and every 512-wide full-row GEMM contracts over K >= 512, so the rule never picks them.)
    if (bm == 48) return launch_t<48, 512, 32, 1, 1, 16, true, 0, 16>(p, st, tm);
    return launch_t<32, 512, 32, 1, 1, 16, true>(p, st, tm);
  }
  const bool bk32 = (p.Cin % 32) == 0;
  // Tile / wave-grid choice (tools/lab sweeps on the path's shapes, MI355X, same-run comparisons).  What wins is
  // many waves per workgroup with ONE 32x32 MFMA tile each: 8 waves as 2x4 over a 64x256 or 64x128 block tile.  The
  // DMA issue cost of a chunk is spread over 8 waves, each barrier interval still holds 16-32 MFMAs per wave, and
  // 2+ workgroups per CU interleave their barrier phases.  (4-wave 2x2 grids with 2x2 tiles per wave: 110-125
  // TFLOP/s on the dominant shape; 2x4 grids: 139-141.)
  const long rows64 = (p.M + 63) / 64, rows32 = (p.M + 31) / 32;
  auto wgs = [&](long rows, int bn) { return rows * ((p.N + bn - 1) / bn); };
  // Narrow outputs (N = 80: mel_linear, the PostNet's last layer): three 32-column tiles instead of a 128-wide tile that
  // is 37 % padding (tools/lab/gemm_lab_n80.hip: k5 512->80, M=16160 99.8 -> 85.2 us; M=64640 360 -> 310 us)
  if (bk32 && p.N > 64 && p.N <= 96 && p.KW * p.Cin >= 1024) {
    if (launch_planner_enabled()) {
      // at every row count beyond the small-grid ladder's 32x64 rung: 32 rows x 96 columns with four K groups, one workgroup
      // per CU (tools/lab/gemm_lab_n80b.hip, k5 512->80, us: M = 5050 46.9 (32x128 KS2) -> 36.8, 9090 87.1 -> 72.4,
      // 12120 84.9 -> 71.1, 32480 158.5 (64x96) -> 140.9; below M ~ 4100 the ladder wins, 29.1 vs 37.2)
      if (wgs(rows32, 64) > 256) {
        // between one and two rounds of that tile (B = 9 ... 12 utterances: 285 workgroups took as long as 512): 48 rows tall, one round
        if (tile16_enabled() && rows32 > 256 && (p.M + 47) / 48 <= 256) return launch_t<48, 96, 32, 4, 1, 3, false, 0, 16>(p, st, tm);
        // between two and three rounds (B = 17 ... 20): 80 rows tall, one round, 15 waves as 5 x 3 with one K loop each
        if (tile16_enabled() && rows32 > 512 && (p.M + 79) / 80 <= 256) return launch_t<80, 96, 32, 1, 5, 3, false, 0, 16>(p, st, tm);
        return launch_t<32, 96, 32, 4, 1, 3>(p, st, tm);
      }
    } else {
      if (rows64 >= 512) return launch_t<64, 96, 32, 1, 2, 3>(p, st, tm);
      if (rows32 >= 400) return launch_t<32, 96, 32, 4, 1, 3>(p, st, tm);
    }
  }
  // Everything from about one step of the 64x128 tile upwards: the step-aware plan (above).  Below that the launch is one
  // partial step whatever the tile, and the small-grid ladder at the end of this function (in-workgroup K split) is faster.
  if (launch_planner_enabled() && allow_split && bk32 && p.N >= 128 && wgs(rows64, 128) > 256) {
    const RowPlan pl = plan_rows(p.M, p.N, p.KW * (p.Cin / 32));
    if (pl.main_rows == 0) return launch_tile(pl.rem, p, st, tm);
    const LaunchTiming t0{tm ? tm->start : nullptr, nullptr}, t1{nullptr, tm ? tm->stop : nullptr};
    const hipError_t e = launch_tile(pl.main, row_range(p, 0, pl.main_rows), st, &t0);
    if (e != hipSuccess) return e;
    return launch_tile(pl.rem, row_range(p, pl.main_rows, p.M - pl.main_rows), st, &t1);
  }
  // Mid-size row counts on the long-K convolutions (FFN k=9, PostNet k=5): a few utterances, or a packed variable-length
  // batch.  Between the small-grid ladder's 512 workgroups and the point where whole rounds of the 64-row tiles average out,
  // the 32x128 tile with two K groups keeps winning (tools/lab/gemm_lab_rem.hip, us: k9 256->1024 M = 2158 125 vs 151 (64x128),
  // 3000 134 vs 160, 4100 204 vs 220, 4771 214 vs 226, then 6000 261 vs 236; k5 512->512 M = 4100 136 vs 164, 4771 139 vs 167,
  // 6000 145 vs 171, then 7296 190 vs 177): its workgroups are half the size, so the partial last round costs half as much.
  // short contraction, wide output (the QKV projection: K = d, N = 3d): 8-16 K steps per tile, so prologue and epilogue weigh
  // as much as the loop and three 64x128 workgroups per CU interleave them better than two 64x256 ones.  Measured in the
  // FORWARD (alternating same-box runs; the lab loop had it the other way round at config 5): config 2 5.476 -> 5.456 ms,
  // config 4 57.50 -> 57.33 ms, config 5 unchanged.
  if (bk32 && p.N >= 512 && p.KW * p.Cin <= 512 && wgs(rows64, 128) > 256) return launch_t<64, 128, 32, 1, 2, 4>(p, st, tm);
  // ... unless a taller 16-wave tile, one workgroup per CU, fills its rounds: fewer passes over the weights, and with one
  // or two rounds there is nothing a second co-resident workgroup could hide.  Settled by alternating whole-forward runs on
  // one box (tools/ab_forward.sh), per launch: 256x256 — decoder k=9 GEMM 547.7 -> 539.7 us at config 2 (254 tiles, one round),
  // 1096 -> 1075 us at config 5 (492 tiles, two rounds), PostNet layer 604 -> 598 us at config 5; 128x256 — PostNet layer
  // 311.9 -> 306.8 us at config 2 (254 tiles).  Rounds that do not fill lose badly (config 4: 540 tiles of 256x256 = 2.1
  // rounds: PostNet layer +30 %; 2156 tiles of 128x256 = 8.4 rounds: k=9 GEMM +6 %), hence the fill tests.
  if (bk32 && p.N >= 512 && wgs(rows64, 256) >= 400) {
    auto fill = [](long n) { return (double)n / (double)(((n + 255) / 256) * 256); };  // of the last round, one workgroup per CU
    const long c256 = wgs((p.M + 255) / 256, 256), c128 = wgs((p.M + 127) / 128, 256);
    if (c256 >= 200 && c256 <= 512 && fill(c256) >= 0.95) return launch_tall(p, st, tm);  // measured for one and two rounds only
    if (c128 <= 512 && fill(c128) >= 0.95) return launch_t<128, 256, 32, 1, 4, 4>(p, st, tm);
    // A row count somewhat above one or two FULL rounds of the 256x256 tile (packed variable-length batches: M is whatever
    // the utterances add up to; a uniform batch of one utterance more than a round holds) and too small for the many-round
    // 64x256 form to average its partial last round away: the full rounds go to the tall tile at its single-round rate, the
    // remaining rows to the rules below as a launch of their own (rows are independent; a convolution's taps reach across the
    // cut through the operand pointers, ConvGemm::m_base keeps the utterance positions).  Long-form ragged batch, M = 21 155,
    // k=9 GEMM: 825 us as 1324 tiles of 64x256, 761 us as one round of 256x256 (16 384 rows) + 4771 rows of 64x128.
    // (Cutting at rounds of the 128x256 tile was measured too and loses to the single launch: M = 10 350 468 vs 413 us,
    // tools/lab/gemm_lab_rem.hip.)
    const long ntn = (p.N + 255) / 256, per = 256 / ntn;  // row tiles of one 256-workgroup round
    if (allow_split && p.epi == EPI_NONE && per >= 1 && p.M > 256 * per && wgs(rows64, 256) < 4 * 512) {
      const long r256 = 256 * per;
      const long n = p.M / r256 > 2 ? 2 : p.M / r256;
      const LaunchTiming t0{tm ? tm->start : nullptr, nullptr}, t1{nullptr, tm ? tm->stop : nullptr};
      const hipError_t e = launch_tall(row_range(p, 0, (int)(n * r256)), st, &t0);
      if (e != hipSuccess) return e;
      return launch_conv_gemm_impl(row_range(p, (int)(n * r256), (int)(p.M - n * r256)), st, false, &t1);
    }
  }
  if (bk32 && p.N >= 512 && wgs(rows64, 256) >= 400) {
    // Few rounds (a batch of ~10 utterances, a packed variable-length batch): a launch's time is a staircase in units of 256
    // workgroups, one per CU — 64x256 137 us per step of 256 tiles, 64x128 73.6 us (k=9 256->1024; tools/lab/gemm_lab_tall.hip) —
    // so when the half-size tile needs fewer than 2 x 0.93 as many steps it wins by up to a step of the big one
    // (M = 10 240: 368 vs 413 us).  Beyond eight steps the partial last step no longer matters and the wider tile's rate does.
    const long sc = (wgs(rows64, 256) + 255) / 256, sd = (wgs(rows64, 128) + 255) / 256;
    if (sc <= 8 && p.N % 128 == 0 && (double)sd * 0.5 < 0.93 * (double)sc) return launch_t<64, 128, 32, 1, 2, 4>(p, st, tm);
    return launch_t<64, 256, 32, 1, 2, 4>(p, st, tm);
  }
  if (bk32 && p.N >= 128 && wgs(rows64, 128) > 256) return launch_t<64, 128, 32, 1, 2, 4>(p, st, tm);
  // The remainder of a split plan stays on tiles WITHOUT an in-workgroup K split: every such tile sums a row's contraction in
  // the same order, so the rows behind the cut carry the same bits as the rows ahead of it (replicas of one utterance inside
  // a batch stay bit-identical wherever the cut falls); the K-split ladder below rounds differently.
  if (!allow_split && bk32 && p.N >= 128) return launch_t<64, 128, 32, 1, 2, 4>(p, st, tm);
  // Fewer output tiles than that (encoder-side GEMMs, single-utterance latency): a workgroup's time is set by how fast
  // ONE CU can pull its operand panels, (BM + BN) * K * 4 bytes, through LDS-DMA, so what matters is to put every CU to
  // work — the smallest tile that still yields <= 256 workgroups (one round, one per CU) — and to spend the rest of the
  // CU's wave slots and LDS on an in-workgroup split of K (KS groups, each with its own double buffer: KS x the bytes
  // in flight).  Ladder measured on the path's shapes in tools/lab/gemm_lab_small.hip (same-run comparisons):
  //   e.g. M=100 k9 256->1024: 64x64 KS4 44.7 us -> 32x32 KS8 19.3;  M=788 k1 1024->256: 23.9 -> 11.9;
  //        M=788 k5 512->512: 51.2 -> 31.8 (32x64 KS4);  M=788 k9 256->1024: 49.5 -> 45.4 (32x128 KS2).
  if (bk32) {
    const int nch = p.KW * (p.Cin / 32);
    if (wgs(rows32, 32) <= 256) return nch >= 16 ? launch_t<32, 32, 32, 8, 1, 1>(p, st, tm) : launch_t<32, 32, 32, 4, 1, 1>(p, st, tm);
    if (wgs(rows32, 64) <= 256) return launch_t<32, 64, 32, 4, 1, 2>(p, st, tm);
    // (row widths the ticketed ladder above serves: the same rungs in the same places, so that the two-launch form of a row epilogue —
    //  this GEMM, then the row kernel — sums every row like the ticketed launch it stands in for: ns_config.row_epilogue, same bits)
    const bool row_width = p.N == 256 || p.N == 512;
    if (row_width && launch_planner_enabled() && tile16_enabled() && wgs((p.M + 47) / 48, 64) <= 256)
      return launch_t<48, 64, 32, 4, 1, 2, false, 0, 16>(p, st, tm);
    // (a square 64 x 64 KS4 tile on 16 waves for long-K launches of about one workgroup per CU — (64 + 64) operand rows per CU instead of
    //  (32 + 128) — measured 47.5 -> 44.9 us in the lab at M = 788 and nothing in the forward: single utterance 0.8443 / 0.8419 ms without,
    //  0.8421 / 0.8408 with, alternating runs; not taken.  tools/lab/gemm_lab_r5b.hip)
    if (wgs(rows32, 128) <= 512) {
      // between one and two rounds of the 32x128 rung (B = 9 ... 12 encoder grids: 288 workgroups took as long as 512): the same
      // rung 48 rows tall, one round (16-row family; tools/lab/gemm_lab_mf16.hip, k9 256->1024: M = 1152 80.5 -> 63.1 us, 1408 79.6 -> 63.9)
      if (!row_width && launch_planner_enabled() && tile16_enabled() && wgs(rows32, 128) > 256 && wgs((p.M + 47) / 48, 128) <= 256)
        return launch_t<48, 128, 32, 2, 1, 4, false, 0, 16>(p, st, tm);
      return launch_t<32, 128, 32, 2, 1, 4>(p, st, tm);
    }
    return launch_t<64, 64, 32>(p, st, tm);
  }
  if (wgs(rows32, 32) <= 512) return launch_t<32, 32, 16, 4, 1, 1>(p, st, tm);
  // Cin = 80 (the PostNet's first layer): wider tiles once there are enough of them (M=16160 77 -> 73.5 us, M=64640 269 -> 250 us)
  if (p.N >= 256 && wgs(rows64, 256) >= 1024) return launch_t<64, 256, 16, 1, 2, 4>(p, st, tm);
  if (p.N >= 128 && wgs(rows64, 128) >= 512) return launch_t<64, 128, 16, 1, 2, 4>(p, st, tm);
  return launch_t<64, 64, 16>(p, st, tm);
}

}  // namespace ns
