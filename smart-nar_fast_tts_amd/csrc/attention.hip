// Fused multi-head self attention on the gfx950 fp32 matrix cores — replaces
// MultiHeadAttention's head split + ScaledDotProductAttention (transformer/SubLayers.py:42-54,
// transformer/Modules.py:14-25): bmm -> /sqrt(d_k) -> masked_fill(-inf at padded KEYS) -> softmax -> bmm.
// The (H*B, S, S) score matrix is never materialised and the attention probabilities (a dead output at
// inference) are not produced.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries for the whole
// key sweep and is alone on its SIMD (the problem offers about one 32-query wave per SIMD at config 2), so the
// softmax VALU work is hidden INSIDE the wave: the loop is software-pipelined — while the matrix pipe runs
// S(t+1)^T = K(t+1) Q^T, the VALU does the online softmax of tile t, then P(t) V(t) follows.
// K/V tiles of 32 keys arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`), K one tile ahead of V, two buffers each.
//
// The trick that removes every P re-layout: compute the TRANSPOSED score tile S^T = K Q^T
// (A = K tile, B = Q^T held in registers for the whole kernel).  In the 32x32 C/D layout each lane then
// holds, for ONE query (col = lane&31), 16 of the 32 keys: key(r,h) = (r&3) + 8*(r>>2) + 4*h.
//   * the softmax row reduction is 16 in-lane values + one cross-half shuffle;
//   * the online-softmax rescale of O^T (same column = same query) is lane-local;
//   * for O^T += V^T P^T the B operand of MFMA step r is literally register p[r]: MFMA's k index is free to
//     be permuted as long as A and B agree, so step r / lane-half h is *defined* to be key(r,h) and the
//     A operand is V[key(r,h)][d], one conflict-free ds_read_b32 per MFMA.
// The K tile's DMA image is lane-linear, so its ds_read_b128 fragment reads are de-conflicted by an XOR swizzle
// on the source side (16-byte slot s of key row r holds chunk s ^ f(r)); V is read a row at a time and stays linear.
// Keys >= lens[b] get -inf before the softmax; key tiles entirely past lens[b] contribute exact zeros and
// are skipped; rows past S are zero-filled by the buffer descriptor's range check.  Padded QUERY rows are
// computed like the reference does.  An utterance with lens[b] == 0 yields NaN rows (0 * inf), as the reference's
// all -inf softmax does (SURVEY.md §8b "Errors").
#include <hip/hip_ext.h>
#include "kernels.h"

namespace ns {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int ATT_STRIP_SPLIT_MAX = 8;  // workgroups sharing one 32-query strip's key axis (k_attention_strip)

// One step of merging key-range partials, out += O_i * w_i / l += l_i * w_i, with the multiply-adds spelled out: the last
// arriver of k_attention_strip and k_attention_merge must produce the same bits from the same partials (the two-launch form
// of a ticketed merge, ns_config.row_epilogue), whatever hipcc would contract in either kernel.
__device__ __forceinline__ f32x4 merge_fma(f32x4 o, float w, f32x4 acc) {
#pragma clang fp contract(off)
  return f32x4{__builtin_fmaf(o[0], w, acc[0]), __builtin_fmaf(o[1], w, acc[1]), __builtin_fmaf(o[2], w, acc[2]), __builtin_fmaf(o[3], w, acc[3])};
}

template <int DK>
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, const long long* __restrict__ lens,
                                                    int S_grid, int d, float c_scale, float* __restrict__ out, int nsplit,
                                                    float* __restrict__ opart, float* __restrict__ mlpart,
                                                    const int* __restrict__ pk_off, const int* __restrict__ pk_win,
                                                    const int* __restrict__ att_off, const int* __restrict__ att_order, int nutt,
                                                    int pk_rows, int pk_heads) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BC = 32;                    // keys per tile
  constexpr int CPR = DK / 4;               // 16-B chunks per tile row
  constexpr int RPI = 64 / CPR;             // tile rows one wave-wide DMA instruction fills
  constexpr int NI = (BC * CPR) / (64 * 4); // DMA instructions per wave per tile (4 waves share the tile)
  constexpr int NG = DK / 8;                // k-groups of the QK^T contraction
  constexpr int NDB = DK / 32;              // 32-wide d blocks of O^T
  constexpr int FSH = (DK == 32) ? 1 : 0;   // swizzle: rows per 256-B bank row = 2 for 128-B rows, else 1
  constexpr int FMSK = (DK == 32) ? 7 : 15;
  static_assert(DK == 32 || DK == 64 || DK == 128, "d_k");
  static_assert(NI >= 1 && RPI * CPR == 64, "tile/DMA geometry");

  // four DISTINCT LDS objects + a 2x unrolled tile loop with static buffer roles: hipcc tracks in-flight LDS-DMA per
  // LDS object, so reading K1/V0 does not wait for the DMA that is filling K0/V1 (see gemm_conv.hip)
  __shared__ __attribute__((aligned(16))) float Ks0[BC * DK];
  __shared__ __attribute__((aligned(16))) float Ks1[BC * DK];
  __shared__ __attribute__((aligned(16))) float Vs0[BC * DK];
  __shared__ __attribute__((aligned(16))) float Vs1[BC * DK];

  // XCD-aware bijective remap (workgroup L runs on XCD L % 8: observed, used for speed only).  In launch order the query
  // tiles of one (batch, head) are consecutive workgroups, i.e. they land on 8 DIFFERENT XCDs and every XCD pulls that
  // head's K and V through its own L2 (profiles/r02: 4.5x the algorithmic bytes at config 2).  Handing XCD x the x-th
  // contiguous slice of the (batch, head, query tile) order instead keeps all query tiles of a head on one L2.
  int bx, hd, b;
  if (att_off) {
    // packed rows: a flat work list, longest utterances first (kernels.h RowMap::att_off); workgroup i belongs to the rank r
    // with att_off[r] <= i < att_off[r + 1], and is query tile (i - att_off[r]) % qtiles of head (i - att_off[r]) / qtiles
    const int i = blockIdx.x;
    int lo = 0, hi = nutt - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (att_off[mid] <= i / nsplit) lo = mid;
      else hi = mid - 1;
    }
    b = att_order[lo];
    // (with a key split every (head, query tile) appears nsplit times: att_off counts unsplit workgroups, the launch has
    //  nsplit times as many and workgroup i is split i % nsplit of unsplit workgroup i / nsplit)
    const int local = i / nsplit - att_off[lo], qtiles = (pk_win[b] + 127) / 128;
    hd = local / qtiles;
    bx = (local - hd * qtiles) * nsplit + i % nsplit;
  } else {
    const int nx = gridDim.x, ny = gridDim.y, nblk = nx * ny * gridDim.z;
    const int L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = L & 7;
    const int pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (L >> 3);
    bx = pos % nx;
    hd = (pos / nx) % ny;
    b = pos / (nx * ny);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, qi = lane & 31;
  // nsplit > 1 (few workgroups, long key axis): workgroup (qt, sp) sweeps only the sp-th contiguous share of the key
  // tiles and leaves an un-normalised partial (O^T, m, l) for k_attention_merge; softmax is exact per part
  const int qt = bx / nsplit, sp = bx - qt * nsplit;
  // packed rows (kernels.h RowMap): utterance b's rows start at pk_off[b] and number pk_win[b]; the grid is sized for the
  // longest window, a query tile past this utterance's window has nothing to do
  const int S = pk_win ? pk_win[b] : S_grid;
  const size_t row0 = pk_off ? (size_t)pk_off[b] : (size_t)b * S_grid;
  if (qt * 128 >= S) return;
  const int q = qt * 128 + wid * 32 + qi;
  const int ld = 3 * d;
  const float* base = qkv + row0 * ld + hd * DK;
  long long len_ll = lens ? lens[b] : (long long)S;
  const int len = (int)(len_ll < S ? len_ll : S);
  const int nkt_all = (len + BC - 1) / BC;
  const int tps = (nkt_all + nsplit - 1) / nsplit;       // key tiles per split
  const int t0 = sp * tps;                               // first key tile of this workgroup
  const int nkt = max(0, min(nkt_all, t0 + tps) - t0);   // its number of key tiles (kt below is local: 0..nkt-1)

  // descriptors over this (batch, head)'s K and V column blocks; rows >= S are out of range -> the DMA writes zeros
  const int nrec = ((S - 1) * ld + DK) * 4;  // bytes from a head's column block in row 0 to its end in row S-1
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(base + d), (short)0, nrec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(base + 2 * d), (short)0, nrec, 0x00020000);

  // per-lane DMA geometry: instruction i of this wave fills tile rows (wid*NI + i)*RPI + lane/CPR, slot lane%CPR
  const int lr = lane / CPR, ls = lane % CPR;
  int vk[NI], vv[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int r = (wid * NI + i) * RPI + lr;
    const int f = (r >> FSH) & FMSK;
    vk[i] = (r * ld + ((ls ^ f) * 4)) * 4;
    vv[i] = (r * ld + ls * 4) * 4;
  }
  const int tile_step = BC * ld * 4;  // bytes per key tile
  auto dma_k = [&](float* Kd, int kt) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr_t)&Kd[(wid * NI + i) * RPI * DK], 16, vk[i] + (t0 + kt) * tile_step, 0, 0, 0);
  };
  auto dma_v = [&](float* Vd, int kt) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr_t)&Vd[(wid * NI + i) * RPI * DK], 16, vv[i] + (t0 + kt) * tile_step, 0, 0, 0);
  };

  // the first tiles are requested before anything else, so that their round trip runs under the Q loads
  if (nkt > 0) {
    dma_k(Ks0, 0);
    dma_v(Vs0, 0);
    if (nkt > 1) dma_k(Ks1, 1);
  }

  // Q^T operand: lane (q, h) keeps Q[q][8g + 4h + e], pre-multiplied by log2(e)/sqrt(d_k) so the scores come out
  // of the MFMA already scaled into the exp2 domain (one multiply per Q element instead of one per score)
  f32x4 qreg[NG];
  {
    // all NG loads in flight together, no branch: a row past S reads row S-1 instead and is scaled by zero (it is never
    // stored).  With `if (q < S)` around each load hipcc emitted load -> s_waitcnt vmcnt(0) NG times in a row — 16 serialized
    // round trips (~10 us) at the head of every workgroup, ahead of the first K / V DMA.
    const float* qrow = base + (size_t)(q < S ? q : S - 1) * ld + 4 * h;
    const float qs = q < S ? c_scale : 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) qreg[g] = *reinterpret_cast<const f32x4*>(qrow + 8 * g);
#pragma unroll
    for (int g = 0; g < NG; ++g) qreg[g] *= qs;
  }

  f32x16 o[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // K fragment read offsets (floats): row qi (the key this lane feeds as MFMA row), slot (2g + h) ^ f(row)
  int koff[NG];
  {
    const int f = (qi >> FSH) & FMSK;
#pragma unroll
    for (int g = 0; g < NG; ++g) koff[g] = qi * DK + (((2 * g + h) ^ f) * 4);
  }

  // S^T[key][q] = sum_d K[key][d] Q[q][d] for the K tile at kp
  auto qk = [&](const float* kp, f32x16& s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + koff[g]);
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qreg[g][e], s, 0, 0, 0);
    }
  };
  // key-padding mask + online softmax of tile kt; s becomes P, returns the O rescale factor
  auto softmax_tile = [&](int kt, f32x16& s) -> float {
    if ((t0 + kt) * BC + BC > len) {  // wave-uniform: only the tile that straddles lens[b] needs the per-key compare
      const int kbase = (t0 + kt) * BC + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + (r & 3) + 8 * (r >> 2);
        s[r] = key < len ? s[r] : -INFINITY;
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    // Lazy reference point: softmax is exact for ANY reference m as long as exp2(s - m) cannot overflow, so the
    // running reference only moves when a score exceeds it by more than 2^RESCALE_LOG2 (P stays <= 2^10, far inside
    // fp32 range and at unchanged relative precision).  After the first tile this almost never fires, which
    // removes the 64-register O^T rescale (and its accumulator-file round trip) from the steady state.
    constexpr float RESCALE_LOG2 = 10.f;
    const bool need = mt > m_run + RESCALE_LOG2;  // (-inf + 10 = -inf: the first finite tile always fires)
    const float m_new = need ? mt : m_run;
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = need ? __builtin_amdgcn_exp2f(m_run - m_use) : 1.0f;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
      psum += s[r];
    }
    // keep the exponentials in THIS scheduling region (the one that holds the QK^T MFMAs): without the pin the
    // optimiser sinks them below pv()'s branch, next to their only consumers
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(s[r]));
    l_run = l_run * alpha + psum;
    m_run = m_new;
    return alpha;
  };
  // g: MFMA row i = 8u + 4h' + j  ->  8u + (j / JP) * 2 JP + h' * JP + j % JP, JP = 4 / NDB rows per 16-byte piece (see pv)
  constexpr int JP = 4 / NDB;
  auto row_label = [](int i) { return (i & ~7) + ((i & 3) / JP) * (2 * JP) + ((i >> 2) & 1) * JP + (i & 3) % JP; };
  // O^T[d][q] = alpha * O^T[d][q] + sum_key V[key][d] P[q][key]   (MFMA step r <-> key(r,h), B operand = p[r])
  auto pv = [&](const float* vtile, const f32x16& pr, float alpha) {
    // O^T block db, row i (= this lane's qi as the A-operand row) is output column d = NDB*g(i) + db — NOT 32*db + i: the NDB
    // values a lane feeds for one key are then adjacent in the V tile, one ds_read_b128 (b64 for d_k = 64) instead of NDB
    // ds_read_b32; MFMA rows are just labels, the stores below use the same labelling.  LDS instruction issue is not free
    // next to the MFMAs (tools/lab/mfma_mix.hip): 32 instead of 48 LDS reads per key tile.
    // ... with the rows of an 8-row group permuted (g below) so that in the C/D layout the two lane halves of a query hold
    // ADJACENT 16-byte pieces: a store instruction then writes 32 contiguous bytes per output row, as before the relabelling
    // (16-byte pieces 64 bytes apart cost 2.5 us per workgroup in the epilogue).
    const float* vp = vtile + (4 * h) * DK + NDB * row_label(qi);
    if (__any(alpha != 1.0f)) {  // wave-uniform; rare after the first tile (lazy reference point)
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      typedef float vrow_t __attribute__((ext_vector_type(NDB)));
      const vrow_t vv = *reinterpret_cast<const vrow_t*>(vp + ((r & 3) + 8 * (r >> 2)) * DK);
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const float v = vv[db];
        o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, pr[r], o[db], 0, 0, 0);
      }
    }
  };

  __syncthreads();

  f32x16 s_a, s_b;  // score tiles of even / odd key tiles (static roles in the 2x unrolled loop: no register copies)
  if (nkt > 0) qk(Ks0, s_a);
  __syncthreads();  // every wave has read K(0) before the loop's first DMA reuses its buffer

  // one key tile kt (not the last): K(kt) and V(kt-1) are dead -> refill their buffers with K(kt+2) / V(kt+1);
  // matrix pipe: next tile's scores, VALU: this tile's softmax, then P(kt) V(kt)
  auto tile = [&](int kt, float* Kdead, const float* Knext, const float* Vcur, float* Vdead, f32x16& s_cur, f32x16& s_next) {
    if (kt + 2 < nkt) dma_k(Kdead, kt + 2);
    dma_v(Vdead, kt + 1);
    qk(Knext, s_next);
    const float alpha = softmax_tile(kt, s_cur);
    pv(Vcur, s_cur, alpha);
    __syncthreads();  // drains the DMA issued above and fences the buffer reuse
  };
  for (int kt = 0; kt + 1 < nkt; kt += 2) {
    tile(kt, Ks0, Ks1, Vs0, Vs1, s_a, s_b);
    if (kt + 2 < nkt) tile(kt + 1, Ks1, Ks0, Vs1, Vs0, s_b, s_a);
  }
  if (nkt > 0) {
    if ((nkt - 1) & 1) {
      const float alpha = softmax_tile(nkt - 1, s_b);
      pv(Vs1, s_b, alpha);
    } else {
      const float alpha = softmax_tile(nkt - 1, s_a);
      pv(Vs0, s_a, alpha);
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  if (opart) {  // split-key mode: un-normalised partial for the merge kernel
    if (q < S) {
      const size_t row = pk_off ? (size_t)sp * pk_rows + row0 + q : (size_t)sp * gridDim.z * S + (size_t)b * S + q;
      const int nheads = pk_off ? pk_heads : (int)gridDim.y;
      float* dst = opart + row * d + hd * DK + 4 * h;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int pj = 0; pj < NDB; ++pj) {  // piece pj of row group u: rows j = pj*JP + e/NDB, block db = e%NDB
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = o[e % NDB][4 * u + pj * JP + e / NDB];
          *reinterpret_cast<f32x4*>(dst + 8 * NDB * u + 8 * pj) = v;
        }
      if (h == 0) {
        mlpart[(row * nheads + hd) * 2] = m_run;
        mlpart[(row * nheads + hd) * 2 + 1] = l_tot;
      }
    }
    return;
  }
  const float inv = 1.0f / l_tot;   // lens[b]==0 -> 0 * inf = NaN, as the reference
  if (q < S) {
    float* dst = out + (row0 + q) * d + hd * DK + 4 * h;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int pj = 0; pj < NDB; ++pj) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[e % NDB][4 * u + pj * JP + e / NDB] * inv;
        *reinterpret_cast<f32x4*>(dst + 8 * NDB * u + 8 * pj) = v;
      }
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// Small grids (single-utterance latency, the encoder's S <= 128): k_attention_strip.  Same arithmetic as k_attention — the
// same S^T = K Q^T / online softmax / O^T += V^T P^T per 32-key tile — with the roles in the workgroup swapped: the four
// waves share ONE strip of 32 queries and each sweeps its own contiguous range of key tiles through its own K / V buffers
// (no barrier in the loop; a wave is alone on its SIMD), then the four (O, m, l) partials are merged through LDS.  A launch
// therefore has S/32 x H x B workgroups instead of S/128 x H x B, and up to `nsplit` of them share a strip's key axis; their
// partials meet in memory and the LAST one to arrive (one ticket per strip, cdna_hip_programming.md Guideline 16 counter
// form: write-through stores, vmcnt drain, barrier, relaxed agent fetch_add / one agent acquire) merges them in split order.
// This replaces split-key k_attention + k_attention_merge on such launches (one launch instead of two, a quarter of the
// partial bytes) and removes the merge launch altogether when nsplit == 1 (the encoder).
// Merging, in-workgroup and across workgroups alike: out = sum_i O_i 2^(m_i - m) / sum_i l_i 2^(m_i - m), m = max_i m_i,
// i in key order — exact for any split (each partial is an exact softmax numerator / denominator relative to its own m_i).
template <int DK>
__global__ __launch_bounds__(256) void k_attention_strip(const float* __restrict__ qkv, const long long* __restrict__ lens,
                                                          int S_grid, int d, float c_scale, float* __restrict__ out, int nsplit,
                                                          float* __restrict__ opart, float* __restrict__ mlpart,
                                                          int* __restrict__ tickets, const int* __restrict__ pk_off,
                                                          const int* __restrict__ pk_win, int pk_rows) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BC = 32, CPR = DK / 4, RPI = 64 / CPR;
  constexpr int NI = (BC * CPR) / 64;  // DMA instructions per tile: ONE wave fills its own tile
  constexpr int NG = DK / 8, NDB = DK / 32;
  constexpr int FSH = (DK == 32) ? 1 : 0, FMSK = (DK == 32) ? 7 : 15;
  constexpr int JP = 4 / NDB;
  static_assert(DK == 32 || DK == 64 || DK == 128, "d_k");
  // one K and one V tile PER WAVE (wave-private: the loop needs no barrier); the compiler tracks LDS-DMA per LDS object, so a
  // read of Ks only waits for the DMAs into Ks.  After the sweep the K region holds the wave's O^T partial, V's head its (m, l).
  __shared__ __attribute__((aligned(16))) float Ks[4 * BC * DK];
  __shared__ __attribute__((aligned(16))) float Vs[4 * BC * DK];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, qi = lane & 31;
  const int nstrips = gridDim.x / nsplit;
  const int strip = blockIdx.x / nsplit, sp = blockIdx.x - strip * nsplit;
  const int hd = blockIdx.y, b = blockIdx.z;
  // packed rows (kernels.h RowMap): utterance b owns rows [pk_off[b], pk_off[b] + pk_win[b]) of the pk_rows packed rows; the grid is
  // sized for the longest window (S_grid), a strip past this utterance's window has nothing to do (no barrier was reached yet)
  const int row0 = pk_off ? pk_off[b] : b * S_grid;
  const int S = pk_off ? pk_win[b] : S_grid;
  if (strip * 32 >= S) return;
  const int q = strip * 32 + qi;
  const int ld = 3 * d;
  const float* base = qkv + (size_t)row0 * ld + hd * DK;
  const long long len_ll = lens ? lens[b] : (long long)S;
  const int len = (int)(len_ll < S ? len_ll : S);
  const int nkt_all = (len + BC - 1) / BC;
  const int nranges = nsplit * 4;                        // key ranges of this strip, in key order: range = sp * 4 + wave
  const int tpr = (nkt_all + nranges - 1) / nranges;     // key tiles per range
  const int t0 = (sp * 4 + wid) * tpr;
  const int nkt = max(0, min(nkt_all, t0 + tpr) - t0);

  const int nrec = ((S - 1) * ld + DK) * 4;
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(base + d), (short)0, nrec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(base + 2 * d), (short)0, nrec, 0x00020000);
  float* const Kw = Ks + wid * BC * DK;
  float* const Vw = Vs + wid * BC * DK;
  const int lr = lane / CPR, ls = lane % CPR;
  const int tile_step = BC * ld * 4;
  auto dma_k = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = i * RPI + lr;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr_t)&Kw[i * RPI * DK], 16, (r * ld + ((ls ^ ((r >> FSH) & FMSK)) * 4)) * 4 + (t0 + kt) * tile_step, 0, 0, 0);
    }
  };
  auto dma_v = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = i * RPI + lr;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr_t)&Vw[i * RPI * DK], 16, (r * ld + ls * 4) * 4 + (t0 + kt) * tile_step, 0, 0, 0);
    }
  };
  if (nkt > 0) { dma_k(0); dma_v(0); }

  f32x4 qreg[NG];
  {
    // all NG loads in flight together, no branch: a row past S reads row S-1 instead and is scaled by zero (it is never
    // stored).  With `if (q < S)` around each load hipcc emitted load -> s_waitcnt vmcnt(0) NG times in a row — 16 serialized
    // round trips (~10 us) at the head of every workgroup, ahead of the first K / V DMA.
    const float* qrow = base + (size_t)(q < S ? q : S - 1) * ld + 4 * h;
    const float qs = q < S ? c_scale : 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) qreg[g] = *reinterpret_cast<const f32x4*>(qrow + 8 * g);
#pragma unroll
    for (int g = 0; g < NG; ++g) qreg[g] *= qs;
  }
  f32x16 o[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  int koff[NG];
  {
    const int f = (qi >> FSH) & FMSK;
#pragma unroll
    for (int g = 0; g < NG; ++g) koff[g] = qi * DK + (((2 * g + h) ^ f) * 4);
  }
  auto row_label = [](int i) { return (i & ~7) + ((i & 3) / JP) * (2 * JP) + ((i >> 2) & 1) * JP + (i & 3) % JP; };
  const float* vp = Vw + (4 * h) * DK + NDB * row_label(qi);

  // the per-tile arithmetic below is k_attention's, statement for statement (see there for the layout argument)
  // DMA completion is waited for by hand (in-order vmcnt): ahead of the scores K(kt) must have landed while V(kt), issued after
  // it, may still fly; ahead of P V the tile V(kt) must have landed while K(kt+1), issued after it, may still fly
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(Kw + koff[g]);
#pragma unroll
      for (int e = 0; e < 4; ++e) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qreg[g][e], sc, 0, 0, 0);
    }
    if (kt + 1 < nkt) {  // K(kt) is consumed (its fragment reads returned before the MFMAs issued): refill it under the softmax
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dma_k(kt + 1);
    }
    if ((t0 + kt) * BC + BC > len) {
      const int kbase = (t0 + kt) * BC + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + (r & 3) + 8 * (r >> 2);
        sc[r] = key < len ? sc[r] : -INFINITY;
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sc[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    constexpr float RESCALE_LOG2 = 10.f;
    const bool need = mt > m_run + RESCALE_LOG2;
    const float m_new = need ? mt : m_run;
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = need ? __builtin_amdgcn_exp2f(m_run - m_use) : 1.0f;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc[r] = __builtin_amdgcn_exp2f(sc[r] - m_use);
      psum += sc[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      typedef float vrow_t __attribute__((ext_vector_type(NDB)));
      const vrow_t vv = *reinterpret_cast<const vrow_t*>(vp + ((r & 3) + 8 * (r >> 2)) * DK);
#pragma unroll
      for (int db = 0; db < NDB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[db], sc[r], o[db], 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dma_v(kt + 1);
    }
  }

  // ---- merge the four waves' partials through LDS: wave w parks (O^T, m, l), then owns row group u = w of every d block
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float l_w = l_run + __shfl_xor(l_run, 32);
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) Kw[(db * 16 + r) * 64 + lane] = o[db][r];
  Vw[lane] = m_run;
  Vw[64 + lane] = l_w;
  __syncthreads();
  float mw[4], lw[4], m_all = -INFINITY;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    mw[w] = Vs[w * BC * DK + lane];
    lw[w] = Vs[w * BC * DK + 64 + lane];
    m_all = fmaxf(m_all, mw[w]);
  }
  const float m_use = (m_all == -INFINITY) ? 0.f : m_all;
  float l_all = 0.f, wgt[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    wgt[w] = __builtin_amdgcn_exp2f(mw[w] - m_use);
    l_all += lw[w] * wgt[w];
  }
  const int u = wid;  // this wave's row group of the 32x32 C/D layout: registers r = 4u .. 4u+3 of every d block
  float mo[NDB][4];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) a += Ks[w * BC * DK + (db * 16 + 4 * u + j) * 64 + lane] * wgt[w];
      mo[db][j] = a;
    }
  // piece pj of row group u: rows j = pj*JP + e/NDB, block db = e%NDB (k_attention's store labelling)
  auto pieces = [&](float scale, float* dst) {
#pragma unroll
    for (int pj = 0; pj < NDB; ++pj) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mo[e % NDB][pj * JP + e / NDB] * scale;
      *reinterpret_cast<f32x4*>(dst + 8 * NDB * u + 8 * pj) = v;
    }
  };
  if (nsplit == 1) {
    if (q < S) pieces(1.0f / l_all, out + ((size_t)row0 + q) * d + hd * DK + 4 * h);  // lens[b]==0 -> 0 * inf = NaN, as the reference
    return;
  }
  // ---- several workgroups share this strip: publish the un-normalised partial, the last arriver merges in split order.
  // Partials travel as 16-byte write-through (sc1) stores and are read back with sc1 loads — all splits' loads in flight at
  // once — so neither side needs a cache write-back / invalidate fence.
  const size_t Mrows = pk_off ? (size_t)pk_rows : (size_t)gridDim.z * S;
  const int H = gridDim.y;
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)opart, (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsML = __builtin_amdgcn_make_buffer_rsrc((void*)mlpart, (short)0, 0x7FFFFFFF, 0x00020000);
  const int qc = q < S ? q : S - 1;  // rows past S: read / write nothing that matters (stores are predicated)
  auto orow = [&](int s2) { return (int)((((size_t)s2 * Mrows + (size_t)row0 + qc) * d + hd * DK + 4 * h + 8 * NDB * u) * 4); };
  auto mlrow = [&](int s2) { return (int)(((((size_t)s2 * Mrows + (size_t)row0 + qc) * H + hd) * 2) * 4); };
  if (q < S) {
#pragma unroll
    for (int pj = 0; pj < NDB; ++pj) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mo[e % NDB][pj * JP + e / NDB];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO, orow(sp) + 32 * pj, 0, 16 /* sc1 */);
    }
    if (u == 0 && h == 0) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{m_all, l_all}), rsML, mlrow(sp), 0, 16);
    }
  }
  if (!tickets) return;  // two-launch form (no ticket block): k_attention_merge combines the same partials in the same order
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* const last = reinterpret_cast<int*>(Vs);  // (the (m, l) words parked there were consumed before the barrier above)
  if (tid == 0)
    *last = __hip_atomic_fetch_add(tickets + ((size_t)b * H + hd) * nstrips + strip, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsplit - 1;
  __syncthreads();
  if (!*last) return;
  // every split's (m, l) and O pieces requested up front; splits past nsplit re-read split 0 with weight 0
  typedef unsigned u32x2b __attribute__((ext_vector_type(2)));
  typedef float f32x2b __attribute__((ext_vector_type(2)));
  f32x2b ml2[ATT_STRIP_SPLIT_MAX];
  f32x4 op[ATT_STRIP_SPLIT_MAX][NDB];
#pragma unroll
  for (int s2 = 0; s2 < ATT_STRIP_SPLIT_MAX; ++s2) {
    const int sc2 = s2 < nsplit ? s2 : 0;
    ml2[s2] = __builtin_bit_cast(f32x2b, __builtin_amdgcn_raw_buffer_load_b64(rsML, mlrow(sc2), 0, 16));
#pragma unroll
    for (int pj = 0; pj < NDB; ++pj) op[s2][pj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsO, orow(sc2) + 32 * pj, 0, 16));
  }
  float mx = -INFINITY;
#pragma unroll
  for (int s2 = 0; s2 < ATT_STRIP_SPLIT_MAX; ++s2) mx = fmaxf(mx, s2 < nsplit ? ml2[s2][0] : -INFINITY);
  const float mx_use = (mx == -INFINITY) ? 0.f : mx;
  f32x4 acc[NDB];
#pragma unroll
  for (int pj = 0; pj < NDB; ++pj) acc[pj] = f32x4{0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < ATT_STRIP_SPLIT_MAX; ++s2) {
    if (s2 < nsplit) {  // wave-uniform
      const float w2 = __builtin_amdgcn_exp2f(ml2[s2][0] - mx_use);
      l = __builtin_fmaf(ml2[s2][1], w2, l);
#pragma unroll
      for (int pj = 0; pj < NDB; ++pj) acc[pj] = merge_fma(op[s2][pj], w2, acc[pj]);
    }
  }
  if (q >= S) return;
  const float inv = 1.0f / l;
  float* dst = out + ((size_t)row0 + q) * d + hd * DK + 4 * h + 8 * NDB * u;
#pragma unroll
  for (int pj = 0; pj < NDB; ++pj) *reinterpret_cast<f32x4*>(dst + 8 * pj) = acc[pj] * inv;
#endif
}

// out[row, c] = sum_sp O_sp[row, c] 2^(m_sp - m) / sum_sp l_sp 2^(m_sp - m), m = max_sp m_sp (per row and head)
__global__ __launch_bounds__(256) void k_attention_merge(const float* __restrict__ opart, const float* __restrict__ mlpart, int M,
                                                          int d, int H, int dk, int nsplit, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  for (int c = lane * 4; c < d; c += 256) {
    const int hd = c / dk;
    float mx = -INFINITY;
    for (int sp = 0; sp < nsplit; ++sp) mx = fmaxf(mx, mlpart[(((size_t)sp * M + m) * H + hd) * 2]);
    const float m_use = (mx == -INFINITY) ? 0.f : mx;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {
      const size_t row = (size_t)sp * M + m;
      const float w = __builtin_amdgcn_exp2f(mlpart[(row * H + hd) * 2] - m_use);
      l = __builtin_fmaf(mlpart[(row * H + hd) * 2 + 1], w, l);
      acc = merge_fma(*reinterpret_cast<const f32x4*>(opart + row * d + c), w, acc);
    }
    const float inv = 1.0f / l;  // no valid key at all -> 0 * inf = NaN, as the reference
    *reinterpret_cast<f32x4*>(out + (size_t)m * d + c) = acc * inv;
  }
}

// Key split of a dense launch that fills the chip unevenly.  k_attention runs ONE workgroup per CU (4 waves, one per SIMD), so
// a launch's time is ceil(workgroups / 256) sweeps of its key axis: 144 workgroups (B = 9) take as long as 256, 272 (B = 17)
// twice as long.  Cutting every workgroup's key axis into n ranges gives n times the workgroups with 1/n of the sweep each,
// and the rounds fill: measured (tools/lab/attn_lab_split.hip, us, S = 1010, d_k = 128) B = 9 141 -> 108 (n = 3), B = 12
// 145 -> 124 (4), B = 17 266 -> 189 (5), B = 20 268 -> 196 (4), B = 24 272 -> 227 (2); B = 5 at S = 3900 959 -> 610 (4).
// The model that reproduces every one of those: rounds * (3.7 + c * key tiles per range) + merge, c = 3.9 us per 32-key tile
// at d_k = 128 (attn_fixed.hip), 2.1 at 64, 1.2 at 32; k_attention_merge ~4 us + its traffic.  n = 1 unless the split is
// worth 6 %.  (Exact for any n: every range leaves an exact partial softmax, merged in key order.)
static int plan_key_split(long blocks, int tiles, int dk, size_t M, int d) {
  const float c = dk == 128 ? 3.9f : dk == 64 ? 2.1f : 1.2f, f = 3.7f;
  auto rounds = [](long wgs) { return (float)((wgs + 255) / 256); };
  const float one = rounds(blocks) * (f + c * (float)tiles);
  int best = 1;
  float best_us = one * 0.94f;
  for (int n = 2; n <= ATT_SPLIT_MAX && n <= tiles; ++n) {
    const int tps = (tiles + n - 1) / n;
    if ((tiles + tps - 1) / tps != n) continue;  // (only splits without an empty range)
    const float us = rounds(blocks * n) * (f + c * (float)tps) + 4.0f + (float)((double)(n + 1) * (double)M * d * 4.0 / 12e6);
    if (us < best_us) { best = n; best_us = us; }
  }
  return best;
}

int attention_split(int B, int S, int H, int dk) {
  if (B <= 0 || S <= 0) return 1;
  const long blocks = (long)((S + 127) / 128) * H * B;
  if (blocks < ATT_SPLIT_MAX_BLOCKS) return ATT_SPLIT_MAX;  // the small-grid paths below size their own split, up to this
  if (!launch_planner_enabled()) return 1;
  return plan_key_split(blocks, (S + 31) / 32, dk, (size_t)B * S, H * dk);
}

// true when a dense launch of this shape can take the ticketed strip path (the only attention path that draws tickets): the
// caller spends attention_ticket_ints(B, S, H) of its phase's ticket block only then — the planner's key split of a LARGE launch
// (k_attention + k_attention_merge) needs the partials scratch but no tickets
// (the SAME acceptance test launch_strips applies: a launch of few workgroups whose strips would each sweep more than 4 key tiles per
//  wave — B = 3 at T ~ 2000 — takes k_attention + k_attention_merge instead, and a slice drawn for it would be burnt from the phase's
//  ticket block for nothing, pushing later GEMM + LayerNorm launches to their two-launch form; round-5 advice)
static int strip_split(long strips, int tiles) {
  int nsplit = (int)((256 + strips - 1) / strips);
  if (nsplit > (tiles + 3) / 4) nsplit = (tiles + 3) / 4;   // at least one key tile per wave
  if (nsplit > ATT_STRIP_SPLIT_MAX) nsplit = ATT_STRIP_SPLIT_MAX;
  return nsplit < 1 ? 1 : nsplit;
}
static int strip_tiles_per_range(int tiles, int nsplit) {
  const int tpr = (tiles + 4 * nsplit - 1) / (4 * nsplit);
  const int n2 = ((tiles + tpr - 1) / tpr + 3) / 4;           // no workgroup of empty ranges
  return (tiles + 4 * n2 - 1) / (4 * n2);
}
bool attention_uses_tickets(int B, int S, int H) {
  if (B <= 0 || S <= 0 || (long)((S + 127) / 128) * H * B >= ATT_SPLIT_MAX_BLOCKS) return false;
  const int tiles = (S + 31) / 32;
  // (with the scratch the caller reserves for a split; a launch that ends up unsplit for lack of scratch draws no ticket either way)
  return strip_tiles_per_range(tiles, strip_split((long)tiles * H * B, tiles)) <= 4;
}

// packed rows: the work list of att_wgs (128-query tile, head) workgroups, longest utterance (S frames) first
int attention_split_packed(int att_wgs, int S, int dk, size_t Mp, int d) {
  if (att_wgs <= 0 || S <= 0) return 1;
  const int tiles = (S + 31) / 32;
  if (!launch_planner_enabled()) {
    if (att_wgs >= ATT_SPLIT_MAX_BLOCKS * 2) return 1;
    int n = (512 + att_wgs - 1) / att_wgs;
    if (n > ATT_SPLIT_MAX) n = ATT_SPLIT_MAX;
    return n > tiles ? tiles : n;
  }
  if (att_wgs < ATT_SPLIT_MAX_BLOCKS) {  // a handful of utterances: enough ranges for ~2 workgroups per CU
    int n = (512 + att_wgs - 1) / att_wgs;
    if (n > ATT_SPLIT_MAX) n = ATT_SPLIT_MAX;
    return n > tiles ? tiles : n;
  }
  return plan_key_split(att_wgs, tiles, dk, Mp, d);
}

// k_attention with optional dispatch-attached timing events (kernels.h LaunchTiming)
template <int DK, typename... Args>
static void launch_k_attention(dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, Args... args) {
  if (e0 || e1) hipExtLaunchKernelGGL((k_attention<DK>), grid, dim3(256), 0, st, e0, e1, 0, args...);
  else hipLaunchKernelGGL((k_attention<DK>), grid, dim3(256), 0, st, args...);
}

// Few workgroups (single-utterance latency, the encoder): a 128-query workgroup's time is its serial sweep over the key tiles.
// First choice, k_attention_strip: a workgroup per 32-query strip whose four waves split the key axis, and up to nsplit such
// workgroups per strip merged by the last arriver — taken when that leaves every wave at most 4 key tiles (beyond that the shared
// K / V tiles of k_attention win: a strip's waves each pull their own).  rm != nullptr: packed rows (kernels.h RowMap; round 5 — phase 1
// of a ragged batch on packed phoneme rows used k_attention + k_attention_merge, one launch more per layer): the grid is sized
// for the longest window S, every utterance's strips address its own rows.  false = not taken (the caller's other paths).
static bool launch_strips(const float* qkv, const long long* lens, int B, int S, int H, int dk, float* out, float* scratch, size_t scratch_floats,
                          int* tickets, hipStream_t st, const RowMap* rm, hipEvent_t ev0, hipEvent_t ev1) {
  const int d = H * dk, tiles = (S + 31) / 32;
  const float c = 1.4426950408889634f / sqrtf((float)dk);
  const size_t M = rm ? (size_t)rm->rows : (size_t)B * S;
  auto part_floats = [&](int n) { return (size_t)n * (M * d + 2 * M * H); };
  const long strips = (long)tiles * H * B;
  int nsplit = (int)((256 + strips - 1) / strips);
  if (nsplit > (tiles + 3) / 4) nsplit = (tiles + 3) / 4;   // at least one key tile per wave
  if (nsplit > ATT_STRIP_SPLIT_MAX) nsplit = ATT_STRIP_SPLIT_MAX;
  if (nsplit > 1 && !scratch) nsplit = 1;
  if (part_floats(nsplit) * 4 >= (1ull << 31)) nsplit = 1;  // 31-bit descriptor offsets over the partials
  while (nsplit > 1 && part_floats(nsplit) > scratch_floats) --nsplit;
  int tpr = (tiles + 4 * nsplit - 1) / (4 * nsplit);
  nsplit = ((tiles + tpr - 1) / tpr + 3) / 4;                // no workgroup of empty ranges
  tpr = (tiles + 4 * nsplit - 1) / (4 * nsplit);
  if (tpr > 4) return false;
  float* opart = nsplit > 1 ? scratch : nullptr;
  float* mlpart = nsplit > 1 ? scratch + (size_t)nsplit * M * d : nullptr;
  const int* off = rm ? rm->off : nullptr;
  const int* win = rm ? rm->win : nullptr;
  const int rows = rm ? rm->rows : 0;
  dim3 grid(tiles * nsplit, H, B), block(256);
  if (ev0) (void)hipEventRecord(ev0, st);  // (small-grid path: plain marker events, this launch is not a roofline case)
  if (dk == 128) hipLaunchKernelGGL((k_attention_strip<128>), grid, block, 0, st, qkv, lens, S, d, c, out, nsplit, opart, mlpart, tickets, off, win, rows);
  else if (dk == 64) hipLaunchKernelGGL((k_attention_strip<64>), grid, block, 0, st, qkv, lens, S, d, c, out, nsplit, opart, mlpart, tickets, off, win, rows);
  else hipLaunchKernelGGL((k_attention_strip<32>), grid, block, 0, st, qkv, lens, S, d, c, out, nsplit, opart, mlpart, tickets, off, win, rows);
  // without a ticket block (ns_config.row_epilogue = two_launch, or the phase's block is spent) the strips' partials are
  // merged by a launch of their own: same layout, same split order, same arithmetic as the last arriver's merge
  if (nsplit > 1 && !tickets) hipLaunchKernelGGL(k_attention_merge, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, opart, mlpart, (int)M, d, H, dk, nsplit, out);
  if (ev1) (void)hipEventRecord(ev1, st);
  return true;
}

hipError_t launch_attention(const float* qkv, const long long* lens, int B, int S, int H, int dk, float* out, float* scratch,
                            size_t scratch_floats, int* tickets, hipStream_t st, const RowMap* rm, const LaunchTiming* tm) {
  if (B <= 0 || S <= 0) return hipSuccess;
  hipEvent_t ev0 = tm ? tm->start : nullptr, ev1 = tm ? tm->stop : nullptr;
  if (rm) {  // packed rows: one workgroup per (128-query tile, head) of every utterance's window, longest utterances first
    const int d = H * dk;
    if ((long long)S * 3 * d * 4 >= (1ll << 31) || (dk != 128 && dk != 64 && dk != 32) || !rm->off || !rm->win) return hipErrorInvalidValue;
    if (!rm->att_off || !rm->att_order || rm->att_wgs <= 0 || rm->rows <= 0) return hipErrorInvalidValue;
    // a handful of short windows (phase 1 of a ragged batch, L <= ~500): the strip kernel on the packed rows, no merge launch
    if ((long)((S + 127) / 128) * H * B < ATT_SPLIT_MAX_BLOCKS && (long long)rm->rows * 3 * d * 4 < (1ll << 31) &&
        launch_strips(qkv, lens, B, S, H, dk, out, scratch, scratch_floats, tickets, st, rm, ev0, ev1))
      return hipGetLastError();
    const float c = 1.4426950408889634f / sqrtf((float)dk);
    // Few workgroups (a handful of ragged utterances): the launch would last as long as the longest utterance's sweep while
    // most CUs idle.  Every workgroup's key axis is cut into nsplit ranges (of ITS utterance's key tiles), the partials are
    // merged by k_attention_merge — the split-key path of the grid, on the work list.
    const size_t Mp = (size_t)rm->rows;
    int nsplit = 1;
    if (scratch) {
      nsplit = attention_split_packed(rm->att_wgs, S, dk, Mp, d);
      while (nsplit > 1 && (size_t)nsplit * (Mp * d + 2 * Mp * H) > scratch_floats) --nsplit;
      if (nsplit < 1) nsplit = 1;
    }
    float* opart = nsplit > 1 ? scratch : nullptr;
    float* mlpart = nsplit > 1 ? scratch + (size_t)nsplit * Mp * d : nullptr;
    dim3 grid(rm->att_wgs * nsplit);
    hipEvent_t k1 = nsplit > 1 ? nullptr : ev1;  // (with a merge launch the stop event rides on the merge)
#define NS_PK rm->off, rm->win, rm->att_off, rm->att_order, B, (int)Mp, H
    if (dk == 128) launch_k_attention<128>(grid, st, ev0, k1, qkv, lens, S, d, c, out, nsplit, opart, mlpart, NS_PK);
    else if (dk == 64) launch_k_attention<64>(grid, st, ev0, k1, qkv, lens, S, d, c, out, nsplit, opart, mlpart, NS_PK);
    else launch_k_attention<32>(grid, st, ev0, k1, qkv, lens, S, d, c, out, nsplit, opart, mlpart, NS_PK);
#undef NS_PK
    if (nsplit > 1) {
      if (ev1) hipExtLaunchKernelGGL(k_attention_merge, dim3((unsigned)((Mp + 3) / 4)), dim3(256), 0, st, nullptr, ev1, 0, opart, mlpart, (int)Mp, d, H, dk, nsplit, out);
      else hipLaunchKernelGGL(k_attention_merge, dim3((unsigned)((Mp + 3) / 4)), dim3(256), 0, st, opart, mlpart, (int)Mp, d, H, dk, nsplit, out);
    }
    return hipGetLastError();
  }
  const int d = H * dk;
  if ((long long)S * 3 * d * 4 >= (1ll << 31)) return hipErrorInvalidValue;  // 31-bit descriptor offsets per utterance
  if (dk != 128 && dk != 64 && dk != 32) return hipErrorInvalidValue;
  const float c = 1.4426950408889634f / sqrtf((float)dk);
  const int qtiles = (S + 127) / 128, tiles = (S + 31) / 32;
  const long blocks = (long)qtiles * H * B;
  const size_t M = (size_t)B * S;
  auto part_floats = [&](int n) { return (size_t)n * (M * d + 2 * M * H); };
  if (blocks < ATT_SPLIT_MAX_BLOCKS && launch_strips(qkv, lens, B, S, H, dk, out, scratch, scratch_floats, tickets, st, nullptr, ev0, ev1)) return hipGetLastError();
  // Otherwise k_attention; still few workgroups (long single utterances): split the key sweep over up to ATT_SPLIT_MAX
  // workgroups per 128-query tile until the launch has ~256 of them, then merge the partials with k_attention_merge.
  // Needs nsplit * (M*d + 2*M*H) floats of scratch.
  int nsplit = 1;
  if (scratch && blocks < ATT_SPLIT_MAX_BLOCKS) {
    nsplit = (int)(256 / blocks);
    if (nsplit > ATT_SPLIT_MAX) nsplit = ATT_SPLIT_MAX;
    if (nsplit > tiles) nsplit = tiles;  // at least one 32-key tile each
    while (nsplit > 1 && part_floats(nsplit) > scratch_floats) --nsplit;
    if (nsplit < 1) nsplit = 1;
    const int tps = (tiles + nsplit - 1) / nsplit;
    nsplit = (tiles + tps - 1) / tps;  // no empty ranges: 25 tiles over 16 ranges are 13 ranges of two
  } else if (scratch && launch_planner_enabled()) {
    // a launch that fills its last round of 256 workgroups badly (plan_key_split above)
    nsplit = plan_key_split(blocks, tiles, dk, M, d);
    while (nsplit > 1 && part_floats(nsplit) > scratch_floats) --nsplit;
    const int tps = (tiles + nsplit - 1) / nsplit;
    nsplit = (tiles + tps - 1) / tps;
  }
  float* opart = nsplit > 1 ? scratch : nullptr;
  float* mlpart = nsplit > 1 ? scratch + (size_t)nsplit * M * d : nullptr;
  dim3 grid(qtiles * nsplit, H, B);
  hipEvent_t k1 = nsplit > 1 ? nullptr : ev1;
  if (dk == 128) launch_k_attention<128>(grid, st, ev0, k1, qkv, lens, S, d, c, out, nsplit, opart, mlpart, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, 0, 0, 0);
  else if (dk == 64) launch_k_attention<64>(grid, st, ev0, k1, qkv, lens, S, d, c, out, nsplit, opart, mlpart, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, 0, 0, 0);
  else launch_k_attention<32>(grid, st, ev0, k1, qkv, lens, S, d, c, out, nsplit, opart, mlpart, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, 0, 0, 0);
  if (nsplit > 1) {
    if (ev1) hipExtLaunchKernelGGL(k_attention_merge, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, nullptr, ev1, 0, opart, mlpart, (int)M, d, H, dk, nsplit, out);
    else hipLaunchKernelGGL(k_attention_merge, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, opart, mlpart, (int)M, d, H, dk, nsplit, out);
  }
  return hipGetLastError();
}

}  // namespace ns
