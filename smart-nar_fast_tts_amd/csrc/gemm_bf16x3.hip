// OPT-IN precision mode "bf16x3" (ns_config.matmul_bf16x3; never the default): the same Conv1D-as-GEMM contraction as
// gemm_conv.hip, computed on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate) from an EXACT
// three-way split of every fp32 operand,
//     x = x_hi + x_mid + x_lo      (each piece a bf16: 8 significant bits, 3 x 8 = the 24 bits of an fp32 mantissa;
//                                   pieces by truncation, so the identity holds bit for bit),
// keeping six of the nine partial products (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi — the three dropped ones are
// below 2^-32 of the product) and accumulating in fp32 inside the MFMA.  Every kept product is exact in fp32 (8 x 8 bits),
// so the only roundings are the accumulator's — the error against an fp64 reference is the same size as the fp32 MFMA
// kernel's own (measured: 7e-7 vs 1.5e-6 max-abs at K = 2304 on N(0,1) x N(0,0.02) operands), but the bits differ, which
// is why this is a separate, labelled mode and not the path the parity tests and the default bench run.
//
// Weights are split once at load into three bf16 planes [3][N][K] (api.hip); activations stay fp32 in HBM, are staged
// fp32 by LDS-DMA exactly like gemm_conv.hip (same zero padding / halo rule, same XOR swizzle) and are split in
// registers after the fragment read (two ANDs and two subtractions per element, hidden under the MFMAs of the other
// waves on the SIMD).
//
// Tiles (one workgroup of 16 waves per CU): 128 x 256 x 32 as 4 x 4 waves, each a 32 x 64 strip (128 KB of LDS: 2 x 16 KB
// of fp32 activations + 2 x 48 KB of weight planes), or 256 x 256 x 32 as 8 x 2 waves with 32 x 128 strips (all 160 KB)
// when the launch still has a workgroup per CU.  Measured (tools/lab/gemm_b3_lab.hip, one MI355X, random operands): the
// k=9 decoder GEMM 645 -> 324 us, PostNet 512->512 k=5 375 -> 205 us; the bf16 matrix pipe is 72 % busy while the chip
// clocks down to ~1.7 GHz under it (fp32 MFMA kernel: 87 % busy at ~2.1 GHz in the same run) — power, not issue, bound.
#include <cstring>

#include "rowln.h"

namespace ns {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr3_t;

[[maybe_unused]] constexpr int OOR3 = (int)0x80000000;

// x = hi + mid + lo with every piece exactly representable as bf16 (upper 16 bits of an fp32, lower 16 zero)
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = __float_as_uint(x) & 0xffff0000u;
  const float r = x - __uint_as_float(hi);
  mid = __float_as_uint(r) & 0xffff0000u;
  lo = __float_as_uint(r - __uint_as_float(mid));
}
// two bf16 (given as fp32 bit patterns with zero low halves) -> one dword, element 0 in the low half
__device__ __forceinline__ unsigned pack2(unsigned e0, unsigned e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

template <int BM, int BN, int WGM, int WGN, bool ROWEPI = false>
__global__ __launch_bounds__(64 * WGM * WGN) void k_conv_gemm_b3(ConvGemm p, int ntn) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BK = 32, NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN, TN = WN / 32;
  static_assert(WM == 32 && TN >= 1 && BM % 8 == 0 && BN % 16 == 0, "wave strip is 32 rows x TN 32-column tiles");
  constexpr int TOTA = BM / 8;   // DMA instructions per A chunk (8 rows of 128 B each)
  constexpr int TOTB = BN / 16;  // per weight plane (16 rows of 64 B each)
  static_assert(TOTA % NW == 0 && TOTB % NW == 0, "DMA work divides evenly over the waves (no branch around a DMA)");
  static_assert(!ROWEPI || (BM <= 64 && BN % 256 == 0 && BM % NW == 0 && 32 * BN * 4 <= 3 * BN * BK * 2),
                "row epilogue: the BM x BN fp32 tile is parked 32 rows per weight-plane buffer");
  constexpr int IA = TOTA / NW, IB = TOTB / NW;

  __shared__ __attribute__((aligned(16))) float As0[BM * BK];
  __shared__ __attribute__((aligned(16))) float As1[BM * BK];
  __shared__ __attribute__((aligned(16))) unsigned short Bs0[3 * BN * BK];
  __shared__ __attribute__((aligned(16))) unsigned short Bs1[3 * BN * BK];

  // XCD-aware bijective remap, as in gemm_conv.hip: an XCD keeps a contiguous group of activation rows in its L2
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
  const int wm0 = (wid / WGN) * WM, wn0 = (wid % WGN) * WN;
  const int Kt = p.KW * p.Cin, cpj = p.Cin / BK, nch = p.KW * cpj;
  const size_t plane = (size_t)p.N * Kt;  // elements per weight plane

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.X + ((ptrdiff_t)m0 - p.pad) * p.ldx), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wb3 + (size_t)n0 * Kt), (short)0, 0x7FFFFFFF, 0x00020000);

  // A: 8 lanes per 128-B row, source-side swizzle f(r) = (r>>1)&7 on the 16-B slot (gemm_conv.hip, BK = 32)
  // byte offset without the tap shift + the range of taps that stay inside the utterance (see gemm_conv.hip)
  int a_base[IA];
  unsigned a_jlo[IA], a_jn[IA];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int r = (wid * IA + i) * 8 + (lane >> 3), m = m0 + r;
    const int t = (m < p.M) ? (m % p.S) : -1;
    a_base[i] = (r * p.ldx + ((lane & 7) ^ ((r >> 1) & 7)) * 4) * 4;
    const int jlo = max(0, p.pad - t), jhi = min(p.KW, p.S + p.pad - t);
    a_jlo[i] = (unsigned)jlo;
    a_jn[i] = (t >= 0 && jhi > jlo) ? (unsigned)(jhi - jlo) : 0u;
  }
  // B planes: 4 lanes per 64-B row (32 bf16), swizzle f(r) = (r>>2)&3 on the 16-B slot (the 64-B-row rule of gemm_conv.hip)
  int vb[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int r = (wid * IB + i) * 16 + (lane >> 2);
    vb[i] = (n0 + r < p.N) ? (r * Kt + ((lane & 3) ^ ((r >> 2) & 3)) * 8) * 2 : OOR3;
  }
  auto dma_chunk = [&](float* As, unsigned short* Bs, int ch) {
    const int cc = ch / p.KW, j = ch - cc * p.KW;  // channel-block major, tap minor (L2 reuse of the activation lines)
    const int soA = (cc * BK + j * p.ldx) * 4;
    const int k0 = j * p.Cin + cc * BK;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int va = ((unsigned)j - a_jlo[i] < a_jn[i]) ? a_base[i] : OOR3;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr3_t)&As[(wid * IA + i) * 8 * BK], 16, va, soA, 0, 0);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const int soB = (int)((pl * plane + k0) * 2);
#pragma unroll
      for (int i = 0; i < IB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr3_t)&Bs[pl * BN * BK + (wid * IB + i) * 16 * BK], 16, vb[i], soB, 0, 0);
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

  // fragment offsets: lane (row = lane&31, h = lane>>5) of K-step t holds k = 16t + 8h + [0,8)
  const int frow = lane & 31, fh = lane >> 5;
  int aoff[2][2], boff[2];  // floats / ushorts
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ar = wm0 + frow;
#pragma unroll
    for (int s = 0; s < 2; ++s) aoff[t][s] = ar * BK + (((4 * t + 2 * fh + s) ^ ((ar >> 1) & 7)) * 4);
    boff[t] = frow * BK + (((2 * t + fh) ^ ((frow >> 2) & 3)) * 8);  // + (wn0 + ni*32) * BK: those rows keep (row>>2)&3 of frow
  }

  auto compute = [&](const float* Ac, const unsigned short* Bc) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4b x0 = *reinterpret_cast<const f32x4b*>(Ac + aoff[t][0]);
      const f32x4b x1 = *reinterpret_cast<const f32x4b*>(Ac + aoff[t][1]);
      unsigned hi[8], mid[8], lo[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        split3(x0[j], hi[j], mid[j], lo[j]);
        split3(x1[j], hi[4 + j], mid[4 + j], lo[4 + j]);
      }
      u32x4 ah, am, al;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ah[j] = pack2(hi[2 * j], hi[2 * j + 1]);
        am[j] = pack2(mid[2 * j], mid[2 * j + 1]);
        al[j] = pack2(lo[2 * j], lo[2 * j + 1]);
      }
      const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am), Al = __builtin_bit_cast(bf16x8, al);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const unsigned short* bp = Bc + (wn0 + ni * 32) * BK + boff[t];
        const bf16x8 Bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp));
        const bf16x8 Bm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + BN * BK));
        const bf16x8 Bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + 2 * BN * BK));
        // smallest terms first
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[ni], 0, 0, 0);
      }
    }
  };

  dma_chunk(As0, Bs0, 0);
  __syncthreads();
  auto step = [&](int ch, const float* Ac, const unsigned short* Bc, float* An, unsigned short* Bn) {
    if (ch + 1 < nch) dma_chunk(An, Bn, ch + 1);
    compute(Ac, Bc);
    __syncthreads();
  };
  for (int ch = 0; ch < nch; ch += 2) {
    step(ch, As0, Bs0, As1, Bs1);
    if (ch + 1 < nch) step(ch + 1, As1, Bs1, As0, Bs0);
  }

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); bias / act / residual as gemm_conv.hip
  const int ecol = lane & 31, erow = (lane >> 5) * 4;
  if constexpr (ROWEPI) {
    // full-row tile (BN == N): the same row epilogue as gemm_conv.hip's (rowln.h) — LayerNorm (+ mask) of act(acc + bias) +
    // residual, one row per wave64; rows [0,32) of the tile are parked in Bs0, [32,64) in Bs1
    auto trow = [&](int ml) -> float* { return reinterpret_cast<float*>(ml < 32 ? Bs0 : Bs1) + (ml & 31) * BN; };
    // residual rows of this wave, all loads in flight before the tile is parked (see gemm_conv.hip)
    constexpr int NV = BN / 256, RPW = BM / NW;
    f32x4 rv[RPW][NV];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int m = m0 + wid * RPW + rr;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        rv[rr][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.resid && m < p.M) rv[rr][i] = *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.ldr + lane * 4 + i * 256);
      }
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int nl = wn0 + ni * 32 + ecol;
      const float bv = p.bias ? p.bias[n0 + nl] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wm0 + (r & 3) + 8 * (r >> 2) + erow;
        float v = acc[ni][r] + bv;
        if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == ACT_TANH) v = tanhf(v);
        trow(ml)[nl] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int ml = wid * RPW + rr, m = m0 + ml;
      if (m < p.M) {
        const int b = m / p.S, t = m - b * p.S;
        const bool masked = p.e.lens && (long long)t >= p.e.lens[b];
        if (masked) {
#pragma unroll
          for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(p.Y + (size_t)m * p.ldy + lane * 4 + i * 256) = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
          f32x4 v[NV];
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            v[i] = *reinterpret_cast<const f32x4*>(trow(ml) + lane * 4 + i * 256);
            if (p.resid) v[i] += rv[rr][i];
          }
          float mean, rstd;
          ln_moments<NV>(v, BN, lane, mean, rstd);
          ln_store<NV>(v, BN, lane, mean, rstd, p.e.ln_g, p.e.ln_b, p.Y + (size_t)m * p.ldy);
        }
      }
    }
  } else {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int n = n0 + wn0 + ni * 32 + ecol;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
      float rs[16];  // residual values first, all loads in flight together (see gemm_conv.hip)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + erow;
        rs[r] = (p.resid && m < p.M) ? p.resid[(size_t)m * p.ldr + n] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + erow;
        if (m >= p.M) continue;
        float v = acc[ni][r] + bv;
        if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == ACT_TANH) v = tanhf(v);
        if (p.resid) v += rs[r];
        p.Y[(size_t)m * p.ldy + n] = v;
      }
    }
  }
#endif
}

bool conv_gemm_b3_ok(int M, int N, int Cin, int KW, int epi) {
  if (Cin % 32 != 0 || N % 16 != 0 || 3ll * N * KW * Cin * 2 >= (1ll << 31)) return false;
  // LayerNorm epilogue: the 64 x 256 full-row tile, one workgroup per CU; plain: 128-row tiles, enough of them to fill the chip
  if (epi == EPI_LN) return N == 256 && (M + 63) / 64 >= 200;
  return epi == EPI_NONE && (long long)((M + 127) / 128) * ((N + 255) / 256) >= 200;
}

hipError_t launch_conv_gemm_b3(const ConvGemm& p, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if (!p.Wb3 || !conv_gemm_b3_ok(p.M, p.N, p.Cin, p.KW, p.epi) || (p.ldx & 3)) return hipErrorInvalidValue;
  if ((long long)(256 + p.KW) * p.ldx >= (1ll << 29)) return hipErrorInvalidValue;
  if (p.epi == EPI_LN) {
    if ((p.ldy & 3) || (p.resid && (p.ldr & 3))) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_conv_gemm_b3<64, 256, 2, 4, true>), dim3((p.M + 63) / 64), dim3(512), 0, st, p, 1);
    return hipGetLastError();
  }
  const int ntn = (p.N + 255) / 256;
  // 256-row tiles halve the weight-plane traffic per flop (k=9 decoder GEMM: 324 vs 355 us) but need a workgroup per CU
  const int ntm256 = (p.M + 255) / 256, ntm128 = (p.M + 127) / 128;
  if ((long long)ntm256 * ntn >= 240)
    hipLaunchKernelGGL((k_conv_gemm_b3<256, 256, 8, 2>), dim3(ntm256 * ntn), dim3(1024), 0, st, p, ntn);
  else
    hipLaunchKernelGGL((k_conv_gemm_b3<128, 256, 4, 4>), dim3(ntm128 * ntn), dim3(1024), 0, st, p, ntn);
  return hipGetLastError();
}

// host-side split of an fp32 weight matrix [n] into three bf16 planes (same truncation split as the device's split3)
void split_weights_b3(const float* w, size_t n, unsigned short* hi, unsigned short* mid, unsigned short* lo) {
  for (size_t i = 0; i < n; ++i) {
    unsigned u;
    memcpy(&u, &w[i], 4);
    const unsigned h = u & 0xffff0000u;
    float hf;
    memcpy(&hf, &h, 4);
    const float r = w[i] - hf;
    unsigned ru;
    memcpy(&ru, &r, 4);
    const unsigned m = ru & 0xffff0000u;
    float mf;
    memcpy(&mf, &m, 4);
    const float l = r - mf;
    unsigned lu;
    memcpy(&lu, &l, 4);
    hi[i] = (unsigned short)(h >> 16);
    mid[i] = (unsigned short)(m >> 16);
    lo[i] = (unsigned short)(lu >> 16);
  }
}

}  // namespace ns
