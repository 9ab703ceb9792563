// Fused multi-head self attention on the gfx950 fp32 matrix cores — replaces
// MultiHeadAttention's head split + ScaledDotProductAttention (transformer/SubLayers.py:42-54,
// transformer/Modules.py:14-25): bmm -> /sqrt(d_k) -> masked_fill(-inf at padded KEYS) -> softmax -> bmm.
// The (H*B, S, S) score matrix is never materialised and the attention probabilities (a dead output at
// inference) are not produced.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries for the whole
// key sweep.  K/V tiles of 32 keys are staged through LDS (double buffered, register-staged prefetch).
//
// The trick that removes every P re-layout: compute the TRANSPOSED score tile S^T = K Q^T
// (A = K tile, B = Q^T held in registers for the whole kernel).  In the 32x32 C/D layout each lane then
// holds, for ONE query (col = lane&31), 16 of the 32 keys: key(r,h) = (r&3) + 8*(r>>2) + 4*h.
//   * the softmax row reduction is 16 in-lane values + one cross-half shuffle;
//   * the online-softmax rescale of O^T (same column = same query) is lane-local;
//   * for O^T += V^T P^T the B operand of MFMA step r is literally register p[r]: MFMA's k index is free to
//     be permuted as long as A and B agree, so step r / lane-half h is *defined* to be key(r,h) and the
//     A operand is V[key(r,h)][d], one conflict-free ds_read_b32 per MFMA.
// Keys >= lens[b] get -inf before the softmax; key tiles entirely past lens[b] contribute exact zeros and
// are skipped.  Padded QUERY rows are computed like the reference does.  An utterance with lens[b] == 0
// yields NaN rows (0/0), as the reference's all -inf softmax does (SURVEY.md §8b "Errors").
#include "kernels.h"

namespace ns {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DK>
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, const long long* __restrict__ lens,
                                                    int S, int d, float c_scale, float* __restrict__ out) {
  constexpr int BC = 32;              // keys per tile
  constexpr int KS = DK + 4;          // K tile row stride (conflict-free ds_read_b128)
  constexpr int TPR = DK / 4;         // float4 lanes per tile row
  constexpr int RPP = 256 / TPR;      // rows per pass
  constexpr int NP = BC / RPP;        // passes per tile
  constexpr int NG = DK / 8;          // k-groups of the QK^T contraction
  constexpr int NDB = DK / 32;        // 32-wide d blocks of O^T

  __shared__ __attribute__((aligned(16))) float Ks[2][BC * KS];
  __shared__ __attribute__((aligned(16))) float Vs[2][BC * DK];

  const int b = blockIdx.z, hd = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, qi = lane & 31;
  const int q = blockIdx.x * 128 + wid * 32 + qi;
  const int ld = 3 * d;
  const float* base = qkv + (size_t)b * S * ld + hd * DK;
  long long len_ll = lens ? lens[b] : (long long)S;
  const int len = (int)(len_ll < S ? len_ll : S);
  const int nkt = (len + BC - 1) / BC;

  // Q^T operand: lane (q, h) keeps Q[q][8g + 4h + e]
  f32x4 qreg[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (q < S) qreg[g] = *reinterpret_cast<const f32x4*>(base + (size_t)q * ld + 8 * g + 4 * h);
    else qreg[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x16 o[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int lrow = tid / TPR, lcol = (tid % TPR) * 4;
  f32x4 rk[NP], rv[NP];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int key = kt * BC + lrow + i * RPP;
      if (key < S) {
        const float* src = base + (size_t)key * ld + lcol;
        rk[i] = *reinterpret_cast<const f32x4*>(src + d);
        rv[i] = *reinterpret_cast<const f32x4*>(src + 2 * d);
      } else {
        rk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      *reinterpret_cast<f32x4*>(&Ks[buf][(lrow + i * RPP) * KS + lcol]) = rk[i];
      *reinterpret_cast<f32x4*>(&Vs[buf][(lrow + i * RPP) * DK + lcol]) = rv[i];
    }
  };

  if (nkt > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);

    // S^T[key][q] = sum_d K[key][d] Q[q][d]
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* kp = &Ks[buf][qi * KS + 4 * h];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qreg[g][e], s, 0, 0, 0);
    }

    // scale (log2 domain), key-padding mask, online softmax
    const int kbase = kt * BC + 4 * h;
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kbase + (r & 3) + 8 * (r >> 2);
      s[r] = key < len ? s[r] * c_scale : -INFINITY;
      mt = fmaxf(mt, s[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_use);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = exp2f(s[r] - m_use);
      psum += s[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;

    // O^T[d][q] += sum_key V[key][d] P[q][key]   (MFMA step r <-> key(r,h), B operand = s[r])
    const float* vp = &Vs[buf][(4 * h) * DK + qi];
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = vp[((r & 3) + 8 * (r >> 2)) * DK + 32 * db];
        o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, s[r], o[db], 0, 0, 0);
      }
    }

    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;   // lens[b]==0 -> 0 * inf = NaN, as the reference
  if (q < S) {
    float* dst = out + ((size_t)b * S + q) * d + hd * DK + 4 * h;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 v = {o[db][4 * u] * inv, o[db][4 * u + 1] * inv, o[db][4 * u + 2] * inv, o[db][4 * u + 3] * inv};
        *reinterpret_cast<f32x4*>(dst + 32 * db + 8 * u) = v;
      }
  }
}

hipError_t launch_attention(const float* qkv, const long long* lens, int B, int S, int H, int dk, float* out, hipStream_t st) {
  if (B <= 0 || S <= 0) return hipSuccess;
  const int d = H * dk;
  const float c = 1.4426950408889634f / sqrtf((float)dk);
  dim3 grid((S + 127) / 128, H, B), block(256);
  if (dk == 128) hipLaunchKernelGGL((k_attention<128>), grid, block, 0, st, qkv, lens, S, d, c, out);
  else if (dk == 64) hipLaunchKernelGGL((k_attention<64>), grid, block, 0, st, qkv, lens, S, d, c, out);
  else if (dk == 32) hipLaunchKernelGGL((k_attention<32>), grid, block, 0, st, qkv, lens, S, d, c, out);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace ns
