// Device helpers shared by the row kernels (rowops.hip) and the full-row GEMM epilogue (gemm_conv.hip): one wave64 per
// activation row, float4 lanes, wavefront shuffles for the reductions.  Both users go through the SAME functions in the
// same order, so a LayerNorm fused into a GEMM epilogue is bit-identical to the separate kernel.
#pragma once
#include "kernels.h"

namespace ns {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr float LN_EPS = 1e-5f;  // torch.nn.LayerNorm default

// torch.bucketize(v, bins, right=False) (model/modules.py:86-88,97-99), wave-cooperative: the index is the number
// of edges e with !(e >= v) — for sorted edges that is the first i with bins[i] >= v, and NaN maps to n_edges.
__device__ __forceinline__ int wave_bucketize(const float* __restrict__ bins, int n_edges, float v, int lane) {
  int cnt = 0;
  for (int k = lane; k < n_edges; k += 64) cnt += !(bins[k] >= v) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  return cnt;
}

// mean / rstd of one row held as up to NV float4 per lane (column c = lane*4 + i*256; entries at c >= C must be zero).
// Two-pass (mean, then centred sum of squares), biased variance, as nn.LayerNorm.
template <int NV>
__device__ __forceinline__ void ln_moments(const f32x4 (&v)[NV], int C, int lane, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dlt = v[i][e] - mean;
        q += dlt * dlt;
      }
    }
  }
  const float var = wave_sum(q) / (float)C;
  rstd = 1.0f / sqrtf(var + LN_EPS);
}

// y[c] = (v[c] - mean) * rstd * g[c] + b[c]
template <int NV>
__device__ __forceinline__ void ln_store(const f32x4 (&v)[NV], int C, int lane, float mean, float rstd, const float* __restrict__ g,
                                         const float* __restrict__ bta, float* __restrict__ y) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < C) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bta + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
      *reinterpret_cast<f32x4*>(y + c) = o;
    }
  }
}

// Tail of VariancePredictor.forward on one LayerNorm'ed row (model/modules.py:273-286): Linear(F->1) -> masked_fill,
// optionally followed by get_pitch_embedding / get_energy_embedding and the UNMASKED residual add of
// VarianceAdaptor.forward (model/modules.py:80-100,139-149):
//   idx = bucketize(pred*control, bins)  (right=False; NaN -> n_bins-1)
//   x_out[m,:] = x_in[m,:] + emb[idx,:]  (+ pos[t,:]: MelDecoder's position add, transformer/Models.py:222,231)
// The row feeds a DISCONTINUOUS consumer (duration rounding, torch.bucketize): layer_norm_2 and the Linear(F->1) dot are
// therefore evaluated in float64 from the fp32 row (mean, centred variance, normalisation, dot: ~1k flops per row, free next
// to the convolution that produced it), so that this tail adds no rounding of its own to the distance from the reference's
// value — what remains is the fp32 summation order of the contractions upstream (profiles/r03_bucket_edge_deviation.md).
// `mean` / `rstd` (the fp32 moments of the shared LayerNorm path) are not used here.
template <int NV>
__device__ __forceinline__ void predictor_row_tail(const f32x4 (&v)[NV], int C, int lane, float /*mean*/, float /*rstd*/, const RowEpilogue& e,
                                                   int m, int t, bool masked) {
  double s1 = 0.0;
#pragma unroll
  for (int i = 0; i < NV; ++i) s1 += ((double)v[i][0] + (double)v[i][1]) + ((double)v[i][2] + (double)v[i][3]);
  const double mu = wave_sum(s1) / (double)C;
  double s2 = 0.0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < C) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double dl = (double)v[i][k] - mu;
        s2 += dl * dl;
      }
    }
  }
  const double rs = 1.0 / sqrt(wave_sum(s2) / (double)C + (double)LN_EPS);
  double dot = 0.0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < C) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(e.ln_g + c);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(e.ln_b + c);
      const f32x4 ww = *reinterpret_cast<const f32x4*>(e.wlin + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) dot += (((double)v[i][k] - mu) * rs * (double)gg[k] + (double)bb[k]) * (double)ww[k];
    }
  }
  float pv = (float)(wave_sum(dot) + (double)e.blin[0]);
  if (masked) pv = 0.f;
  // model/modules.py:82-89: with a target the embedding comes from bucketize(target) and the prediction is returned
  // unscaled; without one prediction = prediction * control and the embedding comes from the scaled prediction
  if (e.target == nullptr) pv *= e.control;
  if (lane == 0) e.pred[m] = pv;
  if (e.emb == nullptr) return;
  const int cnt = wave_bucketize(e.bins, e.n_edges, e.target ? e.target[m] : pv, lane);
  const float* er = e.emb + (size_t)cnt * e.D;
  for (int c = lane * 4; c < e.D; c += 256) {
    f32x4 a = *reinterpret_cast<const f32x4*>(e.x_in + (size_t)m * e.D + c);
    const f32x4 e4 = *reinterpret_cast<const f32x4*>(er + c);
    a += e4;
    if (e.pos) a += *reinterpret_cast<const f32x4*>(e.pos + (size_t)t * e.D + c);
    *reinterpret_cast<f32x4*>(e.x_out + (size_t)m * e.D + c) = a;
  }
}

}  // namespace ns
