// Row-wise (HBM-bound) kernels of the path: one wave64 per activation row, float4 (16 B/lane) accesses,
// wavefront shuffles for the reductions.  Each one cites the reference lines it restates.
#include "rowln.h"

namespace ns {

// The first kernel of a forward phase also zeroes that phase's ticket counters (gemm_conv.hip TICKET, attention.hip): the
// launches that draw tickets come later on the same stream, so the kernel boundary orders the zeroes before them, and no
// memset node / extra launch is needed.
__device__ __forceinline__ void zero_words(int* __restrict__ z, int n) {
  if (!z) return;
  const int nthr = gridDim.x * gridDim.y * blockDim.x;
  for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += nthr) z[i] = 0;
}

// Loads row `x` (C floats, C % 4 == 0, C <= 1024) as up to NV float4 per lane and returns mean / rstd.
template <int NV>
__device__ __forceinline__ void ln_stats(const float* x, int C, int lane, f32x4 (&v)[NV], float& mean, float& rstd) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < C) v[i] = *reinterpret_cast<const f32x4*>(x + c);
    else v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  ln_moments<NV>(v, C, lane, mean, rstd);
}

// y = LN(x)*g + b ; rows at t >= lens[b] -> 0.  Covers `layer_norm(output + residual)` followed by
// FFTBlock's masked_fill (transformer/SubLayers.py:57,93 + transformer/Layers.py:43,46) — the residual add is
// done by the producing GEMM's epilogue — and the predictors' layer_norm_1 (model/modules.py:260).
template <int NV>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ g,
                                                    const float* __restrict__ bta, float* __restrict__ y, int M, int C,
                                                    int S, const long long* __restrict__ lens, const int* __restrict__ row_b,
                                                    const int* __restrict__ row_t) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int bb = row_b ? row_b[m] : m / S, tt = row_t ? row_t[m] : m % S;  // (packed rows: kernels.h RowMap)
  if (lens && (long long)tt >= lens[bb]) {  // masked_fill(mask, 0): a padded row is written, never read
    for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<f32x4*>(y + (size_t)m * C + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  f32x4 v[NV];
  float mean, rstd;
  ln_stats<NV>(x + (size_t)m * C, C, lane, v, mean, rstd);
  ln_store<NV>(v, C, lane, mean, rstd, g, bta, y + (size_t)m * C);
}

hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, int M, int C, int S,
                            const long long* lens, hipStream_t st, const RowMap* rm) {
  if (M <= 0) return hipSuccess;
  if (C % 4 != 0 || C > 1024) return hipErrorInvalidValue;
  dim3 grid((M + 3) / 4), block(256);
  const int* rb = rm ? rm->row_b : nullptr;
  const int* rt = rm ? rm->row_t : nullptr;
  if (C <= 256) hipLaunchKernelGGL((k_layernorm<1>), grid, block, 0, st, x, g, b, y, M, C, S, lens, rb, rt);
  else if (C <= 512) hipLaunchKernelGGL((k_layernorm<2>), grid, block, 0, st, x, g, b, y, M, C, S, lens, rb, rt);
  else hipLaunchKernelGGL((k_layernorm<4>), grid, block, 0, st, x, g, b, y, M, C, S, lens, rb, rt);
  return hipGetLastError();
}

// Tail of VariancePredictor.forward (model/modules.py:273-286): layer_norm_2 -> Linear(F->1) -> squeeze ->
// masked_fill(mask, 0).  Optionally fused with get_pitch_embedding / get_energy_embedding and the UNMASKED
// residual add of VarianceAdaptor.forward (model/modules.py:80-100,139-149):
//   idx = bucketize(pred*control, bins)  (right=False: number of bin edges e with !(e >= v); NaN -> n_bins-1)
//   x_out[m,:] = x_in[m,:] + emb[idx,:]  (+ pos[t,:]: MelDecoder's position add, transformer/Models.py:222,231)
template <int NV>
__global__ __launch_bounds__(256) void k_ln_linear_embed(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ bta, const float* __restrict__ wlin,
                                                          const float* __restrict__ blin, float* __restrict__ pred, int M,
                                                          int C, int S, const long long* __restrict__ lens, float control,
                                                          const float* __restrict__ target,
                                                          const float* __restrict__ bins, int n_edges,
                                                          const float* __restrict__ emb, const float* __restrict__ x_in,
                                                          const float* __restrict__ pos, float* __restrict__ x_out, int D,
                                                          const int* __restrict__ row_b, const int* __restrict__ row_t) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  f32x4 v[NV];
  float mean, rstd;
  ln_stats<NV>(x + (size_t)m * C, C, lane, v, mean, rstd);
  RowEpilogue e;
  e.y_out = nullptr; e.ticket = nullptr; e.row_b = nullptr; e.row_t = nullptr;  // (unused by predictor_row_tail)
  e.ln_g = g; e.ln_b = bta; e.lens = lens; e.wlin = wlin; e.blin = blin; e.pred = pred; e.control = control; e.target = target;
  e.bins = bins; e.n_edges = n_edges; e.emb = emb; e.x_in = x_in; e.pos = pos; e.x_out = x_out; e.D = D;
  const int bb = row_b ? row_b[m] : m / S, t = row_t ? row_t[m] : m % S;  // (packed rows: kernels.h RowMap)
  predictor_row_tail<NV>(v, C, lane, mean, rstd, e, m, t, lens && (long long)t >= lens[bb]);
}

hipError_t launch_ln_linear_embed(const float* x, const float* g, const float* b, const float* wlin, const float* blin,
                                  float* pred, int M, int C, int S, const long long* lens, float control,
                                  const float* target, const float* bins, int n_bins, const float* emb, const float* x_in, const float* pos,
                                  float* x_out, int D, hipStream_t st, const RowMap* rm) {
  if (M <= 0) return hipSuccess;
  if (C % 4 != 0 || C > 1024 || (emb && D % 4 != 0)) return hipErrorInvalidValue;
  dim3 grid((M + 3) / 4), block(256);
  const int n_edges = n_bins - 1;
  const int* rb = rm ? rm->row_b : nullptr;
  const int* rt = rm ? rm->row_t : nullptr;
#define NS_ARGS x, g, b, wlin, blin, pred, M, C, S, lens, control, target, bins, n_edges, emb, x_in, pos, x_out, D, rb, rt
  if (C <= 256) hipLaunchKernelGGL((k_ln_linear_embed<1>), grid, block, 0, st, NS_ARGS);
  else if (C <= 512) hipLaunchKernelGGL((k_ln_linear_embed<2>), grid, block, 0, st, NS_ARGS);
  else hipLaunchKernelGGL((k_ln_linear_embed<4>), grid, block, 0, st, NS_ARGS);
#undef NS_ARGS
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_bucketize(const float* __restrict__ v, int n, const float* __restrict__ bins,
                                                    int n_edges, long long* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int c = wave_bucketize(bins, n_edges, v[i], lane);
  if (lane == 0) idx[i] = c;
}
hipError_t launch_bucketize(const float* v, int n, const float* bins, int n_edges, long long* idx, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_bucketize, dim3((n + 3) / 4), dim3(256), 0, st, v, n, bins, n_edges, idx);
  return hipGetLastError();
}

// TxtEncoder.forward input: src_word_emb(src_seq) + position_enc[:, :L] (transformer/Models.py:82-91).
// A token id outside [0, n_vocab) (nn.Embedding raises IndexError) reads row 0 here and is reported by the duration tail.
__global__ __launch_bounds__(256) void k_embed_pos(const long long* __restrict__ texts, const float* __restrict__ emb,
                                                    const float* __restrict__ pos, float* __restrict__ out, int M, int S, int D,
                                                    int n_vocab, int* __restrict__ zero, int nzero) {
  zero_words(zero, nzero);
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  long long tok = texts[m];
  if (tok < 0 || tok >= (long long)n_vocab) tok = 0;
  const int t = m % S;
  for (int c = lane * 4; c < D; c += 256) {
    f32x4 a = *reinterpret_cast<const f32x4*>(emb + (size_t)tok * D + c);
    a += *reinterpret_cast<const f32x4*>(pos + (size_t)t * D + c);
    *reinterpret_cast<f32x4*>(out + (size_t)m * D + c) = a;
  }
}
hipError_t launch_embed_pos(const long long* texts, const float* emb, const float* pos, float* out, int M, int S, int D, int n_vocab,
                            int* zero, int nzero, hipStream_t st) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_embed_pos, dim3((M + 3) / 4), dim3(256), 0, st, texts, emb, pos, out, M, S, D, n_vocab, zero, nzero);
  return hipGetLastError();
}

// Host values -> a device vector WITHOUT a copy command: up to 128 int64 ride in the kernel's argument block (1 KB), the
// kernel stores them.  A pinned-staging hipMemcpyAsync for 128 bytes costs a forward ~35 us (blit + its stream dependency), a
// pageable torch .to(device) ~80 us; this is one ~3 us launch at the head of the phase that needs the lengths on the device.
struct LensBlock { long long v[128]; };
__global__ void k_store_lens(LensBlock blk, int n, long long* __restrict__ dst) {
  const int i = threadIdx.x;
  if (i < n) dst[i] = blk.v[i];
}
hipError_t launch_store_lens(const long long* host, int n, long long* dst, hipStream_t st) {
  for (int o = 0; o < n; o += 128) {
    LensBlock blk;
    const int k = n - o < 128 ? n - o : 128;
    for (int i = 0; i < k; ++i) blk.v[i] = host[o + i];
    hipLaunchKernelGGL(k_store_lens, dim3(1), dim3(128), 0, st, blk, k, dst + o);
  }
  return hipGetLastError();
}

static void plan_pointers(int* plan, int B, int Mp, RowMap* rm);
// The same on packed phoneme rows (kernels.h RowMap, api.hip forward_durations): row m is phoneme row_t[m] of utterance row_b[m].
// Also zeroes the phase's ticket counters (the plan kernels ahead of it do not).
__global__ __launch_bounds__(256) void k_embed_pos_packed(const long long* __restrict__ texts, const float* __restrict__ emb,
                                                           const float* __restrict__ pos, float* __restrict__ out, int Mp, int L, int D,
                                                           int n_vocab, int B, const int* __restrict__ off, const int* __restrict__ win,
                                                           int* __restrict__ row_b, int* __restrict__ row_t, int* __restrict__ row_w,
                                                           int* __restrict__ zero, int nzero) {
  zero_words(zero, nzero);
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= Mp) return;
  int lo = 0, hi = B - 1;  // largest b with off[b] <= m; this kernel also writes the row maps every later kernel reads
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= m) lo = mid;
    else hi = mid - 1;
  }
  const int t = m - off[lo];
  if (lane == 0) { row_b[m] = lo; row_t[m] = t; row_w[m] = win[lo]; }
  long long tok = texts[(size_t)lo * L + t];
  if (tok < 0 || tok >= (long long)n_vocab) tok = 0;
  for (int c = lane * 4; c < D; c += 256) {
    f32x4 a = *reinterpret_cast<const f32x4*>(emb + (size_t)tok * D + c);
    a += *reinterpret_cast<const f32x4*>(pos + (size_t)t * D + c);
    *reinterpret_cast<f32x4*>(out + (size_t)m * D + c) = a;
  }
}
hipError_t launch_embed_pos_packed(const long long* texts, const float* emb, const float* pos, float* out, const RowMap& rm, int B, int Mp, int L,
                                   int D, int n_vocab, int* zero, int nzero, hipStream_t st) {
  if (Mp <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_embed_pos_packed, dim3((Mp + 3) / 4), dim3(256), 0, st, texts, emb, pos, out, Mp, L, D, n_vocab, B, rm.off, rm.win,
                     const_cast<int*>(rm.row_b), const_cast<int*>(rm.row_t), const_cast<int*>(rm.row_w), zero, nzero);
  return hipGetLastError();
}

// dense [B*S, D] rows from packed rows: row (b, t) = the packed row off[b] + t when t < min(lens[b], win[b]), zeros otherwise
// (what the reference's masked_fill leaves on a padded phoneme: transformer/Layers.py:43,46, model/modules.py:283-284)
__global__ __launch_bounds__(256) void k_unpack_rows(const int* __restrict__ off, const int* __restrict__ win, const long long* __restrict__ lens,
                                                      int B, int S, int D, const float* __restrict__ src, float* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= B * S) return;
  const int b = m / S, t = m - b * S;
  long long len = lens ? lens[b] : (long long)S;
  const bool live = t < win[b] && (long long)t < len;
  const float* s = src + ((size_t)off[b] + t) * D;
  float* d = dst + (size_t)m * D;
  if ((D & 3) == 0) {
    for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<f32x4*>(d + c) = live ? *reinterpret_cast<const f32x4*>(s + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    for (int c = lane; c < D; c += 64) d[c] = live ? s[c] : 0.f;
  }
}
hipError_t launch_unpack_rows(const RowMap& rm, const long long* lens, int B, int S, int D, const float* src, float* dst, hipStream_t st) {
  if (B <= 0 || S <= 0 || D <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_unpack_rows, dim3((B * S + 3) / 4), dim3(256), 0, st, rm.off, rm.win, lens, B, S, D, src, dst);
  return hipGetLastError();
}

// phase 1's two padded tensors in one launch: rows [B*S, D] (zeros past a window) and a vector [B*S] (zeros at t >= lens[b])
__global__ __launch_bounds__(256) void k_unpack_phase1(const int* __restrict__ off, const int* __restrict__ win, const long long* __restrict__ lens,
                                                        int B, int S, int D, const float* __restrict__ rows_p, float* __restrict__ rows,
                                                        const float* __restrict__ vec_p, float* __restrict__ vec) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= B * S) return;
  const int b = m / S, t = m - b * S;
  const bool in_win = t < win[b];
  const size_t src = (size_t)off[b] + t;
  if (lane == 0) vec[m] = (in_win && (long long)t < lens[b]) ? vec_p[src] : 0.f;
  for (int c = lane * 4; c < D; c += 256)
    *reinterpret_cast<f32x4*>(rows + (size_t)m * D + c) = in_win ? *reinterpret_cast<const f32x4*>(rows_p + src * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
}
hipError_t launch_unpack_phase1(const RowMap& rm, const long long* lens, int B, int S, int D, const float* rows_p, float* rows, const float* vec_p,
                                float* vec, hipStream_t st) {
  if (B <= 0 || S <= 0 || D <= 0 || (D & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_unpack_phase1, dim3((B * S + 3) / 4), dim3(256), 0, st, rm.off, rm.win, lens, B, S, D, rows_p, rows, vec_p, vec);
  return hipGetLastError();
}

// MelDecoder.forward input: enc_seq + position table (transformer/Models.py:218-235).
__global__ __launch_bounds__(256) void k_add_pos(const float* __restrict__ x, const float* __restrict__ pos,
                                                  float* __restrict__ out, int M, int S, int D, const int* __restrict__ row_t) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int t = row_t ? row_t[m] : m % S;
  for (int c = lane * 4; c < D; c += 256) {
    f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)m * D + c);
    a += *reinterpret_cast<const f32x4*>(pos + (size_t)t * D + c);
    *reinterpret_cast<f32x4*>(out + (size_t)m * D + c) = a;
  }
}
hipError_t launch_add_pos(const float* x, const float* pos, float* out, int M, int S, int D, hipStream_t st, const RowMap* rm) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_add_pos, dim3((M + 3) / 4), dim3(256), 0, st, x, pos, out, M, S, D, rm ? rm->row_t : nullptr);
  return hipGetLastError();
}

// get_mask_from_lengths (utils/tools.py:89-97): mask[b,t] = t >= lens[b]; 1 = padding.
__global__ void k_mask(const long long* __restrict__ lens, int B, int max_len, uint8_t* __restrict__ mask) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * max_len) return;
  const int b = (int)(i / max_len), t = (int)(i % max_len);
  mask[i] = (long long)t >= lens[b] ? 1 : 0;
}
hipError_t launch_mask_from_lengths(const long long* lens, int B, int max_len, uint8_t* mask, hipStream_t st) {
  const long long n = (long long)B * max_len;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lens, B, max_len, mask);
  return hipGetLastError();
}

// get_sinusoid_encoding_table (transformer/Models.py:10-30): the angle and sin/cos are evaluated in float64
// and only then cast to float32, exactly as numpy does in the reference.
__global__ void k_sinusoid(int n_pos, int d, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_pos * d) return;
  const int p = (int)(i / d), j = (int)(i % d);
  const double ang = (double)p / pow(10000.0, (double)(2 * (j / 2)) / (double)d);
  out[i] = (float)((j & 1) ? cos(ang) : sin(ang));
}
hipError_t launch_sinusoid(int n_pos, int d, float* out, hipStream_t st) {
  const long long n = (long long)n_pos * d;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_sinusoid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_pos, d, out);
  return hipGetLastError();
}

// duration_rounded = clamp(round(exp(log_d) - 1) * d_control, min=0) (model/modules.py:132-135).
// torch.round is round-half-to-even -> rintf; the clamp keeps -0.0 and NaN like torch.clamp does.
__device__ __forceinline__ float duration_round(float log_d, float d_control) {
  const float r = rintf(expf(log_d) - 1.0f) * d_control;
  return (r < 0.f) ? 0.f : r;
}
__global__ void k_duration_round(const float* __restrict__ log_d, int n, float d_control, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = duration_round(log_d[i], d_control);
}
hipError_t launch_duration_round(const float* log_d, int n, float d_control, float* d_rounded, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_duration_round, dim3((n + 255) / 256), dim3(256), 0, st, log_d, n, d_control, d_rounded);
  return hipGetLastError();
}

// One body, two instantiations.  One workgroup per utterance walks its L phonemes 256 at a time:
//   TAIL = false  (ns_op_duration_scan): LengthRegulator.expand's repeat counts (model/modules.py:221-223), max(int(d), 0)
//                 with int() truncating toward zero, their inclusive prefix sums, mel_len[b] = total (:209-211);
//   TAIL = true   (the forward's phase-1 tail, ONE launch): additionally produces its own input — the rounded durations
//                 from log_d (two copies: the caller's output and the workspace copy phase 2 reads) — and the source
//                 mask (utils/tools.py:89-97), and reports a token id outside [0, n_vocab) as mel_len[b] = -1
//                 (nn.Embedding raises IndexError there; the host raises it after its one read of mel_lens).
template <bool TAIL>
__global__ __launch_bounds__(256) void k_duration_scan(const float* __restrict__ in, int L, int32_t* __restrict__ cum,
                                                        long long* __restrict__ mel_lens, const long long* __restrict__ src_lens,
                                                        float d_control, float* __restrict__ d_rounded, float* __restrict__ d_keep,
                                                        uint8_t* __restrict__ src_mask, const long long* __restrict__ texts,
                                                        int n_vocab, long long* __restrict__ mel_lens_host) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  __shared__ int bad_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  long long len = 0;
  if constexpr (TAIL) len = src_lens[b];
  if (tid == 0) { carry_s = 0; bad_s = 0; }
  __syncthreads();
  for (int l0 = 0; l0 < L; l0 += 256) {
    const int l = l0 + tid;
    int v = 0;
    if (l < L) {
      const size_t i = (size_t)b * L + l;
      float dr;
      if constexpr (TAIL) {
        dr = duration_round(in[i], d_control);
        d_rounded[i] = dr;
        d_keep[i] = dr;
        src_mask[i] = (long long)l >= len ? 1 : 0;
        if (texts) {
          const long long tok = texts[i];
          if (tok < 0 || tok >= (long long)n_vocab) bad_s = 1;  // benign race: every writer stores 1
        }
      } else {
        dr = in[i];
      }
      const int ri = (int)dr;
      v = ri > 0 ? ri : 0;
    }
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(inc, o);
      if (lane >= o) inc += n;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int off = carry_s;
    for (int w = 0; w < wid; ++w) off += wsum[w];
    if (l < L) cum[(size_t)b * L + l] = inc + off;
    __syncthreads();
    if (tid == 255) carry_s = inc + off;
    __syncthreads();
  }
  if (tid == 0) {
    const long long n = bad_s ? -1ll : (long long)carry_s;
    mel_lens[b] = n;
    // optional second copy straight into device-visible (pinned) HOST memory: the caller's one read of mel_lens then needs
    // a stream synchronisation only, no device-to-host copy behind the kernel
    if (mel_lens_host) mel_lens_host[b] = n;
  }
}
hipError_t launch_duration_scan(const float* d_rounded, int B, int L, int32_t* cum, long long* mel_lens, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_duration_scan<false>), dim3(B), dim3(256), 0, st, d_rounded, L, cum, mel_lens, nullptr, 1.0f, nullptr, nullptr,
                     nullptr, nullptr, 0, nullptr);
  return hipGetLastError();
}
hipError_t launch_duration_tail(const float* log_d, const long long* src_lens, const long long* texts, int n_vocab, int B, int L,
                                float d_control, float* d_rounded, float* d_keep, int32_t* cum, long long* mel_lens, uint8_t* src_mask,
                                long long* mel_lens_host, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_duration_scan<true>), dim3(B), dim3(256), 0, st, log_d, L, cum, mel_lens, src_lens, d_control, d_rounded, d_keep,
                     src_mask, texts, n_vocab, mel_lens_host);
  return hipGetLastError();
}

// ns_forward_mel's per-utterance status word (include/nar_fs2.h NS_STATUS_*): bit 0 = the utterance is longer than the T the
// caller chose (its frames past T are cut off), bit 1 = phase 1 flagged a token id outside the vocabulary (mel_lens[b] = -1)
__device__ __forceinline__ int32_t forward_status(long long total, int T, long long mel_len) {
  return (total > (long long)T ? 1 : 0) | (mel_len < 0 ? 2 : 0);
}

// LengthRegulator.LR + pad (model/modules.py:201-218, utils/tools.py:288-306) as a gather: output frame t of
// utterance b copies encoder row i = first index with cum[b][i] > t; frames at t >= mel_len[b] are zero.
__global__ __launch_bounds__(256) void k_length_regulate(const float* __restrict__ x, const int32_t* __restrict__ cum, int L,
                                                          int D, int T, int M, float* __restrict__ out,
                                                          uint8_t* __restrict__ mel_mask, const long long* __restrict__ mel_lens,
                                                          int32_t* __restrict__ status, int* __restrict__ zero, int nzero) {
  zero_words(zero, nzero);
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int b = m / T, t = m % T;
  const int32_t* cb = cum + (size_t)b * L;
  const int total = L > 0 ? cb[L - 1] : 0;
  if (mel_mask && lane == 0) mel_mask[m] = t >= total ? 1 : 0;  // get_mask_from_lengths(mel_len): total IS mel_len[b]
  // one writer per utterance: what a caller that chose T without reading mel_lens (capacity mode) must be able to find out later
  if (status && t == 0 && lane == 0) status[b] = forward_status(total, T, mel_lens ? mel_lens[b] : 0ll);
  float* dst = out + (size_t)m * D;
  if (t >= total) {
    for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<f32x4*>(dst + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  int lo = 0, hi = L - 1;  // smallest i with cb[i] > t
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cb[mid] > t) hi = mid;
    else lo = mid + 1;
  }
  const float* src = x + ((size_t)b * L + lo) * D;
  for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(src + c);
}
// T == 0 (nothing to regulate): only the status words
__global__ void k_status_only(const int32_t* __restrict__ cum, int L, int B, const long long* __restrict__ mel_lens,
                              int32_t* __restrict__ status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) status[b] = forward_status(L > 0 ? cum[(size_t)b * L + L - 1] : 0, 0, mel_lens ? mel_lens[b] : 0ll);
}
hipError_t launch_length_regulate(const float* x, const int32_t* cum, int B, int L, int D, int T, float* out, uint8_t* mel_mask,
                                  const long long* mel_lens, int32_t* status, int* zero, int nzero, hipStream_t st) {
  const int M = B * T;
  if (M <= 0) {
    if (B > 0 && status) hipLaunchKernelGGL(k_status_only, dim3((B + 63) / 64), dim3(64), 0, st, cum, L, B, mel_lens, status);
    return hipGetLastError();
  }
  if (D % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_length_regulate, dim3((M + 3) / 4), dim3(256), 0, st, x, cum, L, D, T, M, out, mel_mask, mel_lens, status, zero, nzero);
  return hipGetLastError();
}

// ---- packed rows (kernels.h RowMap): plan, gather, unpack ------------------------------------------------------------------
// Plan, one block: win[b] = min(max(mel_lens[b], 0) + PACK_GUARD, T); off = exclusive scan (B is a batch size: serial is fine);
// attention work list: utterances ranked by descending window (ties by index), att_off = exclusive scan of
// ceil(win / 128) * H in rank order.
__global__ __launch_bounds__(256) void k_pack_plan(const long long* __restrict__ mel_lens, int B, int T, int H, int* __restrict__ off,
                                                    int* __restrict__ win, int* __restrict__ att_off, int* __restrict__ att_order, int guard) {
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    long long l = mel_lens[b];
    if (l < 0) l = 0;
    l += guard;
    win[b] = (int)(l < (long long)T ? l : (long long)T);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int w = win[b];
    int rank = 0;
    for (int o = 0; o < B; ++o) rank += (win[o] > w || (win[o] == w && o < b)) ? 1 : 0;
    att_order[rank] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int o = 0, a = 0;
    for (int b = 0; b < B; ++b) {
      off[b] = o;
      o += win[b];
      att_off[b] = a;
      a += ((win[att_order[b]] + 127) / 128) * H;
    }
    off[B] = o;
    att_off[B] = a;
  }
}

// the row maps alone (the Gaussian regulator's packed form gathers nothing: it computes its rows in place)
__global__ void k_pack_rows(const int* __restrict__ off, const int* __restrict__ win, int B, int Mp, int* __restrict__ row_b,
                            int* __restrict__ row_t, int* __restrict__ row_w) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Mp) return;
  int lo = 0, hi = B - 1;  // largest b with off[b] <= m
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= m) lo = mid;
    else hi = mid - 1;
  }
  row_b[m] = lo; row_t[m] = m - off[lo]; row_w[m] = win[lo];
}

static void plan_pointers(int* plan, int B, int Mp, RowMap* rm) {
  int* off = plan;
  int* win = off + B + 1;
  int* att_off = win + B + 1;
  int* att_order = att_off + B + 1;
  int* row_b = att_order + B + 1;
  rm->off = off; rm->win = win; rm->att_off = att_off; rm->att_order = att_order;
  rm->row_b = row_b; rm->row_t = row_b + Mp; rm->row_w = row_b + 2 * (size_t)Mp;
}

// the plan alone (offsets, windows, attention work list); the row maps are written by the kernel that opens the phase
hipError_t launch_pack_plan_only(const long long* lens, int B, int T, int H, int Mp, int* plan, RowMap* rm, hipStream_t st, int guard) {
  if (B <= 0 || Mp <= 0 || !plan || !rm || !lens) return hipErrorInvalidValue;
  plan_pointers(plan, B, Mp, rm);
  hipLaunchKernelGGL(k_pack_plan, dim3(1), dim3(256), 0, st, lens, B, T, H, const_cast<int*>(rm->off), const_cast<int*>(rm->win),
                     const_cast<int*>(rm->att_off), const_cast<int*>(rm->att_order), guard);
  return hipGetLastError();
}

hipError_t launch_pack_plan(const long long* mel_lens, int B, int T, int H, int Mp, int* plan, RowMap* rm, hipStream_t st, int guard) {
  if (B <= 0 || Mp <= 0 || !plan || !rm || !mel_lens) return hipErrorInvalidValue;
  plan_pointers(plan, B, Mp, rm);
  hipLaunchKernelGGL(k_pack_plan, dim3(1), dim3(256), 0, st, mel_lens, B, T, H, const_cast<int*>(rm->off), const_cast<int*>(rm->win),
                     const_cast<int*>(rm->att_off), const_cast<int*>(rm->att_order), guard);
  hipLaunchKernelGGL(k_pack_rows, dim3((Mp + 255) / 256), dim3(256), 0, st, rm->off, rm->win, B, Mp, const_cast<int*>(rm->row_b),
                     const_cast<int*>(rm->row_t), const_cast<int*>(rm->row_w));
  return hipGetLastError();
}

// LengthRegulator.LR + pad (model/modules.py:201-218) into the packed layout: row m belongs to the utterance b with
// off[b] <= m < off[b+1] (binary search), frame t = m - off[b]; also writes the row maps every later kernel reads.
__global__ __launch_bounds__(256) void k_length_regulate_packed(const float* __restrict__ x, const int32_t* __restrict__ cum, int B, int L,
                                                                 int D, int T, int Mp, float* __restrict__ out,
                                                                 const long long* __restrict__ mel_lens, int32_t* __restrict__ status,
                                                                 const int* __restrict__ off, const int* __restrict__ win,
                                                                 int* __restrict__ row_b, int* __restrict__ row_t, int* __restrict__ row_w,
                                                                 int* __restrict__ zero, int nzero) {
  zero_words(zero, nzero);
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= Mp) return;
  int lo = 0, hi = B - 1;  // largest b with off[b] <= m
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= m) lo = mid;
    else hi = mid - 1;
  }
  const int b = lo, t = m - off[b];
  if (lane == 0) { row_b[m] = b; row_t[m] = t; row_w[m] = win[b]; }
  const int32_t* cb = cum + (size_t)b * L;
  const int total = L > 0 ? cb[L - 1] : 0;
  if (status && t == 0 && lane == 0) status[b] = forward_status(total, T, mel_lens ? mel_lens[b] : 0ll);
  float* dst = out + (size_t)m * D;
  if (t >= total) {
    for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<f32x4*>(dst + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  lo = 0; hi = L - 1;  // smallest i with cb[i] > t
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cb[mid] > t) hi = mid;
    else lo = mid + 1;
  }
  const float* src = x + ((size_t)b * L + lo) * D;
  for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(src + c);
}

hipError_t launch_length_regulate_packed(const float* x, const int32_t* cum, int B, int L, int D, int T, int Mp, int H, float* out,
                                         const long long* mel_lens, int32_t* status, int* zero, int nzero, int* plan, RowMap* rm,
                                         hipStream_t st) {
  if (B <= 0 || Mp <= 0 || D % 4 != 0 || !plan || !rm || !mel_lens) return hipErrorInvalidValue;
  plan_pointers(plan, B, Mp, rm);
  int* off = const_cast<int*>(rm->off);
  int* win = const_cast<int*>(rm->win);
  hipLaunchKernelGGL(k_pack_plan, dim3(1), dim3(256), 0, st, mel_lens, B, T, H, off, win, const_cast<int*>(rm->att_off),
                     const_cast<int*>(rm->att_order), PACK_GUARD);  // (att_wgs, rows: the caller's, from its host copy of the lengths)
  hipLaunchKernelGGL(k_length_regulate_packed, dim3((Mp + 3) / 4), dim3(256), 0, st, x, cum, B, L, D, T, Mp, out, mel_lens, status, off, win,
                     const_cast<int*>(rm->row_b), const_cast<int*>(rm->row_t), const_cast<int*>(rm->row_w), zero, nzero);
  return hipGetLastError();
}

__global__ void k_broadcast_row(const float* __restrict__ row, float* __restrict__ dst, int total, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) dst[i] = row[i % n];
}
hipError_t launch_broadcast_row(const float* row, float* dst, int rows, int n, hipStream_t st) {
  if (rows <= 0 || n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_broadcast_row, dim3((rows * n + 255) / 256), dim3(256), 0, st, row, dst, rows * n, n);
  return hipGetLastError();
}

// dst[m] = src[row_b[m] * T + row_t[m]]: a padded [B, T] vector (forward()'s p_targets / e_targets) onto the packed rows
__global__ void k_pack_vector(const int* __restrict__ row_b, const int* __restrict__ row_t, int T, const float* __restrict__ src,
                              float* __restrict__ dst, int Mp) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < Mp) dst[m] = src[(size_t)row_b[m] * T + row_t[m]];
}
hipError_t launch_pack_vector(const RowMap& rm, int T, const float* src, float* dst, int Mp, hipStream_t st) {
  if (Mp <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_vector, dim3((Mp + 255) / 256), dim3(256), 0, st, rm.row_b, rm.row_t, T, src, dst, Mp);
  return hipGetLastError();
}

// The caller's padded [B, T, .] outputs from the packed rows.  Frame t of utterance b (window w = win[b], length len):
//   mel       t < w: the packed row; else the mel_linear bias (mel_linear of a zeroed decoder row, fastspeech2_align.py:83)
//   p / e     t < w: the packed value (0 past len: the predictors' mask); else 0
//   mel_mask  t >= len (utils/tools.py:89-97)
//   postnet   the PostNet has no mask between its layers (transformer/Layers.py:169-177): in the reference a padded frame's value
//             depends on how far it is from the utterance's last valid frame and from the end of the padded axis — at most
//             10 frames either way (5 convolutions of reach 2); in between every frame is the same vector.  w == T (the window
//             is the whole axis): every packed row is what the dense computation gives.  w < T (then len + PACK_GUARD <= T):
//             t < len + 10 the packed row (its reach stays inside the window's first len + 20 frames); t >= T - 10 row
//             1 + t - (T - 10) of post_const (frames that see the end of the axis); else row 0 (deep padding).  post_const is
//             the PostNet applied to an all-padding utterance (api.hip postnet_constants).
__global__ __launch_bounds__(256) void k_unpack_outputs(const int* __restrict__ off, const int* __restrict__ win, int B, int T, int n_mel,
                                                         const long long* __restrict__ mel_lens, const float* __restrict__ mel_p,
                                                         const float* __restrict__ post_p, const float* __restrict__ p_p,
                                                         const float* __restrict__ e_p, const float* __restrict__ mel_bias,
                                                         const float* __restrict__ post_const, float* __restrict__ mel,
                                                         float* __restrict__ post, float* __restrict__ p_pred, float* __restrict__ e_pred,
                                                         uint8_t* __restrict__ mel_mask) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);  // row of the padded grid
  if (m >= B * T) return;
  const int b = m / T, t = m - b * T;
  const int w = win[b];
  long long len_ll = mel_lens[b];
  const int len = (int)(len_ll < 0 ? 0 : (len_ll < (long long)T ? len_ll : (long long)T));
  const size_t src = (size_t)off[b] + t;
  const bool in_win = t < w;
  if (lane == 0) {
    if (mel_mask) mel_mask[m] = t >= len ? 1 : 0;
    if (p_pred) p_pred[m] = in_win ? p_p[src] : 0.f;
    if (e_pred) e_pred[m] = in_win ? e_p[src] : 0.f;
  }
  const float* post_src;
  if (w == T || t < len + 10) post_src = post_p + src * n_mel;
  else if (t >= T - 10) post_src = post_const + (size_t)(1 + t - (T - 10)) * n_mel;
  else post_src = post_const;
  const float* mel_src = in_win ? mel_p + src * n_mel : mel_bias;
  for (int c = lane; c < n_mel; c += 64) {
    mel[(size_t)m * n_mel + c] = mel_src[c];
    post[(size_t)m * n_mel + c] = post_src[c];
  }
}

hipError_t launch_unpack_outputs(const RowMap& rm, int B, int T, int n_mel, const long long* mel_lens, const float* mel_p, const float* post_p,
                                 const float* p_p, const float* e_p, const float* mel_bias, const float* post_const, float* mel,
                                 float* post, float* p_pred, float* e_pred, uint8_t* mel_mask, hipStream_t st) {
  if (B <= 0 || T <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_unpack_outputs, dim3((unsigned)(((size_t)B * T + 3) / 4)), dim3(256), 0, st, rm.off, rm.win, B, T, n_mel, mel_lens, mel_p,
                     post_p, p_p, e_p, mel_bias, post_const, mel, post, p_pred, e_pred, mel_mask);
  return hipGetLastError();
}

// GaussianUpsampling.forward (model/modules.py:166-192; defined but never called by the reference forward,
// SURVEY.md F1).  c = cumsum(d) - d/2 ; w[b,l,t] = exp(-0.01 (t-c)^2) / (sum_l exp(..) + 1e-20) ; out = w^T x.
// t spans [0, T) with T = max_b sum(d): rows past an utterance's own length are NOT zeroed; rows in
// [T, T_out) are the zero padding of pad(output, max_len).
__global__ __launch_bounds__(256) void k_gauss_centers(const float* __restrict__ dur, int L, float* __restrict__ centers,
                                                        float* __restrict__ s, const long long* __restrict__ own_len, int T,
                                                        int32_t* __restrict__ status, int* __restrict__ zero, int nzero) {
  zero_words(zero, nzero);
  // sequential fp32 cumsum per utterance, as torch.cumsum on CPU accumulates
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  if (status) status[b] = own_len ? forward_status(own_len[b], T, own_len[b]) : 0;
  float e = 0.f;
  for (int l = 0; l < L; ++l) {
    const float dl = dur[(size_t)b * L + l];
    e += dl;
    centers[(size_t)b * L + l] = e - 0.5f * dl;
  }
  s[b] = e;
}
// One workgroup = 32 consecutive output frames of one utterance x all D channels: out[32, D] = w^T[32, L] x[L, D] on the
// fp32 matrix cores.  The normalised weights of the 32 frames are staged in LDS one chunk of GU_LC phonemes at a time
// (row stride GU_LC + 1: the A-fragment reads — 32 frames at one phoneme — hit 32 different banks); the B operand
// x[l, n0 + lane] is read straight from L2 (x[b] is re-read by every tile of its utterance and stays resident).
// v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain, so walking l upwards reproduces the serial
// `acc = fmaf(w[l], x[l], acc)` of a scalar loop bit for bit.  Cost O(T L D / 32) matrix instructions instead of O(T L D)
// scalar FMAs with one L2 sweep of x per FRAME (the round-1 kernel).
typedef float f32x16r __attribute__((ext_vector_type(16)));
constexpr int GU_LC = 256;
__global__ __launch_bounds__(256) void k_gauss_upsample(const float* __restrict__ x, const float* __restrict__ centers, int L,
                                                         int D, int T, int T_out_grid, float* __restrict__ out,
                                                         float* __restrict__ w, const long long* __restrict__ own_len,
                                                         const int* __restrict__ pk_off, const int* __restrict__ pk_win) {
#if defined(__HIP_DEVICE_COMPILE__)  // the MFMA builtin does not exist in the host pass; it only needs the stub
  __shared__ float wt[32 * (GU_LC + 1)];
  __shared__ float w2s[32];
  const int b = blockIdx.y, t0 = blockIdx.x * 32, tid = threadIdx.x, lane = tid & 63;
  // packed rows (kernels.h RowMap): utterance b's frames are rows [pk_off[b], pk_off[b] + pk_win[b]) of `out`
  const int T_out = pk_win ? pk_win[b] : T_out_grid;
  const size_t row0 = pk_off ? (size_t)pk_off[b] : (size_t)b * T_out_grid;
  if (t0 >= T_out) return;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tt = tid & 31, sub = tid >> 5;  // frame of this thread inside the tile, and which eighth of the phonemes it walks
  const float* cb = centers + (size_t)b * L;
  const float tf = (float)(t0 + tt);
  // pass 1: w2[t] = sum_l exp(-0.01 (t - c_l)^2) + 1e-20 (model/modules.py:184-186)
  float part = 0.f;
  for (int l = sub; l < L; l += 8) {
    const float df = tf - cb[l];
    part += expf(-0.01f * (df * df));
  }
  wt[tt * 9 + sub] = part;
  __syncthreads();
  if (tid < 32) {
    const float* p8 = wt + tid * 9;
    w2s[tid] = (((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]))) + 1e-20f;
  }
  __syncthreads();
  const float w2 = w2s[tt];
  // frames this tile must leave as zeros: t >= T is pad(output, max_len); with own_len (the wired-in length regulator)
  // also t >= the utterance's own length.  own_len == nullptr keeps the reference module's behaviour (NOT zeroed).
  const int t_lim = own_len ? (int)(own_len[b] < (long long)T ? own_len[b] : (long long)T) : T;
  const int nt = (D + 31) / 32;  // 32-channel tiles (the last may be partial); wave wv owns tiles wv, wv+4, ... (up to 4 per pass)
  for (int nbase = 0; nbase < nt; nbase += 16) {
    f32x16r acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int lc = 0; lc < L; lc += GU_LC) {
      const int ln = min(GU_LC, L - lc), lpad = (ln + 1) & ~1;
      for (int j = sub; j < lpad; j += 8) {
        float wvv = 0.f;
        if (j < ln) {
          const float df = tf - cb[lc + j];
          wvv = expf(-0.01f * (df * df)) / w2;
          if (w && nbase == 0 && t0 + tt < T) w[((size_t)b * L + lc + j) * T + t0 + tt] = wvv;
        }
        wt[tt * (GU_LC + 1) + j] = wvv;
      }
      __syncthreads();
      const float* arow = wt + (lane & 31) * (GU_LC + 1) + (lane >> 5);
      const float* xrow = x + ((size_t)b * L + lc + (lane >> 5)) * D + (lane & 31);
      // (this wave's tiles are nbase + wv + 4 j, j < njt: ONE scalar count instead of four per-tile range predicates and
      //  channel-range masks kept live across the loop — with expf's constants that was 18 spilled SGPRs, round 5)
      const int njt = min(4, (nt - nbase - wv + 3) >> 2);
      const int col = (nbase + wv) * 32 + (lane & 31);  // channel of tile j: col + 128 j
      for (int l2 = 0; l2 < lpad; l2 += 2) {
        const float av = arow[l2];
        const bool in = lc + l2 + (lane >> 5) < L;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < njt) {
            const float bv = (in && col + 128 * j < D) ? xrow[(size_t)l2 * D + (nbase + wv + 4 * j) * 32] : 0.f;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
    // (the 16 rows' range predicates do not depend on the tile pass: the compiler hoisted all 32 of them out of the nbase loop
    //  and held them as SGPR pairs across the contraction — the 18 spilled SGPRs of round 5.  An opaque base keeps them here.)
    int tb = t0 + 4 * (lane >> 5);
    asm("" : "+v"(tb));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tile = nbase + wv + 4 * j;
      if (tile >= nt) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = tb + (r & 3) + 8 * (r >> 2);
        if (t < T_out && tile * 32 + (lane & 31) < D) out[(row0 + t) * D + tile * 32 + (lane & 31)] = t < t_lim ? acc[j][r] : 0.f;
      }
    }
  }
#endif
}
hipError_t launch_gaussian_upsampling(const float* x, const float* dur, int B, int L, int D, int T, int T_out, float* out,
                                      float* s, float* w, const long long* own_len, int32_t* status, int* zero, int nzero,
                                      hipStream_t st, const RowMap* rm) {
  if (B <= 0 || T_out <= 0) return hipSuccess;
  if (rm && w) return hipErrorInvalidValue;  // (the weight tensor is a [B, L, T] grid output of the stand-alone module)
  // s holds B sums followed by B*L Gaussian centres (scratch)
  float* centers = s + B;
  hipLaunchKernelGGL(k_gauss_centers, dim3(B), dim3(64), 0, st, dur, L, centers, s, own_len, T, status, zero, nzero);
  hipLaunchKernelGGL(k_gauss_upsample, dim3((T_out + 31) / 32, B), dim3(256), 0, st, x, centers, L, D, T, T_out, out, w, own_len,
                     rm ? rm->off : nullptr, rm ? rm->win : nullptr);
  return hipGetLastError();
}

}  // namespace ns
