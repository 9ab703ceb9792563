// Internal launch interface between the C-ABI (api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ns {

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

// Packed rows (variable-length batches, api.hip forward_mel packed mode): the M activation rows are the utterances' WINDOWS
// laid end to end — utterance b owns rows [off[b], off[b] + win[b]), win[b] = min(len[b] + guard, T) — instead of the dense
// [B, S] grid in which the reference computes (and then discards) every padded frame.  Per row: its utterance, its position
// in the utterance, and its utterance's window length.  All nullptr = dense grid: b = m / S, t = m % S, window S.
struct RowMap {
  const int* row_b; const int* row_t; const int* row_w;  // [M] each
  const int* off; const int* win;                        // [B + 1], [B]
  // attention work list (attention.hip, packed launches): utterances in order of DESCENDING window, att_order[r] = the r-th
  // longest; att_off[r] = first workgroup of rank r, att_off[B] = att_wgs = sum_b ceil(win[b] / 128) * H.  Longest sweeps first:
  // the hardware hands workgroups to CUs in launch order, so the launch finishes when the work does, not when the XCD that drew
  // the longest utterance does.
  const int* att_off; const int* att_order; int att_wgs;
  int rows;                                              // M (host side)
};

// Row epilogue of a FULL-ROW tile (N == the tile width, 256 or 512): what the reference applies to every output row right
// after the contraction, done while the row is still on chip instead of by a second kernel over [M, N].
//   EPI_LN      Y[m,:] = LayerNorm_N(v[m,:]) * ln_g + ln_b, rows at t >= lens[b] written as zeros when lens != nullptr
//               (`layer_norm(output + residual)` + FFTBlock's masked_fill, transformer/SubLayers.py:57,93 + Layers.py:43,46;
//               the predictors' layer_norm_1, model/modules.py:260)
//   EPI_LN_PRED pred[m] = mask ? 0 : dot(LayerNorm_N(v[m,:]), wlin) + blin, optionally followed by the bucketize +
//               embedding (+ position) add into x_out [M, D] (VariancePredictor tail, model/modules.py:273-286,80-100,139-149);
//               Y is not written
// v = act(contraction + bias) + resid, exactly what the plain epilogue would have stored.
enum RowEpi : int { EPI_NONE = 0, EPI_LN = 1, EPI_LN_PRED = 2 };
struct RowEpilogue {
  const float* ln_g; const float* ln_b;
  const long long* lens;
  const float* wlin; const float* blin; float* pred; float control; const float* target;
  const float* bins; int n_edges; const float* emb; const float* x_in; const float* pos; float* x_out; int D;
  // ticketed form (small grids, gemm_conv.hip TICKET): the GEMM tiles N with BN < N, stores the raw rows into ConvGemm::Y
  // ([M, N], ldy == N) and the LAST workgroup to finish a row block applies the row epilogue, EPI_LN writing y_out [M, N].
  // ticket: conv_gemm_ticket_ints(M) zeroed ints, used by this launch only; nullptr selects the full-row tile.
  float* y_out; int* ticket;
  const int* row_b; const int* row_t;  // packed rows (RowMap): utterance / position of row m; nullptr = m / S, m % S
};
inline int conv_gemm_ticket_ints(int M) { return (M + 31) / 32; }
bool conv_gemm_ticket_ok(int M, int N, int Cin);  // shapes the ticketed form covers (row widths 256 / 512)

// Y[m, n] = act( sum_{j<KW} sum_{c<Cin} X[m + j - pad, c] * W[n][j*Cin + c] + bias[n] ) + resid[m, n]
// rows of X outside the utterance's [0, S) window read as zero ("same" zero padding of nn.Conv1d).
struct ConvGemm {
  const float* X; int ldx;
  const float* W;               // packed [N][KW*Cin], row stride ldw
  int ldw;                      // floats between weight rows; 0 = KW*Cin (dense)
  const unsigned short* Wb3;    // optional: the same weights as three bf16 planes [3][N][KW*Cin] (gemm_bf16x3.hip), else nullptr
  const float* bias;            // [N] or nullptr
  const float* resid; int ldr;  // [M, N] or nullptr
  float* Y; int ldy;
  int M, N, Cin, KW, pad, S;
  int m_base;                   // rows of the full matrix ahead of X / Y / resid row 0 (a launch over a row range of a larger
                                // problem, gemm_conv.hip split plan): the utterance position of row m is that of row m_base + m.
                                // Plain epilogues only (epi == EPI_NONE).
  int act;
  int epi;                      // RowEpi; != EPI_NONE requires conv_gemm_row_epilogue_ok(p)
  RowEpilogue e;
  RowMap rm;                    // packed rows: tap windows come from row_t / row_w instead of m % S / S
};
// Optional timing of one launch_conv_gemm / launch_attention call: the events ride ON the dispatch packets
// (hipExtLaunchKernel's start / stop events: the kernel's own begin / end timestamps), so timing a launch adds no marker
// packet and no idle gap to the stream — hipEventRecord pairs around the heavy launches cost ~6 us each, 3 % of a forward.
// A plan of several launches gets `start` on its first and `stop` on its last kernel.  nullptr / {nullptr, nullptr} = untimed.
struct LaunchTiming { hipEvent_t start, stop; };
hipError_t launch_conv_gemm(const ConvGemm& p, hipStream_t st, const LaunchTiming* tm = nullptr);
// {bm, bn, rows} of the main launch, {bm, bn, rows} of the remainder (0 = none), the MFMA tile edge of the launch(es) (32 / 16),
// the cost model's estimate in us
bool conv_gemm_plan(int M, int N, int Cin, int KW, int out[8]);
bool conv_gemm_tile16_enabled();  // the 16-row tile family is in use (planner on, NS_TILE16 != 0)
int conv_gemm_row_tile(int M, int N, int K = 256);  // height of the full-row (LayerNorm epilogue) tile for M rows of N = 256 / 512 columns, contraction length K
int conv_gemm_acc_chunk();  // k values per accumulation chunk of the long contractions (gemm_conv.hip ACC2; NS_ACC_CHUNK, 0 = one sequential sum)
// NS_PLAN=0 in the environment: the round-3 one-tile-per-launch rules (A/B runs of the planner; read once)
bool launch_planner_enabled();
// opt-in "bf16x3" precision mode (gemm_bf16x3.hip): same contraction from an exact 3-way bf16 split of both operands
bool conv_gemm_b3_ok(int M, int N, int Cin, int KW, int epi);  // epi: EPI_NONE or EPI_LN
hipError_t launch_conv_gemm_b3(const ConvGemm& p, hipStream_t st);
void split_weights_b3(const float* w, size_t n, unsigned short* hi, unsigned short* mid, unsigned short* lo);
// true when launch_conv_gemm has a full-row tile for this shape (so p.epi may be set); otherwise the caller runs the
// plain GEMM followed by the row kernel
bool conv_gemm_row_epilogue_ok(int M, int N, int Cin);

// Fused multi-head self attention over the packed projection buffer qkv [B*S, 3*d]
// (cols [0,d) = Q, [d,2d) = K, [2d,3d) = V, head h at offset h*dk inside each).
// out [B*S, d] = merge_heads( softmax(Q K^T / sqrt(dk) + (-inf at keys >= lens[b])) V )
// scratch (optional, scratch_floats floats): enables the split-key path, taken by launches of fewer than
// ATT_SPLIT_MAX_BLOCKS workgroups: up to ATT_SPLIT_MAX key ranges, each needing B*S*(H*dk + 2*H) floats
// (16, not 8: a single 788-frame utterance has 14 (query tile, head) pairs x 25 key tiles; 13 two-tile ranges instead of 8
//  four-tile ranges take the decoder attention from 24.5 to ~19 us, single-utterance p50 1.08 -> 1.06 ms, same box)
constexpr int ATT_SPLIT_MAX = 16, ATT_SPLIT_MAX_BLOCKS = 128;
// tickets (nullable): attention_ticket_ints(B, S, H) ZEROED ints for the strip kernel's last-arriver merge (small grids)
inline int attention_ticket_ints(int B, int S, int H) { return B * H * ((S + 31) / 32); }
bool attention_uses_tickets(int B, int S, int H);  // false: launch_attention(B, S, H, ...) never touches `tickets` (pass nullptr)
// rm (packed rows): utterance b's rows start at rm->off[b] and number rm->win[b] <= S (S = the longest window); no split-key path
hipError_t launch_attention(const float* qkv, const long long* lens, int B, int S, int H, int dk, float* out, float* scratch,
                            size_t scratch_floats, int* tickets, hipStream_t st, const RowMap* rm = nullptr, const LaunchTiming* tm = nullptr);
// key ranges per 128-query tile the dense launch of (B, S, H, dk) will use when it has the scratch for them (1 = no split):
// the caller sizes `scratch` as attention_split(...) * B*S*(H*dk + 2*H) floats
int attention_split(int B, int S, int H, int dk);
// the same for a packed launch: att_wgs workgroups on the work list, S = the longest window, Mp packed rows
int attention_split_packed(int att_wgs, int S, int dk, size_t Mp, int d);

// ---- row kernels (rowops.hip) -----------------------------------------------------------------
// y = LayerNorm_C(x) * g + b ; rows with t >= lens[b] are written as zero when lens != nullptr
hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, int M, int C, int S,
                            const long long* lens, hipStream_t st, const RowMap* rm = nullptr);
// pred[m] = mask ? 0 : dot(LayerNorm_C(x[m]), wlin) + blin            (variance predictor tail)
// if emb != nullptr additionally  x_out[m,:] = x_in[m,:] + emb[bucketize(pred[m]*control, bins)] (+ pos[t,:])
hipError_t launch_ln_linear_embed(const float* x, const float* g, const float* b, const float* wlin, const float* blin,
                                  float* pred, int M, int C, int S, const long long* lens, float control,
                                  const float* target, const float* bins, int n_bins, const float* emb, const float* x_in, const float* pos,
                                  float* x_out, int D, hipStream_t st, const RowMap* rm = nullptr);
// out[m,:] = emb[texts[m],:] + pos[t,:]
// token ids outside [0, n_vocab) read row 0 (and are reported by launch_duration_tail)
// zero / nzero (nullable): ticket counters of the forward phase this kernel opens, zeroed by it (rowops.hip zero_words);
// the same pair on launch_length_regulate / launch_gaussian_upsampling
hipError_t launch_embed_pos(const long long* texts, const float* emb, const float* pos, float* out, int M, int S, int D, int n_vocab,
                            int* zero, int nzero, hipStream_t st);
hipError_t launch_add_pos(const float* x, const float* pos, float* out, int M, int S, int D, hipStream_t st, const RowMap* rm = nullptr);
hipError_t launch_bucketize(const float* v, int n, const float* bins, int n_edges, long long* idx, hipStream_t st);
hipError_t launch_mask_from_lengths(const long long* lens, int B, int max_len, uint8_t* mask, hipStream_t st);
hipError_t launch_sinusoid(int n_pos, int d, float* out, hipStream_t st);
hipError_t launch_duration_round(const float* log_d, int n, float d_control, float* d_rounded, hipStream_t st);
hipError_t launch_duration_scan(const float* d_rounded, int B, int L, int32_t* cum, long long* mel_lens, hipStream_t st);
// mel_mask (nullable): also writes get_mask_from_lengths(mel_len) for the [B,T] frame grid
// status (nullable, [B] int32): per-utterance NS_STATUS_* bits of ns_forward_mel (needs mel_lens for the bad-token bit)
hipError_t launch_length_regulate(const float* x, const int32_t* cum, int B, int L, int D, int T, float* out, uint8_t* mel_mask,
                                  const long long* mel_lens, int32_t* status, int* zero, int nzero, hipStream_t st);
// phase-1 tail in one launch: src mask, duration_round (two copies), duration_scan; mel_lens[b] = -1 when utterance b
// holds a token id outside [0, n_vocab) (texts may be nullptr: no check)
hipError_t launch_duration_tail(const float* log_d, const long long* src_lens, const long long* texts, int n_vocab, int B, int L,
                                float d_control, float* d_rounded, float* d_keep, int32_t* cum, long long* mel_lens, uint8_t* src_mask,
                                long long* mel_lens_host /* nullable: device-visible host copy */, hipStream_t st);
// Packed variant of launch_length_regulate (kernels.h RowMap).  Builds the plan first: win[b] = min(max(mel_lens[b], 0) + guard, T),
// off = exclusive scan, row maps for the Mp = sum(win) rows (the caller computed the same Mp from its host copy of mel_lens);
// then gathers the encoder rows into the packed layout (frames at t >= mel_len[b] are zero).  status as launch_length_regulate.
// plan: int storage for off [B+1], win [B], row_b / row_t / row_w [Mp] — pack_plan_ints(B, Mp) ints; *rm receives the pointers.
constexpr int PACK_GUARD = 20;  // frames kept past an utterance's end: the PostNet's reach (5 layers x 2) twice over, see api.hip
inline size_t pack_plan_ints(int B, size_t Mp) { return (size_t)4 * B + 4 + 3 * Mp; }
// H: attention heads of the stack that will run on these rows (the plan's attention work list is per head)
hipError_t launch_length_regulate_packed(const float* x, const int32_t* cum, int B, int L, int D, int T, int Mp, int H, float* out,
                                         const long long* mel_lens, int32_t* status, int* zero, int nzero, int* plan, RowMap* rm,
                                         hipStream_t st);
// dst[r, :] = row[:] for r < rows (n % 4 == 0)
hipError_t launch_broadcast_row(const float* row, float* dst, int rows, int n, hipStream_t st);
hipError_t launch_pack_vector(const RowMap& rm, int T, const float* src, float* dst, int Mp, hipStream_t st);
// the packing plan and row maps alone (launch_length_regulate_packed builds them as a side effect of its gather)
hipError_t launch_pack_plan(const long long* mel_lens, int B, int T, int H, int Mp, int* plan, RowMap* rm, hipStream_t st, int guard = PACK_GUARD);
// Phase 1 on packed PHONEME rows (api.hip forward_durations): utterance b keeps min(src_len[b] + PHONEME_GUARD, L) rows.  In the
// FFT blocks a valid phoneme never reads a padded one except as zeros (masked_fill ahead of every convolution, -inf keys); the
// variance predictors have no mask between their two convolutions (model/modules.py:245-286), so the last valid
// phoneme reads ONE row past the utterance's end — a row that must be computed, from zeroed encoder output, like the reference does.
constexpr int PHONEME_GUARD = 2;
hipError_t launch_store_lens(const long long* host, int n, long long* dst, hipStream_t st);  // dst[i] = host[i], values passed as kernel arguments
hipError_t launch_pack_plan_only(const long long* lens, int B, int T, int H, int Mp, int* plan, RowMap* rm, hipStream_t st, int guard);
// (also writes the row maps of *rm: launch_pack_plan_only + this kernel = the whole plan)
hipError_t launch_embed_pos_packed(const long long* texts, const float* emb, const float* pos, float* out, const RowMap& rm, int B, int Mp, int L,
                                   int D, int n_vocab, int* zero, int nzero, hipStream_t st);
hipError_t launch_unpack_phase1(const RowMap& rm, const long long* lens, int B, int S, int D, const float* rows_p, float* rows, const float* vec_p,
                                float* vec, hipStream_t st);
// dst [B*S, D] (any D) = the packed rows of src where t < min(lens[b], win[b]), zeros elsewhere
hipError_t launch_unpack_rows(const RowMap& rm, const long long* lens, int B, int S, int D, const float* src, float* dst, hipStream_t st);
// padded outputs from packed rows (api.hip forward_mel): see k_unpack_outputs in rowops.hip
hipError_t launch_unpack_outputs(const RowMap& rm, int B, int T, int n_mel, const long long* mel_lens, const float* mel_p, const float* post_p,
                                 const float* p_p, const float* e_p, const float* mel_bias, const float* post_const, float* mel,
                                 float* post, float* p_pred, float* e_pred, uint8_t* mel_mask, hipStream_t st);
hipError_t launch_gaussian_upsampling(const float* x, const float* dur, int B, int L, int D, int T, int T_out,
                                      float* out, float* s, float* w, const long long* own_len, int32_t* status, int* zero, int nzero,
                                      hipStream_t st, const RowMap* rm = nullptr);  // rm: out is the packed layout (w must be nullptr)

}  // namespace ns
