/*
 * nar_fs2.h — C ABI of the MI355X-native FastSpeech2 inference forward.
 *
 * The reference (SMART-TTS/SMART-NAR_Fast_TTS) has no FFI / plugin layer: its
 * boundary for this path is the Python class FastSpeech2Align
 * (model/fastspeech2_align.py:13-100).  This header is the C-ABI a binding for
 * that class sits on: plain pointers and sizes, an explicit hipStream_t passed
 * as void*, int status returns (0 = ok; the Python side raises RuntimeError
 * with ns_last_error()).  The library allocates NO device memory: weights live
 * in a caller-provided arena and every call takes a caller-provided workspace
 * (PyTorch is only the allocator / stream owner on the Python side).
 *
 * THREADING.  One ns_model serves ONE host thread at a time: the model carries per-call state — the measurement slots of
 * ns_profile_enable and the row counts ns_last_phase1_rows / ns_last_phase2_rows report — that a forward writes while it
 * enqueues its launches (everything else a forward needs, the packed-row context included, lives in the caller's
 * workspaces).  Calls on the same model from two host threads must be serialised by the caller; different models (one per
 * thread, or one per process as bench.py does per GPU) are independent, and one thread may drive several HIP streams
 * with one model as long as each stream has its own workspaces.  ns_last_error() is per thread.  This matches the
 * reference's caller: single-threaded, synchronous, one module instance (synthesize.py:59-76).
 *
 * All tensors are dense row-major float32 unless stated; token ids and lengths
 * are int64 (what torch.long hands over); masks are uint8 with 1 = padding
 * (utils/tools.py:89-97: True = padding).
 */
#ifndef NAR_FS2_H
#define NAR_FS2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this header's contract, returned by ns_abi_version(): 6 = ns_plan_row_tile_k, ns_acc_chunk, ns_abi_version;
 * 5 = ns_plan_gemm writes out[8] (round 5; unversioned then). */
#define NS_ABI_VERSION 6

typedef struct ns_model ns_model;

/* Mirrors the keys FastSpeech2Align.__init__ reads from model.yaml / preprocess.yaml
 * (model/fastspeech2_align.py:16-28, model/modules.py:20-77, transformer/Models.py:36-71,176-210)
 * plus the sizes the reference hard-codes (PostNet 80/512/k5/x5: transformer/Layers.py:112-118). */
typedef struct ns_config {
  int32_t n_vocab;          /* len(symbols)+1 = 361, transformer/Models.py:40 */
  int32_t max_seq_len;      /* 1000 */
  int32_t d_enc, n_enc_layer, n_enc_head;
  int32_t d_dec, n_dec_layer, n_dec_head;
  int32_t d_inner;          /* conv_filter_size 1024 */
  int32_t ffn_k1, ffn_k2;   /* conv_kernel_size [9, 1] */
  int32_t vp_filter, vp_kernel; /* variance_predictor 256 / 3 */
  int32_t n_bins;           /* 256 */
  int32_t n_mel;            /* 80 */
  int32_t postnet_dim, postnet_k, postnet_n; /* 512, 5, 5 */
  int32_t pitch_frame_level, energy_frame_level; /* 1 = frame_level (shipped config), 0 = phoneme_level */
  int32_t length_regulator; /* 0 = LengthRegulator (what the reference wires, model/modules.py:22); 1 = EXTENSION: the
                               reference's unused GaussianUpsampling (model/modules.py:162-192) in its place */
  int32_t matmul_bf16x3;    /* 0 = every contraction on the fp32 matrix cores (the default, the reference's arithmetic);
                               1 = OPT-IN: the decoder stack's projections / FFN convolutions and the PostNet 512->512
                               convolutions (downstream of every discrete duration / bucket decision; only launches large
                               enough to fill the chip with 64..256-row tiles, smaller ones stay fp32) run from an exact
                               3-way bf16 split of both operands on the bf16 matrix cores, 6 products, fp32 accumulation:
                               fp32-sized error, different bits, ~1.8x faster on those layers (csrc/gemm_bf16x3.hip) */
  int32_t row_epilogue;     /* 0 = the default: on small grids LayerNorm / the predictor tail / attention's key-range merge run as
                               TICKETED last-arriver epilogues inside the producing launch (csrc/gemm_conv.hip TICKET,
                               csrc/attention.hip); 1 = "two_launch": the same row functions as separate launches, no ticket is
                               ever drawn.  Same bits either way — the switch exists so that tests can A/B the ticket protocol
                               (tests/test_gpu_stress.py).  The full-row tile of large launches needs no ticket and is not affected */
  int32_t phase1_packing;   /* ns_forward_durations_packed: 0 = auto — pack the phoneme rows when >= 10 % of the [B, L] grid is padding AND the
                               smaller row count gives the fullest CU fewer rows of the phase's dominant launch (a whole round of its 32-row
                               K-split tile less, or one round of the 48-row form instead of two of the 32-row one); 1 = pack whenever >= 10 % is padding (tests exercise the path on small shapes);
                               2 = never */
} ns_config;

/* ---- lifetime ------------------------------------------------------------------------------ */
const char* ns_last_error(void);
int ns_create(const ns_config* cfg, ns_model** out);
void ns_destroy(ns_model* m);

/* Bytes of device memory the prepared weights need; the caller allocates it (torch.empty) and binds it.
 * The arena is position independent (offsets only), so rank 0 can fill it and RCCL-broadcast the bytes. */
size_t ns_arena_bytes(const ns_model* m);
int ns_bind_arena(ns_model* m, void* dev_arena, size_t bytes);

/* load_state_dict(): one call per state-dict entry, reference key names and torch-native layouts
 * (Linear [out,in], Conv1d [out,in,k]; utils/model.py:21-22).  `host` is HOST memory.  Unknown
 * "mel_encoder.*" keys and "*.num_batches_tracked" are accepted and ignored (returns 0);
 * any other unknown key or a shape mismatch is an error. */
int ns_set_weight(ns_model* m, const char* name, const float* host, const int64_t* shape, int ndim);
/* The same key / rank / shape validation WITHOUT touching the model (nothing staged, a loaded model stays loaded):
 * load_state_dict() checks every entry with this first, so that a rejected state dict leaves the previous weights in use
 * (nn.Module.load_state_dict raises before/without corrupting the module, utils/model.py:21-22). */
int ns_check_weight(ns_model* m, const char* name, const int64_t* shape, int ndim);
/* After the last ns_set_weight: repack (conv [out,in,k] -> [out,k,in]; fused QKV), fold eval-mode
 * BatchNorm into the PostNet convs, upload into the arena.  Fails if an inference key is missing. */
int ns_finalize_weights(ns_model* m, void* stream);
/* Non-root ranks: arena bytes arrived by broadcast; mark the model ready without ns_set_weight. */
int ns_adopt_arena(ns_model* m);

/* ---- the forward: model/fastspeech2_align.py:30-100, inference branch -------------------------- */
size_t ns_encoder_ws_bytes(const ns_model* m, int B, int L);
size_t ns_decoder_ws_bytes(const ns_model* m, int B, int L, int T);

/* Phase 1: get_mask_from_lengths + TxtEncoder + duration predictor + rounding + duration scan.
 * Writes log_d [B,L], d_rounded [B,L] (float32, may hold -0.0), src_mask [B,L], mel_lens [B] (int64).
 * The encoder output and the duration prefix sums stay in ws_enc for phase 2.
 * mel_lens_host (nullable): device-visible HOST memory (hipHostMalloc / a pinned torch tensor), [B] int64; the kernel that
 * produces mel_lens writes a second copy there, so the caller's read needs only a stream synchronisation, no D2H copy.
 * A token id outside [0, n_vocab) (nn.Embedding raises IndexError, transformer/Models.py:89) is reported as
 * mel_lens[b] = -1 for its utterance; the kernels read embedding row 0 for it, nothing out of bounds.
 * The caller reads mel_lens back (the one unavoidable device->host read: the output tensors are
 * shaped by max(mel_lens), model/modules.py:136-137) and allocates the phase-2 outputs. */
/* phoneme_level pitch / energy (preprocess.yaml `feature`, model/modules.py:117-126) are predicted here, on the
 * encoder output: p_control / e_control / p_targets / e_targets / p_pred / e_pred ([B,L]) are used by THIS call for a
 * phoneme_level feature and by ns_forward_mel ([B,T]) for a frame_level one; pass NULL where not applicable. */
int ns_forward_durations(ns_model* m, const int64_t* texts, const int64_t* src_lens, int B, int L,
                         float d_control, float p_control, float e_control, const float* p_targets, const float* e_targets,
                         void* ws_enc, size_t ws_enc_bytes,
                         float* log_d, float* d_rounded, uint8_t* src_mask, int64_t* mel_lens, float* p_pred, float* e_pred,
                         int64_t* mel_lens_host, void* stream);
/* The same with the HOST copy of src_lens (what a caller that builds its batches on the host has anyway: dataset.py:182-191,
 * utils/tools.py:254-264): ragged phoneme counts then run phase 1 on PACKED phoneme rows — utterance b keeps
 * min(src_lens[b] + 2, L) rows instead of L (transformer/Models.py:73-100 computes, then zeroes, every padded phoneme) — when that
 * saves >= 10 % of the rows and both variance features are frame_level.  Same outputs, padded like the reference's; values agree
 * with ns_forward_durations to fp32 summation order (another row count picks other tiles on the small-grid ladder).
 * ns_last_phase1_rows: the row count the most recent phase 1 ran on (B*L, or the packed rows). */
int ns_forward_durations_packed(ns_model* m, const int64_t* texts, const int64_t* src_lens, const int64_t* src_lens_host, int B, int L,
                                float d_control, float p_control, float e_control, const float* p_targets, const float* e_targets,
                                void* ws_enc, size_t ws_enc_bytes,
                                float* log_d, float* d_rounded, uint8_t* src_mask, int64_t* mel_lens, float* p_pred, float* e_pred,
                                int64_t* mel_lens_host, void* stream);
int64_t ns_last_phase1_rows(const ns_model* m);
/* Helper for callers whose lengths live on the host: dev[i] = host[i] (int64) on `stream`, the values riding in a kernel's
 * argument block — one ~3 us launch, no copy command (a pinned-staging async copy of these 128 bytes costs a forward ~35 us of
 * blit + stream dependency, a pageable copy ~80 us).  host is read before the call returns. */
int ns_upload_lengths(const int64_t* host, int n, int64_t* dev, void* stream);

/* Phase 2: LengthRegulator + frame-level pitch/energy + MelDecoder + mel_linear + PostNet (+ residual).
 * T is max(mel_lens), or a caller-chosen capacity (max_mel_len, model/modules.py:128-131,204-213 semantics: the mel axis is
 * padded and masked to it).  mel_lens stays on the device: a caller that fixes T up front can enqueue this call right
 * behind ns_forward_durations with NO host read in between (capacity mode).
 * Writes mel [B,T,n_mel], postnet_mel [B,T,n_mel], p_pred [B,T], e_pred [B,T], mel_mask [B,T] and
 * status [B] (int32, device or pinned host memory, REQUIRED): per utterance a bit set of
 *   NS_STATUS_TRUNCATED  mel_lens[b] > T: the frames past T were cut off (the rest of the row is still well defined)
 *   NS_STATUS_BAD_TOKEN  phase 1 reported a token id outside [0, n_vocab) for this utterance (mel_lens[b] = -1)
 * so that neither condition can pass silently when the caller never reads mel_lens before this call. */
#define NS_STATUS_TRUNCATED 1
#define NS_STATUS_BAD_TOKEN 2
/* p_targets / e_targets ([B,T], nullable): forward()'s p_targets / e_targets — when given, the embedding is
 * taken from bucketize(target) and the prediction is returned unscaled (model/modules.py:82-84,93-95). */
int ns_forward_mel(ns_model* m, int B, int L, int T, const int64_t* mel_lens, float p_control, float e_control,
                   const float* p_targets, const float* e_targets, const void* ws_enc, void* ws_dec, size_t ws_dec_bytes,
                   float* mel, float* postnet_mel, float* p_pred, float* e_pred, uint8_t* mel_mask, int32_t* status, void* stream);

/* Phase 2 on PACKED rows, for variable-length batches.  The reference computes every frame of the padded [B, T] grid and then
 * zeroes or ignores the frames past each utterance's length (transformer/Layers.py:43,46, model/modules.py:283-284); this
 * entry point keeps, per utterance, only a window of min(mel_lens[b] + 20, T) frames, lays the windows end to end and runs the
 * same kernels on sum(windows) rows instead of B*T.  Same arguments, same padded outputs: valid frames are computed by the
 * same arithmetic (they differ from ns_forward_mel's only by fp32 summation order where a launch picks another tile shape for
 * the smaller problem), padded frames are rebuilt — mel = the mel_linear bias, predictions 0, PostNet frames within 10 of an
 * utterance's end computed, all others constants of the weights (the PostNet has no mask between its layers).
 * mel_lens_host: the caller's HOST copy of mel_lens (what ns_forward_durations wrote to its mel_lens_host): the row count of
 * the launches comes from it, so this is the synchronous path's entry point.  The call falls back to the dense grid by
 * itself when packing does not apply (bf16x3 mode) or saves less than 10 % of the rows (20 % on grids of up to 20 000 rows, where the grid is one full round of the tallest tile). */
int ns_forward_mel_packed(ns_model* m, int B, int L, int T, const int64_t* mel_lens, const int64_t* mel_lens_host,
                          float p_control, float e_control, const float* p_targets, const float* e_targets, const void* ws_enc,
                          void* ws_dec, size_t ws_dec_bytes, float* mel, float* postnet_mel, float* p_pred, float* e_pred,
                          uint8_t* mel_mask, int32_t* status, void* stream);
/* Activation rows phase 2 of the most recent ns_forward_mel / ns_forward_mel_packed call on this model ran on: B*T on the
 * dense grid, the sum of the windows when packed (measurement / tests). */
int64_t ns_last_phase2_rows(const ns_model* m);

/* ---- per-operator entry points (the rows of SURVEY.md §8a; used by the parity tests and by bench.py's
 *      dominant-kernel timing).  `prefix` is the reference module path, e.g.
 *      "mel_decoder.layer_stack.0.slf_attn".  lens[b] = valid length of utterance b (keys/rows >= it are padding). */
size_t ns_op_ws_bytes(const ns_model* m, int B, int S);
/* a1  utils/tools.py:89-97 */
int ns_op_mask_from_lengths(const int64_t* lens, int B, int max_len, uint8_t* mask, void* stream);
/* a2  transformer/Models.py:10-30 */
int ns_op_sinusoid_table(int n_position, int d_hid, float* out, void* stream);
/* a3  transformer/Models.py:73-100 */
int ns_op_txt_encoder(ns_model* m, const int64_t* texts, const int64_t* lens, int B, int L, float* out, void* ws, size_t ws_bytes, void* stream);
/* a4+a5  transformer/SubLayers.py:29-59, transformer/Modules.py:14-25 */
int ns_op_multi_head_attention(ns_model* m, const char* prefix, const float* x, const int64_t* lens, int B, int S, float* out, void* ws, size_t ws_bytes, void* stream);
/* a6  transformer/SubLayers.py:87-95 */
int ns_op_positionwise_ffn(ns_model* m, const char* prefix, const float* x, int B, int S, float* out, void* ws, size_t ws_bytes, void* stream);
/* a7  transformer/Layers.py:39-48 */
int ns_op_fft_block(ns_model* m, const char* prefix, const float* x, const int64_t* lens, int B, int S, float* out, void* ws, size_t ws_bytes, void* stream);
/* a8  model/modules.py:278-286 */
int ns_op_variance_predictor(ns_model* m, const char* prefix, const float* x, const int64_t* lens, int B, int S, float* out, void* ws, size_t ws_bytes, void* stream);
/* a9  model/modules.py:132-135 */
int ns_op_duration_round(const float* log_d, int n, float d_control, float* d_rounded, void* stream);
/* a10 model/modules.py:201-230 + utils/tools.py:288-306: step 1 prefix sums + mel_lens, step 2 gather to [B,T,D] */
int ns_op_duration_scan(const float* d_rounded, int B, int L, int32_t* cum, int64_t* mel_lens, void* stream);
int ns_op_length_regulate(const float* x, const int32_t* cum, int B, int L, int D, int T, float* out, void* stream);
/* a11 model/modules.py:80-100,139-149: which = 0 pitch, 1 energy; x_out = x + embedding[bucketize(pred*control)] (unmasked add) */
int ns_op_variance_embedding(ns_model* m, int which, const float* x, const int64_t* lens, int B, int S, float control, const float* target /* nullable */, float* pred, float* x_out, void* ws, size_t ws_bytes, void* stream);
/* a11 torch.bucketize(values, bins, right=False) as called at model/modules.py:86-88,97-99 (the same device routine
 * the fused ns_op_variance_embedding kernel uses); idx[i] in [0, n_edges] */
int ns_op_bucketize(const float* values, int n, const float* bins, int n_edges, int64_t* idx, void* stream);
/* a12 model/modules.py:166-192 (dead code in the reference forward): out [B,T_out,D] (rows >= T zero),
 * w [B,L,T] (may be NULL), s: B*(L+1) floats — s[0..B) = sum of durations, the rest is scratch for the Gaussian centres */
int ns_op_gaussian_upsampling(const float* x, const float* durations, int B, int L, int D, int T, int T_out, float* out, float* s, float* w, void* stream);
/* a13 transformer/Models.py:212-244 */
int ns_op_mel_decoder(ns_model* m, const float* x, const int64_t* lens, int B, int T, float* out, void* ws, size_t ws_bytes, void* stream);
/* a14 model/fastspeech2_align.py:24-27,83 */
int ns_op_mel_linear(ns_model* m, const float* x, int B, int T, float* out, void* stream);
/* a15 transformer/Layers.py:169-177 (returns postnet(x) WITHOUT the residual, like PostNet.forward) */
int ns_op_postnet(ns_model* m, const float* mel, int B, int T, float* out, void* ws, size_t ws_bytes, void* stream);

/* bench.py's roofline leg: launches only the dominant kernel (the FFN k=9 Conv1D-as-GEMM of `prefix`.w_1,
 * bias+ReLU epilogue) on [B,S,d] -> [B,S,d_inner]; flops = 2*B*S*k*d*d_inner. */
int ns_op_ffn_conv1(ns_model* m, const char* prefix, const float* x, int B, int S, float* hidden, void* stream);
/* a5 alone (transformer/Modules.py:14-25 on already projected heads): qkv [B*S, 3*H*dk] with columns [0,d) = Q,
 * [d,2d) = K, [2d,3d) = V (head h at h*dk inside each), out [B*S, H*dk] = merged heads of
 * softmax(Q K^T / sqrt(dk) + (-inf at keys >= lens[b])) V.  dk in {32, 64, 128}. */
/* scratch (nullable): device memory for the split-key path the kernel takes when a launch has few workgroups
 * (single-utterance latency); 8 * (B*S*H*dk + 2*B*S*H) floats always suffice. */
int ns_op_attention_core(const float* qkv, const int64_t* lens, int B, int S, int H, int dk, float* out, void* scratch,
                         size_t scratch_bytes, void* stream);

/* Introspection of the step-aware launch plan (csrc/gemm_conv.hip plan_rows, csrc/attention.hip plan_key_split); host-side, no GPU
 * needed.  ns_plan_gemm: how a plain Conv1D-as-GEMM of M rows, N output channels, kernel size KW over Cin input channels
 * (transformer/SubLayers.py:87-95 over an arbitrary B*T) is launched: out = {BM, BN, rows} of the main launch, {BM, BN, rows} of
 * the remainder launch (zeros: a single launch), the edge of the MFMA tile the launch(es) are built from — 32, or 16 for the 16-row
 * family whose tile HEIGHT (any multiple of 16) is chosen for the row count — and the cost model's estimate in microseconds.
 * Returns 1 when the planner covers the shape, 0 when it is left to the small-grid K-split ladder (few tiles) or the
 * narrow-channel rules (then out is zeroed).  All rows of one GEMM are summed in one order: a cut is only ever made between tiles
 * of the 32-row family, which share it, so it never shows in the bits.
 * ns_plan_row_tile: height of the full-row tile (GEMM + LayerNorm / predictor-tail epilogue, N = 256 or 512 = one activation
 * row) for M rows: 32, or 48 / 80 / 112 (16-row family) when that gives the fullest CU fewer rows; 0 for other widths.
 * ns_plan_attention_split: key ranges per 128-query tile of a dense attention launch (1 = none; 16 = the small-grid paths' own
 * sizing, the value the workspace is reserved for). */
int ns_plan_gemm(int M, int N, int Cin, int KW, int32_t out[8]);
int ns_plan_row_tile(int M, int N);            /* = ns_plan_row_tile_k(M, N, N): the attention output projection's contraction */
int ns_plan_row_tile_k(int M, int N, int K);   /* contraction length K = KW * Cin: K > 256 keeps to the heights with chunked accumulation */
/* k values per accumulation chunk of the long contractions (K > 256; csrc/gemm_conv.hip ACC2): partial sums of this many
 * products are formed from zero and then added to the running total, so that the matrix cores round against short sums.
 * 64 by default; NS_ACC_CHUNK in the environment (a multiple of 64, or 0 = one sequential sum per output: A/B runs). */
int ns_acc_chunk(void);
/* Version of this header's contract (NS_ABI_VERSION below): bumped whenever a signature, an output-array length or a struct
 * layout changes, so that a caller built against an older header can refuse to run instead of overrunning a buffer
 * (round 5 grew ns_plan_gemm's out[6] to out[8]). */
int ns_abi_version(void);
int ns_plan_attention_split(int B, int S, int H, int dk);

/* Measurement hook for bench.py's roofline legs: while enabled, the launches of the three heaviest kernels inside
 * ns_forward_mel carry hipEvents ON THEIR OWN DISPATCH PACKETS (hipExtLaunchKernel start / stop events: the kernel's begin and
 * end timestamps, no marker packet on the stream; a timed launch still costs the stream ~5 us, so bench.py times slot 0 inside
 * its timed region and the other two in a separate pass), one slot each:
 *   slot 0  the FFT blocks' k=9 Conv1D-as-GEMM (PositionwiseFeedForward.w_1, transformer/SubLayers.py:70-75; the dominant
 *           kernel): flops = 2*rows*k*d*d_inner per launch
 *   slot 1  the fused attention (transformer/Modules.py:14-25): flops = 4*rows*T*d per launch
 *   slot 2  the PostNet's 512->512 k=5 convolutions (transformer/Layers.py:120-152): flops = 2*rows*k*512*512 per launch
 * ns_profile_read_slot waits for the slot's events and returns the summed kernel time, the summed algorithmic flops and
 * the launch count, then resets the slot.  ns_profile_read is slot 0. */
#define NS_PROFILE_SLOTS 3
#define NS_PROFILE_OFF 0
#define NS_PROFILE_ALL 1
#define NS_PROFILE_SLOT(i) (2 << (i)) /* OR several together to time exactly those slots */
#define NS_PROFILE_KEEP 0x10000        /* OR into `on`: keep the events recorded so far (bench.py times every 5th forward of its
                                          timed region: NS_PROFILE_KEEP | NS_PROFILE_SLOT(0) on those, NS_PROFILE_KEEP alone between) */
int ns_profile_enable(ns_model* m, int on);
int ns_profile_read(ns_model* m, double* total_ms, double* total_flops, int64_t* launches);
int ns_profile_read_slot(ns_model* m, int slot, double* total_ms, double* total_flops, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* NAR_FS2_H */
