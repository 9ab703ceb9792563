// C-ABI (include/nar_fs2.h) over the gfx950 kernels: weight registry/packing, workspace planning and the
// launch sequence of the FastSpeech2Align inference forward (model/fastspeech2_align.py:30-100).
// Host-side only; every byte of device memory is provided by the caller.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/nar_fs2.h"
#include "kernels.h"

using namespace ns;

static thread_local std::string g_err;
static int fail(const std::string& s) { g_err = s; return 1; }
#define NS_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)
#define NS_TRY(expr)              \
  do {                            \
    int rc_ = (expr);             \
    if (rc_) return rc_;          \
  } while (0)

namespace {

struct LayerW { size_t qkv_w, qkv_b, fc_w, fc_b, ln1_g, ln1_b, w1, w1_b, w2, w2_b, ln2_g, ln2_b, qkv_b3, fc_b3, w1_b3, w2_b3; };  // *_b3: bf16x3 planes or NO_B3
struct PredW { size_t c1, c1_b, ln1_g, ln1_b, c2, c2_b, ln2_g, ln2_b, lin_w, lin_b; int cin; };
struct PostW { size_t w, b, w_b3; int cin, cout; };
constexpr size_t NO_B3 = (size_t)-1;

struct Arena {
  size_t n = 0;  // floats
  size_t take(size_t floats) { size_t o = n; n += (floats + 63) & ~(size_t)63; return o; }
};

struct Staged { std::vector<int64_t> shape; std::vector<float> data; bool set = false; bool optional = false; };

}  // namespace

struct ns_model {
  ns_config cfg;
  std::vector<LayerW> enc, dec;
  PredW pred[3];
  size_t hdr;  // arena header: magic, layout version, arena size, config hash (arena_header below)
  size_t emb, enc_pos, dec_pos, pitch_bins, energy_bins, pitch_emb, energy_emb, mel_w, mel_b;
  size_t pn_in, pn_hid, pn_const;  // derived at ns_finalize_weights: the PostNet over an all-padding utterance (packed rows, forward_mel)
  size_t pos_long;                 // derived at ns_finalize_weights: the sinusoid table regenerated for POS_LONG_ROWS positions (position_rows)
  std::vector<PostW> post;
  Arena ar;
  float* arena = nullptr;
  bool ready = false;
  std::map<std::string, Staged> staged;
  const float* P(size_t off) const { return arena + off; }
  // optional HIP-event timing of the three heaviest launch groups inside the real forward (bench.py's roofline legs):
  // slot 0 = FFN w_1 (k=9 Conv1D-as-GEMM, the dominant kernel), 1 = fused attention, 2 = PostNet 512->512 k=5 layers.
  // Measurement state: the only per-call state the MODEL carries (with the two row counts below) — the packed-row context of a
  // forward travels in its Scratch, not in a global — which is why one model serves one host thread at a time (nar_fs2.h).
  struct ProfSlot { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; double flops = 0.0; };
  long long last_rows = 0;  // rows phase 2 of the most recent ns_forward_mel[_packed] ran on (B*T, or the packed windows)
  long long last_rows1 = 0; // rows phase 1 of the most recent ns_forward_durations[_packed] ran on (B*L, or the packed phoneme rows)
  unsigned prof = 0;  // bit i: slot i is timed (ns_profile_enable)
  bool prof_active = false;
  ProfSlot prof_slot[NS_PROFILE_SLOTS];
};

static const char* kPredNames[3] = {"duration", "pitch", "energy"};

// The arena is position independent and travels between processes as bytes (ns_adopt_arena after an RCCL broadcast, or a file).
// Its first 64 bytes say what it is: a magic word, the LAYOUT VERSION of this library (bump it whenever an offset, a packing
// rule or a derived constant in the arena changes), the arena size and a hash of the configuration — so that bytes packed by
// another build, for another configuration, or never finalized are refused instead of silently misread.
constexpr uint32_t ARENA_MAGIC = 0x3246534eu;  // "NSF2"
constexpr uint32_t ARENA_LAYOUT_VERSION = 6;   // 6: config hash without the run-time switches (same offsets as 5, which hashed them); 5: long position table; 4: header added
constexpr int ARENA_HDR_WORDS = 16;
static void arena_header(const ns_model* m, uint32_t* w) {
  memset(w, 0, ARENA_HDR_WORDS * sizeof(uint32_t));
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the config struct — the fields that shape the arena: the run-time switches
  ns_config lay = m->cfg;               // (how epilogues are launched, when phase 1 packs) are zeroed, ranks may differ in them
  lay.row_epilogue = 0;
  lay.phase1_packing = 0;
  const unsigned char* cb = reinterpret_cast<const unsigned char*>(&lay);
  for (size_t i = 0; i < sizeof(ns_config); ++i) { h ^= cb[i]; h *= 1099511628211ull; }
  w[0] = ARENA_MAGIC; w[1] = ARENA_LAYOUT_VERSION;
  w[2] = (uint32_t)(m->ar.n & 0xffffffffu); w[3] = (uint32_t)((uint64_t)m->ar.n >> 32);
  w[4] = (uint32_t)(h & 0xffffffffu); w[5] = (uint32_t)(h >> 32);
  w[6] = 1;  // finalized (postnet constants computed)
}
// Sequences longer than max_seq_len get a position table regenerated on the fly in the reference (transformer/Models.py:82-87,
// 218-225: get_sinusoid_encoding_table for the whole length, per call).  The table is a pure function of (position, d), so it is
// generated ONCE per weight load for this many positions, by the same kernel the per-call path uses (bit-identical), and kept in
// the arena; only longer sequences still rebuild per call.  Config 2 (T_pad 1010 > 1000) saves a 6.6 us launch per forward.
constexpr int POS_LONG_ROWS = 8192;
constexpr int PN_CONST_ROWS = 32;  // synthetic all-padding utterance: rows [10, 22) are deep padding, [22, 32) see the end of the axis

static void expect(ns_model* m, const std::string& name, std::vector<int64_t> shape, bool optional = false) {
  Staged s; s.shape = std::move(shape); s.optional = optional; m->staged[name] = std::move(s);
}

static void plan_stack(ns_model* m, const char* prefix, int n_layer, int d, std::vector<LayerW>& out, bool decoder) {
  const ns_config& c = m->cfg;
  for (int i = 0; i < n_layer; ++i) {
    std::string p = std::string(prefix) + ".layer_stack." + std::to_string(i);
    for (const char* w : {"w_qs", "w_ks", "w_vs", "fc"}) {
      expect(m, p + ".slf_attn." + w + ".weight", {d, d});
      expect(m, p + ".slf_attn." + w + ".bias", {d});
    }
    expect(m, p + ".slf_attn.layer_norm.weight", {d});
    expect(m, p + ".slf_attn.layer_norm.bias", {d});
    expect(m, p + ".pos_ffn.w_1.weight", {c.d_inner, d, c.ffn_k1});
    expect(m, p + ".pos_ffn.w_1.bias", {c.d_inner});
    expect(m, p + ".pos_ffn.w_2.weight", {d, c.d_inner, c.ffn_k2});
    expect(m, p + ".pos_ffn.w_2.bias", {d});
    expect(m, p + ".pos_ffn.layer_norm.weight", {d});
    expect(m, p + ".pos_ffn.layer_norm.bias", {d});
    LayerW L;
    L.qkv_w = m->ar.take((size_t)3 * d * d); L.qkv_b = m->ar.take(3 * d);
    L.fc_w = m->ar.take((size_t)d * d); L.fc_b = m->ar.take(d);
    L.ln1_g = m->ar.take(d); L.ln1_b = m->ar.take(d);
    L.w1 = m->ar.take((size_t)c.d_inner * c.ffn_k1 * d); L.w1_b = m->ar.take(c.d_inner);
    L.w2 = m->ar.take((size_t)d * c.ffn_k2 * c.d_inner); L.w2_b = m->ar.take(d);
    L.ln2_g = m->ar.take(d); L.ln2_b = m->ar.take(d);
    // opt-in bf16x3 mode: three bf16 planes of the (packed) k=9 weights of the DECODER stack — 1.5x their fp32 size
    // (decoder stack only: everything upstream of the duration / pitch / energy decisions stays exact fp32)
    const bool b3 = c.matmul_bf16x3 && decoder;
    auto planes = [&](size_t n) { return b3 ? m->ar.take((3 * n + 1) / 2) : NO_B3; };
    L.qkv_b3 = planes((size_t)3 * d * d);
    L.fc_b3 = planes((size_t)d * d);
    L.w1_b3 = planes((size_t)c.d_inner * c.ffn_k1 * d);
    L.w2_b3 = planes((size_t)d * c.ffn_k2 * c.d_inner);
    out.push_back(L);
  }
}

extern "C" const char* ns_last_error(void) { return g_err.c_str(); }

extern "C" int ns_create(const ns_config* cfg, ns_model** out) {
  if (!cfg || !out) return fail("ns_create: null argument");
  const ns_config& c = *cfg;
  if (c.d_enc != c.d_dec) return fail("ns_create: encoder_hidden != decoder_hidden is not supported (the variance adaptor adds encoder-width embeddings to the decoder input)");
  if (c.d_enc % 32 || c.d_inner % 32 || c.vp_filter % 32 || c.postnet_dim % 32 || c.n_mel % 16)
    return fail("ns_create: channel widths must be multiples of 32 (n_mel: 16)");
  for (int pair = 0; pair < 2; ++pair) {
    const int d = pair ? c.d_dec : c.d_enc, h = pair ? c.n_dec_head : c.n_enc_head;
    if (h <= 0 || d % h) return fail("ns_create: hidden size not divisible by head count");
    const int dk = d / h;
    if (dk != 32 && dk != 64 && dk != 128) return fail("ns_create: d_k must be 32, 64 or 128");
  }
  if (!(c.ffn_k1 & 1) || !(c.ffn_k2 & 1) || !(c.vp_kernel & 1) || !(c.postnet_k & 1))
    return fail("ns_create: kernel sizes must be odd");
  if (c.length_regulator != 0 && c.length_regulator != 1) return fail("ns_create: length_regulator must be 0 (hard) or 1 (gaussian)");
  if (c.matmul_bf16x3 != 0 && c.matmul_bf16x3 != 1) return fail("ns_create: matmul_bf16x3 must be 0 (fp32) or 1 (bf16x3)");
  if (c.row_epilogue != 0 && c.row_epilogue != 1) return fail("ns_create: row_epilogue must be 0 (fused) or 1 (two_launch)");
  if (c.phase1_packing < 0 || c.phase1_packing > 2) return fail("ns_create: phase1_packing must be 0 (auto), 1 (always) or 2 (never)");
  if (c.vp_kernel != 3) return fail("ns_create: variance predictor conv1d_2 hard-codes padding=1 (model/modules.py:267); kernel_size must be 3");
  ns_model* m = new ns_model();
  m->cfg = c;
  const int d = c.d_enc, npos = c.max_seq_len + 1;
  m->hdr = m->ar.take(ARENA_HDR_WORDS);
  expect(m, "txt_encoder.src_word_emb.weight", {c.n_vocab, d});
  expect(m, "txt_encoder.position_enc", {1, npos, d}, true);
  expect(m, "mel_decoder.position_enc", {1, npos, c.d_dec}, true);
  m->emb = m->ar.take((size_t)c.n_vocab * d);
  m->enc_pos = m->ar.take((size_t)npos * d);
  m->dec_pos = m->ar.take((size_t)npos * c.d_dec);
  plan_stack(m, "txt_encoder", c.n_enc_layer, d, m->enc, false);
  plan_stack(m, "mel_decoder", c.n_dec_layer, c.d_dec, m->dec, true);
  const int F = c.vp_filter, K = c.vp_kernel;
  for (int i = 0; i < 3; ++i) {
    std::string p = std::string("variance_adaptor.") + kPredNames[i] + "_predictor";
    expect(m, p + ".conv_layer.conv1d_1.conv.weight", {F, d, K});
    expect(m, p + ".conv_layer.conv1d_1.conv.bias", {F});
    expect(m, p + ".conv_layer.layer_norm_1.weight", {F});
    expect(m, p + ".conv_layer.layer_norm_1.bias", {F});
    expect(m, p + ".conv_layer.conv1d_2.conv.weight", {F, F, K});
    expect(m, p + ".conv_layer.conv1d_2.conv.bias", {F});
    expect(m, p + ".conv_layer.layer_norm_2.weight", {F});
    expect(m, p + ".conv_layer.layer_norm_2.bias", {F});
    expect(m, p + ".linear_layer.weight", {1, F});
    expect(m, p + ".linear_layer.bias", {1});
    PredW& w = m->pred[i];
    w.cin = d;
    w.c1 = m->ar.take((size_t)F * K * d); w.c1_b = m->ar.take(F);
    w.ln1_g = m->ar.take(F); w.ln1_b = m->ar.take(F);
    w.c2 = m->ar.take((size_t)F * K * F); w.c2_b = m->ar.take(F);
    w.ln2_g = m->ar.take(F); w.ln2_b = m->ar.take(F);
    w.lin_w = m->ar.take(F); w.lin_b = m->ar.take(1);
  }
  expect(m, "variance_adaptor.pitch_bins", {c.n_bins - 1});
  expect(m, "variance_adaptor.energy_bins", {c.n_bins - 1});
  expect(m, "variance_adaptor.pitch_embedding.weight", {c.n_bins, d});
  expect(m, "variance_adaptor.energy_embedding.weight", {c.n_bins, d});
  m->pitch_bins = m->ar.take(c.n_bins); m->energy_bins = m->ar.take(c.n_bins);
  m->pitch_emb = m->ar.take((size_t)c.n_bins * d); m->energy_emb = m->ar.take((size_t)c.n_bins * d);
  expect(m, "mel_linear.weight", {c.n_mel, c.d_dec});
  expect(m, "mel_linear.bias", {c.n_mel});
  m->mel_w = m->ar.take((size_t)c.n_mel * c.d_dec); m->mel_b = m->ar.take(c.n_mel);
  for (int i = 0; i < c.postnet_n; ++i) {
    const int cin = i == 0 ? c.n_mel : c.postnet_dim, cout = i == c.postnet_n - 1 ? c.n_mel : c.postnet_dim;
    std::string p = "postnet.convolutions." + std::to_string(i);
    expect(m, p + ".0.conv.weight", {cout, cin, c.postnet_k});
    expect(m, p + ".0.conv.bias", {cout});
    for (const char* s : {"weight", "bias", "running_mean", "running_var"}) expect(m, p + ".1." + s, {cout});
    PostW w; w.cin = cin; w.cout = cout;
    w.w = m->ar.take((size_t)cout * c.postnet_k * cin); w.b = m->ar.take(cout);
    w.w_b3 = (c.matmul_bf16x3 && cin == c.postnet_dim && cout == c.postnet_dim) ? m->ar.take(((size_t)3 * cout * c.postnet_k * cin + 1) / 2) : NO_B3;
    m->post.push_back(w);
  }
  // PostNet constants for packed rows (forward_mel): input, ping-pong scratch and output of one PostNet run over
  // PN_CONST_ROWS all-padding frames; part of the arena so that they travel with the weights (ns_adopt_arena)
  m->pn_in = m->ar.take((size_t)PN_CONST_ROWS * c.n_mel);
  m->pn_hid = m->ar.take((size_t)2 * PN_CONST_ROWS * c.postnet_dim);
  m->pn_const = m->ar.take((size_t)PN_CONST_ROWS * c.n_mel);
  m->pos_long = m->ar.take((size_t)POS_LONG_ROWS * c.d_dec);  // (d_enc == d_dec: one table serves both stacks)
  *out = m;
  return 0;
}

extern "C" void ns_destroy(ns_model* m) {
  if (!m) return;
  for (auto& ps : m->prof_slot)
    for (auto& ev : ps.ev) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  delete m;
}
extern "C" size_t ns_arena_bytes(const ns_model* m) { return m ? m->ar.n * sizeof(float) : 0; }

extern "C" int ns_bind_arena(ns_model* m, void* dev, size_t bytes) {
  if (!m || !dev) return fail("ns_bind_arena: null argument");
  if (bytes < ns_arena_bytes(m)) return fail("ns_bind_arena: arena too small");
  if ((uintptr_t)dev & 255) return fail("ns_bind_arena: arena must be 256-byte aligned");
  m->arena = (float*)dev;
  m->ready = false;
  return 0;
}

extern "C" int ns_adopt_arena(ns_model* m) {
  if (!m || !m->arena) return fail("ns_adopt_arena: no arena bound");
  // the bytes may have arrived on any stream (an RCCL broadcast): wait for the device, then read the header back
  // (on the device that OWNS the arena, which need not be the caller's current one; the caller's device is restored)
  uint32_t got[ARENA_HDR_WORDS], want[ARENA_HDR_WORDS];
  int cur_dev = 0;
  NS_HIP(hipGetDevice(&cur_dev));
  hipPointerAttribute_t at;
  NS_HIP(hipPointerGetAttributes(&at, m->arena));
  struct DevGuard { int d; bool on; ~DevGuard() { if (on) (void)hipSetDevice(d); } } guard{cur_dev, at.device != cur_dev};
  if (guard.on) NS_HIP(hipSetDevice(at.device));
  NS_HIP(hipDeviceSynchronize());
  NS_HIP(hipMemcpy(got, m->arena + m->hdr, sizeof(got), hipMemcpyDeviceToHost));
  arena_header(m, want);
  if (got[0] != ARENA_MAGIC) return fail("ns_adopt_arena: the arena does not start with this library's magic word (not a finalized ns_model arena)");
  if (got[1] != want[1]) return fail("ns_adopt_arena: arena layout version " + std::to_string(got[1]) + ", this library packs version " + std::to_string(want[1]));
  if (got[2] != want[2] || got[3] != want[3]) return fail("ns_adopt_arena: arena size differs from ns_arena_bytes of this model");
  if (got[4] != want[4] || got[5] != want[5]) return fail("ns_adopt_arena: the arena was packed for a different ns_config");
  if (got[6] != 1) return fail("ns_adopt_arena: the arena was not finalized (ns_finalize_weights did not complete)");
  for (auto& kv : m->staged) { kv.second.data.clear(); kv.second.data.shrink_to_fit(); kv.second.set = false; }
  m->ready = true;
  return 0;
}

static bool starts_with(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }
static bool ends_with(const std::string& s, const char* p) {
  size_t n = strlen(p); return s.size() >= n && s.compare(s.size() - n, n, p) == 0;
}

// name / rank / shape validation shared by ns_check_weight (no side effect) and ns_set_weight; *slot = nullptr for an ignored key
static int lookup_weight(ns_model* m, const char* name_c, const int64_t* shape, int ndim, Staged** slot, size_t* count, const char* who) {
  *slot = nullptr;
  if (!m || !name_c) return fail(std::string(who) + ": null argument");
  std::string name(name_c);
  // training-only aligner weights live in every checkpoint; accept and ignore
  if (starts_with(name, "mel_encoder.") || ends_with(name, ".num_batches_tracked")) return 0;
  auto it = m->staged.find(name);
  if (it == m->staged.end()) return fail(std::string(who) + ": unexpected key '" + name + "'");
  Staged& s = it->second;
  if ((int)s.shape.size() != ndim) return fail(std::string(who) + ": rank mismatch for '" + name + "'");
  if (ndim > 0 && !shape) return fail(std::string(who) + ": null shape for '" + name + "'");
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] != s.shape[i]) {
      return fail(std::string(who) + ": size mismatch for '" + name + "': dim " + std::to_string(i) + " is " +
                  std::to_string(shape[i]) + ", expected " + std::to_string(s.shape[i]));
    }
    n *= (size_t)shape[i];
  }
  *slot = &s;
  *count = n;
  return 0;
}

extern "C" int ns_check_weight(ns_model* m, const char* name, const int64_t* shape, int ndim) {
  Staged* s; size_t n;
  return lookup_weight(m, name, shape, ndim, &s, &n, "ns_check_weight");
}

extern "C" int ns_set_weight(ns_model* m, const char* name_c, const float* host, const int64_t* shape, int ndim) {
  Staged* s; size_t n;
  NS_TRY(lookup_weight(m, name_c, shape, ndim, &s, &n, "ns_set_weight"));
  if (!s) return 0;
  if (!host) return fail(std::string("ns_set_weight: null data for '") + name_c + "'");
  s->data.assign(host, host + n);
  s->set = true;
  m->ready = false;
  return 0;
}

// conv weight [out][in][k] (torch) -> [out][k][in] (tap-major K for the implicit GEMM), optional per-out scale
static void pack_conv(const std::vector<float>& w, int cout, int cin, int k, float* dst, const double* scale = nullptr) {
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < cin; ++c)
      for (int j = 0; j < k; ++j) {
        double v = w[((size_t)o * cin + c) * k + j];
        if (scale) v *= scale[o];
        dst[((size_t)o * k + j) * cin + c] = (float)v;
      }
}

static void host_sinusoid(int n_pos, int d, float* dst) {  // transformer/Models.py:10-30
  for (int p = 0; p < n_pos; ++p)
    for (int j = 0; j < d; ++j) {
      const double ang = (double)p / std::pow(10000.0, (double)(2 * (j / 2)) / (double)d);
      dst[(size_t)p * d + j] = (float)((j & 1) ? std::cos(ang) : std::sin(ang));
    }
}

static int postnet_constants(ns_model* m, hipStream_t st);  // (below, next to postnet())
extern "C" int ns_finalize_weights(ns_model* m, void* stream) {
  if (!m) return fail("ns_finalize_weights: null model");
  if (!m->arena) return fail("ns_finalize_weights: bind an arena first (ns_bind_arena)");
  std::string missing;
  for (auto& kv : m->staged)
    if (!kv.second.set && !kv.second.optional) missing += (missing.empty() ? "" : ", ") + kv.first;
  if (!missing.empty()) return fail("ns_finalize_weights: missing keys: " + missing);
  const ns_config& c = m->cfg;
  std::vector<float> img(m->ar.n, 0.f);
  auto S = [&](const std::string& k) -> const std::vector<float>& { return m->staged[k].data; };
  auto cp = [&](size_t off, const std::string& k) { const auto& v = S(k); memcpy(&img[off], v.data(), v.size() * sizeof(float)); };

  arena_header(m, reinterpret_cast<uint32_t*>(&img[m->hdr]));
  cp(m->emb, "txt_encoder.src_word_emb.weight");
  const int npos = c.max_seq_len + 1;
  if (m->staged["txt_encoder.position_enc"].set) cp(m->enc_pos, "txt_encoder.position_enc");
  else host_sinusoid(npos, c.d_enc, &img[m->enc_pos]);
  if (m->staged["mel_decoder.position_enc"].set) cp(m->dec_pos, "mel_decoder.position_enc");
  else host_sinusoid(npos, c.d_dec, &img[m->dec_pos]);

  auto do_stack = [&](const char* prefix, std::vector<LayerW>& Ls, int d) {
    for (size_t i = 0; i < Ls.size(); ++i) {
      const LayerW& L = Ls[i];
      std::string p = std::string(prefix) + ".layer_stack." + std::to_string(i);
      const char* qkv[3] = {"w_qs", "w_ks", "w_vs"};
      for (int t = 0; t < 3; ++t) {  // fused projection: rows [0,d) = Q, [d,2d) = K, [2d,3d) = V
        memcpy(&img[L.qkv_w + (size_t)t * d * d], S(p + ".slf_attn." + qkv[t] + ".weight").data(), (size_t)d * d * 4);
        memcpy(&img[L.qkv_b + (size_t)t * d], S(p + ".slf_attn." + qkv[t] + ".bias").data(), (size_t)d * 4);
      }
      cp(L.fc_w, p + ".slf_attn.fc.weight"); cp(L.fc_b, p + ".slf_attn.fc.bias");
      cp(L.ln1_g, p + ".slf_attn.layer_norm.weight"); cp(L.ln1_b, p + ".slf_attn.layer_norm.bias");
      pack_conv(S(p + ".pos_ffn.w_1.weight"), c.d_inner, d, c.ffn_k1, &img[L.w1]); cp(L.w1_b, p + ".pos_ffn.w_1.bias");
      cp(L.ln2_g, p + ".pos_ffn.layer_norm.weight"); cp(L.ln2_b, p + ".pos_ffn.layer_norm.bias");
      pack_conv(S(p + ".pos_ffn.w_2.weight"), d, c.d_inner, c.ffn_k2, &img[L.w2]); cp(L.w2_b, p + ".pos_ffn.w_2.bias");
      // opt-in bf16x3 mode: the SAME packed fp32 values, split exactly into three bf16 planes
      auto split = [&](size_t src, size_t dst, size_t n) {
        if (dst == NO_B3) return;
        unsigned short* pl = reinterpret_cast<unsigned short*>(&img[dst]);
        split_weights_b3(&img[src], n, pl, pl + n, pl + 2 * n);
      };
      split(L.qkv_w, L.qkv_b3, (size_t)3 * d * d);
      split(L.fc_w, L.fc_b3, (size_t)d * d);
      split(L.w1, L.w1_b3, (size_t)c.d_inner * c.ffn_k1 * d);
      split(L.w2, L.w2_b3, (size_t)d * c.ffn_k2 * c.d_inner);
    }
  };
  do_stack("txt_encoder", m->enc, c.d_enc);
  do_stack("mel_decoder", m->dec, c.d_dec);

  for (int i = 0; i < 3; ++i) {
    const PredW& w = m->pred[i];
    std::string p = std::string("variance_adaptor.") + kPredNames[i] + "_predictor";
    pack_conv(S(p + ".conv_layer.conv1d_1.conv.weight"), c.vp_filter, w.cin, c.vp_kernel, &img[w.c1]);
    cp(w.c1_b, p + ".conv_layer.conv1d_1.conv.bias");
    cp(w.ln1_g, p + ".conv_layer.layer_norm_1.weight"); cp(w.ln1_b, p + ".conv_layer.layer_norm_1.bias");
    pack_conv(S(p + ".conv_layer.conv1d_2.conv.weight"), c.vp_filter, c.vp_filter, c.vp_kernel, &img[w.c2]);
    cp(w.c2_b, p + ".conv_layer.conv1d_2.conv.bias");
    cp(w.ln2_g, p + ".conv_layer.layer_norm_2.weight"); cp(w.ln2_b, p + ".conv_layer.layer_norm_2.bias");
    cp(w.lin_w, p + ".linear_layer.weight"); cp(w.lin_b, p + ".linear_layer.bias");
  }
  cp(m->pitch_bins, "variance_adaptor.pitch_bins"); cp(m->energy_bins, "variance_adaptor.energy_bins");
  cp(m->pitch_emb, "variance_adaptor.pitch_embedding.weight"); cp(m->energy_emb, "variance_adaptor.energy_embedding.weight");
  cp(m->mel_w, "mel_linear.weight"); cp(m->mel_b, "mel_linear.bias");

  // PostNet: fold eval-mode BatchNorm1d (running stats, eps 1e-5) into the conv (transformer/Layers.py:120-167)
  for (size_t i = 0; i < m->post.size(); ++i) {
    const PostW& w = m->post[i];
    std::string p = "postnet.convolutions." + std::to_string(i);
    const auto &g = S(p + ".1.weight"), &b = S(p + ".1.bias"), &mu = S(p + ".1.running_mean"), &var = S(p + ".1.running_var");
    const auto& cb = S(p + ".0.conv.bias");
    std::vector<double> sc(w.cout);
    for (int o = 0; o < w.cout; ++o) {
      sc[o] = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
      img[w.b + o] = (float)(((double)cb[o] - (double)mu[o]) * sc[o] + (double)b[o]);
    }
    pack_conv(S(p + ".0.conv.weight"), w.cout, w.cin, c.postnet_k, &img[w.w], sc.data());
    if (w.w_b3 != NO_B3) {
      const size_t n = (size_t)w.cout * c.postnet_k * w.cin;
      unsigned short* pl = reinterpret_cast<unsigned short*>(&img[w.w_b3]);
      split_weights_b3(&img[w.w], n, pl, pl + n, pl + 2 * n);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  NS_HIP(hipMemcpyAsync(m->arena, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, st));
  NS_TRY(postnet_constants(m, st));
  NS_HIP(launch_sinusoid(POS_LONG_ROWS, c.d_dec, m->arena + m->pos_long, st));
  NS_HIP(hipStreamSynchronize(st));  // img is a local; also makes load_state_dict() synchronous like the reference's
  for (auto& kv : m->staged) { kv.second.data.clear(); kv.second.data.shrink_to_fit(); }
  m->ready = true;
  return 0;
}

// ------------------------------------------------------------------------------------------- workspace
namespace {
struct Bump {
  char* base; size_t off = 0, cap;
  Bump(void* p, size_t c) : base((char*)p), cap(c) {}
  float* f(size_t n) { return (float*)raw(n * sizeof(float)); }
  void* raw(size_t bytes) {
    size_t o = off; off += (bytes + 255) & ~(size_t)255;
    return base ? base + o : nullptr;
  }
};

// ticket counters of one forward phase (gemm_conv.hip TICKET, attention.hip): zeroed as a block by the phase's first kernel,
// every ticketed launch then takes its own slice — no reset, no reuse inside a phase
constexpr int TICKET_INTS = 16384;

struct PackedCtx { RowMap rm; int Mp; };

struct Scratch {  // per-stack temporaries for M rows
  const PackedCtx* pk;  // packed-row context of the phase this scratch serves (see cur_rm below); nullptr = dense grid
  float *xa, *xb, *qkv, *att, *t1, *x1, *hid, *vp1, *vp2, *pos_ext, *att_part;
  size_t att_part_floats;
  int* tickets; int tickets_used;
  int* take_tickets(int n) {  // nullptr when the block is spent (the caller then takes the two-launch form)
    if (!tickets || tickets_used + n > TICKET_INTS) return nullptr;
    int* t = tickets + tickets_used;
    tickets_used += n;
    return t;
  }
};

static size_t imax(size_t a, size_t b) { return a > b ? a : b; }

static Scratch carve(const ns_config& c, Bump& bp, size_t M, int S, bool no_split = false) {
  Scratch s;
  s.pk = nullptr;
  const size_t d = c.d_enc;
  s.xa = bp.f(M * d); s.xb = bp.f(M * d);
  s.qkv = bp.f(M * 3 * d); s.att = bp.f(M * d); s.t1 = bp.f(M * d); s.x1 = bp.f(M * d);
  s.hid = bp.f(M * imax(c.d_inner, 2 * (size_t)c.postnet_dim));
  s.vp1 = bp.f(M * c.vp_filter); s.vp2 = bp.f(M * c.vp_filter);
  s.pos_ext = S > c.max_seq_len ? bp.f((size_t)S * d) : nullptr;
  // attention's split-key partials: only launches with few workgroups take that path (kernels.h)
  // attention's split-key partials: few workgroups (small grids), or a last round of 256 that fills badly (attention.hip
  // plan_key_split) — as many key ranges as either stack's launch of this shape will ask for
  const int hmax = c.n_enc_head > c.n_dec_head ? c.n_enc_head : c.n_dec_head;
  int nsp = 1;
  if (!no_split && S > 0) {  // (packed rows size their own split: forward_mel)
    const int Bg = (int)(M / (size_t)S);
    const int ne = attention_split(Bg, S, c.n_enc_head, c.d_enc / c.n_enc_head), nd = attention_split(Bg, S, c.n_dec_head, c.d_dec / c.n_dec_head);
    nsp = ne > nd ? ne : nd;
  }
  s.att_part_floats = nsp > 1 ? (size_t)nsp * (M * d + 2 * M * hmax) : 0;
  s.att_part = s.att_part_floats ? bp.f(s.att_part_floats) : nullptr;
  s.tickets = (int*)bp.raw(TICKET_INTS * sizeof(int));
  if (c.row_epilogue == 1) s.tickets = nullptr;  // "two_launch": take_tickets() then always answers "spent" (the block stays reserved)
  s.tickets_used = 0;
  return s;
}
}  // namespace

extern "C" size_t ns_op_ws_bytes(const ns_model* m, int B, int S) {
  if (!m || B <= 0 || S <= 0) return 256;
  Bump bp(nullptr, 0);
  carve(m->cfg, bp, (size_t)B * S, S);
  return bp.off + 256;
}
static size_t phase1_packed_extra_bytes(const ns_config& c, int B, int L);
extern "C" size_t ns_encoder_ws_bytes(const ns_model* m, int B, int L) {
  if (!m || B <= 0 || L <= 0) return 256;
  Bump bp(nullptr, 0);
  bp.f((size_t)B * L * m->cfg.d_enc);  // encoder output (kept for phase 2)
  bp.raw((size_t)B * L * sizeof(int32_t));  // duration prefix sums (kept for phase 2)
  bp.f((size_t)B * L);                       // rounded durations (kept for phase 2: Gaussian length regulator)
  carve(m->cfg, bp, (size_t)B * L, L);
  return bp.off + phase1_packed_extra_bytes(m->cfg, B, L) + 256;  // (+ ns_forward_durations_packed's plan and packed outputs)
}
static size_t packed_extra_bytes(const ns_config& c, int B, int T);
extern "C" size_t ns_decoder_ws_bytes(const ns_model* m, int B, int L, int T) {
  (void)L;
  if (!m || B <= 0 || T <= 0) return ns_op_ws_bytes(m, B, T);
  return ns_op_ws_bytes(m, B, T) + packed_extra_bytes(m->cfg, B, T);  // (ns_forward_mel_packed's plan and staging; see forward_mel)
}

// ------------------------------------------------------------------------------------------- building blocks
static int check_ready(const ns_model* m) {
  if (!m) return fail("null model");
  if (!m->ready) return fail("weights not loaded: call ns_set_weight for every key, then ns_finalize_weights (or ns_adopt_arena)");
  return 0;
}

// Packed rows (kernels.h RowMap): a forward that runs a phase on packed rows hangs its context on that phase's Scratch (sc.pk);
// every GEMM, row kernel and attention launch issued with that Scratch addresses rows through the map, and "B utterances of S
// rows" means the Mp packed rows.  nullptr = the dense [B, S] grid.
static const RowMap* cur_rm(const Scratch& sc) { return sc.pk ? &sc.pk->rm : nullptr; }
static int rows_of(const Scratch& sc, int B, int S) { return sc.pk ? sc.pk->Mp : B * S; }

static int gemm(const Scratch& sc, const float* X, int ldx, const float* W, const float* bias, const float* resid, int ldr, float* Y, int ldy,
                int M, int N, int Cin, int KW, int S, int act, hipStream_t st, const RowEpilogue* epi = nullptr, int epi_mode = EPI_NONE,
                const unsigned short* Wb3 = nullptr, const LaunchTiming* tm = nullptr) {
  ConvGemm p;
  memset(&p, 0, sizeof(p));
  p.X = X; p.ldx = ldx; p.W = W; p.bias = bias; p.resid = resid; p.ldr = ldr; p.Y = Y; p.ldy = ldy;
  p.M = M; p.N = N; p.Cin = Cin; p.KW = KW; p.pad = (KW - 1) / 2; p.S = S; p.act = act;
  p.epi = epi ? epi_mode : EPI_NONE;
  if (epi) p.e = *epi;
  if (sc.pk) { p.rm = sc.pk->rm; p.e.row_b = sc.pk->rm.row_b; p.e.row_t = sc.pk->rm.row_t; }
  // opt-in bf16x3 planes exist for this weight AND the launch is large enough for the 128-row tiles: split-bf16 matrix cores
  if (Wb3 && conv_gemm_b3_ok(M, N, Cin, KW, p.epi)) {
    p.Wb3 = Wb3;
    if (tm && tm->start) NS_HIP(hipEventRecord(tm->start, st));  // (the experiment mode keeps marker events around its launch)
    NS_HIP(launch_conv_gemm_b3(p, st));
    if (tm && tm->stop) NS_HIP(hipEventRecord(tm->stop, st));
    return 0;
  }
  NS_HIP(launch_conv_gemm(p, st, tm));
  return 0;
}

// A GEMM whose N columns are one whole activation row can run the row's LayerNorm in its epilogue (kernels.h
// RowEpilogue).  The full-row tile is at least 32 rows tall (48 / 80 / 112 between the steps: gemm_conv.hip conv_gemm_row_tile), so it
// is taken once the launch has about a workgroup per CU;
// below that the many-small-tiles + split-K ladder followed by the row kernel is faster (tools/lab/gemm_lab_ln.hip:
// M=16160 K=1024 93 -> 85 us, K=256 39.5 -> 31 us; M=2048 K=1024 21.6 -> 43 us).
static bool fuse_row_epilogue(int M, int N, int Cin) {
  return conv_gemm_row_epilogue_ok(M, N, Cin) && (M + 31) / 32 >= 200;
}

// Y = mask(LayerNorm(act(conv(X)) + resid)) in ONE launch: the full-row tile when the launch is large enough, else the
// small-grid ladder with the ticketed row epilogue (raw rows through tmp, the last workgroup of a row block normalises it).
// Two launches (GEMM -> tmp -> k_layernorm) only in the opt-in bf16x3 mode for widths its LayerNorm tile does not cover,
// or when the phase's ticket block is spent.
static int gemm_ln(const float* X, int ldx, const float* W, const float* bias, const float* resid, float* tmp, float* Y,
                   int M, int N, int Cin, int KW, int S, int act, const float* g, const float* b, const long long* lens,
                   Scratch& sc, hipStream_t st, const unsigned short* Wb3 = nullptr) {
  RowEpilogue e;
  memset(&e, 0, sizeof(e));
  e.ln_g = g; e.ln_b = b; e.lens = lens;
  if (Wb3 && !conv_gemm_b3_ok(M, N, Cin, KW, EPI_LN) && conv_gemm_b3_ok(M, N, Cin, KW, EPI_NONE)) {
    NS_TRY(gemm(sc, X, ldx, W, bias, resid, N, tmp, N, M, N, Cin, KW, S, act, st, nullptr, EPI_NONE, Wb3));
    NS_HIP(launch_layernorm(tmp, g, b, Y, M, N, S, lens, st, cur_rm(sc)));
    return 0;
  }
  // (64x64 tiles with the ticketed epilogue in place of the full-row tile when its steps of 256 tiles fit the row count badly
  //  were measured in round 4 and lose: at 572 workgroups the last arrivers' row work costs +12 us for a LayerNorm and +25 us
  //  for a predictor tail, more than the finer steps save — B = 9: conv+LN 57 vs 56 us, conv+tail 80-86 vs 62, w_2 70.7 vs 70)
  if (fuse_row_epilogue(M, N, Cin)) return gemm(sc, X, ldx, W, bias, resid, N, Y, N, M, N, Cin, KW, S, act, st, &e, EPI_LN, Wb3);
  if (conv_gemm_ticket_ok(M, N, Cin) && (e.ticket = sc.take_tickets(conv_gemm_ticket_ints(M))) != nullptr) {
    e.y_out = Y;
    return gemm(sc, X, ldx, W, bias, resid, N, tmp, N, M, N, Cin, KW, S, act, st, &e, EPI_LN);
  }
  NS_TRY(gemm(sc, X, ldx, W, bias, resid, N, tmp, N, M, N, Cin, KW, S, act, st));
  NS_HIP(launch_layernorm(tmp, g, b, Y, M, N, S, lens, st, cur_rm(sc)));
  return 0;
}

// Timing of one launch group inside the real forward; a no-op unless ns_profile_enable(1) and the forward is inside its
// timed section (the decoder stack / PostNet of ns_forward_mel).  The events ride on the group's own dispatch packets
// (kernels.h LaunchTiming): no marker packet, no gap on the stream — marker pairs around the 11 timed launches of a forward
// cost it ~130 us (3 % at config 2, round 3's bench line paid that inside its timed region).
struct ProfScope {
  ns_model* m; int slot; double flops; bool on;
  LaunchTiming tm{nullptr, nullptr};
  ProfScope(ns_model* m_, int slot_, double flops_) : m(m_), slot(slot_), flops(flops_), on(m_->prof_active && ((m_->prof >> slot_) & 1u)) {}
  int begin() {
    if (!on) return 0;
    ns_model::ProfSlot& ps = m->prof_slot[slot];
    if (ps.used == ps.ev.size()) {
      hipEvent_t a, b;
      NS_HIP(hipEventCreate(&a));
      NS_HIP(hipEventCreate(&b));
      ps.ev.emplace_back(a, b);
    }
    tm = LaunchTiming{ps.ev[ps.used].first, ps.ev[ps.used].second};
    return 0;
  }
  const LaunchTiming* timing() const { return on ? &tm : nullptr; }
  void end() {
    if (!on) return;
    ns_model::ProfSlot& ps = m->prof_slot[slot];
    ps.used++;
    ps.flops += flops;
  }
};

// MultiHeadAttention.forward (transformer/SubLayers.py:29-59); out = LayerNorm(fc(attn) + x), masked when mask_rows
static int mha(ns_model* m, const LayerW& L, int d, int H, const float* x, const long long* lens, int B, int S,
               float* out, bool mask_rows, Scratch& sc, hipStream_t st) {
  const int M = rows_of(sc, B, S);
  auto b3 = [&](size_t off) { return off != NO_B3 ? reinterpret_cast<const unsigned short*>(m->P(off)) : nullptr; };
  NS_TRY(gemm(sc, x, d, m->P(L.qkv_w), m->P(L.qkv_b), nullptr, 0, sc.qkv, 3 * d, M, 3 * d, d, 1, S, ACT_NONE, st, nullptr, EPI_NONE, b3(L.qkv_b3)));
  {
    ProfScope ps(m, 1, 4.0 * (double)M * (double)S * (double)d);
    NS_TRY(ps.begin());
    NS_HIP(launch_attention(sc.qkv, lens, B, S, H, d / H, sc.att, sc.att_part, sc.att_part_floats,
                            (sc.att_part && attention_uses_tickets(B, S, H)) ? sc.take_tickets(attention_ticket_ints(B, S, H)) : nullptr, st, cur_rm(sc),
                            ps.timing()));
    ps.end();
  }
  return gemm_ln(sc.att, d, m->P(L.fc_w), m->P(L.fc_b), x, sc.t1, out, M, d, d, 1, S, ACT_NONE, m->P(L.ln1_g), m->P(L.ln1_b),
                 mask_rows ? lens : nullptr, sc, st, b3(L.fc_b3));
}

// PositionwiseFeedForward.forward (transformer/SubLayers.py:87-95)
static int ffn(ns_model* m, const LayerW& L, int d, const float* x, const long long* lens, int B, int S, float* out,
               bool mask_rows, Scratch& sc, hipStream_t st) {
  const ns_config& c = m->cfg;
  const int M = rows_of(sc, B, S);
  {
    ProfScope ps(m, 0, 2.0 * (double)M * (double)c.ffn_k1 * (double)d * (double)c.d_inner);
    NS_TRY(ps.begin());
    NS_TRY(gemm(sc, x, d, m->P(L.w1), m->P(L.w1_b), nullptr, 0, sc.hid, c.d_inner, M, c.d_inner, d, c.ffn_k1, S, ACT_RELU, st, nullptr, EPI_NONE,
                L.w1_b3 != NO_B3 ? reinterpret_cast<const unsigned short*>(m->P(L.w1_b3)) : nullptr, ps.timing()));
    ps.end();
  }
  return gemm_ln(sc.hid, c.d_inner, m->P(L.w2), m->P(L.w2_b), x, sc.t1, out, M, d, c.d_inner, c.ffn_k2, S, ACT_NONE, m->P(L.ln2_g),
                 m->P(L.ln2_b), mask_rows ? lens : nullptr, sc, st,
                 L.w2_b3 != NO_B3 ? reinterpret_cast<const unsigned short*>(m->P(L.w2_b3)) : nullptr);
}

// FFTBlock.forward (transformer/Layers.py:39-48): both masked_fill's are fused into the LayerNorm kernels
static int fft_block(ns_model* m, const LayerW& L, int d, int H, const float* x, const long long* lens, int B, int S,
                     float* out, Scratch& sc, hipStream_t st) {
  NS_TRY(mha(m, L, d, H, x, lens, B, S, sc.x1, true, sc, st));
  NS_TRY(ffn(m, L, d, sc.x1, lens, B, S, out, true, sc, st));
  return 0;
}

// position rows [0,S): cached parameter when S <= max_seq_len, else rebuilt (transformer/Models.py:82-91,218-235)
static int position_rows(const ns_model* m, size_t cached_off, int S, int d, Scratch& sc, const float** pos, hipStream_t st) {
  if (S > m->cfg.max_seq_len && S <= POS_LONG_ROWS) {
    *pos = m->P(m->pos_long);  // rows [0, S) of the regenerated table (its rows do not depend on how many were generated)
  } else if (S > m->cfg.max_seq_len) {
    NS_HIP(launch_sinusoid(S, d, sc.pos_ext, st));
    *pos = sc.pos_ext;
  } else {
    *pos = m->P(cached_off);
  }
  return 0;
}

// VariancePredictor.forward (model/modules.py:278-286) with the optional fused embedding add
static int predictor(const ns_model* m, const PredW& w, const float* x, const long long* lens, int B, int S, float control,
                     const float* target,
                     float* pred, const float* bins, const float* emb, const float* pos, float* x_out, Scratch& sc,
                     hipStream_t st) {
  const ns_config& c = m->cfg;
  const int M = rows_of(sc, B, S), F = c.vp_filter;
  // conv1d_1 -> relu -> layer_norm_1 (no mask between the layers: model/modules.py:245-274)
  NS_TRY(gemm_ln(x, w.cin, m->P(w.c1), m->P(w.c1_b), nullptr, sc.vp1, sc.vp2, M, F, w.cin, c.vp_kernel, S, ACT_RELU, m->P(w.ln1_g),
                 m->P(w.ln1_b), nullptr, sc, st));
  // conv1d_2 -> relu -> layer_norm_2 -> linear -> mask (-> bucketize + embedding add): the whole tail rides on conv1d_2's
  // row epilogue — the full-row tile when the launch is large, the ticketed form on small grids
  RowEpilogue e;
  memset(&e, 0, sizeof(e));
  e.ln_g = m->P(w.ln2_g); e.ln_b = m->P(w.ln2_b); e.lens = lens; e.wlin = m->P(w.lin_w); e.blin = m->P(w.lin_b); e.pred = pred;
  e.control = control; e.target = target; e.bins = bins; e.n_edges = c.n_bins - 1; e.emb = emb; e.x_in = x; e.pos = pos; e.x_out = x_out;
  e.D = w.cin;
  if (fuse_row_epilogue(M, F, F))
    return gemm(sc, sc.vp2, F, m->P(w.c2), m->P(w.c2_b), nullptr, 0, nullptr, F, M, F, F, c.vp_kernel, S, ACT_RELU, st, &e, EPI_LN_PRED);
  if (conv_gemm_ticket_ok(M, F, F) && (e.ticket = sc.take_tickets(conv_gemm_ticket_ints(M))) != nullptr)
    return gemm(sc, sc.vp2, F, m->P(w.c2), m->P(w.c2_b), nullptr, 0, sc.vp1, F, M, F, F, c.vp_kernel, S, ACT_RELU, st, &e, EPI_LN_PRED);
  NS_TRY(gemm(sc, sc.vp2, F, m->P(w.c2), m->P(w.c2_b), nullptr, 0, sc.vp1, F, M, F, F, c.vp_kernel, S, ACT_RELU, st));
  NS_HIP(launch_ln_linear_embed(sc.vp1, m->P(w.ln2_g), m->P(w.ln2_b), m->P(w.lin_w), m->P(w.lin_b), pred, M, F, S, lens,
                                control, target, bins, c.n_bins, emb, x, pos, x_out, w.cin, st, cur_rm(sc)));
  return 0;
}

// PostNet.forward (transformer/Layers.py:169-177); resid != nullptr adds `+ output` of fastspeech2_align.py:85
static int postnet(ns_model* m, const float* mel, int B, int T, const float* resid, float* out, Scratch& sc, hipStream_t st) {
  const ns_config& c = m->cfg;
  const int M = rows_of(sc, B, T);
  float* ping = sc.hid;
  float* pong = sc.hid + (size_t)M * c.postnet_dim;
  const float* cur = mel;
  int ld = c.n_mel;
  for (size_t i = 0; i < m->post.size(); ++i) {
    const PostW& w = m->post[i];
    const bool last = i + 1 == m->post.size();
    float* dst = last ? out : ((i & 1) ? pong : ping);
    const bool mid = w.cin == c.postnet_dim && w.cout == c.postnet_dim;
    ProfScope ps(m, 2, 2.0 * (double)M * (double)c.postnet_k * (double)w.cin * (double)w.cout);
    if (mid) NS_TRY(ps.begin());
    NS_TRY(gemm(sc, cur, ld, m->P(w.w), m->P(w.b), last ? resid : nullptr, c.n_mel, dst, w.cout, M, w.cout, w.cin, c.postnet_k, T,
                last ? ACT_NONE : ACT_TANH, st, nullptr, EPI_NONE,
                w.w_b3 != NO_B3 ? reinterpret_cast<const unsigned short*>(m->P(w.w_b3)) : nullptr, mid ? ps.timing() : nullptr));
    if (mid) ps.end();
    cur = dst;
    ld = w.cout;
  }
  return 0;
}

// PostNet constants for packed rows (rowops.hip k_unpack_outputs): the PostNet over an all-padding utterance — every input
// frame is the mel_linear bias (mel_linear of a zeroed decoder row, fastspeech2_align.py:83), zero padding at both ends of a
// 32-frame axis.  Frame 21 of the result is deep padding (its 10-frame reach sees neither end), frames 22..31 see the end of
// the axis.  Run once per weight load, into the arena.
static int postnet_constants(ns_model* m, hipStream_t st) {
  const ns_config& c = m->cfg;
  float* in = m->arena + m->pn_in;
  NS_HIP(launch_broadcast_row(m->P(m->mel_b), in, PN_CONST_ROWS, c.n_mel, st));
  Scratch sc;
  memset(&sc, 0, sizeof(sc));
  sc.hid = m->arena + m->pn_hid;
  return postnet(m, in, 1, PN_CONST_ROWS, in, m->arena + m->pn_const, sc, st);
}

static int encoder(ns_model* m, const long long* texts, const long long* lens, int B, int L, float* out, Scratch& sc,
                   hipStream_t st) {
  const ns_config& c = m->cfg;
  const int M = rows_of(sc, B, L), d = c.d_enc;
  const float* pos;
  NS_TRY(position_rows(m, m->enc_pos, L, d, sc, &pos, st));
  float* cur = m->enc.empty() ? out : sc.xa;
  // (+ zeroes the phase's tickets)
  if (sc.pk) NS_HIP(launch_embed_pos_packed(texts, m->P(m->emb), pos, cur, sc.pk->rm, B, M, L, d, c.n_vocab, sc.tickets, TICKET_INTS, st));
  else NS_HIP(launch_embed_pos(texts, m->P(m->emb), pos, cur, M, L, d, c.n_vocab, sc.tickets, TICKET_INTS, st));
  for (size_t i = 0; i < m->enc.size(); ++i) {
    float* dst = (i + 1 == m->enc.size()) ? out : (cur == sc.xa ? sc.xb : sc.xa);
    NS_TRY(fft_block(m, m->enc[i], d, c.n_enc_head, cur, lens, B, L, dst, sc, st));
    cur = dst;
  }
  return 0;
}

// MelDecoder's layer stack on an input that already carries the position rows
static int decoder_stack(ns_model* m, float* x, const long long* lens, int B, int T, float* out, Scratch& sc, hipStream_t st) {
  const ns_config& c = m->cfg;
  float* cur = x;
  float* alt = (x == sc.xa) ? sc.xb : sc.xa;
  for (size_t i = 0; i < m->dec.size(); ++i) {
    float* dst = (i + 1 == m->dec.size()) ? out : alt;
    NS_TRY(fft_block(m, m->dec[i], c.d_dec, c.n_dec_head, cur, lens, B, T, dst, sc, st));
    if (dst != out) { alt = cur; cur = dst; }
  }
  if (m->dec.empty() && out != x) NS_HIP(hipMemcpyAsync(out, x, (size_t)rows_of(sc, B, T) * c.d_dec * 4, hipMemcpyDeviceToDevice, st));
  return 0;
}

// ------------------------------------------------------------------------------------------- the forward
// Phase 1: encoder -> [phoneme-level pitch / energy] -> duration predictor -> rounding, prefix sums, mel_lens.
// lens_host (nullable): the host copy of src_lens.  With it, and when the utterances' phoneme counts leave >= 10 % of the [B, L]
// grid as padding, everything up to the duration predictor runs on PACKED phoneme rows (kernels.h PHONEME_GUARD: exact, see
// there) and the padded [B, L] tensors the caller and phase 2 expect are rebuilt before the duration tail.
static size_t phase1_packed_rows(const int64_t* lens_host, int B, int L) {
  size_t mp = 0;
  for (int b = 0; b < B; ++b) {
    long long l = (lens_host[b] < 0 ? 0 : lens_host[b]) + PHONEME_GUARD;
    mp += (size_t)(l < (long long)L ? l : (long long)L);
  }
  return mp;
}
static size_t phase1_packed_extra_bytes(const ns_config& c, int B, int L) {  // plan + packed encoder output + packed log-durations
  Bump bp(nullptr, 0);
  const size_t M = (size_t)B * L;
  bp.raw(pack_plan_ints(B, M) * sizeof(int));
  bp.f(M * c.d_enc); bp.f(M);
  return bp.off;
}

static int forward_durations(ns_model* m, const int64_t* texts, const int64_t* src_lens, const int64_t* lens_host, int B, int L, float d_control,
                             float p_control, float e_control, const float* p_targets, const float* e_targets,
                             void* ws_enc, size_t ws_bytes, float* log_d, float* d_rounded, uint8_t* src_mask,
                             int64_t* mel_lens, float* p_pred, float* e_pred, int64_t* mel_lens_host, void* stream) {
  NS_TRY(check_ready(m));
  if (B <= 0 || L <= 0) return fail("ns_forward_durations: empty batch");
  if (ws_bytes < ns_encoder_ws_bytes(m, B, L)) return fail("ns_forward_durations: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const ns_config& c = m->cfg;
  if (!c.pitch_frame_level && !p_pred) return fail("ns_forward_durations: phoneme_level pitch needs a p_pred [B,L] output");
  if (!c.energy_frame_level && !e_pred) return fail("ns_forward_durations: phoneme_level energy needs an e_pred [B,L] output");
  Bump bp(ws_enc, ws_bytes);
  float* enc_out = bp.f((size_t)B * L * c.d_enc);
  int32_t* cum = (int32_t*)bp.raw((size_t)B * L * sizeof(int32_t));
  float* dur_keep = bp.f((size_t)B * L);
  const long long* lens = (const long long*)src_lens;
  size_t Mp = 0;
  // (phoneme-level pitch / energy add an embedding row to PADDED phonemes too — model/modules.py:117-126, an unmasked add whose
  //  index comes from the caller's target at that very position — so those configurations keep the grid; the shipped one is frame_level)
  bool packed = lens_host && !m->enc.empty() && launch_planner_enabled() && c.pitch_frame_level && c.energy_frame_level && c.phase1_packing != 2;
  if (packed) {
    // Phase-1 launches are small grids: their time is steps of 256 workgroups x a K loop, not rows (measured, config-2 shape with
    // 0.66 of the rows: 4.32 ms packed vs 4.29 ms on the grid — the same two steps everywhere, plus the plan / unpack / attention-merge
    // launches).  So pack only when the rows drop by a whole step of the launch that dominates the phase, the FFN k=9 GEMM on
    // 32x128 tiles (4 x ~38 us per step saved against ~50 us of extra launches) — or from two steps to one of its 48-row form
    // (4 x ~18 us) — and by at least 10 %.
    Mp = phase1_packed_rows(lens_host, B, L);
    // rows the fullest CU gets from that launch on the small-grid ladder: rounds of the 32 x 128 rung, or — round 5 — ONE round of its
    // 48-row form when that fits (the ragged config-2 shape: 1388 packed rows = 232 workgroups of 48 rows against 512 of 32 on the grid)
    auto cu_rows = [&](size_t rows) {
      const size_t ntn = (size_t)c.d_inner / 128, w32 = (rows + 31) / 32 * ntn, w48 = (rows + 47) / 48 * ntn;
      return (conv_gemm_tile16_enabled() && w32 > 256 && w48 <= 256) ? (size_t)48 : 32 * ((w32 + 255) / 256);
    };
    packed = Mp > 0 && Mp * 10 <= (size_t)B * L * 9 && (c.phase1_packing == 1 || cu_rows(Mp) < cu_rows((size_t)B * L));
  }
  const size_t Mrows = packed ? Mp : (size_t)B * L;
  m->last_rows1 = (long long)Mrows;
  Scratch sc = carve(c, bp, Mrows, L, packed);
  PackedCtx pk;
  memset(&pk, 0, sizeof(pk));
  float *enc_w = enc_out, *logd_w = log_d;
  if (packed) {
    const int M = (int)Mp, d = c.d_enc;
    int base = 0;  // the attention work list's length, from the same lengths the device plan reads
    for (int b = 0; b < B; ++b) {
      long long l = (lens_host[b] < 0 ? 0 : lens_host[b]) + PHONEME_GUARD;
      base += (int)(((l < L ? l : L) + 127) / 128) * c.n_enc_head;
    }
    size_t n = (size_t)attention_split_packed(base, L, d / c.n_enc_head, Mp, d);
    const size_t per = Mp * d + 2 * Mp * c.n_enc_head;
    const size_t extra = phase1_packed_extra_bytes(c, B, L);
    const size_t avail = ws_bytes > bp.off + extra ? (ws_bytes - bp.off - extra) / sizeof(float) : 0;
    while (n > 1 && n * per > avail) --n;
    if (n > 1) { sc.att_part_floats = n * per; sc.att_part = bp.f(sc.att_part_floats); }
    int* plan = (int*)bp.raw(pack_plan_ints(B, Mp) * sizeof(int));
    enc_w = bp.f(Mp * d); logd_w = bp.f(Mp);
    if (bp.off > ws_bytes) return fail("ns_forward_durations: workspace too small (packed rows)");
    pk.Mp = M;
    pk.rm.rows = M;
    pk.rm.att_wgs = base;
    NS_HIP(launch_pack_plan_only(lens, B, L, c.n_enc_head, M, plan, &pk.rm, st, PHONEME_GUARD));  // (row maps: the embedding kernel)
  }
  {
    sc.pk = packed ? &pk : nullptr;
    NS_TRY(encoder(m, (const long long*)texts, lens, B, L, enc_w, sc, st));
    NS_TRY(predictor(m, m->pred[0], enc_w, lens, B, L, 1.0f, nullptr, logd_w, nullptr, nullptr, nullptr, nullptr, sc, st));
    // phoneme_level features are predicted on the encoder output, before the length regulator, pitch first, and
    // added in place (model/modules.py:117-126); the duration predictor above saw x before these adds (:116)
    if (!c.pitch_frame_level)
      NS_TRY(predictor(m, m->pred[1], enc_w, lens, B, L, p_control, p_targets, p_pred, m->P(m->pitch_bins), m->P(m->pitch_emb), nullptr, enc_w, sc, st));
    if (!c.energy_frame_level)
      NS_TRY(predictor(m, m->pred[2], enc_w, lens, B, L, e_control, e_targets, e_pred, m->P(m->energy_bins), m->P(m->energy_emb), nullptr, enc_w, sc, st));
  }
  if (packed) {  // the padded tensors phase 2 and the caller read
    // (masked rows are zeros already; rows past a window become zeros; log_d is 0 at every padded phoneme)
    NS_HIP(launch_unpack_phase1(pk.rm, lens, B, L, c.d_enc, enc_w, enc_out, logd_w, log_d, st));
  }
  // src mask (utils/tools.py:89-97), duration rounding (model/modules.py:132-135), repeat counts + prefix sums + mel_len
  // (:209-223): one launch
  NS_HIP(launch_duration_tail(log_d, lens, (const long long*)texts, c.n_vocab, B, L, d_control, d_rounded, dur_keep, cum,
                              (long long*)mel_lens, src_mask, (long long*)mel_lens_host, st));
  return 0;
}

extern "C" int ns_forward_durations(ns_model* m, const int64_t* texts, const int64_t* src_lens, int B, int L, float d_control,
                                    float p_control, float e_control, const float* p_targets, const float* e_targets,
                                    void* ws_enc, size_t ws_bytes, float* log_d, float* d_rounded, uint8_t* src_mask,
                                    int64_t* mel_lens, float* p_pred, float* e_pred, int64_t* mel_lens_host, void* stream) {
  return forward_durations(m, texts, src_lens, nullptr, B, L, d_control, p_control, e_control, p_targets, e_targets, ws_enc, ws_bytes, log_d,
                           d_rounded, src_mask, mel_lens, p_pred, e_pred, mel_lens_host, stream);
}

extern "C" int ns_upload_lengths(const int64_t* host, int n, int64_t* dev, void* stream) {
  if (n < 0 || (n > 0 && (!host || !dev))) return fail("ns_upload_lengths: null argument");
  NS_HIP(launch_store_lens((const long long*)host, n, (long long*)dev, (hipStream_t)stream));
  return 0;
}

extern "C" int ns_forward_durations_packed(ns_model* m, const int64_t* texts, const int64_t* src_lens, const int64_t* src_lens_host, int B, int L,
                                           float d_control, float p_control, float e_control, const float* p_targets, const float* e_targets,
                                           void* ws_enc, size_t ws_bytes, float* log_d, float* d_rounded, uint8_t* src_mask,
                                           int64_t* mel_lens, float* p_pred, float* e_pred, int64_t* mel_lens_host, void* stream) {
  if (!src_lens_host) return fail("ns_forward_durations_packed: src_lens_host (the host copy of src_lens) is required");
  return forward_durations(m, texts, src_lens, src_lens_host, B, L, d_control, p_control, e_control, p_targets, e_targets, ws_enc, ws_bytes, log_d,
                           d_rounded, src_mask, mel_lens, p_pred, e_pred, mel_lens_host, stream);
}

// Packed phase 2 (kernels.h RowMap).  The reference runs everything behind the length regulator on the dense [B, T] grid and
// zeroes / ignores the frames past each utterance's length (transformer/Layers.py:43,46, model/modules.py:283-284; SURVEY.md
// F3); with variable lengths most of a batch's rows can be such padding.  Here utterance b keeps only its window of
// min(len[b] + PACK_GUARD, T) frames, the windows are laid end to end (Mp rows instead of B*T) and the same kernels run on
// them; the padded outputs the caller expects are rebuilt at the end (rowops.hip k_unpack_outputs).  What makes this exact:
//   * FFT blocks: a valid frame never reads a padded one except as zeros (masked_fill before every convolution, -inf keys);
//     a window's end is the convolution's zero padding, exactly what the zeroed frames were.
//   * predictors (no mask between the layers): a valid frame reaches 1 frame past the utterance's end, the guard keeps 20.
//   * PostNet (no mask at all, input = the mel_linear bias on padded frames): a valid frame reaches 10 frames past the end
//     (5 layers, reach 2); the guard's first 10 frames are computed from a window that extends 10 further, so they too are
//     what the dense computation gives; frames beyond are constants of the weights (postnet_constants below).
// Same arithmetic per row, so valid frames differ from the dense path's only where a launch picks another tile shape for
// the smaller M (fp32 summation order, ~1e-6).  Needs the lengths on the host (row count), hence the separate entry point.
static size_t packed_rows(const int64_t* lens_host, int B, int T) {
  size_t mp = 0;
  for (int b = 0; b < B; ++b) {
    long long l = lens_host[b] < 0 ? 0 : lens_host[b];
    l += PACK_GUARD;
    mp += (size_t)(l < (long long)T ? l : (long long)T);
  }
  return mp;
}
static size_t packed_extra_bytes(const ns_config& c, int B, int T) {  // on top of carve(): plan, packed outputs, PostNet constants
  Bump bp(nullptr, 0);
  const size_t M = (size_t)B * T;
  bp.raw(pack_plan_ints(B, M) * sizeof(int));
  bp.f(M * c.n_mel); bp.f(M * c.n_mel); bp.f(M); bp.f(M); bp.f(M); bp.f(M);
  return bp.off;
}

static int forward_mel(ns_model* m, int B, int L, int T, const int64_t* mel_lens, const int64_t* lens_host, float p_control, float e_control,
                       const float* p_targets, const float* e_targets,
                       const void* ws_enc, void* ws_dec, size_t ws_bytes, float* mel, float* postnet_mel, float* p_pred,
                       float* e_pred, uint8_t* mel_mask, int32_t* status, void* stream) {
  NS_TRY(check_ready(m));
  if (B <= 0 || L <= 0) return fail("ns_forward_mel: empty batch");
  if (!status) return fail("ns_forward_mel: status [B] is required (a T smaller than an utterance's length must not go unnoticed)");
  if (T < 0) return fail("ns_forward_mel: negative T");
  if (T == 0) {  // all durations zero: [B,0,*] outputs; only the status words (an utterance with frames would be cut off entirely)
    Bump be0(const_cast<void*>(ws_enc), (size_t)-1);
    be0.f((size_t)B * L * m->cfg.d_enc);
    const int32_t* cum0 = (const int32_t*)be0.raw((size_t)B * L * sizeof(int32_t));
    NS_HIP(launch_length_regulate(nullptr, cum0, B, L, m->cfg.d_enc, 0, nullptr, nullptr, (const long long*)mel_lens, status, nullptr, 0, (hipStream_t)stream));
    return 0;
  }
  if (ws_bytes < ns_decoder_ws_bytes(m, B, L, T)) return fail("ns_forward_mel: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const ns_config& c = m->cfg;
  Bump be(const_cast<void*>(ws_enc), (size_t)-1);
  const float* enc_out = be.f((size_t)B * L * c.d_enc);
  const int32_t* cum = (const int32_t*)be.raw((size_t)B * L * sizeof(int32_t));
  const float* dur_keep = be.f((size_t)B * L);
  const long long* lens = (const long long*)mel_lens;
  const int d = c.d_dec;
  // packed rows: the exact fp32 path, and only when the windows are a real saving over the grid
  size_t Mp = 0;
  bool packed = lens_host && !c.matmul_bf16x3 && !m->dec.empty();
  if (packed) {
    Mp = packed_rows(lens_host, B, T);
    // measured, packed against grid over length distributions (profiles/r03_packed_vs_grid_by_padding.txt): any saving of 10 %
    // pays on grids of 30 000 rows and more (0.89 of the rows: 1.07x, 0.85: 1.06-1.12x); a grid of ~16 000 rows is ONE full round
    // of the 256x256 tile, the packed rows fill no round, and the break-even is at 0.80 of the rows (0.80: 1.005x, 0.88: 0.99x)
    const size_t grid_rows = (size_t)B * T;
    packed = Mp > 0 && Mp < ((size_t)1 << 30) && Mp * 10 <= grid_rows * (grid_rows <= 20000 ? 8 : 9);
  }
  const size_t Mrows = packed ? Mp : (size_t)B * T;
  const int M = (int)Mrows;
  m->last_rows = (long long)Mrows;
  Bump bp(ws_dec, ws_bytes);
  Scratch sc = carve(c, bp, Mrows, T, packed);
  if (packed) {  // split-key partials for a work list of few workgroups (attention.hip packed launch): whatever the workspace still holds
    size_t base = 0;
    for (int b = 0; b < B; ++b) {
      long long l = (lens_host[b] < 0 ? 0 : lens_host[b]) + PACK_GUARD;
      base += (size_t)(((l < T ? l : T) + 127) / 128) * c.n_dec_head;
    }
    size_t n = (size_t)attention_split_packed((int)base, T, d / c.n_dec_head, Mrows, d);
    if (n > 1) {
      const size_t per = Mrows * d + 2 * Mrows * c.n_dec_head;  // floats per key range
      const size_t avail = ws_bytes > bp.off + packed_extra_bytes(c, B, T) ? (ws_bytes - bp.off - packed_extra_bytes(c, B, T)) / sizeof(float) : 0;
      while (n > 1 && n * per > avail) --n;
      if (n > 1) { sc.att_part_floats = n * per; sc.att_part = bp.f(sc.att_part_floats); }
    }
  }

  PackedCtx pk;
  memset(&pk, 0, sizeof(pk));
  float *mel_p = nullptr, *post_p = nullptr, *pp_p = nullptr, *ep_p = nullptr;
  if (packed) {
    int* plan = (int*)bp.raw(pack_plan_ints(B, Mp) * sizeof(int));
    mel_p = bp.f(Mp * c.n_mel); post_p = bp.f(Mp * c.n_mel); pp_p = bp.f(Mp); ep_p = bp.f(Mp);
    float* pt_p = bp.f(Mp);
    float* et_p = bp.f(Mp);
    if (bp.off > ws_bytes) return fail("ns_forward_mel_packed: workspace too small");
    pk.Mp = M;
    pk.rm.rows = M;
    if (c.length_regulator == 1) {  // extension: GaussianUpsampling in the LengthRegulator's place, on the packed rows
      if ((size_t)B * (L + 1) > (size_t)M * c.vp_filter) return fail("ns_forward_mel: workspace too small for the Gaussian centres");
      NS_HIP(launch_pack_plan(lens, B, T, c.n_dec_head, M, plan, &pk.rm, st));
      NS_HIP(launch_gaussian_upsampling(enc_out, dur_keep, B, L, c.d_enc, T, T, sc.xa, sc.vp2, nullptr, lens, status, sc.tickets, TICKET_INTS, st, &pk.rm));
    } else {
      NS_HIP(launch_length_regulate_packed(enc_out, cum, B, L, c.d_enc, T, M, c.n_dec_head, sc.xa, lens, status, sc.tickets, TICKET_INTS, plan, &pk.rm, st));
    }
    for (int b = 0; b < B; ++b) {  // the attention work list's length, from the same lengths the device plan reads
      long long l = (lens_host[b] < 0 ? 0 : lens_host[b]) + PACK_GUARD;
      if (l > T) l = T;
      pk.rm.att_wgs += (int)((l + 127) / 128) * c.n_dec_head;
    }
    // frame-level targets arrive on the padded [B, T] grid
    if (p_targets && c.pitch_frame_level) { NS_HIP(launch_pack_vector(pk.rm, T, p_targets, pt_p, M, st)); p_targets = pt_p; }
    if (e_targets && c.energy_frame_level) { NS_HIP(launch_pack_vector(pk.rm, T, e_targets, et_p, M, st)); e_targets = et_p; }
  } else if (c.length_regulator == 1) {
    NS_HIP(launch_mask_from_lengths(lens, B, T, mel_mask, st));
    // extension: GaussianUpsampling (model/modules.py:166-192) in the LengthRegulator's place;
    // mel_len = sum of the rounded durations, frames past an utterance's own length are zero like pad()'s
    if ((size_t)B * (L + 1) > (size_t)M * c.vp_filter) return fail("ns_forward_mel: workspace too small for the Gaussian centres");
    NS_HIP(launch_gaussian_upsampling(enc_out, dur_keep, B, L, c.d_enc, T, T, sc.xa, sc.vp2, nullptr, lens, status, sc.tickets, TICKET_INTS, st));
  } else {
    NS_HIP(launch_length_regulate(enc_out, cum, B, L, c.d_enc, T, sc.xa, mel_mask, lens, status, sc.tickets, TICKET_INTS, st));  // + mel mask, status, ticket zeroing
  }
  sc.pk = packed ? &pk : nullptr;
  float* const mel_dst = packed ? mel_p : mel;
  float* const post_dst = packed ? post_p : postnet_mel;
  float* const pp_dst = packed ? pp_p : p_pred;
  float* const ep_dst = packed ? ep_p : e_pred;
  const float* pos;
  NS_TRY(position_rows(m, m->dec_pos, T, d, sc, &pos, st));
  // frame-level pitch then energy (model/modules.py:139-149); MelDecoder's position add rides on the last
  // frame-level embedding kernel (or is a kernel of its own when both features are phoneme_level)
  float* cur = sc.xa;
  float* alt = sc.xb;
  if (c.pitch_frame_level) {
    if (!p_pred) return fail("ns_forward_mel: frame_level pitch needs a p_pred [B,T] output");
    NS_TRY(predictor(m, m->pred[1], cur, lens, B, T, p_control, p_targets, pp_dst, m->P(m->pitch_bins), m->P(m->pitch_emb),
                     c.energy_frame_level ? nullptr : pos, alt, sc, st));
    float* t = cur; cur = alt; alt = t;
  }
  if (c.energy_frame_level) {
    if (!e_pred) return fail("ns_forward_mel: frame_level energy needs an e_pred [B,T] output");
    NS_TRY(predictor(m, m->pred[2], cur, lens, B, T, e_control, e_targets, ep_dst, m->P(m->energy_bins), m->P(m->energy_emb), pos,
                     alt, sc, st));
    float* t = cur; cur = alt; alt = t;
  }
  if (!c.pitch_frame_level && !c.energy_frame_level) {
    NS_HIP(launch_add_pos(cur, pos, alt, M, T, d, st, cur_rm(sc)));
    float* t = cur; cur = alt; alt = t;
  }
  m->prof_active = m->prof != 0;  // time only phase 2's launches: one shape per slot (the encoder runs the same kernels at B*L rows)
  int rc = decoder_stack(m, cur, lens, B, T, sc.att, sc, st);
  // note: decoder_stack's last layer writes into sc.att only after its own attention output was consumed
  if (!rc) rc = gemm(sc, sc.att, d, m->P(m->mel_w), m->P(m->mel_b), nullptr, 0, mel_dst, c.n_mel, M, c.n_mel, d, 1, T, ACT_NONE, st);
  if (!rc) rc = postnet(m, mel_dst, B, T, mel_dst, post_dst, sc, st);
  m->prof_active = false;
  if (!rc && packed) {
    hipError_t e = launch_unpack_outputs(pk.rm, B, T, c.n_mel, lens, mel_p, post_p, c.pitch_frame_level ? pp_p : nullptr,
                                         c.energy_frame_level ? ep_p : nullptr, m->P(m->mel_b), m->P(m->pn_const) + (size_t)21 * c.n_mel, mel, postnet_mel,
                                         c.pitch_frame_level ? p_pred : nullptr, c.energy_frame_level ? e_pred : nullptr, mel_mask, st);
    if (e != hipSuccess) rc = fail(std::string("launch_unpack_outputs: ") + hipGetErrorString(e));
  }
  return rc;
}

extern "C" int64_t ns_last_phase2_rows(const ns_model* m) { return m ? (int64_t)m->last_rows : 0; }
extern "C" int64_t ns_last_phase1_rows(const ns_model* m) { return m ? (int64_t)m->last_rows1 : 0; }

extern "C" int ns_forward_mel(ns_model* m, int B, int L, int T, const int64_t* mel_lens, float p_control, float e_control,
                              const float* p_targets, const float* e_targets,
                              const void* ws_enc, void* ws_dec, size_t ws_bytes, float* mel, float* postnet_mel, float* p_pred,
                              float* e_pred, uint8_t* mel_mask, int32_t* status, void* stream) {
  return forward_mel(m, B, L, T, mel_lens, nullptr, p_control, e_control, p_targets, e_targets, ws_enc, ws_dec, ws_bytes, mel, postnet_mel,
                     p_pred, e_pred, mel_mask, status, stream);
}

extern "C" int ns_forward_mel_packed(ns_model* m, int B, int L, int T, const int64_t* mel_lens, const int64_t* mel_lens_host,
                                     float p_control, float e_control, const float* p_targets, const float* e_targets,
                                     const void* ws_enc, void* ws_dec, size_t ws_bytes, float* mel, float* postnet_mel, float* p_pred,
                                     float* e_pred, uint8_t* mel_mask, int32_t* status, void* stream) {
  if (!mel_lens_host) return fail("ns_forward_mel_packed: mel_lens_host (the host copy of mel_lens) is required");
  return forward_mel(m, B, L, T, mel_lens, mel_lens_host, p_control, e_control, p_targets, e_targets, ws_enc, ws_dec, ws_bytes, mel,
                     postnet_mel, p_pred, e_pred, mel_mask, status, stream);
}

// ------------------------------------------------------------------------------------------- per-op entry points
static int find_layer(ns_model* m, const char* prefix_c, const LayerW** L, int* d, int* H, const char* suffix) {
  std::string p(prefix_c ? prefix_c : "");
  if (suffix && *suffix) {
    if (!ends_with(p, suffix)) return fail("prefix '" + p + "' does not end with '" + suffix + "'");
    p.resize(p.size() - strlen(suffix));
  }
  const bool enc = starts_with(p, "txt_encoder.layer_stack."), dec = starts_with(p, "mel_decoder.layer_stack.");
  if (!enc && !dec) return fail("unknown module prefix '" + std::string(prefix_c ? prefix_c : "") + "'");
  const int idx = atoi(p.c_str() + (enc ? strlen("txt_encoder.layer_stack.") : strlen("mel_decoder.layer_stack.")));
  auto& v = enc ? m->enc : m->dec;
  if (idx < 0 || idx >= (int)v.size()) return fail("layer index out of range in '" + p + "'");
  *L = &v[idx];
  *d = enc ? m->cfg.d_enc : m->cfg.d_dec;
  *H = enc ? m->cfg.n_enc_head : m->cfg.n_dec_head;
  return 0;
}

#define NS_OP_PROLOGUE(B_, S_)                                                        \
  NS_TRY(check_ready(m));                                                             \
  if ((B_) <= 0 || (S_) <= 0) return fail("empty input");                             \
  if (ws_bytes < ns_op_ws_bytes(m, (B_), (S_))) return fail("workspace too small");   \
  hipStream_t st = (hipStream_t)stream;                                               \
  Bump bp(ws, ws_bytes);                                                              \
  Scratch sc = carve(m->cfg, bp, (size_t)(B_) * (S_), (S_));                          \
  if (sc.tickets) NS_HIP(hipMemsetAsync(sc.tickets, 0, TICKET_INTS * sizeof(int), st)); /* a forward's first kernel does this itself */

extern "C" int ns_op_mask_from_lengths(const int64_t* lens, int B, int max_len, uint8_t* mask, void* stream) {
  NS_HIP(launch_mask_from_lengths((const long long*)lens, B, max_len, mask, (hipStream_t)stream));
  return 0;
}
extern "C" int ns_op_sinusoid_table(int n_position, int d_hid, float* out, void* stream) {
  NS_HIP(launch_sinusoid(n_position, d_hid, out, (hipStream_t)stream));
  return 0;
}
extern "C" int ns_op_txt_encoder(ns_model* m, const int64_t* texts, const int64_t* lens, int B, int L, float* out, void* ws,
                                 size_t ws_bytes, void* stream) {
  NS_OP_PROLOGUE(B, L);
  return encoder(m, (const long long*)texts, (const long long*)lens, B, L, out, sc, st);
}
extern "C" int ns_op_multi_head_attention(ns_model* m, const char* prefix, const float* x, const int64_t* lens, int B, int S,
                                          float* out, void* ws, size_t ws_bytes, void* stream) {
  NS_OP_PROLOGUE(B, S);
  const LayerW* L; int d, H;
  NS_TRY(find_layer(m, prefix, &L, &d, &H, ".slf_attn"));
  return mha(m, *L, d, H, x, (const long long*)lens, B, S, out, false, sc, st);
}
extern "C" int ns_op_positionwise_ffn(ns_model* m, const char* prefix, const float* x, int B, int S, float* out, void* ws,
                                      size_t ws_bytes, void* stream) {
  NS_OP_PROLOGUE(B, S);
  const LayerW* L; int d, H;
  NS_TRY(find_layer(m, prefix, &L, &d, &H, ".pos_ffn"));
  return ffn(m, *L, d, x, nullptr, B, S, out, false, sc, st);
}
extern "C" int ns_op_fft_block(ns_model* m, const char* prefix, const float* x, const int64_t* lens, int B, int S, float* out,
                               void* ws, size_t ws_bytes, void* stream) {
  NS_OP_PROLOGUE(B, S);
  const LayerW* L; int d, H;
  NS_TRY(find_layer(m, prefix, &L, &d, &H, ""));
  return fft_block(m, *L, d, H, x, (const long long*)lens, B, S, out, sc, st);
}
static int find_pred(const char* prefix, int* idx) {
  std::string p(prefix ? prefix : "");
  for (int i = 0; i < 3; ++i)
    if (p == std::string("variance_adaptor.") + kPredNames[i] + "_predictor") { *idx = i; return 0; }
  return fail("unknown predictor prefix '" + p + "'");
}
extern "C" int ns_op_variance_predictor(ns_model* m, const char* prefix, const float* x, const int64_t* lens, int B, int S,
                                        float* out, void* ws, size_t ws_bytes, void* stream) {
  NS_OP_PROLOGUE(B, S);
  int i;
  NS_TRY(find_pred(prefix, &i));
  return predictor(m, m->pred[i], x, (const long long*)lens, B, S, 1.0f, nullptr, out, nullptr, nullptr, nullptr, nullptr, sc, st);
}
extern "C" int ns_op_duration_round(const float* log_d, int n, float d_control, float* d_rounded, void* stream) {
  NS_HIP(launch_duration_round(log_d, n, d_control, d_rounded, (hipStream_t)stream));
  return 0;
}
extern "C" int ns_op_duration_scan(const float* d_rounded, int B, int L, int32_t* cum, int64_t* mel_lens, void* stream) {
  NS_HIP(launch_duration_scan(d_rounded, B, L, cum, (long long*)mel_lens, (hipStream_t)stream));
  return 0;
}
extern "C" int ns_op_length_regulate(const float* x, const int32_t* cum, int B, int L, int D, int T, float* out, void* stream) {
  NS_HIP(launch_length_regulate(x, cum, B, L, D, T, out, nullptr, nullptr, nullptr, nullptr, 0, (hipStream_t)stream));
  return 0;
}
extern "C" int ns_op_variance_embedding(ns_model* m, int which, const float* x, const int64_t* lens, int B, int S, float control,
                                        const float* target, float* pred, float* x_out, void* ws, size_t ws_bytes, void* stream) {
  NS_OP_PROLOGUE(B, S);
  if (which != 0 && which != 1) return fail("which must be 0 (pitch) or 1 (energy)");
  const size_t bins = which ? m->energy_bins : m->pitch_bins, emb = which ? m->energy_emb : m->pitch_emb;
  return predictor(m, m->pred[1 + which], x, (const long long*)lens, B, S, control, target, pred, m->P(bins), m->P(emb), nullptr, x_out,
                   sc, st);
}
extern "C" int ns_plan_gemm(int M, int N, int Cin, int KW, int32_t out[8]) {
  int o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool planned = conv_gemm_plan(M, N, Cin, KW, o);
  if (out) for (int i = 0; i < 8; ++i) out[i] = planned ? o[i] : 0;
  return planned ? 1 : 0;
}
extern "C" int ns_plan_row_tile_k(int M, int N, int K) {
  if (M <= 0 || K <= 0 || (N != 256 && N != 512)) return 0;
  return conv_gemm_row_tile(M, N, K);
}
// (the two-argument form answers for the shortest contraction a full-row GEMM of that width has: the attention output
// projection, K = N)
extern "C" int ns_plan_row_tile(int M, int N) { return ns_plan_row_tile_k(M, N, N); }
extern "C" int ns_acc_chunk(void) { return conv_gemm_acc_chunk(); }
extern "C" int ns_abi_version(void) { return NS_ABI_VERSION; }
extern "C" int ns_plan_attention_split(int B, int S, int H, int dk) { return attention_split(B, S, H, dk); }

extern "C" int ns_profile_enable(ns_model* m, int on) {
  if (!m) return fail("ns_profile_enable: null model");
  const bool keep = on > 0 && (on & NS_PROFILE_KEEP) != 0;  // change the set of timed slots, keep what was recorded so far
  if (on > 0) on &= ~NS_PROFILE_KEEP;
  m->prof = on == 1 ? (1u << NS_PROFILE_SLOTS) - 1u : on < 0 ? 0u : ((unsigned)on >> 1);  // 1 = every slot; 2 * mask = those slots
  if (!keep)
    for (auto& ps : m->prof_slot) { ps.used = 0; ps.flops = 0.0; }
  return 0;
}
extern "C" int ns_profile_read_slot(ns_model* m, int slot, double* total_ms, double* total_flops, int64_t* launches) {
  if (!m) return fail("ns_profile_read_slot: null model");
  if (slot < 0 || slot >= NS_PROFILE_SLOTS) return fail("ns_profile_read_slot: slot out of range");
  ns_model::ProfSlot& ps = m->prof_slot[slot];
  double ms = 0.0;
  for (size_t i = 0; i < ps.used; ++i) {
    NS_HIP(hipEventSynchronize(ps.ev[i].second));
    float e = 0.f;
    NS_HIP(hipEventElapsedTime(&e, ps.ev[i].first, ps.ev[i].second));
    ms += e;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = ps.flops;
  if (launches) *launches = (int64_t)ps.used;
  ps.used = 0;
  ps.flops = 0.0;
  return 0;
}
extern "C" int ns_profile_read(ns_model* m, double* total_ms, double* total_flops, int64_t* launches) {
  return ns_profile_read_slot(m, 0, total_ms, total_flops, launches);
}
extern "C" int ns_op_bucketize(const float* values, int n, const float* bins, int n_edges, int64_t* idx, void* stream) {
  NS_HIP(launch_bucketize(values, n, bins, n_edges, (long long*)idx, (hipStream_t)stream));
  return 0;
}
extern "C" int ns_op_gaussian_upsampling(const float* x, const float* durations, int B, int L, int D, int T, int T_out, float* out,
                                         float* s, float* w, void* stream) {
  if (T_out < T) return fail("ns_op_gaussian_upsampling: T_out < T");
  NS_HIP(launch_gaussian_upsampling(x, durations, B, L, D, T, T_out, out, s, w, nullptr, nullptr, nullptr, 0, (hipStream_t)stream));
  return 0;
}
extern "C" int ns_op_mel_decoder(ns_model* m, const float* x, const int64_t* lens, int B, int T, float* out, void* ws,
                                 size_t ws_bytes, void* stream) {
  NS_OP_PROLOGUE(B, T);
  const float* pos;
  NS_TRY(position_rows(m, m->dec_pos, T, m->cfg.d_dec, sc, &pos, st));
  NS_HIP(launch_add_pos(x, pos, sc.xa, B * T, T, m->cfg.d_dec, st));
  return decoder_stack(m, sc.xa, (const long long*)lens, B, T, out, sc, st);
}
extern "C" int ns_op_mel_linear(ns_model* m, const float* x, int B, int T, float* out, void* stream) {
  NS_TRY(check_ready(m));
  const ns_config& c = m->cfg;
  Scratch sc;
  memset(&sc, 0, sizeof(sc));  // (a plain GEMM on the dense grid: no temporaries, no packed context)
  return gemm(sc, x, c.d_dec, m->P(m->mel_w), m->P(m->mel_b), nullptr, 0, out, c.n_mel, B * T, c.n_mel, c.d_dec, 1, T, ACT_NONE,
              (hipStream_t)stream);
}
extern "C" int ns_op_postnet(ns_model* m, const float* mel, int B, int T, float* out, void* ws, size_t ws_bytes, void* stream) {
  NS_OP_PROLOGUE(B, T);
  return postnet(m, mel, B, T, nullptr, out, sc, st);
}
extern "C" int ns_op_attention_core(const float* qkv, const int64_t* lens, int B, int S, int H, int dk, float* out, void* scratch,
                                    size_t scratch_bytes, void* stream) {
  // the strip kernel's tickets are carved from the END of the caller's scratch and zeroed here (a forward's opening kernel
  // does that for its own launches)
  float* sp = (float*)scratch;
  size_t floats = scratch_bytes / sizeof(float);
  int* tickets = nullptr;
  const size_t tk = ((size_t)attention_ticket_ints(B, S, H) + 63) & ~(size_t)63;
  if (sp && B > 0 && S > 0 && floats > tk) {
    floats -= tk;
    tickets = (int*)(sp + floats);
    NS_HIP(hipMemsetAsync(tickets, 0, tk * sizeof(int), (hipStream_t)stream));
  }
  NS_HIP(launch_attention(qkv, (const long long*)lens, B, S, H, dk, out, sp, floats, tickets, (hipStream_t)stream));
  return 0;
}
extern "C" int ns_op_ffn_conv1(ns_model* m, const char* prefix, const float* x, int B, int S, float* hidden, void* stream) {
  NS_TRY(check_ready(m));
  const LayerW* L; int d, H;
  NS_TRY(find_layer(m, prefix, &L, &d, &H, ".pos_ffn"));
  const ns_config& c = m->cfg;
  Scratch sc;
  memset(&sc, 0, sizeof(sc));
  return gemm(sc, x, d, m->P(L->w1), m->P(L->w1_b), nullptr, 0, hidden, c.d_inner, B * S, c.d_inner, d, c.ffn_k1, S, ACT_RELU,
              (hipStream_t)stream);
}
