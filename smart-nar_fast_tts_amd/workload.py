"""Synthetic workload definition: configs, weights, stats and inputs.

The reference ships no checkpoint, no ``stats.json`` and no test inputs
(SURVEY.md F6/F8), so every parity and bench run needs a reproducible synthetic
utterance batch and weight set.  Everything here is derived from a seed through
``numpy.random.RandomState`` (frozen legacy stream) so the SAME weights can be
rebuilt in the golden generator (which loads them into the imported reference),
in the oracle, in the HIP path and on the GPU box without shipping 116 MB.

Key names and tensor layouts follow the reference's ``state_dict``
(SURVEY.md §8b):  Linear ``[out,in]``, Conv1d ``[out,in,k]``.

This is workload definition, not reference behaviour: in particular the
duration predictor's output bias is set to ``log(frames_per_phoneme+1)`` because
plain random-init weights yield ~0.35 frames per phoneme.
"""
from __future__ import annotations

import copy
import math
import re
from collections import OrderedDict

import numpy as np

N_SYMBOLS = 360  # len(text.symbols.symbols) in the reference [ran]; vocab = N_SYMBOLS + 1 (transformer/Models.py:40)
N_MEL = 80
POSTNET_DIM = 512  # transformer/Layers.py:112-118 (hard-coded)
POSTNET_K = 5
POSTNET_N = 5

# config/LJSpeech/model.yaml:1-25 restated (values only; it is data, not code)
LJSPEECH_MODEL_CONFIG = {
    "transformer": {
        "encoder_layer": 4,
        "encoder_head": 2,
        "encoder_hidden": 256,
        "decoder_layer": 4,
        "decoder_head": 2,
        "decoder_hidden": 256,
        "conv_filter_size": 1024,
        "conv_kernel_size": [9, 1],
        "encoder_dropout": 0.2,
        "decoder_dropout": 0.2,
    },
    "variance_predictor": {"filter_size": 256, "kernel_size": 3, "dropout": 0.5},
    "variance_embedding": {
        # the shipped yaml says pitch "log"; with normalised pitch that gives NaN bins,
        # so stats below use a positive pitch range which keeps "log" well defined.
        "pitch_quantization": "log",
        "energy_quantization": "linear",
        "n_bins": 256,
    },
    "multi_speaker": False,
    "max_seq_len": 1000,
}

# config/LJSpeech/preprocess.yaml — only the keys the model constructor reads (model/modules.py:26-46)
LJSPEECH_PREPROCESS_CONFIG = {
    "path": {"preprocessed_path": "./preprocessed_data/LJSpeech"},
    "preprocessing": {
        "mel": {"n_mel_channels": N_MEL},
        "pitch": {"feature": "frame_level", "normalization": True},
        "energy": {"feature": "frame_level", "normalization": True},
    },
}

# synthetic stats.json: [min, max, mean, std]
SYNTH_STATS = {"pitch": [50.0, 600.0, 200.0, 50.0], "energy": [-1.5, 9.0, 30.0, 20.0]}


def model_config(name: str = "ljspeech") -> dict:
    """Named model configs used by BASELINE.json's configs list."""
    mc = copy.deepcopy(LJSPEECH_MODEL_CONFIG)
    if name == "ljspeech":
        return mc
    if name.endswith("+gaussian"):
        mc = model_config(name[:-len("+gaussian")])
        mc["length_regulator"] = "gaussian"
        return mc
    if name == "d512":  # BASELINE config 4: d_model=512, 6+6 layers, 8 heads
        t = mc["transformer"]
        t.update(encoder_layer=6, decoder_layer=6, encoder_head=8, decoder_head=8,
                 encoder_hidden=512, decoder_hidden=512)
        mc["variance_predictor"]["filter_size"] = 256
        return mc
    if name == "tiny":  # fixtures: real widths, 1+1 layers
        t = mc["transformer"]
        t.update(encoder_layer=1, decoder_layer=1)
        return mc
    if name in ("tiny512", "tiny_h4"):  # fuzzing: the d_k = 64 / d_k = 32 attention paths and 512-wide rows, 1+1 layers
        t = mc["transformer"]
        if name == "tiny512":
            t.update(encoder_layer=1, decoder_layer=1, encoder_head=8, decoder_head=8, encoder_hidden=512, decoder_hidden=512)
        else:
            t.update(encoder_layer=1, decoder_layer=1, encoder_head=8, decoder_head=4)  # 256 wide: d_k 32 / 64
        return mc
    raise KeyError(name)


def preprocess_config(pitch: str = "frame_level", energy: str = "frame_level") -> dict:
    """preprocess.yaml restated; ``pitch`` / ``energy`` select the feature level (model/modules.py:26-33)."""
    pc = copy.deepcopy(LJSPEECH_PREPROCESS_CONFIG)
    pc["preprocessing"]["pitch"]["feature"] = pitch
    pc["preprocessing"]["energy"]["feature"] = energy
    return pc


def sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """Vectorised restatement of transformer/Models.py:10-30: angle in float64,
    sin on even / cos on odd columns, cast to float32."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    denom = np.power(10000.0, 2 * (j // 2) / d_hid)
    tab = pos / denom[None, :]
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return tab.astype(np.float32)


def variance_bins(model_cfg: dict, stats: dict = SYNTH_STATS):
    """pitch_bins / energy_bins as model/modules.py:48-71 builds them.

    torch.linspace(fp32) then exp in fp32 — reproduced with torch to be
    bit-identical with what the reference would register as parameters."""
    import torch

    n_bins = model_cfg["variance_embedding"]["n_bins"]
    out = {}
    for name in ("pitch", "energy"):
        lo, hi = stats[name][:2]
        if model_cfg["variance_embedding"][f"{name}_quantization"] == "log":
            b = torch.exp(torch.linspace(np.log(lo), np.log(hi), n_bins - 1))
        else:
            b = torch.linspace(lo, hi, n_bins - 1)
        out[name] = b.numpy().astype(np.float32)
    return out["pitch"], out["energy"]


def _uniform(rs, shape, bound):
    return rs.uniform(-bound, bound, size=shape).astype(np.float32)


def synth_state_dict(model_cfg: dict, seed: int = 0, frames_per_phoneme: float = 8.0,
                     dur_weight_scale: float = 0.25, stats: dict = SYNTH_STATS) -> "OrderedDict[str, np.ndarray]":
    """Inference-subset state dict (no ``mel_encoder.*``) with reference key names.

    Distributions follow torch's default initialisers in scale (uniform
    +-1/sqrt(fan_in) for Linear/Conv, N(0,1) embeddings) but LayerNorm affine
    and BatchNorm running statistics are made non-trivial so that the parity
    tests exercise them."""
    rs = np.random.RandomState(seed)
    t = model_cfg["transformer"]
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def linear(prefix, out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        sd[prefix + ".weight"] = _uniform(rs, (out_f, in_f), b)
        sd[prefix + ".bias"] = _uniform(rs, (out_f,), b)

    def conv(prefix, out_c, in_c, k):
        b = 1.0 / math.sqrt(in_c * k)
        sd[prefix + ".weight"] = _uniform(rs, (out_c, in_c, k), b)
        sd[prefix + ".bias"] = _uniform(rs, (out_c,), b)

    def lnorm(prefix, d):
        sd[prefix + ".weight"] = (1.0 + 0.1 * rs.standard_normal(d)).astype(np.float32)
        sd[prefix + ".bias"] = (0.05 * rs.standard_normal(d)).astype(np.float32)

    def stack(prefix, n_layers, d, n_head, d_inner, ks):
        for i in range(n_layers):
            p = f"{prefix}.layer_stack.{i}"
            for w in ("w_qs", "w_ks", "w_vs"):
                linear(f"{p}.slf_attn.{w}", d, d)
            lnorm(f"{p}.slf_attn.layer_norm", d)
            linear(f"{p}.slf_attn.fc", d, d)
            conv(f"{p}.pos_ffn.w_1", d_inner, d, ks[0])
            conv(f"{p}.pos_ffn.w_2", d, d_inner, ks[1])
            lnorm(f"{p}.pos_ffn.layer_norm", d)

    d_enc, d_dec = t["encoder_hidden"], t["decoder_hidden"]
    n_pos = model_cfg["max_seq_len"] + 1
    sd["txt_encoder.position_enc"] = sinusoid_table(n_pos, d_enc)[None]
    emb = rs.standard_normal((N_SYMBOLS + 1, d_enc)).astype(np.float32)
    emb[0] = 0.0  # padding_idx=0 (transformer/Models.py:55-57)
    sd["txt_encoder.src_word_emb.weight"] = emb
    stack("txt_encoder", t["encoder_layer"], d_enc, t["encoder_head"], t["conv_filter_size"], t["conv_kernel_size"])

    pb, eb = variance_bins(model_cfg, stats)
    sd["variance_adaptor.pitch_bins"] = pb
    sd["variance_adaptor.energy_bins"] = eb
    vp = model_cfg["variance_predictor"]
    for name in ("duration", "pitch", "energy"):
        p = f"variance_adaptor.{name}_predictor"
        conv(f"{p}.conv_layer.conv1d_1.conv", vp["filter_size"], d_enc, vp["kernel_size"])
        lnorm(f"{p}.conv_layer.layer_norm_1", vp["filter_size"])
        conv(f"{p}.conv_layer.conv1d_2.conv", vp["filter_size"], vp["filter_size"], vp["kernel_size"])
        lnorm(f"{p}.conv_layer.layer_norm_2", vp["filter_size"])
        linear(f"{p}.linear_layer", 1, vp["filter_size"])
    # F2 workload edit: ~frames_per_phoneme frames per phoneme with a moderate spread
    p = "variance_adaptor.duration_predictor.linear_layer"
    sd[p + ".weight"] = (sd[p + ".weight"] * dur_weight_scale).astype(np.float32)
    sd[p + ".bias"] = np.array([math.log(frames_per_phoneme + 1.0)], dtype=np.float32)
    # spread pitch/energy predictions over several buckets so bucketize is exercised
    for name, scale, shift in (("pitch", 400.0, 250.0), ("energy", 8.0, 3.0)):
        p = f"variance_adaptor.{name}_predictor.linear_layer"
        sd[p + ".weight"] = (sd[p + ".weight"] * scale).astype(np.float32)
        sd[p + ".bias"] = np.array([shift], dtype=np.float32)
    n_bins = model_cfg["variance_embedding"]["n_bins"]
    sd["variance_adaptor.pitch_embedding.weight"] = rs.standard_normal((n_bins, d_enc)).astype(np.float32)
    sd["variance_adaptor.energy_embedding.weight"] = rs.standard_normal((n_bins, d_enc)).astype(np.float32)

    sd["mel_decoder.position_enc"] = sinusoid_table(n_pos, d_dec)[None]
    stack("mel_decoder", t["decoder_layer"], d_dec, t["decoder_head"], t["conv_filter_size"], t["conv_kernel_size"])

    linear("mel_linear", N_MEL, d_dec)

    chans = [N_MEL] + [POSTNET_DIM] * (POSTNET_N - 1) + [N_MEL]
    for i in range(POSTNET_N):
        conv(f"postnet.convolutions.{i}.0.conv", chans[i + 1], chans[i], POSTNET_K)
        c = chans[i + 1]
        sd[f"postnet.convolutions.{i}.1.weight"] = (1.0 + 0.1 * rs.standard_normal(c)).astype(np.float32)
        sd[f"postnet.convolutions.{i}.1.bias"] = (0.05 * rs.standard_normal(c)).astype(np.float32)
        sd[f"postnet.convolutions.{i}.1.running_mean"] = (0.05 * rs.standard_normal(c)).astype(np.float32)
        sd[f"postnet.convolutions.{i}.1.running_var"] = rs.uniform(0.5, 1.5, size=c).astype(np.float32)
        sd[f"postnet.convolutions.{i}.1.num_batches_tracked"] = np.array(0, dtype=np.int64)
    return sd


def inference_shapes(model_cfg: dict) -> "OrderedDict[str, tuple]":
    """Name -> shape of every state-dict entry the inference forward REQUIRES (reference key names).
    ``position_enc`` tables (deterministic, regenerated when absent), ``num_batches_tracked`` and ``mel_encoder.*`` are not
    required and not listed."""
    t = model_cfg["transformer"]
    vp = model_cfg["variance_predictor"]
    out: "OrderedDict[str, tuple]" = OrderedDict()

    def wb(prefix, *wshape):
        out[prefix + ".weight"] = tuple(wshape)
        out[prefix + ".bias"] = (wshape[0],)

    def stack(prefix, n_layers, d):
        for i in range(n_layers):
            p = f"{prefix}.layer_stack.{i}"
            for w in ("w_qs", "w_ks", "w_vs", "fc"):
                wb(f"{p}.slf_attn.{w}", d, d)
            out[f"{p}.slf_attn.layer_norm.weight"] = out[f"{p}.slf_attn.layer_norm.bias"] = (d,)
            wb(f"{p}.pos_ffn.w_1", t["conv_filter_size"], d, t["conv_kernel_size"][0])
            wb(f"{p}.pos_ffn.w_2", d, t["conv_filter_size"], t["conv_kernel_size"][1])
            out[f"{p}.pos_ffn.layer_norm.weight"] = out[f"{p}.pos_ffn.layer_norm.bias"] = (d,)

    d_enc, d_dec, n_bins = t["encoder_hidden"], t["decoder_hidden"], model_cfg["variance_embedding"]["n_bins"]
    out["txt_encoder.src_word_emb.weight"] = (N_SYMBOLS + 1, d_enc)
    stack("txt_encoder", t["encoder_layer"], d_enc)
    out["variance_adaptor.pitch_bins"] = out["variance_adaptor.energy_bins"] = (n_bins - 1,)
    F, K = vp["filter_size"], vp["kernel_size"]
    for name in ("duration", "pitch", "energy"):
        p = f"variance_adaptor.{name}_predictor"
        wb(f"{p}.conv_layer.conv1d_1.conv", F, d_enc, K)
        out[f"{p}.conv_layer.layer_norm_1.weight"] = out[f"{p}.conv_layer.layer_norm_1.bias"] = (F,)
        wb(f"{p}.conv_layer.conv1d_2.conv", F, F, K)
        out[f"{p}.conv_layer.layer_norm_2.weight"] = out[f"{p}.conv_layer.layer_norm_2.bias"] = (F,)
        wb(f"{p}.linear_layer", 1, F)
    out["variance_adaptor.pitch_embedding.weight"] = out["variance_adaptor.energy_embedding.weight"] = (n_bins, d_enc)
    stack("mel_decoder", t["decoder_layer"], d_dec)
    wb("mel_linear", N_MEL, d_dec)
    chans = [N_MEL] + [POSTNET_DIM] * (POSTNET_N - 1) + [N_MEL]
    for i in range(POSTNET_N):
        wb(f"postnet.convolutions.{i}.0.conv", chans[i + 1], chans[i], POSTNET_K)
        for s in ("weight", "bias", "running_mean", "running_var"):
            out[f"postnet.convolutions.{i}.1.{s}"] = (chans[i + 1],)
    return out


def inference_keys(model_cfg: dict):
    return list(inference_shapes(model_cfg))


_NORM_KEY = re.compile(r"(.*\.layer_norm(_\d)?|postnet\.convolutions\.\d+\.1)\.(weight|bias|running_mean|running_var)")


def default_init_state_dict(model_cfg: dict, stats: dict) -> "OrderedDict[str, np.ndarray]":
    """What the reference constructor leaves in the module (model/fastspeech2_align.py:16-28): every layer is a stock
    ``torch.nn`` module with torch's default initialiser — Linear / Conv1d ``kaiming_uniform_(a=sqrt(5))`` = U(+-1/sqrt(fan_in))
    for weight and bias, Embedding N(0,1) with the padding row zeroed, LayerNorm / BatchNorm affine (1, 0), running stats
    (0, 1) — drawn here from torch's GLOBAL generator so that ``torch.manual_seed`` controls it as it does there (the draw
    order differs from the reference's module construction order, so the values are equivalent in distribution, not
    equal).  Bins come from ``stats.json`` (model/modules.py:41-71).  Inference subset only (no ``mel_encoder``)."""
    import torch

    shapes = inference_shapes(model_cfg)
    pb, eb = variance_bins(model_cfg, stats)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, shape in shapes.items():
        if k.endswith("pitch_bins"):
            out[k] = pb
        elif k.endswith("energy_bins"):
            out[k] = eb
        elif _NORM_KEY.fullmatch(k):  # LayerNorm / BatchNorm1d parameters and running statistics — and nothing else
            one = k.endswith(".weight") or k.endswith("running_var")
            out[k] = np.full(shape, 1.0 if one else 0.0, dtype=np.float32)
        elif k.endswith("embedding.weight") or k.endswith("src_word_emb.weight"):
            e = torch.randn(*shape).numpy()
            if k.endswith("src_word_emb.weight"):
                e[0] = 0.0  # padding_idx
            out[k] = e
        else:  # Linear [out, in] / Conv1d [out, in, k] weight, or their bias [out] (fan_in from the matching weight)
            w = shapes[k[:-len(".bias")] + ".weight"] if k.endswith(".bias") else shape
            bound = 1.0 / math.sqrt(int(np.prod(w[1:])))
            out[k] = torch.empty(*shape).uniform_(-bound, bound).numpy()
    return out


def synth_inputs(batch: int, max_src_len: int, seed: int = 0, src_lens=None):
    """texts ~ randint(1, 361) zero-padded, src_lens (default all = L), speakers = 0."""
    rs = np.random.RandomState(seed + 1000003)
    texts = rs.randint(1, N_SYMBOLS + 1, size=(batch, max_src_len)).astype(np.int64)
    if src_lens is None:
        lens = np.full((batch,), max_src_len, dtype=np.int64)
    else:
        lens = np.asarray(src_lens, dtype=np.int64)
        assert lens.shape == (batch,) and lens.max() <= max_src_len
    for b in range(batch):
        texts[b, lens[b]:] = 0
    speakers = np.zeros((batch,), dtype=np.int64)
    return speakers, texts, lens, int(max_src_len)


# BASELINE.json "configs" restated as named workloads
WORKLOADS = {
    # name: (model config, batch, L, frames/phoneme)
    "cfg1_single": ("ljspeech", 1, 100, 8.0),
    "cfg2_b16": ("ljspeech", 16, 128, 8.0),
    "cfg3_b128_sharded": ("ljspeech", 128, 128, 8.0),
    "cfg4_d512": ("d512", 64, 128, 8.0),
    "cfg5_longform": ("ljspeech", 8, 128, 31.0),
    # config 5 names "Gaussian-upsample": the same batch with the reference's (unwired) GaussianUpsampling module as
    # the length regulator — the extension of SURVEY.md §8 f1; the plain cfg5_longform is the reference's real path
    "cfg5_longform_gaussian": ("ljspeech+gaussian", 8, 128, 31.0),
    # not a BASELINE config: a handful of utterances (B*T ~ 4000 rows), the size between the single-utterance and the
    # chip-filling regime; used for tile-rule A/Bs (tools/ab_forward.sh) only
    "mid_b4": ("ljspeech", 4, 128, 8.0),
    "mid_b10": ("ljspeech", 10, 128, 8.0),
}


def algorithmic_flops_per_frame(model_cfg: dict, S_dec: int, S_enc: int, frames_per_phoneme: float) -> float:
    """SURVEY.md §8(d) FLOP model (2*MAC) per mel frame at T_pad == T."""
    t = model_cfg["transformer"]
    vp = model_cfg["variance_predictor"]

    def fft(d, S, d_inner, ks):
        return 8 * d * d + 4 * S * d + 2 * ks[0] * d * d_inner + 2 * ks[1] * d_inner * d

    def pred(d, f, k):
        return 2 * k * d * f + 2 * k * f * f + 2 * f

    d_e, d_d = t["encoder_hidden"], t["decoder_hidden"]
    dec = t["decoder_layer"] * fft(d_d, S_dec, t["conv_filter_size"], t["conv_kernel_size"])
    enc = t["encoder_layer"] * fft(d_e, S_enc, t["conv_filter_size"], t["conv_kernel_size"]) / frames_per_phoneme
    pe = 2 * pred(d_e, vp["filter_size"], vp["kernel_size"])
    du = pred(d_e, vp["filter_size"], vp["kernel_size"]) / frames_per_phoneme
    post = 2 * POSTNET_K * (N_MEL * POSTNET_DIM + (POSTNET_N - 2) * POSTNET_DIM ** 2 + POSTNET_DIM * N_MEL)
    lin = 2 * d_d * N_MEL
    return float(dec + enc + pe + du + post + lin)
