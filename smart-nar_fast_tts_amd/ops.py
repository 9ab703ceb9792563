"""Per-operator Python wrappers over the C-ABI's ``ns_op_*`` entry points — one per row of
SURVEY.md §8(a).  They exist so the parity tests can check every stage of the path against
the oracle in isolation, and so ``bench.py`` can time the dominant kernel alone.

Every function takes CUDA tensors, launches on torch's current stream and returns fresh
tensors.  ``model`` is a :class:`smart_nar_fast_tts_amd.model.FastSpeech2Align` with loaded
weights; ``prefix`` is the reference's module path (e.g. ``"mel_decoder.layer_stack.0.slf_attn"``);
``lens`` is the int64 ``[B]`` valid-length vector from which the reference builds its masks.
"""
from __future__ import annotations

import math

import torch

from . import _lib


def _ws(model, B, S):
    n = model._lib.ns_op_ws_bytes(model._h, B, S)
    return model._workspace("op", n)


def _st(t):
    return _lib.stream_ptr(t.device)


def mask_from_lengths(lens: torch.Tensor, max_len: int | None = None) -> torch.Tensor:
    """utils/tools.py:89-97 (True = padding).  max_len=None costs a host sync, as in the reference."""
    lib = _lib.load()
    lens = lens.long().contiguous()
    if max_len is None:
        max_len = int(lens.max().item())
    out = torch.empty(lens.shape[0], max_len, dtype=torch.bool, device=lens.device)
    _lib.check(lib.ns_op_mask_from_lengths(_lib.ptr(lens), lens.shape[0], int(max_len), _lib.ptr(out), _st(lens)), "mask")
    return out


def sinusoid_table(n_position: int, d_hid: int, device="cuda") -> torch.Tensor:
    """transformer/Models.py:10-30."""
    lib = _lib.load()
    out = torch.empty(n_position, d_hid, dtype=torch.float32, device=device)
    _lib.check(lib.ns_op_sinusoid_table(n_position, d_hid, _lib.ptr(out), _st(out)), "sinusoid")
    return out


def txt_encoder(model, texts, lens):
    B, L = texts.shape
    texts, lens = texts.long().contiguous(), lens.long().contiguous()
    out = torch.empty(B, L, model._cfg.d_enc, dtype=torch.float32, device=texts.device)
    ws = _ws(model, B, L)
    _lib.check(model._lib.ns_op_txt_encoder(model._h, _lib.ptr(texts), _lib.ptr(lens), B, L, _lib.ptr(out), _lib.ptr(ws),
                                            ws.numel(), _st(texts)), "txt_encoder")
    return out


def _prefixed(fn_name, model, prefix, x, lens):
    B, S, _ = x.shape
    x = x.contiguous()
    out = torch.empty_like(x)
    ws = _ws(model, B, S)
    fn = getattr(model._lib, fn_name)
    if lens is None:
        rc = fn(model._h, prefix.encode(), _lib.ptr(x), B, S, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _st(x))
    else:
        lens = lens.long().contiguous()
        rc = fn(model._h, prefix.encode(), _lib.ptr(x), _lib.ptr(lens), B, S, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _st(x))
    _lib.check(rc, fn_name)
    return out


def multi_head_attention(model, prefix, x, lens):
    """transformer/SubLayers.py:29-59 (self attention: q = k = v = x); returns LayerNorm(fc(attn) + x)."""
    return _prefixed("ns_op_multi_head_attention", model, prefix, x, lens)


def positionwise_ffn(model, prefix, x):
    """transformer/SubLayers.py:87-95."""
    return _prefixed("ns_op_positionwise_ffn", model, prefix, x, None)


def fft_block(model, prefix, x, lens):
    """transformer/Layers.py:39-48."""
    return _prefixed("ns_op_fft_block", model, prefix, x, lens)


def variance_predictor(model, prefix, x, lens):
    """model/modules.py:278-286; returns [B,S]."""
    B, S, _ = x.shape
    x, lens = x.contiguous(), lens.long().contiguous()
    out = torch.empty(B, S, dtype=torch.float32, device=x.device)
    ws = _ws(model, B, S)
    _lib.check(model._lib.ns_op_variance_predictor(model._h, prefix.encode(), _lib.ptr(x), _lib.ptr(lens), B, S, _lib.ptr(out),
                                                   _lib.ptr(ws), ws.numel(), _st(x)), "variance_predictor")
    return out


def duration_round(log_d, d_control: float = 1.0):
    """model/modules.py:132-135."""
    lib = _lib.load()
    log_d = log_d.contiguous()
    out = torch.empty_like(log_d)
    _lib.check(lib.ns_op_duration_round(_lib.ptr(log_d), log_d.numel(), float(d_control), _lib.ptr(out), _st(log_d)), "duration_round")
    return out


def length_regulate(x, duration, max_len=None):
    """LengthRegulator.forward (model/modules.py:201-230): returns (output [B,T,D], mel_len int64 [B])."""
    lib = _lib.load()
    B, L, D = x.shape
    x, duration = x.contiguous(), duration.contiguous().float()
    cum = torch.empty(B, L, dtype=torch.int32, device=x.device)
    mel_len = torch.empty(B, dtype=torch.long, device=x.device)
    _lib.check(lib.ns_op_duration_scan(_lib.ptr(duration), B, L, _lib.ptr(cum), _lib.ptr(mel_len), _st(x)), "duration_scan")
    T = int(max_len) if max_len else int(mel_len.max().item())
    out = torch.empty(B, T, D, dtype=torch.float32, device=x.device)
    _lib.check(lib.ns_op_length_regulate(_lib.ptr(x), _lib.ptr(cum), B, L, D, T, _lib.ptr(out), _st(x)), "length_regulate")
    return out, mel_len


def variance_embedding(model, which: str, x, lens, control: float = 1.0, target=None):
    """get_pitch_embedding / get_energy_embedding + the unmasked add (model/modules.py:80-100,139-149):
    returns (prediction [B,S], x + embedding).  With ``target`` the embedding comes from bucketize(target)."""
    B, S, _ = x.shape
    x, lens = x.contiguous(), lens.long().contiguous()
    pred = torch.empty(B, S, dtype=torch.float32, device=x.device)
    x_out = torch.empty_like(x)
    ws = _ws(model, B, S)
    if target is not None:
        target = target.contiguous().float()
    _lib.check(model._lib.ns_op_variance_embedding(model._h, {"pitch": 0, "energy": 1}[which], _lib.ptr(x), _lib.ptr(lens), B, S,
                                                   float(control), _lib.ptr(target), _lib.ptr(pred), _lib.ptr(x_out),
                                                   _lib.ptr(ws), ws.numel(), _st(x)), "variance_embedding")
    return pred, x_out


def bucketize(values, bins):
    """torch.bucketize(values, bins) with right=False, as model/modules.py:86-88,97-99 calls it."""
    lib = _lib.load()
    values, bins = values.contiguous().float(), bins.contiguous().float()
    out = torch.empty(values.shape, dtype=torch.long, device=values.device)
    _lib.check(lib.ns_op_bucketize(_lib.ptr(values), values.numel(), _lib.ptr(bins), bins.numel(), _lib.ptr(out), _st(values)), "bucketize")
    return out


def gaussian_upsampling(x, durations, max_len=None):
    """GaussianUpsampling.forward (model/modules.py:166-192): returns (output, s [B,1], w [B,L,T])."""
    lib = _lib.load()
    B, L, D = x.shape
    x, durations = x.contiguous(), durations.contiguous().float()
    # torch.arange(0, torch.max(s)) in the reference (also a host sync there): ceil(max s) frames for a fractional sum
    T = int(math.ceil(durations.sum(dim=-1).max().item()))
    T_out = int(max_len) if max_len else T
    out = torch.empty(B, T_out, D, dtype=torch.float32, device=x.device)
    s = torch.empty(B * (L + 1), dtype=torch.float32, device=x.device)
    w = torch.empty(B, L, T, dtype=torch.float32, device=x.device)
    _lib.check(lib.ns_op_gaussian_upsampling(_lib.ptr(x), _lib.ptr(durations), B, L, D, T, T_out, _lib.ptr(out), _lib.ptr(s),
                                             _lib.ptr(w), _st(x)), "gaussian_upsampling")
    return out, s[:B].reshape(B, 1).clone(), w


def mel_decoder(model, x, lens):
    """transformer/Models.py:212-244."""
    B, T, _ = x.shape
    x, lens = x.contiguous(), lens.long().contiguous()
    out = torch.empty_like(x)
    ws = _ws(model, B, T)
    _lib.check(model._lib.ns_op_mel_decoder(model._h, _lib.ptr(x), _lib.ptr(lens), B, T, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                            _st(x)), "mel_decoder")
    return out


def mel_linear(model, x):
    B, T, _ = x.shape
    x = x.contiguous()
    out = torch.empty(B, T, model._cfg.n_mel, dtype=torch.float32, device=x.device)
    _lib.check(model._lib.ns_op_mel_linear(model._h, _lib.ptr(x), B, T, _lib.ptr(out), _st(x)), "mel_linear")
    return out


def postnet(model, mel):
    """PostNet.forward (transformer/Layers.py:169-177), without the residual."""
    B, T, _ = mel.shape
    mel = mel.contiguous()
    out = torch.empty_like(mel)
    ws = _ws(model, B, T)
    _lib.check(model._lib.ns_op_postnet(model._h, _lib.ptr(mel), B, T, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _st(mel)), "postnet")
    return out


def ffn_conv1(model, prefix, x, out=None):
    """The path's dominant kernel alone: relu(w_1(x)) of PositionwiseFeedForward (k=9 Conv1D-as-GEMM)."""
    B, S, _ = x.shape
    if out is None:
        out = torch.empty(B, S, model._cfg.d_inner, dtype=torch.float32, device=x.device)
    _lib.check(model._lib.ns_op_ffn_conv1(model._h, prefix.encode(), _lib.ptr(x), B, S, _lib.ptr(out), _st(x)), "ffn_conv1")
    return out


def attention_core(qkv, lens, n_head: int, split_scratch: bool = True):
    """ScaledDotProductAttention on already-projected, head-packed q/k/v (transformer/Modules.py:14-25):
    qkv [B,S,3*d] (Q | K | V, head h at h*dk inside each) -> merged heads [B,S,d].  ``split_scratch`` hands the
    kernel the scratch it needs to take its split-key path on small grids."""
    lib = _lib.load()
    B, S, d3 = qkv.shape
    d = d3 // 3
    qkv = qkv.contiguous()
    out = torch.empty(B, S, d, dtype=torch.float32, device=qkv.device)
    lens_p = _lib.ptr(lens.long().contiguous()) if lens is not None else _lib.ptr(None)
    scratch = torch.empty(8 * (B * S * d + 2 * B * S * n_head), dtype=torch.float32, device=qkv.device) if split_scratch else None
    _lib.check(lib.ns_op_attention_core(_lib.ptr(qkv), lens_p, B, S, n_head, d // n_head, _lib.ptr(out), _lib.ptr(scratch),
                                        0 if scratch is None else scratch.numel() * 4, _st(qkv)), "attention_core")
    return out
