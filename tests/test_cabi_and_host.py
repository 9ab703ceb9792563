"""CPU-only checks: the C-ABI library builds/loads and exports every symbol include/nar_fs2.h declares,
host-side validation (config / state-dict errors) and the workload definition.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()  # hipcc cross-compiles gfx950 without a GPU; no-op when up to date
    import smart_nar_fast_tts_amd._lib as L

    return L, L.load()


def test_header_symbols_are_bound_and_exported(lib):
    L, so = lib
    text = open(os.path.join(ROOT, "include", "nar_fs2.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(ns_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 30
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    for name in declared:
        assert hasattr(so, name), name


def test_config_struct_matches_header(lib):
    L, so = lib
    text = open(os.path.join(ROOT, "include", "nar_fs2.h")).read()
    body = re.search(r"typedef struct ns_config \{(.*?)\} ns_config;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [f.strip() for decl in re.findall(r"int32_t ([^;]+);", body) for f in decl.split(",")]
    assert fields == [f for f, _ in L.NsConfig._fields_]


def _cfg(L, **over):
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import config_struct

    c = config_struct(wl.preprocess_config(), wl.model_config("tiny"))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def test_create_validates_config(lib):
    L, so = lib
    h = C.c_void_p()
    assert so.ns_create(C.byref(_cfg(L)), C.byref(h)) == 0
    assert so.ns_arena_bytes(h) > 30e6
    so.ns_destroy(h)
    for over, msg in ((dict(n_enc_head=3), "divisible"), (dict(n_dec_head=16), "d_k"), (dict(ffn_k1=8), "odd"),
                      (dict(vp_kernel=5), "padding=1"), (dict(d_dec=512), "encoder_hidden")):
        assert so.ns_create(C.byref(_cfg(L, **over)), C.byref(h)) != 0
        assert msg in so.ns_last_error().decode(), (over, so.ns_last_error())


def test_set_weight_validates_keys_and_shapes(lib):
    L, so = lib
    h = C.c_void_p()
    assert so.ns_create(C.byref(_cfg(L)), C.byref(h)) == 0
    a = np.zeros((80, 256), np.float32)

    def setw(name, arr):
        shape = (C.c_int64 * arr.ndim)(*arr.shape)
        return so.ns_set_weight(h, name.encode(), C.c_void_p(arr.ctypes.data), shape, arr.ndim)

    assert setw("mel_linear.weight", a) == 0
    assert setw("mel_linear.weight", a[:, :100].copy()) != 0 and "size mismatch" in so.ns_last_error().decode()
    assert setw("mel_linear.weight", a.reshape(-1)) != 0 and "rank mismatch" in so.ns_last_error().decode()
    assert setw("no.such.key", a) != 0 and "unexpected key" in so.ns_last_error().decode()
    # training-only aligner weights and BN counters are accepted and ignored
    assert setw("mel_encoder.prenet.w_1.weight", a) == 0
    assert setw("postnet.convolutions.0.1.num_batches_tracked", np.zeros((), np.float32)) == 0
    # ns_check_weight: the same verdicts without staging anything
    def chk(name, arr):
        return so.ns_check_weight(h, name.encode(), (C.c_int64 * arr.ndim)(*arr.shape), arr.ndim)

    assert chk("mel_linear.weight", a) == 0 and chk("mel_encoder.prenet.w_1.weight", a) == 0
    assert chk("mel_linear.weight", a[:, :100]) != 0 and "size mismatch" in so.ns_last_error().decode()
    assert chk("mel_linear.weight", a.reshape(-1)) != 0 and "rank mismatch" in so.ns_last_error().decode()
    assert chk("no.such.key", a) != 0 and "unexpected key" in so.ns_last_error().decode()
    # finalize without an arena / with missing keys fails loudly (host-side checks come first)
    assert so.ns_finalize_weights(h, None) != 0 and "arena" in so.ns_last_error().decode()
    so.ns_destroy(h)


def test_forward_refuses_without_weights(lib):
    L, so = lib
    h = C.c_void_p()
    assert so.ns_create(C.byref(_cfg(L)), C.byref(h)) == 0
    rc = so.ns_forward_durations(h, None, None, 1, 4, 1.0, 1.0, 1.0, None, None, None, 0, None, None, None, None, None, None,
                                 None, None)
    assert rc != 0 and "weights not loaded" in so.ns_last_error().decode()
    so.ns_destroy(h)


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of computing on the host."""
    import torch

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    m = FastSpeech2Align(wl.preprocess_config(), wl.model_config("tiny"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.to("cpu")
    sp, tx, ln, L = wl.synth_inputs(1, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.from_numpy(sp), torch.from_numpy(tx), torch.from_numpy(ln), L)
    pkg = os.path.join(ROOT, "smart-nar_fast_tts_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle|fs2_oracle", src, flags=re.M), \
                f"{fn}: the product path must not import the oracle"


def test_workload_definition():
    import smart_nar_fast_tts_amd.workload as wl

    cfg = wl.model_config()
    a, b = wl.synth_state_dict(cfg, seed=0), wl.synth_state_dict(cfg, seed=0)
    assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
    n = sum(v.size for k, v in a.items() if "position_enc" not in k and "num_batches" not in k)
    assert abs(n - 28.9e6) < 0.1e6  # 28.9 M inference parameters
    # SURVEY.md §8(d) FLOP model: 40.6 MFLOP per frame at config 2, 109.0 at config 4
    assert abs(wl.algorithmic_flops_per_frame(cfg, 1024, 128, 8.0) / 1e6 - 40.6) < 0.1
    assert abs(wl.algorithmic_flops_per_frame(wl.model_config("d512"), 1024, 128, 8.0) / 1e6 - 109.0) < 0.5
    sp, tx, ln, L = wl.synth_inputs(4, 10, seed=1, src_lens=[10, 3, 0, 7])
    assert tx.shape == (4, 10) and (tx[1, 3:] == 0).all() and (tx[2] == 0).all() and tx[0].min() >= 1
    # the first rows of a larger batch are the same utterances (sharding relies on it)
    _, tx2, _, _ = wl.synth_inputs(8, 10, seed=1)
    _, tx1, _, _ = wl.synth_inputs(4, 10, seed=1)
    assert np.array_equal(tx2[:4], tx1)


def _stats_dir(tmp_path):
    import json

    import smart_nar_fast_tts_amd.workload as wl

    os.makedirs(tmp_path / "pre", exist_ok=True)
    with open(tmp_path / "pre" / "stats.json", "w") as f:
        json.dump(wl.SYNTH_STATS, f)
    pc = wl.preprocess_config()
    pc["path"]["preprocessed_path"] = str(tmp_path / "pre")
    return pc


def test_stats_json_does_not_make_a_partial_state_dict(tmp_path, monkeypatch):
    """with <preprocessed_path>/stats.json present — every real deployment — the constructor must not
    leave a 2-key state dict behind that `.to(device)` then tries to upload (`missing keys: ...`) before a checkpoint
    can be loaded (checkpoint.get_model: construct -> .to(device) -> load).  Host-side: the upload calls are recorded."""
    import torch

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    m = FastSpeech2Align(_stats_dir(tmp_path), cfg)
    assert len(m._sd) == 0 and not m._loaded and m._stats is not None
    calls = []
    monkeypatch.setattr(FastSpeech2Align, "_bind_arena", lambda self: calls.append("bind"))
    monkeypatch.setattr(FastSpeech2Align, "_upload", lambda self, staged=False: calls.append("upload"))
    m.to(torch.device("cuda", 0))       # get_model's order: .to(device) BEFORE load_state_dict
    assert calls == []                   # nothing to upload yet, and no arena to re-bind
    # a checkpoint WITHOUT the bin buffers still loads: stats.json supplies them as defaults
    sd = {k: v for k, v in wl.synth_state_dict(cfg).items() if not k.endswith("_bins")}
    m.load_state_dict(sd)
    assert calls == ["upload"] and m._loaded
    pb, eb = wl.variance_bins(cfg, wl.SYNTH_STATS)
    assert np.array_equal(m._sd["variance_adaptor.pitch_bins"], pb) and np.array_equal(m._sd["variance_adaptor.energy_bins"], eb)
    # a checkpoint's own bins override the stats defaults
    sd2 = dict(wl.synth_state_dict(cfg))
    sd2["variance_adaptor.pitch_bins"] = (pb * 2).astype(np.float32)
    m.load_state_dict(sd2)
    assert np.array_equal(m._sd["variance_adaptor.pitch_bins"], pb * 2)


def test_load_state_dict_rejects_bad_keys_without_poisoning(tmp_path, monkeypatch):
    """a key is validated by the native side BEFORE it is kept, so one unexpected key cannot poison
    later uploads; a device change re-binds the arena (ADVICE medium) and re-uploads a loaded state dict."""
    import torch

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    m = FastSpeech2Align(wl.preprocess_config(), cfg)
    calls = []
    monkeypatch.setattr(FastSpeech2Align, "_bind_arena", lambda self: calls.append("bind") or setattr(self, "_arena", object()))
    monkeypatch.setattr(FastSpeech2Align, "_upload", lambda self, staged=False: calls.append("upload"))
    m._device = torch.device("cuda", 0)
    good = wl.synth_state_dict(cfg)
    with pytest.raises(RuntimeError, match="unexpected key"):
        m.load_state_dict(dict(good, **{"not.a.key": np.zeros(3, np.float32)}))
    assert len(m._sd) == 0 and not m._loaded and calls == []
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(dict(good, **{"mel_linear.bias": np.zeros(81, np.float32)}))
    assert len(m._sd) == 0
    m.load_state_dict(good)
    assert m._loaded and calls == ["upload"] and "not.a.key" not in m._sd
    # a rejected load on an ALREADY LOADED model must not touch native state (every successful
    # ns_set_weight overwrites a staging buffer and marks the handle not-ready): no staging call, no upload
    staged = []
    real_stage = FastSpeech2Align._stage
    monkeypatch.setattr(FastSpeech2Align, "_stage", lambda self, k, a: staged.append(k) or real_stage(self, k, a))
    before = {k: v.copy() for k, v in m._sd.items()}
    for bad, msg in (({"not.a.key": np.zeros(3, np.float32)}, "unexpected key"),
                     ({"mel_linear.bias": np.zeros(81, np.float32)}, "size mismatch")):
        with pytest.raises(RuntimeError, match=msg):
            m.load_state_dict(dict({"mel_linear.weight": np.ones((80, 256), np.float32)}, **bad))
    assert staged == [] and calls == ["upload"] and all(np.array_equal(before[k], m._sd[k]) for k in before)
    # strict=False: the unexpected key is skipped and reported, the rest is loaded (nn.Module.load_state_dict semantics)
    missing, unexpected = m.load_state_dict({"mel_linear.bias": np.full(80, 0.5, np.float32), "not.a.key": np.zeros(3, np.float32)}, strict=False)
    assert missing == [] and unexpected == ["not.a.key"] and (m._sd["mel_linear.bias"] == 0.5).all() and calls == ["upload", "upload"]
    calls.pop()
    # a first load that lacks keys: strict raises naming them, strict=False fills them like the constructor would
    m2 = FastSpeech2Align(wl.preprocess_config(), cfg)
    m2._device = torch.device("cuda", 0)
    part = {k: v for k, v in good.items() if not k.startswith("mel_linear.")}
    with pytest.raises(RuntimeError, match="missing key.*mel_linear.weight"):
        m2.load_state_dict(part)
    assert not m2._loaded
    m2._stats = wl.SYNTH_STATS
    m2.load_state_dict(part, strict=False)
    assert m2._loaded and m2._sd["mel_linear.weight"].shape == (80, 256) and m2._sd["mel_linear.weight"].std() > 0
    calls.pop()
    m._arena = object()
    m.to(torch.device("cuda", 1))  # device change: never keep the old device's arena bound
    assert calls == ["upload", "bind", "upload"] and m._device == torch.device("cuda", 1) and len(m._ws) == 0


def test_workspace_cache_is_bounded():
    """scratch sets are keyed by stream handle; the cache must not grow with every new stream."""
    from collections import OrderedDict

    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cap = FastSpeech2Align.MAX_WORKSPACE_STREAMS
    ws = OrderedDict()
    for i in range(50):  # what a forward on a fresh stream does per call, without a device: four entries per stream
        for kind in ("enc", "pin", "lens", "dec"):
            ws[(kind, i)] = i
            ws.move_to_end((kind, i))
            FastSpeech2Align._evict_streams(ws, cap)
    assert len(ws) == 4 * cap and {k[1] for k in ws} == set(range(50 - cap, 50))
    # round-robin over as many streams as the bound keeps: nothing is ever evicted (the stream pool of batching.synthesize)
    ws = OrderedDict()
    for i in range(5 * cap):
        for kind in ("enc", "pin", "lens", "dec"):
            ws[(kind, i % cap)] = ws.get((kind, i % cap), i)
            ws.move_to_end((kind, i % cap))
            FastSpeech2Align._evict_streams(ws, cap)
    assert len(ws) == 4 * cap and all(v < cap for v in ws.values())
    # a stream is judged by its MOST RECENT use: touching one entry of the oldest stream keeps its whole set
    ws = OrderedDict(((kind, st), 0) for st in range(cap) for kind in ("enc", "dec"))
    ws.move_to_end(("enc", 0))
    ws[("enc", cap)] = 0
    FastSpeech2Align._evict_streams(ws, cap)
    assert ("dec", 0) in ws and ("enc", 1) not in ws and ("dec", 1) not in ws


def test_default_init_state_dict_follows_torch_initialisers():
    """Random-init construction (model/fastspeech2_align.py:16-28): same key set and shapes as a checkpoint, torch's
    default initialiser bounds, controlled by torch.manual_seed."""
    import torch

    import smart_nar_fast_tts_amd.workload as wl

    cfg = wl.model_config("tiny")
    torch.manual_seed(3)
    a = wl.default_init_state_dict(cfg, wl.SYNTH_STATS)
    torch.manual_seed(3)
    b = wl.default_init_state_dict(cfg, wl.SYNTH_STATS)
    ref = {k: v for k, v in wl.synth_state_dict(cfg).items() if not k.endswith("position_enc") and not k.endswith("num_batches_tracked")}
    assert set(a) == set(ref) == set(wl.inference_shapes(cfg)) and all(a[k].shape == ref[k].shape for k in a)
    # every Linear / Conv1d weight AND bias is a random draw — only LayerNorm / BatchNorm entries may be constant
    # (round 2 matched ".1." as a substring and turned PostNet conv 1 into all-ones)
    for k, v in a.items():
        is_norm = "layer_norm" in k or re.fullmatch(r"postnet\.convolutions\.\d+\.1\..*", k)
        if v.size > 1 and not is_norm and not k.endswith("_bins"):
            assert v.min() != v.max(), f"{k} is constant"
    assert np.abs(a["postnet.convolutions.1.0.conv.weight"]).max() <= 1 / np.sqrt(512 * 5) and a["postnet.convolutions.1.0.conv.weight"].std() > 0.01
    assert all(np.array_equal(a[k], b[k]) for k in a)
    w = a["mel_decoder.layer_stack.0.pos_ffn.w_1.weight"]  # Conv1d [1024, 256, 9]: U(+-1/sqrt(256*9))
    assert w.shape == (1024, 256, 9) and 0.99 / 48 < np.abs(w).max() <= 1 / 48
    assert np.abs(a["mel_decoder.layer_stack.0.pos_ffn.w_1.bias"]).max() <= 1 / 48
    assert (a["txt_encoder.src_word_emb.weight"][0] == 0).all() and abs(a["txt_encoder.src_word_emb.weight"][1:].std() - 1) < 0.05
    assert (a["postnet.convolutions.2.1.running_var"] == 1).all() and (a["postnet.convolutions.2.1.running_mean"] == 0).all()
    assert (a["variance_adaptor.pitch_predictor.conv_layer.layer_norm_1.weight"] == 1).all()


def test_output_blocks_are_one_allocation_with_aligned_views():
    """model._OutputBlock: forward() makes ONE allocation per phase and cuts the caller's tensors out of it; phase 2's block
    is allocated at a guessed capacity and laid out again for the actual T (which must fit)."""
    import torch

    from smart_nar_fast_tts_amd.model import _OutputBlock

    f32, u8, i64 = torch.float32, torch.bool, torch.long

    def specs(B, T):
        return [("mel", (B, T, 80), f32), ("post", (B, T, 80), f32), ("mel_masks", (B, T), u8), ("p_pred", (B, T), f32), ("lens", (B,), i64)]

    blk = _OutputBlock(specs(3, 40), "cpu")
    cap = blk.nbytes
    base = blk.buf.data_ptr()
    blk.layout(specs(3, 33))  # the actual T turned out smaller than the capacity
    assert blk.nbytes <= cap
    seen = []
    for name, shape, dtype in specs(3, 33):
        v = blk.view(name)
        assert tuple(v.shape) == shape and v.dtype == dtype and v.is_contiguous()
        off = v.data_ptr() - base
        assert off % 256 == 0 and off == blk.ptr(name).value - base and off + v.numel() * v.element_size() <= cap
        seen.append((off, off + v.numel() * v.element_size()))
    seen.sort()
    assert all(a[1] <= b[0] for a, b in zip(seen, seen[1:])), "views overlap"
    blk.view("mel").fill_(1.0)
    blk.view("post").fill_(2.0)
    assert float(blk.view("mel").sum()) == 3 * 33 * 80 and float(blk.view("post").sum()) == 2 * 3 * 33 * 80  # (no aliasing)
    assert blk.ptr("absent").value in (None, 0)


def test_module_introspection_surface():
    """What code that walks an nn.Module finds on the drop-in (model/fastspeech2_align.py:13-28; utils/model.py:31-35 counts
    parameters): parameters() / named_parameters() over the host copies, BatchNorm running statistics as buffers, modules()
    = the object itself, float() a no-op and the casts that cannot be honoured raising clearly."""
    import torch

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align, ForwardOutput

    cfg = wl.model_config("tiny")
    m = FastSpeech2Align(wl.preprocess_config(), cfg)
    sd = wl.synth_state_dict(cfg)
    m.load_state_dict(sd)  # (no GPU here: the state dict is validated and kept on the host, uploaded on first use of a device)
    named = dict(m.named_parameters())
    bufs = dict(m.named_buffers())
    is_buf = lambda k: k.endswith((".running_mean", ".running_var", ".num_batches_tracked"))  # noqa: E731
    assert set(named) == {k for k in m._sd if not is_buf(k)} and set(bufs) == {k for k in m._sd if is_buf(k)}
    assert len(bufs) == 2 * 5 and all(torch.is_tensor(p) and p.dtype == torch.float32 and not p.requires_grad for p in named.values())
    # utils/model.py:31-35 get_param_num: sum(param.numel() for param in model.parameters())
    assert sum(p.numel() for p in m.parameters()) == sum(int(np.prod(np.shape(v))) for k, v in m._sd.items() if not is_buf(k))
    np.testing.assert_array_equal(named["mel_linear.weight"].numpy(), np.asarray(sd["mel_linear.weight"], dtype=np.float32))
    assert list(m.modules()) == [m] and list(m.children()) == [] and dict(m.named_modules()) == {"": m}
    assert m.float() is m and m.eval() is m
    for cast in (m.half, m.bfloat16, m.double):
        with pytest.raises(NotImplementedError, match="float32"):
            cast()
    with pytest.raises(NotImplementedError, match="wrap forward"):
        m.register_forward_hook(lambda *a: None)
    with pytest.raises(ValueError, match="outputs"):
        FastSpeech2Align(wl.preprocess_config(), dict(cfg, outputs="copies"))
    # the forward's return type is the reference's 12-tuple for every positional consumer
    out = ForwardOutput(tuple(range(12)), status=None)
    assert isinstance(out, tuple) and len(out) == 12 and out[9] == 9 and out.check() == [] and tuple(out) == tuple(range(12))
    a, b, *rest = out
    assert (a, b) == (0, 1) and len(rest) == 10


def test_introspection_before_a_load_does_not_count_as_loaded(tmp_path, monkeypatch):
    """parameters() / named_parameters() / buffers() on a model that has loaded nothing show the
    constructor-equivalent initialisation WITHOUT turning it into "loaded" state: a truncated checkpoint under strict=True still
    names its missing keys — also after the model ran on that initialisation — and what introspection showed is what the first
    forward uploads (one draw, not two)."""
    import torch

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    m = FastSpeech2Align(_stats_dir(tmp_path), cfg)
    monkeypatch.setattr(FastSpeech2Align, "_bind_arena", lambda self: setattr(self, "_arena", object()))
    monkeypatch.setattr(FastSpeech2Align, "_upload", lambda self, staged=False: None)
    m._device = torch.device("cuda", 0)
    shown = dict(m.named_parameters())
    assert len(shown) > 80 and len(m._sd) == 0 and not m._loaded and m._user_keys == set()
    # state_dict() of a fresh model holds the same full set (parameters + buffers), like nn.Module's — not {} (round-5 advice) —
    # and looking at it does not count as a load either
    fresh = m.state_dict()
    assert set(shown) <= set(fresh) and len(fresh) == len(shown) + len(dict(m.named_buffers())) and len(m._sd) == 0
    np.testing.assert_array_equal(fresh["mel_linear.weight"].numpy(), shown["mel_linear.weight"].numpy())
    good = wl.synth_state_dict(cfg)
    part = {k: v for k, v in good.items() if not k.startswith("mel_linear.")}
    with pytest.raises(RuntimeError, match="missing key.*mel_linear.weight"):
        m.load_state_dict(part)  # the introspection cache must not stand in for the absent keys
    assert not m._loaded and len(m._sd) == 0
    m._ensure_weights()  # what the first forward does on a model nobody loaded: uploads the SAME draw parameters() showed
    assert m._loaded and m._user_keys == set()
    np.testing.assert_array_equal(m._sd["mel_linear.weight"], shown["mel_linear.weight"].numpy())
    np.testing.assert_array_equal(dict(m.named_parameters())["mel_linear.weight"].numpy(), shown["mel_linear.weight"].numpy())
    with pytest.raises(RuntimeError, match="missing key.*mel_linear.weight"):
        m.load_state_dict(part)  # still the first CALLER-supplied load: random-init leftovers are not "loaded"
    np.testing.assert_array_equal(m._sd["mel_linear.weight"], shown["mel_linear.weight"].numpy())
    missing, _ = m.load_state_dict(part, strict=False)  # explicit opt-in: the absent keys keep the constructor's values
    assert missing == ["mel_linear.weight", "mel_linear.bias"]
    np.testing.assert_array_equal(m._sd["mel_linear.weight"], shown["mel_linear.weight"].numpy())
    np.testing.assert_array_equal(m._sd["mel_linear.bias"], shown["mel_linear.bias"].numpy())
    m.load_state_dict({k: v for k, v in good.items() if k.startswith("mel_linear.")})  # the rest arrives: nothing missing any more
    assert m._user_keys >= set(wl.inference_keys(cfg))
    m.load_state_dict({"mel_linear.bias": np.zeros(80, np.float32)})  # partial update of a fully caller-loaded model: fine


def test_launch_plan_invariants_and_settled_choices(lib):
    """The step-aware launch plan (include/nar_fs2.h ns_plan_gemm / ns_plan_attention_split) is host logic: checked here without a
    GPU.  Invariants for every row count: the launches cover exactly M rows, a cut falls on a whole number of the main tile's rows and
    of its full steps of 256 workgroups, the remainder's tile is no taller than the main one.  And the choices that round 3 settled by
    alternating whole-forward runs at the BASELINE configurations must come out of the cost model unchanged."""
    _, lib = lib

    def plan(M, N, Cin, KW):
        o = (C.c_int32 * 8)()
        return lib.ns_plan_gemm(M, N, Cin, KW, o), list(o)

    shapes = [(1024, 256, 9), (512, 512, 5), (768, 256, 1), (256, 1024, 1)]  # FFN w_1, PostNet 512->512, QKV, FFN w_2 (plain form)
    fam16 = 0
    for N, Cin, KW in shapes:
        for M in list(range(3000, 36000, 317)) + [8192, 8193, 16384, 16385, 65536 + 1088]:
            ok, (bm, bn, rows, rbm, rbn, rrows, mf, us) = plan(M, N, Cin, KW)
            if not ok:
                assert ((M + 63) // 64) * ((N + 127) // 128) <= 256, (M, N)
                continue
            assert rows + rrows == M and mf in (16, 32) and us > 0 and bn in (64, 128, 256), (M, N, bm, bn)
            long_k = Cin * KW > 256  # long contractions keep to the tiles with the second accumulator set (<= 32 registers per lane)
            if mf == 16:  # the 16-row family: ONE launch, any multiple of 16 rows per tile, 128 or 256 columns
                fam16 += 1
                assert rrows == 0 and bm % 16 == 0 and 48 <= bm <= (128 if long_k else 256) and bn in (128, 256) and N % bn == 0, (M, N, bm, bn)
                continue
            assert bm in (32, 64, 128) + (() if long_k else (256,)), (M, N, bm, bn)
            if rrows:
                ntn = -(-N // bn)
                per = (256 // ntn) * bm  # rows of one full step of the main tile
                assert rows % per == 0 and rows % bm == 0 and rbm <= bm and rbm * rbn <= bm * bn, (M, N, bm, bn, rows, rbm, rbn, rrows)
            if Cin * KW <= 512:
                assert bm <= 64, ("short contractions stay on the 8-wave tiles", M, N, bm, bn)
    assert fam16 > 50, fam16  # between the steps of the tall tiles the family is what the plan picks
    assert lib.ns_abi_version() == 6 and lib.ns_acc_chunk() == 64
    # BASELINE configurations (B*T_pad rows): config 2, config 5, config 4.  Round 6: the k=9 / k=5 contractions accumulate in
    # chunks, which needs a second accumulator set — the 256x256 tile (64 registers per lane) has no room, two rounds of 128x256 do
    assert plan(16160, 1024, 256, 9)[1][:7] == [128, 256, 16160, 0, 0, 0, 32]      # FFN w_1: two rounds of the 128x256 tile
    assert plan(16160, 512, 512, 5)[1][:7] == [128, 256, 16160, 0, 0, 0, 32]       # PostNet 512->512: one round of 128x256
    assert plan(16160, 768, 256, 1)[1][:7] == [192, 256, 16160, 0, 0, 0, 16]       # QKV (K = 256, one chunk: the tall tiles stay): one step of 255 tall tiles
    assert plan(31248, 1024, 256, 9)[1][:7] == [128, 256, 31248, 0, 0, 0, 32]      # config 5: four rounds
    ok, p4 = plan(66624, 1024, 512, 9)                                       # config 4: many rounds of a tile of at most 128 rows
    assert ok and p4[0] <= 128 and p4[1] == 256 and p4[2] + p4[5] == 66624
    # between the steps (B = 9, 10 utterances of 1010 frames; the ragged config-2 batch's packed rows): one launch of a tile as tall as
    # the rows ask for, at most 128 rows — 9090 rows x 4 column tiles = 190 row tiles of 48 rows = 3 per CU
    assert plan(9090, 1024, 256, 9)[1][:7] == [48, 256, 9090, 0, 0, 0, 16]
    assert plan(10100, 1024, 256, 9)[1][:7] == [80, 256, 10100, 0, 0, 0, 16]
    assert plan(10490, 1024, 256, 9)[1][6] == 16 and plan(9090, 512, 512, 5)[1][6] == 16
    # the full-row (GEMM + LayerNorm epilogue) tile: as tall as the fullest CU needs, 32 on ties; 512 columns: 32 or 48 rows only
    # (every 512-wide full-row GEMM contracts over K >= 512, and the taller 512-column tiles have no room for the second set)
    rt = {M: lib.ns_plan_row_tile(M, 256) for M in (8080, 9090, 10490, 11110, 12120, 16160, 17170, 20200, 31248)}
    assert rt == {8080: 32, 9090: 48, 10490: 48, 11110: 48, 12120: 48, 16160: 32, 17170: 80, 20200: 80, 31248: 32}, rt
    assert all(lib.ns_plan_row_tile_k(M, 256, 1024) == v for M, v in rt.items())
    assert lib.ns_plan_row_tile(66624, 512) == 32 and lib.ns_plan_row_tile(9090, 512) == 48 and lib.ns_plan_row_tile(9090, 300) == 0
    assert lib.ns_plan_row_tile(17170, 512) == 32 and lib.ns_plan_row_tile_k(25250, 512, 1024) == 32
    # "time follows the rows": the model's cost per utterance for the dominant launch never jumps by more than 6 % from B to B + 1
    # utterances of 1010 frames (round 4's plans: +19 % at B = 8 -> 9, +16 % at 16 -> 17 in the measured forward), and it is
    # monotone in the rows up to one microsecond of rounding
    per = {B: plan(B * 1010, 1024, 256, 9)[1][7] / B for B in range(5, 33)}
    worst = max(per[B + 1] / per[B] for B in range(5, 32))
    assert worst <= 1.06, (worst, per)
    us = [plan(M, 1024, 256, 9)[1][7] for M in range(5000, 33000, 250)]
    assert all(b >= a - 1 for a, b in zip(us, us[1:])), us
    # below the planner's range: the small-grid K-split ladder (encoder rows, single utterances)
    assert plan(2048, 1024, 256, 9)[0] == 0 and plan(788, 512, 512, 5)[0] == 0 and plan(16160, 80, 512, 5)[0] == 0
    # attention: one workgroup per CU -> a key split only when the last round of 256 fills badly
    split = {B: lib.ns_plan_attention_split(B, 1010, 2, 128) for B in (9, 12, 16, 17, 20, 24, 32)}
    assert split[16] == 1 and split[32] == 1 and split[9] == 3 and split[17] >= 4 and split[20] == 4 and split[24] == 2, split
    assert lib.ns_plan_attention_split(1, 788, 2, 128) == 16 and lib.ns_plan_attention_split(64, 1041, 8, 64) == 1


def test_header_is_plain_c_and_host_entry_points_work_from_c(lib, tmp_path):
    """include/nar_fs2.h is the drop-in boundary: it must compile as C99 (not only as the C++ the library is written in) and the
    host-side entry points must work from a plain C program through dlopen — no Python, no torch, no GPU."""
    import subprocess

    L, _ = lib
    exe = tmp_path / "host_only"
    src = os.path.join(ROOT, "tests", "cabi", "host_only.c")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", str(exe), "-ldl"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(exe), L.LIB_PATH if hasattr(L, "LIB_PATH") else os.path.join(ROOT, "smart-nar_fast_tts_amd", "csrc", "libnarfs2.so")],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "C caller ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    # the struct the C side sees is the struct the ctypes binding declares
    assert f"sizeof(ns_config)={C.sizeof(L.NsConfig)} " in r.stdout


def test_no_kernel_spills_registers():
    """Register hygiene gate (round-5 review): no kernel of the GEMM / row-operator sources spills a VGPR or an SGPR or uses
    scratch memory — tools/kernel_resources.py compiles the source for gfx950 (hipcc cross-compiles without a GPU) and reads the
    kernels' metadata.  Round 5 shipped 36-43 spilled SGPRs in the two-pass full-row tiles, 10-18 in two ticketed rungs and 18 in
    k_gauss_upsample; profiles/r06_kernel_resources.txt is the table for all five sources."""
    import subprocess
    import sys

    for src in ("gemm_conv.hip", "rowops.hip"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), os.path.join(ROOT, "smart-nar_fast_tts_amd", "csrc", src),
                            "--assert-no-spill"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (src, r.stdout[-2000:], r.stderr[-2000:])
