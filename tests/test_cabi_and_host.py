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
    # finalize without an arena / with missing keys fails loudly (host-side checks come first)
    assert so.ns_finalize_weights(h, None) != 0 and "arena" in so.ns_last_error().decode()
    so.ns_destroy(h)


def test_forward_refuses_without_weights(lib):
    L, so = lib
    h = C.c_void_p()
    assert so.ns_create(C.byref(_cfg(L)), C.byref(h)) == 0
    rc = so.ns_forward_durations(h, None, None, 1, 4, 1.0, 1.0, 1.0, None, None, None, 0, None, None, None, None, None, None,
                                 None)
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
