"""§8 f2 (batch construction / output slicing) and §8 f3 (checkpoint reader): host-side logic on CPU, plus a GPU
round trip through a checkpoint file."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import load_golden


def _data(z, meta):
    off = np.concatenate([[0], np.cumsum(z["phone_lens"])])
    return [(meta["ids"][i], int(z["speaker_ids"][i]), z["phones"][off[i]:off[i + 1]], meta["raw_texts"][i])
            for i in range(len(meta["ids"]))]


def test_collate_pad_expand_match_reference():
    from smart_nar_fast_tts_amd import batching

    meta, z = load_golden("kat_batching")
    ids, raw_texts, speakers, texts, text_lens, max_len = batching.collate(_data(z, meta))
    assert ids == meta["ids"] and raw_texts == meta["raw_texts"] and int(max_len) == meta["max_len"]
    for got, key in ((speakers, "speakers"), (texts, "texts"), (text_lens, "text_lens")):
        assert got.dtype == z[key].dtype and np.array_equal(got, z[key]), key
    assert np.array_equal(batching.pad_1D([d[2] for d in _data(z, meta)], 5), z["pad1d"])
    assert np.array_equal(batching.expand(z["expand_vals"], z["expand_durs"]), z["expand_out"])
    dev = batching.to_device((ids, raw_texts, speakers, texts, text_lens, max_len), "cpu")
    assert dev[2].dtype == torch.long and dev[3].dtype == torch.long and dev[3].shape == (5, 19) and dev[5] == max_len


def test_bucket_by_length_partitions():
    from smart_nar_fast_tts_amd.batching import bucket_by_length

    rs = np.random.RandomState(0)
    lens = rs.randint(5, 200, size=97).tolist()
    for max_batch, frac in ((16, 0.1), (4, 0.0), (128, 1.0)):
        b = bucket_by_length(lens, max_batch, frac)
        assert sorted(i for g in b for i in g) == list(range(97))
        for g in b:
            assert 1 <= len(g) <= max_batch
            longest = max(lens[i] for i in g)
            assert all(longest - lens[i] <= frac * longest for i in g)
    assert len(bucket_by_length(lens, 128, 1.0)) == 1
    # step-friendly cutting (round 5): a row-based rule that asks the launch plan's own cost model (ns_plan_gemm, host-side) whether
    # two forwards are cheaper than one — no list of batch sizes tuned at one utterance length
    from smart_nar_fast_tts_amd.batching import forward_cost_us, step_friendly_cuts, step_friendly_sizes

    for n in range(0, 40):
        for mb in (1, 3, 8, 16, 32):
            c = step_friendly_sizes(n, mb)
            assert sum(c) == n and all(1 <= x <= mb for x in c)
    # with tiles as tall as the rows ask for, a uniform group is cheapest whole at every utterance length: T ~ 300, 1000, 3000
    for frames in (300, 1000, 3000):
        for n in (5, 9, 11, 13, 17, 20, 27):
            assert step_friendly_sizes(n, 32, frames_per_utterance=frames) == [n], (frames, n, step_friendly_sizes(n, 32, frames_per_utterance=frames))
    assert step_friendly_sizes(33, 32) in ([32, 1], [17, 16], [16, 17], [1, 32]) or len(step_friendly_sizes(33, 32)) == 2
    # the modelled cost grows with the rows and a cut is only taken where it removes padding worth more than a forward's fixed cost
    assert forward_cost_us(4000) < forward_cost_us(9090) < forward_cost_us(16160) < forward_cost_us(40000)
    assert step_friendly_cuts([3000] + [300] * 12, 32) == [1, 12]          # one long utterance would pad twelve short ones 10x
    assert step_friendly_cuts([1000, 990, 985, 980], 32) == [4]             # 2 % of padding is not worth a second forward
    with pytest.raises(ValueError, match="descending"):
        step_friendly_cuts([100, 200], 4)
    # any cost function can be plugged in (a staircase in B: round 4's chip, as a check of the dynamic program)
    stair = lambda rows: 1000.0 * -(-rows // 8000)  # noqa: E731
    assert step_friendly_cuts([1000] * 9, 16, cost=stair) == [9] and step_friendly_cuts([1000] * 9, 8, cost=stair) in ([8, 1], [1, 8])
    b = bucket_by_length(lens, 16, 1.0, step_friendly=True)
    assert sorted(i for g in b for i in g) == list(range(len(lens))) and all(1 <= len(g) <= 16 for g in b)
    for g in b:  # groups stay sorted by descending length (a batch is padded to its first member)
        assert all(lens[g[i]] >= lens[g[i + 1]] for i in range(len(g) - 1))


def test_inference_state_dict_filters_training_state():
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.checkpoint import inference_state_dict

    sd = {k: torch.from_numpy(np.array(v)) for k, v in wl.synth_state_dict(wl.model_config("tiny")).items()}
    full = dict(sd)
    full["mel_encoder.prenet.w_1.weight"] = torch.zeros(256, 80)
    full["mel_encoder.layer_stack.0.crs_attn.w_qs.weight"] = torch.zeros(256, 256)
    ckpt = {"model": {("module." + k): v for k, v in full.items()}, "optimizer": {"state": {}, "param_groups": []}}
    got = inference_state_dict(ckpt)
    assert set(got) == {k for k in sd if not k.endswith("num_batches_tracked")}
    assert all(torch.equal(got[k], sd[k]) for k in got)


@pytest.mark.gpu
def test_checkpoint_file_round_trip_and_synthesize(tmp_path):
    """A {step}.pth.tar written the way train.py:149-159 writes it -> get_model -> synthesize() gives the same
    per-utterance mels as loading the state dict directly; phoneme_level predictions are expanded to frame rate."""
    import types

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd import batching
    from smart_nar_fast_tts_amd.checkpoint import get_model
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    pc = wl.preprocess_config("phoneme_level", "frame_level")
    sd = {k: torch.from_numpy(np.array(v)) for k, v in wl.synth_state_dict(cfg, frames_per_phoneme=4.0).items()}
    ckpt = {"model": dict(sd, **{"mel_encoder.prenet.w_1.weight": torch.zeros(256, 80)}), "optimizer": {"state": {}}}
    os.makedirs(tmp_path / "ckpt")
    torch.save(ckpt, tmp_path / "ckpt" / "1000.pth.tar")
    args = types.SimpleNamespace(restore_step=1000)
    model = get_model(args, (pc, cfg, {"path": {"ckpt_path": str(tmp_path / "ckpt")}}), torch.device("cuda"))
    direct = FastSpeech2Align(pc, cfg).to("cuda").eval()
    direct.load_state_dict(sd)

    meta, z = load_golden("kat_batching")
    data = _data(z, meta)
    batchs = [batching.collate([data[i] for i in g]) for g in batching.bucket_by_length(z["phone_lens"].tolist(), 3, 0.5)]
    res = batching.synthesize(model, batchs, pc, "cuda")
    assert sorted(r["basename"] for r in res) == sorted(meta["ids"])
    for r in res:
        i = meta["ids"].index(r["basename"])
        assert r["src_len"] == int(z["phone_lens"][i]) and tuple(r["mel"].shape) == (r["mel_len"], 80)
        assert r["duration"].shape == (r["src_len"],) and int(np.maximum(r["duration"], 0).sum()) == r["mel_len"]
        assert r["pitch"].shape == (r["mel_len"],) and r["energy"].shape == (r["mel_len"],)
    # same batch through the directly loaded model: identical bits (same kernels, same weights)
    b0 = batching.to_device(batchs[0], "cuda")
    with torch.no_grad():
        o1, o2 = model(*b0[2:]), direct(*b0[2:])
    assert torch.equal(o1[1], o2[1]) and torch.equal(o1[9], o2[9])


@pytest.mark.gpu
def test_stream_pipelined_synthesize_is_identical():
    """synthesize(streams=3) runs consecutive batches concurrently on different HIP streams, each with its own
    scratch: every per-utterance result must be bit-identical to the one-stream run (many small batches, several
    rounds, so forwards really overlap and any shared temporary would show)."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd import batching
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    pc = wl.preprocess_config()
    model = FastSpeech2Align(pc, cfg).to("cuda").eval()
    model.load_state_dict(wl.synth_state_dict(cfg, frames_per_phoneme=6.0))
    rng = np.random.RandomState(11)
    data = [(f"utt{i}", 0, rng.randint(1, 300, size=int(rng.randint(5, 90))).astype(np.int64), "") for i in range(40)]
    batchs = [batching.collate(data[i:i + 2]) for i in range(0, 40, 2)]
    one = batching.synthesize(model, batchs, pc, "cuda")
    for _ in range(3):
        many = batching.synthesize(model, batchs, pc, "cuda", streams=3)
        assert [r["basename"] for r in many] == [r["basename"] for r in one]
        for a, b in zip(one, many):
            assert a["mel_len"] == b["mel_len"] and torch.equal(a["mel"], b["mel"]), a["basename"]
            assert np.array_equal(a["pitch"], b["pitch"]) and np.array_equal(a["duration"], b["duration"])
    # src_lens kept on the host (to_device(host_lens=True)): the forward uploads them itself (ns_upload_lengths: kernel arguments,
    # stream-ordered, so pipelined streams cannot see each other's lengths); batches this small never pack phase 1, so the
    # results are the one-stream run's, bit for bit
    for st in (1, 3):
        host = batching.synthesize(model, batchs, pc, "cuda", streams=st, host_lens=True)
        for a, b in zip(one, host):
            assert a["mel_len"] == b["mel_len"] and a["src_len"] == b["src_len"] and torch.equal(a["mel"], b["mel"]), (st, a["basename"])
            assert np.array_equal(a["duration"], b["duration"])
    # capacity mode on the streams (max_mel_len): no host wait inside a forward; every batch runs as the reference would with
    # max_len = max_mel_len (model/modules.py:128-131).  That is bit-identical to the synchronous global-pad forward on the grid;
    # against the un-padded run only utterances that already had >= 10 frames of padding are comparable — a batch's longest utterance GAINS
    # padding, which changes its output in the reference too.
    cap = max(r["mel_len"] for r in one) + 7
    capd = batching.synthesize(model, batchs, pc, "cuda", streams=4, max_mel_len=cap)
    model.packed_rows = False
    try:
        padded = []
        for batch in batchs:
            bd = batching.to_device(batch, "cuda")
            with torch.no_grad():
                padded.extend(batching.split_outputs(bd, model(*(bd[2:]), max_mel_len=lambda t: cap), pc))
    finally:
        model.packed_rows = True
    # (and one within the PostNet's reach of its batch's old padded end — 5 layers x 2 frames — saw the grid's zero edge there)
    t_pad = [max(one[j - j % 2]["mel_len"], one[j - j % 2 + 1]["mel_len"]) for j in range(40)]
    n_cmp = 0
    for j, (a, g, b) in enumerate(zip(one, padded, capd)):
        assert a["mel_len"] == b["mel_len"] and np.array_equal(a["duration"], b["duration"]), a["basename"]
        assert torch.equal(g["mel"], b["mel"]) and np.array_equal(g["pitch"], b["pitch"]), a["basename"]
        if a["mel_len"] + 10 <= t_pad[j]:
            n_cmp += 1
            assert float((a["mel"] - b["mel"]).abs().max()) <= 2e-5, a["basename"]
    assert n_cmp >= 10
    with pytest.raises(ValueError, match="cut off"):
        batching.synthesize(model, batchs, pc, "cuda", streams=2, max_mel_len=max(r["mel_len"] for r in one) - 1)
    with pytest.raises(ValueError, match="streams"):
        batching.synthesize(model, batchs, pc, "cuda", max_mel_len=cap)


@pytest.mark.gpu
def test_get_model_flow_with_stats_json(tmp_path):
    """utils/model.py:11-35's order — construct, .to(device), THEN load the checkpoint — with a
    stats.json in <preprocessed_path>, as every real deployment has (model/modules.py:41-46)."""
    import types

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.checkpoint import get_model
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    pc = wl.preprocess_config()
    os.makedirs(tmp_path / "pre")
    with open(tmp_path / "pre" / "stats.json", "w") as f:
        json.dump(wl.SYNTH_STATS, f)
    pc["path"]["preprocessed_path"] = str(tmp_path / "pre")
    sd = {k: torch.from_numpy(np.array(v)) for k, v in wl.synth_state_dict(cfg, frames_per_phoneme=4.0).items()}
    os.makedirs(tmp_path / "ckpt")
    torch.save({"model": sd, "optimizer": {}}, tmp_path / "ckpt" / "7.pth.tar")
    model = get_model(types.SimpleNamespace(restore_step=7), (pc, cfg, {"path": {"ckpt_path": str(tmp_path / "ckpt")}}),
                      torch.device("cuda"))
    direct = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    direct.load_state_dict(sd)
    sp, tx, ln, L = wl.synth_inputs(2, 12, seed=3, src_lens=[12, 9])
    a = [torch.from_numpy(x).cuda() for x in (sp, tx, ln)]
    with torch.no_grad():
        o1, o2 = model(*a, L), direct(*a, L)
    assert torch.equal(o1[1], o2[1]) and torch.equal(o1[9], o2[9]) and o1[1].shape[1] > 0

    # random-init construction (model/fastspeech2_align.py:16-28): with stats.json and NO checkpoint the module runs
    torch.manual_seed(0)
    rnd = FastSpeech2Align(pc, cfg).to("cuda").eval()
    with torch.no_grad():
        o3 = rnd(*a, L)
    assert o3[9].shape == (2,) and torch.isfinite(o3[0]).all() and torch.isfinite(o3[4]).all()
    # without stats.json (the reference constructor raises FileNotFoundError there) the forward refuses
    with pytest.raises(RuntimeError, match="weights not loaded"):
        FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda")(*a, L)


@pytest.mark.gpu
def test_token_id_out_of_range_raises_like_embedding():
    """nn.Embedding raises IndexError for an id outside [0, n_vocab) (transformer/Models.py:89); the
    kernels must not read out of bounds and the wrapper raises after its one host read."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    m.load_state_dict(wl.synth_state_dict(cfg, frames_per_phoneme=4.0))
    sp, tx, ln, L = wl.synth_inputs(3, 10, seed=1)
    for bad in (wl.N_SYMBOLS + 1, 10 ** 12, -1):
        t2 = tx.copy()
        t2[1, 4] = bad
        with pytest.raises(IndexError, match=r"utterance\(s\) \[1\]"):
            m(torch.from_numpy(sp).cuda(), torch.from_numpy(t2).cuda(), torch.from_numpy(ln).cuda(), L)
    t2 = tx.copy()
    t2[0, 0] = wl.N_SYMBOLS  # the largest valid id
    out = m(torch.from_numpy(sp).cuda(), torch.from_numpy(t2).cuda(), torch.from_numpy(ln).cuda(), L)
    assert int(out[9].min()) >= 0


@pytest.mark.gpu
def test_adopted_arena_survives_a_device_rebind():
    """a model whose weights arrived as packed bytes (adopt_arena, no host copy) must not keep a
    stale arena pointer when its device handle changes; re-binding moves the bytes and the results stay identical."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    src = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    src.load_state_dict(wl.synth_state_dict(cfg, frames_per_phoneme=4.0))
    dst = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    dst.arena_tensor().copy_(src.arena_tensor())
    dst.adopt_arena()
    sp, tx, ln, L = wl.synth_inputs(2, 9, seed=5)
    a = [torch.from_numpy(x).cuda() for x in (sp, tx, ln)]
    ref = src(*a, L)
    assert torch.equal(dst(*a, L)[1], ref[1])
    old = dst._arena
    dst._device = None  # what a move from another device looks like to to(): the handle's device differs
    dst.to("cuda")
    assert dst._arena is not old and dst._arena.data_ptr() != old.data_ptr()
    del old
    torch.cuda.empty_cache()
    assert torch.equal(dst(*a, L)[1], ref[1])


@pytest.mark.gpu
def test_adopt_arena_refuses_bytes_that_are_not_this_models_arena():
    """the arena travels between processes as bytes; bytes that were never finalized, that belong to another
    configuration, or that another layout version packed must be refused by ns_adopt_arena instead of silently misread
    (round 3 grew the layout by the PostNet constants: an old same-size arena would have left them garbage)."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    src = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    src.load_state_dict(wl.synth_state_dict(cfg, frames_per_phoneme=4.0))
    # never finalized: zeros
    dst = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    dst.arena_tensor().zero_()
    with pytest.raises(RuntimeError, match="magic"):
        dst.adopt_arena()
    # right bytes, one header word off: the layout version
    dst.arena_tensor().copy_(src.arena_tensor())
    hdr = dst.arena_tensor()[:64].view(torch.int32)
    ver = int(hdr[1])
    hdr[1] = ver - 1
    with pytest.raises(RuntimeError, match="layout version"):
        dst.adopt_arena()
    hdr[1] = ver
    dst.adopt_arena()  # intact again: accepted
    # another configuration whose arena happens to be handed over: refused by size or by the config hash
    cfg2 = wl.model_config("tiny")
    cfg2["variance_embedding"] = dict(cfg2["variance_embedding"], n_bins=cfg2["variance_embedding"]["n_bins"] * 2)
    other = FastSpeech2Align(wl.preprocess_config(), cfg2).to("cuda").eval()
    n = min(other.arena_tensor().numel(), src.arena_tensor().numel())
    other.arena_tensor()[:n].copy_(src.arena_tensor()[:n])
    with pytest.raises(RuntimeError, match="arena size|different ns_config"):
        other.adopt_arena()
