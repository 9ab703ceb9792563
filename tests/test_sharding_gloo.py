"""Multi-GPU host logic on CPU: world_size 2, gloo.  Covers the utterance split, the one-shot
weight-bytes broadcast and the global-pad all-reduce MAX that bench.py / the model use over RCCL on the GPU box."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # weights: rank 0 holds the packed bytes, everyone ends up with them after ONE broadcast
        ref = torch.arange(100003, dtype=torch.int64).to(torch.uint8)
        buf = ref.clone() if rank == 0 else torch.zeros_like(ref)
        sharding.broadcast_bytes(buf, src=0)
        ok_bcast = bool(torch.equal(buf, ref))
        # utterances: contiguous, disjoint, balanced, max_src_len stays global
        sp, tx, ln, L = wl.synth_inputs(7, 12, seed=0, src_lens=[12, 5, 9, 1, 12, 3, 8])
        s_sp, s_tx, s_ln, s_L = sharding.shard_batch(sp, tx, ln, world, rank)
        lo, hi = sharding.shard_bounds(7, world, rank)
        # global-pad: MAX over ranks of the local longest mel
        gmax = sharding.global_max(torch.tensor(100 + 10 * rank))
        # balanced split (LPT on the phoneme counts): every rank computes the same partition from the same host lengths; the
        # gathered per-rank rows go back into the batch's order through gather_order
        b_sp, b_tx, b_ln, b_L, b_idx = sharding.shard_batch(sp, tx, ln, world, rank, balance="phonemes", return_index=True)
        gathered = [None] * world
        dist.all_gather_object(gathered, (b_idx.tolist(), b_tx.tolist(), int(b_ln.sum())))
        parts = [g[0] for g in gathered]
        inv = sharding.gather_order(parts)
        rows = [row for g in gathered for row in g[1]]
        restored = [rows[int(j)] for j in inv]
        q.put((rank, ok_bcast, (lo, hi), s_tx.tolist() == tx[lo:hi].tolist(), s_L == L, gmax,
               (parts, restored == tx.tolist(), [g[2] for g in gathered], b_L == L, b_tx.tolist() == tx[b_idx].tolist())))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_partition():
    from smart_nar_fast_tts_amd.sharding import shard_bounds

    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[2] for r in res] == [(0, 4), (4, 7)]
    for rank, ok_bcast, span, same_rows, same_L, gmax, bal in res:
        assert ok_bcast and same_rows and same_L
        assert gmax == 110  # max(100, 110)
        parts, restored, loads, same_L2, own_rows = bal
        assert restored and same_L2 and own_rows
        assert sorted(i for p in parts for i in p) == list(range(7)) and [len(p) for p in parts] == [4, 3]  # a partition, counts kept
        # src_lens [12, 5, 9, 1, 12, 3, 8] = 50 phonemes: contiguous gives 27 / 23, LPT 25 / 25
        assert loads == [25, 25], loads


def test_balanced_split_properties():
    """sharding.shard_indices(balance="phonemes"): a partition with the contiguous split's counts, never worse than it in the
    heaviest shard's phoneme load, deterministic, and the identity for one rank; gather_order inverts it."""
    import numpy as np

    from smart_nar_fast_tts_amd.sharding import gather_order, shard_bounds, shard_indices

    rs = np.random.RandomState(3)
    for n, w in ((0, 2), (1, 2), (7, 2), (16, 8), (128, 8), (129, 8), (50, 3)):
        lens = rs.randint(1, 129, size=n)
        cnt = shard_indices(lens, w, "count")
        bal = shard_indices(lens, w, "phonemes")
        assert [len(p) for p in bal] == [shard_bounds(n, w, r)[1] - shard_bounds(n, w, r)[0] for r in range(w)]
        assert sorted(int(i) for p in bal for i in p) == list(range(n))
        assert all(np.all(np.diff(p) > 0) for p in bal if len(p) > 1)
        heavy = lambda parts: max([int(lens[p].sum()) for p in parts] + [0])  # noqa: E731
        assert heavy(bal) <= heavy(cnt)
        assert all(np.array_equal(a, b) for a, b in zip(bal, shard_indices(lens, w, "phonemes")))
        inv = gather_order(bal)
        cat = np.concatenate(bal) if n else np.zeros(0, dtype=np.int64)
        assert np.array_equal(cat[inv], np.arange(n))
    assert np.array_equal(shard_indices([5, 9, 2], 1, "phonemes")[0], [0, 1, 2])
    lens = np.random.RandomState(7).randint(16, 129, size=128)
    loads = [int(lens[p].sum()) for p in shard_indices(lens, 8, "phonemes")]
    assert max(loads) - min(loads) <= 8, loads  # (contiguous: a spread of hundreds on the same draw)


def _gpu_worker(rank, world, port, q):
    """bench.py's N > 1 flow with both ranks on cuda:0 and gloo instead of RCCL: rank 0 packs the weights, ONE broadcast
    of the arena bytes, the other rank adopts them, every rank runs its own shard."""
    import numpy as np

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd import sharding
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        cfg = wl.model_config("tiny")
        model = FastSpeech2Align(wl.preprocess_config(), cfg).to(dev).eval()
        sd = wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=4.0) if rank == 0 else None
        arena = model.arena_tensor()
        try:
            sharding.broadcast_weights(model, sd, src=0)
        except RuntimeError:  # a gloo build without device-tensor support: stage the same bytes through the host
            if rank == 0:
                model.load_state_dict(sd)
            host = model.arena_tensor().cpu()
            sharding.broadcast_bytes(host, src=0)
            model.arena_tensor().copy_(host)
            if rank != 0:
                model.adopt_arena()
        assert model.arena_tensor().data_ptr() == arena.data_ptr() or rank == 0
        sp, tx, ln, _ = wl.synth_inputs(6, 40, seed=3)
        s_sp, s_tx, s_ln, L = sharding.shard_batch(sp, tx, ln, world, rank)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        with torch.no_grad():
            out = model(to(s_sp), to(s_tx), to(s_ln), L)
            # global-pad mode: every shard pads to the global longest mel
            outg = model(to(s_sp), to(s_tx), to(s_ln), L, max_mel_len=sharding.global_max)
        torch.cuda.synchronize()
        q.put((rank, out[1].cpu().numpy(), out[9].cpu().numpy(), int(outg[1].shape[1])))
    finally:
        dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_rank_weight_broadcast_and_shards_on_gpu():
    """The multi-GPU flow end to end on the one GPU a test box has: the rank that ADOPTED the broadcast arena must produce
    exactly what a single process with the state dict loaded produces on the same shard."""
    import numpy as np

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd import sharding
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = wl.model_config("tiny")
    model = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    model.load_state_dict(wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=4.0))
    sp, tx, ln, _ = wl.synth_inputs(6, 40, seed=3)
    tmax = 0
    for rank, post, mel_lens, t_global in res:
        s_sp, s_tx, s_ln, L = sharding.shard_batch(sp, tx, ln, world, rank)
        with torch.no_grad():
            ref = model(*(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (s_sp, s_tx, s_ln)), L)
        assert np.array_equal(mel_lens, ref[9].cpu().numpy())
        assert np.array_equal(post, ref[1].cpu().numpy()), f"rank {rank}: adopted weights give different bits"
        tmax = max(tmax, int(mel_lens.max()))
    assert all(r[3] == tmax for r in res), "global-pad mode: every shard pads to the global longest mel"
