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
        q.put((rank, ok_bcast, (lo, hi), s_tx.tolist() == tx[lo:hi].tolist(), s_L == L, gmax))
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
    for rank, ok_bcast, span, same_rows, same_L, gmax in res:
        assert ok_bcast and same_rows and same_L
        assert gmax == 110  # max(100, 110)


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
