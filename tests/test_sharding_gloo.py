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
