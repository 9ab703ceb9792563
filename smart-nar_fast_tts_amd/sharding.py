"""Multi-GPU host logic: utterances shard across ranks, one process per GPU.

* weights: rank 0 packs the arena once, the bytes are replicated with ONE ``torch.distributed.broadcast``
  (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests);
* per-shard mode (default): no collective on the data path;
* global-pad mode (optional): one all-reduce MAX of a single int32 per forward so that every shard pads its
  mel axis to the full batch's T_pad, as the reference run on the whole batch would.

The entry point a caller uses is :func:`synthesize_sharded` (synthesize.py:59-76 for N ranks): every rank passes the SAME
host batch, takes its share, runs the forward and gets its utterances back (optionally gathered on rank 0 in the batch's
order).  It is built so that no rank can leave another one waiting in a collective: a rank with ZERO utterances (a tail batch
with fewer utterances than ranks) and a rank whose forward RAISED both still join every exchange (:func:`forward_shard`).
"""
from __future__ import annotations

import sys

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous, balanced split of ``n_items`` utterances: the first ``n_items % world_size`` ranks get one more."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_indices(src_lens, world_size: int, balance: str = "count"):
    """Which utterances each rank takes: a list of ``world_size`` ascending index arrays that partition ``range(len(src_lens))``.

    * ``"count"``: the contiguous split of :func:`shard_bounds` (SURVEY.md §8e "contiguous split of the B utterances").
    * ``"phonemes"``: longest-processing-time-first on the phoneme counts — utterances in order of descending ``src_lens``, each
      to the rank with the fewest phonemes so far that still has room (every rank keeps the count of the contiguous split, so
      batch shapes stay what they were).  Frames are not known before the duration predictor has run (SURVEY.md §8e "scaling
      risks: variable T_pad per shard ... cannot be known before the predictor runs"); phonemes are the host-side proxy, and
      with phase 2 on packed rows a shard's work IS its sum of frames, not B x T_pad.  Deterministic: ties go to the lower
      rank, equal lengths keep their order."""
    import numpy as np

    lens = np.asarray(src_lens).reshape(-1)
    n = int(lens.shape[0])
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if balance == "count":
        return [np.arange(*shard_bounds(n, world_size, r)) for r in range(world_size)]
    if balance != "phonemes":
        raise ValueError("balance must be 'count' or 'phonemes'")
    room = [shard_bounds(n, world_size, r)[1] - shard_bounds(n, world_size, r)[0] for r in range(world_size)]
    load = [0] * world_size
    take = [[] for _ in range(world_size)]
    for i in np.argsort(-lens, kind="stable"):
        r = min((r for r in range(world_size) if len(take[r]) < room[r]), key=lambda r: (load[r], r))
        take[r].append(int(i))
        load[r] += int(lens[i])
    return [np.array(sorted(t), dtype=np.int64) for t in take]


def gather_order(parts):
    """Inverse of a :func:`shard_indices` partition: ``inv`` such that ``concat(rank outputs in rank order)[inv]`` is the batch in
    its original order (row ``inv[i]`` of the concatenation is utterance ``i``)."""
    import numpy as np

    flat = np.concatenate([np.asarray(p, dtype=np.int64) for p in parts]) if parts else np.zeros(0, dtype=np.int64)
    inv = np.empty_like(flat)
    inv[flat] = np.arange(flat.shape[0], dtype=np.int64)
    return inv


def shard_batch(speakers, texts, src_lens, world_size: int, rank: int, balance: str = "count", return_index: bool = False):
    """This rank's share of a host-side batch; max_src_len is kept GLOBAL (it is an input, not data dependent,
    so phoneme-side padding needs no collective).  ``balance``: see :func:`shard_indices`; ``return_index`` appends the global
    indices of this rank's utterances (with :func:`gather_order` the gathered outputs go back into the batch's order)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    if balance == "count":  # a contiguous slice: views, no copy
        lo, hi = shard_bounds(len(src_lens), world_size, rank)
        out = (speakers[lo:hi], texts[lo:hi], src_lens[lo:hi], int(texts.shape[1]))
        if return_index:
            import numpy as np
            out += (np.arange(lo, hi),)
        return out
    idx = shard_indices(src_lens, world_size, balance)[rank]
    out = (speakers[idx], texts[idx], src_lens[idx], int(texts.shape[1]))
    return out + (idx,) if return_index else out


def broadcast_bytes(buf: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Replicate a flat uint8 buffer from ``src`` to every rank in place (one collective, no ring of small ones)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(buf, src=src, group=group)
    return buf


def broadcast_weights(model, state_dict=None, src: int = 0, group=None):
    """Rank ``src`` calls with the state dict (packs + uploads); every other rank passes None and adopts the bytes."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if rank == src:
        if state_dict is None:
            raise ValueError("the source rank needs the state dict")
        model.load_state_dict(state_dict)
    arena = model.arena_tensor()
    broadcast_bytes(arena, src=src, group=group)
    if rank != src:
        model.adopt_arena()
    return model


def global_max(local_max: torch.Tensor, group=None) -> int:
    """all-reduce MAX of one integer (global-pad mode); returns a Python int (the caller needs it to shape outputs)."""
    t = local_max.reshape(1).to(torch.int32).clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


# ---------------------------------------------------------------------------------------------------------------------
# The sharded entry point (synthesize.py:59-76 for N ranks)
# ---------------------------------------------------------------------------------------------------------------------

def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _collective_device(group=None, device=None):
    """Where a tensor must live to go through this group's collectives: RCCL ("nccl") moves device memory only; gloo (the CPU
    tests and the one-GPU rigs) takes host tensors in every build."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
        d = torch.device(device) if device is not None else None
        return d if (d is not None and d.type == "cuda" and d.index is not None) else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class PeerFailure(RuntimeError):
    """Another rank's forward raised; this rank's own shard was fine.  Raised on EVERY healthy rank of a global-pad exchange so
    that the job fails together instead of hanging in the next collective."""


def exchange_pad(local_max: int, failed: bool, device=None, group=None):
    """Global-pad mode's per-forward exchange: ONE all-reduce MAX over ``[longest mel of this shard, failure flag]``.  A rank with
    no utterances contributes 0; a rank whose phase 1 raised contributes (0, 1).  Returns ``(global_max, any_failed)``."""
    world, _ = _world(group)
    if world == 1:
        return int(local_max), bool(failed)
    t = torch.tensor([int(local_max), 1 if failed else 0], dtype=torch.int32, device=_collective_device(group, device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    g = t.tolist()
    return int(g[0]), bool(g[1])


class Shard:
    """One rank's share of a host batch, prepared once (inputs resident on the device): what :func:`forward_shard` runs.

    ``index``: global positions of this rank's utterances (ascending); ``parts``: every rank's positions (the same list on every
    rank — the partition is a pure function of the host lengths); ``batch``: the to_device'd 6-tuple (ids, raw_texts, speakers,
    texts, src_lens, max_src_len) or None for a rank that got nothing; ``n_global``: utterances in the whole batch."""

    def __init__(self, index, parts, batch, n_global, world, rank, max_src_len, all_ids):
        self.index, self.parts, self.batch, self.n_global, self.world, self.rank = index, parts, batch, n_global, world, rank
        self.max_src_len, self.all_ids = int(max_src_len), list(all_ids)

    @property
    def empty(self) -> bool:
        return self.batch is None

    def __len__(self):
        return int(len(self.index))


def prepare_shard(batch, device="cuda", balance: str = "count", host_lens: bool = True, group=None, world_size=None, rank=None) -> Shard:
    """Split the host 6-tuple ``(ids, raw_texts, speakers, texts, text_lens, max_text_len)`` (dataset.py:182-191, what
    ``batching.collate`` returns; identical on every rank) and move this rank's rows to ``device``.  ``max_src_len`` stays the GLOBAL
    one (it is an input: phoneme-side padding needs no collective).  ``host_lens`` keeps the shard's ``src_lens`` on the host so that
    phase 1 may run on packed phoneme rows (``batching.to_device``)."""
    import numpy as np

    from .batching import to_device

    w, r = _world(group)
    world = w if world_size is None else int(world_size)
    rank = r if rank is None else int(rank)
    ids, raw_texts, speakers, texts, text_lens, max_len = batch
    text_lens = np.asarray(text_lens).reshape(-1)
    n = int(text_lens.shape[0])
    parts = shard_indices(text_lens, world, balance)
    idx = parts[rank]
    texts = np.asarray(texts)
    max_len = int(texts.shape[1]) if texts.ndim == 2 else int(max_len)  # (the padded width forward() sees, = max(text_lens) after collate)
    if len(idx) == 0:
        return Shard(idx, parts, None, n, world, rank, max_len, ids)
    mine = ([ids[i] for i in idx], [raw_texts[i] for i in idx], np.asarray(speakers)[idx], np.ascontiguousarray(texts[idx]),
            np.ascontiguousarray(text_lens[idx]), int(texts.shape[1]))
    return Shard(idx, parts, to_device(mine, device, host_lens), n, world, rank, max_len, ids)


def forward_shard(model, shard: Shard, global_pad: bool = False, group=None, **forward_kw):
    """This rank's forward, joining every collective of the step whatever happens locally.

    * per-shard mode (``global_pad=False``): no collective on the data path; an empty shard returns None at once.
    * global-pad mode: ONE all-reduce per forward (:func:`exchange_pad`).  A rank with zero utterances contributes 0 and returns
      None; a rank whose forward raises BEFORE its exchange (bad token id, empty utterance list ...) still joins it with the
      failure flag set, then re-raises; every other rank raises :class:`PeerFailure` after the exchange.  Nobody is left waiting.

    Returns the forward's 12-tuple (``ForwardOutput``) or None for an empty shard."""
    if not global_pad or shard.world == 1:
        if shard.empty:
            return None
        return model(*(shard.batch[2:]), **forward_kw)
    dev = None if shard.empty else shard.batch[3].device
    if shard.empty:
        _, any_failed = exchange_pad(0, False, dev, group)
        if any_failed:
            raise PeerFailure("another rank's forward failed (this rank had no utterances)")
        return None
    joined = []

    def pad_to(local_max):
        g, any_failed = exchange_pad(int(local_max), False, dev, group)
        joined.append(g)
        if any_failed:
            raise PeerFailure("another rank's forward failed before the global-pad exchange")
        return g

    try:
        return model(*(shard.batch[2:]), max_mel_len=pad_to, **forward_kw)
    except Exception as e:
        if not joined:  # raised before the exchange: join it (flag set) so the other ranks do not wait for this one
            exchange_pad(0, True, dev, group)
            e._ns_peers_know = True
        raise


def _frame_block(out, p_frame: bool, e_frame: bool):
    """[valid frames of this shard, n_mel (+1 pitch) (+1 energy)] fp32, utterance after utterance: the PostNet mel with the
    frame-level predictions as extra columns (what utils/tools.py:158-171 slices per utterance)."""
    keep = ~out[7]
    cols = [out[1][keep]]
    if p_frame:
        cols.append(out[2][keep].unsqueeze(1))
    if e_frame:
        cols.append(out[3][keep].unsqueeze(1))
    return torch.cat(cols, dim=1) if len(cols) > 1 else cols[0]


def synthesize_sharded(model, batch, preprocess_config, device="cuda", balance: str = "count", global_pad: bool = False,
                       gather: bool = False, host_lens: bool = True, p_control: float = 1.0, e_control: float = 1.0, group=None):
    """synthesize.py:59-76 across the ranks of ``group``: every rank calls this with the SAME host batch (the 6-tuple of
    ``batching.collate``); each takes its share (``balance``: :func:`shard_indices`), runs the forward and returns

        ``(results, index)``

    ``results``: per-utterance dicts as ``batching.split_outputs`` builds them (``mel`` [mel_len, n_mel] on the device, ``duration``,
    ``pitch``, ``energy``, ``src_len``, ``mel_len``, ``basename``) plus ``"index"`` = position in the global batch; ``index``: the same
    positions as an array.  With ``gather=True`` rank 0 instead returns ALL utterances in the batch's original order (its own
    slices plus the other ranks' rows, received root <- peer with one point-to-point transfer per rank: frames are ragged, so there
    is no padded all-gather) and ``index = arange(n)``; the other ranks still return their own.

    ``global_pad``: pad every shard's mel axis to the whole batch's longest utterance (one 2-int all-reduce MAX per forward), so
    that the concatenated ranks equal the reference run on the whole batch (SURVEY.md §8e secondary parity); the default
    per-shard mode has no data-path collective.  A rank that gets ZERO utterances (``n < world_size``) joins every collective
    and returns ``([], [])``."""
    import numpy as np

    from .batching import split_outputs

    shard = prepare_shard(batch, device, balance, host_lens, group)
    try:
        with torch.no_grad():
            out = forward_shard(model, shard, global_pad, group, p_control=p_control, e_control=e_control)
    except PeerFailure:
        raise  # the pad exchange told EVERY rank: nobody goes on to the gather
    except Exception:
        if gather and shard.world > 1:
            # this rank alone knows: the others are on their way to the gather's all-gather.  Join it, marked, then re-raise.
            # (A failure that the global-pad exchange already announced is different: the others raised PeerFailure and are NOT
            # on their way — this rank must not enter a collective nobody else joins.)
            if not getattr(sys.exc_info()[1], "_ns_peers_know", False):
                _gather_to_root(shard, None, [], preprocess_config, device, group, failed=True)
        raise
    results = []
    if out is not None:
        results = split_outputs(shard.batch, out, preprocess_config)
        for item, gi in zip(results, shard.index):
            item["index"] = int(gi)
    if not gather or shard.world == 1:
        return results, np.asarray(shard.index, dtype=np.int64)
    return _gather_to_root(shard, out, results, preprocess_config, device, group)


def _gather_to_root(shard: Shard, out, results, preprocess_config, device, group=None, failed: bool = False):
    """Rank 0 <- every peer: a fixed-size all-gather of the per-utterance integers and phoneme-rate rows (small), then ONE
    point-to-point transfer per peer of its frame block (ragged).  Ranks with nothing to send take part in the all-gather only.
    Row 0 of the all-gathered block is a header: a rank whose forward raised joins with ``failed`` set, and every other rank
    raises :class:`PeerFailure` before any point-to-point transfer is posted."""
    import numpy as np

    from .batching import expand

    pp = preprocess_config["preprocessing"]
    p_frame, e_frame = pp["pitch"]["feature"] == "frame_level", pp["energy"]["feature"] == "frame_level"
    world, rank, L = shard.world, shard.rank, shard.max_src_len
    room = max(len(p) for p in shard.parts)
    cdev = _collective_device(group, device if out is None else out[1].device)
    # phoneme-rate block, fixed shape on every rank: [room, 2 + 3 L] = mel_len, src_len, d_rounded[L], pitch[L], energy[L]
    head = torch.zeros((room + 1, 2 + 3 * L), dtype=torch.float64, device=cdev)
    head[0, 0] = 1.0 if failed else 0.0
    meta = head[1:]
    if out is not None:
        n = len(shard)
        meta[:n, 0] = out[9].to(cdev, torch.float64)
        meta[:n, 1] = torch.as_tensor(np.asarray(out[8].cpu() if torch.is_tensor(out[8]) else out[8]), dtype=torch.float64).to(cdev)
        meta[:n, 2:2 + L] = out[5].to(cdev, torch.float64)
        if not p_frame:
            meta[:n, 2 + L:2 + 2 * L] = out[2].to(cdev, torch.float64)
        if not e_frame:
            meta[:n, 2 + 2 * L:2 + 3 * L] = out[3].to(cdev, torch.float64)
    heads = [torch.empty_like(head) for _ in range(world)]
    dist.all_gather(heads, head, group=group)
    if failed:
        return None
    bad = [r for r in range(world) if float(heads[r][0, 0]) != 0.0]
    if bad:
        raise PeerFailure(f"the forward of rank(s) {bad} failed; nothing was gathered")
    metas = [h[1:] for h in heads]
    frames_of = [int(m[:len(p), 0].sum().item()) for m, p in zip(metas, shard.parts)]
    width = int(preprocess_config["preprocessing"]["mel"]["n_mel_channels"]) + int(p_frame) + int(e_frame)
    mine = None
    if out is not None and frames_of[rank] > 0:
        mine = _frame_block(out, p_frame, e_frame).to(cdev).contiguous()
    if rank != 0:
        if mine is not None:
            dist.send(mine, dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return results, np.asarray(shard.index, dtype=np.int64)
    blocks = [mine if mine is not None else torch.empty((0, width), dtype=torch.float32, device=cdev)]
    reqs = []
    for r in range(1, world):
        blocks.append(torch.empty((frames_of[r], width), dtype=torch.float32, device=cdev))
        if frames_of[r] > 0:
            src = dist.get_global_rank(group, r) if group is not None else r
            reqs.append(dist.irecv(blocks[r], src=src, group=group))
    for q in reqs:
        q.wait()
    ids, n_mel = shard.all_ids, width - int(p_frame) - int(e_frame)
    allr = [None] * shard.n_global
    for r in range(world):
        m, blk, off = metas[r].cpu().numpy(), blocks[r], 0
        for j, gi in enumerate(shard.parts[r]):
            mel_len, src_len = int(m[j, 0]), int(m[j, 1])
            fr = blk[off:off + mel_len]
            off += mel_len
            duration = m[j, 2:2 + src_len].astype(np.float32)
            item = {"basename": ids[int(gi)], "mel": fr[:, :n_mel], "duration": duration, "src_len": src_len, "mel_len": mel_len, "index": int(gi)}
            item["pitch"] = fr[:, n_mel].cpu().numpy() if p_frame else expand(m[j, 2 + L:2 + L + src_len].astype(np.float32), duration)
            item["energy"] = (fr[:, n_mel + int(p_frame)].cpu().numpy() if e_frame
                              else expand(m[j, 2 + 2 * L:2 + 2 * L + src_len].astype(np.float32), duration))
            allr[int(gi)] = item
    return allr, np.arange(shard.n_global, dtype=np.int64)


def spin_budget_us(local_world_size: int, cores: int | None = None, default: float = 300.0) -> float:
    """How long a forward may busy-wait for the mid-forward hand-over (``FastSpeech2Align.SPIN_US``) when ``local_world_size`` ranks
    share one host.  The spin saves a ~50 us wake-up per forward and costs up to ``default`` us of a core per forward: harmless while
    every rank has a core to spin on and one for everything else (cores >= 2 x ranks), harmful once ranks outnumber that — a
    spinning rank then takes the time slice another rank's launch thread needs.  ``NS_SPIN_US`` in the environment overrides."""
    import os

    if "NS_SPIN_US" in os.environ:
        return float(os.environ["NS_SPIN_US"])
    cores = cores if cores is not None else len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if local_world_size <= 1:
        return float(default)
    return float(default) if cores >= 2 * local_world_size else 0.0


def configure_spin(model, local_world_size: int | None = None, cores: int | None = None) -> float:
    """Set ``model.SPIN_US`` from the number of ranks on this host (``LOCAL_WORLD_SIZE`` as torchrun exports it) and return it."""
    import os

    if local_world_size is None:
        local_world_size = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    model.SPIN_US = spin_budget_us(local_world_size, cores)
    return model.SPIN_US
