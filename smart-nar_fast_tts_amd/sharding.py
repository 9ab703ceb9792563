"""Multi-GPU host logic: utterances shard across ranks, one process per GPU.

* weights: rank 0 packs the arena once, the bytes are replicated with ONE ``torch.distributed.broadcast``
  (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests);
* per-shard mode (default): no collective on the data path;
* global-pad mode (optional): one all-reduce MAX of a single int32 per forward so that every shard pads its
  mel axis to the full batch's T_pad, as the reference run on the whole batch would.
"""
from __future__ import annotations

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
