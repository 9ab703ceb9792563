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


def shard_batch(speakers, texts, src_lens, world_size: int, rank: int):
    """This rank's slice of a host-side batch; max_src_len is kept GLOBAL (it is an input, not data dependent,
    so phoneme-side padding needs no collective)."""
    lo, hi = shard_bounds(len(src_lens), world_size, rank)
    return speakers[lo:hi], texts[lo:hi], src_lens[lo:hi], int(texts.shape[1])


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
