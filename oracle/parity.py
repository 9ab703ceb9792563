"""TEST INFRASTRUCTURE (checker only — never imported by the product package).

Classification of the two discontinuous stages of the path, shared by the parity tests and by the checker that sits
beside bench.py's measurement:

* ``round(exp(log_d) - 1)``                 model/modules.py:132-135   -> duration flips
* ``torch.bucketize(pitch | energy, bins)``  model/modules.py:86-88,97-99 -> bucket flips

Any fp32 evaluation whose summation order differs from torch's CPU kernels may take the other branch when the value
sits within rounding noise of the boundary.  A flip is *legal* only there: the reference's own value must lie within
``EDGE_REL`` (relative) of a bin edge.  A flip anywhere else is a real error.
"""
from __future__ import annotations

import numpy as np

# Bound on |implementation - reference| / max(|reference|, 1) for pitch / energy values inside the bin range — and therefore
# on how far from a bin edge (in the same normalisation) a reference value can sit while the two sides still take different
# buckets.  It is set from a RECORDED distribution, not from one observed maximum: tools/bucket_edge_deviation.py measures the
# deviation of the HIP path on every in-range frame of the five BASELINE pins (profiles/r03_bucket_edge_deviation.md: 48 000
# frames each for pitch and energy; median 7e-7 / 1.1e-6, p99.9 1.4e-5 / 1.9e-5, worst 2.28e-5 / 2.12e-5) and the bound is
# 2x the worst value, rounded — round 3.  Round 4 re-measured the distribution on the final build (profiles/r04_bucket_edge_deviation.md:
# worst 2.31e-5 / 2.00e-5 after the row arithmetic's multiply-adds were spelled out) and PINNED the bound closer to it, at 3e-5:
# the deviation itself is asserted on every in-range frame, so what the bound still has to absorb is the 30 % between the worst
# frame of 48 000 and the next build's, not a factor of two.  Round 2's 2e-5 sat BELOW the implementation's own worst case (its
# flips all happened to lie closer to an edge than that).  The tests assert the deviation itself on every in-range frame (max_rel_deviation below), not
# only the position of the frames that flipped.  Evaluating the predictors' LayerNorm + Linear tail in float64 does not
# move these figures (2.27e-5 / 2.11e-5): the deviation is the fp32 summation order of the contractions upstream.
EDGE_REL = 3e-5


def in_range(ref: np.ndarray, bins: np.ndarray, valid: np.ndarray) -> np.ndarray:
    """Frames where a bucket decision can change at all: valid, and the reference value inside the bin range (1 % margin)."""
    b = np.asarray(bins, dtype=np.float64)
    r = np.asarray(ref, dtype=np.float64)
    lo, hi = b[0] - 0.01 * max(abs(b[0]), 1.0), b[-1] + 0.01 * max(abs(b[-1]), 1.0)
    return valid & (r >= lo) & (r <= hi)


def max_rel_deviation(got: np.ndarray, ref: np.ndarray, bins: np.ndarray, valid: np.ndarray) -> float:
    """max over the in-range frames of |got - ref| / max(|ref|, 1); 0.0 when there is no such frame."""
    sel = in_range(ref, bins, valid)
    if not sel.any():
        return 0.0
    g, r = np.asarray(got, dtype=np.float64)[sel], np.asarray(ref, dtype=np.float64)[sel]
    return float((np.abs(g - r) / np.maximum(np.abs(r), 1.0)).max())


def edge_distance(values: np.ndarray, bins: np.ndarray) -> np.ndarray:
    """Absolute distance of every value to its nearest bin edge."""
    v = np.asarray(values, dtype=np.float64)
    b = np.asarray(bins, dtype=np.float64)
    i = np.clip(np.searchsorted(b, v, side="left"), 1, len(b) - 1)
    return np.minimum(np.abs(v - b[i - 1]), np.abs(v - b[i]))


def edge_rel(values: np.ndarray, bins: np.ndarray) -> np.ndarray:
    """Distance to the nearest bin edge relative to max(|value|, 1) (what the fixtures store as p_edge_rel / e_edge_rel)."""
    return edge_distance(values, bins) / np.maximum(np.abs(np.asarray(values, dtype=np.float64)), 1.0)


def bucketize(values: np.ndarray, bins: np.ndarray) -> np.ndarray:
    """torch.bucketize(values, bins, right=False) for finite values."""
    return np.searchsorted(np.asarray(bins), np.asarray(values), side="left")


def classify_bucket_flips(got: np.ndarray, ref: np.ndarray, bins: np.ndarray, valid: np.ndarray, edge_rel_ref=None):
    """Compare the bucket decisions of ``got`` with those of the reference values ``ref`` on the ``valid`` frames.

    Returns ``(flips, off_edge, by_more_than_one)``: how many decisions differ, how many of those differ although the
    reference value is NOT within EDGE_REL of an edge (must be 0), and how many moved by more than one bucket (must be 0)."""
    gi, ri = bucketize(got, bins), bucketize(ref, bins)
    flip = (gi != ri) & valid
    rel = edge_rel(ref, bins) if edge_rel_ref is None else edge_rel_ref
    off = flip & ~(rel < EDGE_REL)
    far = flip & (np.abs(gi - ri) > 1)
    return int(flip.sum()), int(off.sum()), int(far.sum())
