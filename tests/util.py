"""Shared helpers for the parity tests (fixture loading, weight rebuild)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, z


_SD_CACHE = {}


def weights_for(meta):
    """Rebuild the seeded weights a fixture was generated with (they are not stored)."""
    import smart_nar_fast_tts_amd.workload as wl

    key = (meta["config"], meta.get("weight_seed", 0), meta["frames_per_phoneme"], meta.get("dur_weight_scale", 0.25))
    if key not in _SD_CACHE:
        cfg = wl.model_config(meta["config"])
        _SD_CACHE.clear()  # one at a time: a full state dict is 116 MB
        _SD_CACHE[key] = (cfg, wl.synth_state_dict(cfg, seed=key[1], frames_per_phoneme=key[2], dur_weight_scale=key[3]))
    return _SD_CACHE[key]
