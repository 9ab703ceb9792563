#!/usr/bin/env python3
"""Phase 1 on packed phoneme rows against the padded [B, L] grid (include/nar_fs2.h ns_config.phase1_packing), ragged batches of
the config-2 shape with src_lens on the host:   python tools/phase1_packing_ab.py [--batches 9,12,16,24,32,48,64]
Three models on one box (auto / always / never), alternating, 3 warm-up + 12 timed forwards each; prints ms per forward and the
rows phase 1 ran on."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import smart_nar_fast_tts_amd.workload as wl  # noqa: E402
from smart_nar_fast_tts_amd.model import FastSpeech2Align  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="9,12,16,24,32,48,64")
    args = ap.parse_args()
    cfg_name, _, L, fpp = wl.WORKLOADS["cfg2_b16"]
    dev = torch.device("cuda", 0)
    sd = wl.synth_state_dict(wl.model_config(cfg_name), seed=0, frames_per_phoneme=fpp)
    models = {}
    for mode in ("auto", "always", "never"):
        m = FastSpeech2Align(wl.preprocess_config(), dict(wl.model_config(cfg_name), phase1_packing=mode)).to(dev).eval()
        m.load_state_dict(sd)
        models[mode] = m
    for B in (int(b) for b in args.batches.split(",")):
        rr = np.random.RandomState(7)
        lens = rr.randint(max(1, L // 8), L + 1, size=B)
        lens[0] = L
        sp, tx, ln, Lmax = wl.synth_inputs(B, L, seed=0, src_lens=lens)
        a = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (sp, tx)]
        lens_host = torch.from_numpy(np.ascontiguousarray(ln))
        out = []
        best = {k: 1e9 for k in models}
        rows = {}
        with torch.no_grad():
            for rep in range(2):
                for mode, m in models.items():
                    for _ in range(3):
                        m(a[0], a[1], lens_host, Lmax)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(12):
                        m(a[0], a[1], lens_host, Lmax)
                    torch.cuda.synchronize()
                    best[mode] = min(best[mode], (time.perf_counter() - t0) / 12 * 1e3)
                    rows[mode] = int(m._lib.ns_last_phase1_rows(m._h))
        print(f"B={B} grid rows {B * L}: " + "  ".join(f"{k}: {best[k]:.3f} ms ({rows[k]} rows)" for k in models), flush=True)


if __name__ == "__main__":
    main()
