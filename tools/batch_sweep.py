#!/usr/bin/env python3
"""Forward time against the batch size on one MI355X (the launch planner's acceptance sweep):

    python tools/batch_sweep.py [--batches 1,2,...,32] [--workload cfg2_b16] [--ragged]     (NS_PLAN=0: round-3 tile rules)

One model, uniform BASELINE-style batches of B utterances (L phonemes, T_pad ~ 8 L), 3 warm-up + 12 timed forwards each,
device-synchronised wall time.  Prints per B: rows of phase 2, ms per forward, valid frames/s, ms per utterance, the
end-to-end fraction of the fp32 MFMA peak, and how ms/utterance moved against the best smaller batch (the planner's
target: never more than +3 % from B = 4 upwards)."""
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
    ap.add_argument("--batches", default=",".join(str(b) for b in range(1, 33)))
    ap.add_argument("--workload", default="cfg2_b16")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--iters", type=int, default=12)
    args = ap.parse_args()
    cfg_name, _, L, fpp = wl.WORKLOADS[args.workload]
    cfg = wl.model_config(cfg_name)
    dev = torch.device("cuda", 0)
    model = FastSpeech2Align(wl.preprocess_config(), cfg).to(dev).eval()
    model.load_state_dict(wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=fpp))
    best = None
    print(f"# {args.workload}{' ragged' if args.ragged else ''}, NS_PLAN={os.environ.get('NS_PLAN', '1')}, {torch.cuda.get_device_name(dev)}")
    for B in (int(b) for b in args.batches.split(",")):
        lens = None
        if args.ragged:
            rr = np.random.RandomState(7)
            lens = rr.randint(max(1, L // 8), L + 1, size=B)
            lens[0] = L
        sp, tx, ln, Lmax = wl.synth_inputs(B, L, seed=0, src_lens=lens)
        a = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (sp, tx, ln)]
        with torch.no_grad():
            for _ in range(3):
                out = model(a[0], a[1], a[2], Lmax)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                out = model(a[0], a[1], a[2], Lmax)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.iters
        frames, T = int(out[9].sum()), int(out[0].shape[1])
        rows = int(model._lib.ns_last_phase2_rows(model._h))
        e2e = wl.algorithmic_flops_per_frame(cfg, T, L, fpp) * frames / dt / 1e12 / 157.3
        per = dt * 1e3 / B
        note = ""
        if best is not None and B >= 4:
            note = f"  {100.0 * (per / best - 1.0):+5.1f} % vs best smaller"
        if B >= 4:
            best = per if best is None else min(best, per)
        print(f"B={B:2d} T {T:4d} rows {rows:6d}: {dt * 1e3:7.3f} ms  {frames / dt / 1e6:6.3f} Mframes/s  {per:6.3f} ms/utt  e2e {e2e:5.3f}{note}", flush=True)


if __name__ == "__main__":
    main()
