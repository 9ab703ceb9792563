"""Small-batch serving on several HIP streams: do independent B=1 forwards overlap?

tools/lab/two_queues.hip shows two queues overlap whenever both kernels' workgroups fit on the chip together.  A B=1 forward is
~80 launches of well under 256 workgroups each, so forwards of different requests on different streams should overlap.  This
measures it: B=1 (and B=2, 4) 100-phoneme forwards in capacity mode (no host wait inside), round-robin over 1, 2, 4, 8 streams
of ONE model (one arena; one workspace set per stream).  Prints utterances/s and the per-forward latency seen on a stream.
    python tools/multi_stream_small.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

wl = importlib.import_module("smart-nar_fast_tts_amd.workload")
FastSpeech2Align = importlib.import_module("smart-nar_fast_tts_amd.model").FastSpeech2Align


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,4")
    ap.add_argument("--phonemes", type=int, default=100)
    ap.add_argument("--streams", default="1,2,4,8")
    ap.add_argument("--n", type=int, default=400)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = wl.model_config("ljspeech")
    m = FastSpeech2Align(wl.preprocess_config(), cfg).to(dev).eval()
    m.load_state_dict(wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=8.0))
    out = {}
    for B in [int(x) for x in args.batches.split(",")]:
        sp, tx, ln, L = wl.synth_inputs(B, args.phonemes, seed=0)
        a = [torch.from_numpy(x).to(dev) for x in (sp, tx, ln)]
        with torch.no_grad():
            cap = int(m(a[0], a[1], a[2], L)[9].max()) if B > 4 else 1024  # large batches: the exact padded length (same work as the synchronous forward)
        for ns in [int(x) for x in args.streams.split(",")]:
            streams = [torch.cuda.Stream(dev) for _ in range(ns)]
            n = args.n
            with torch.no_grad():
                for i in range(4 * ns):
                    with torch.cuda.stream(streams[i % ns]):
                        m(a[0], a[1], a[2], L, max_mel_len=cap, async_status=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    with torch.cuda.stream(streams[i % ns]):
                        o = m(a[0], a[1], a[2], L, max_mel_len=cap, async_status=True)
                t_enq = time.perf_counter() - t0
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            o.check()
            print(f"B={B} streams={ns}: {n * B / dt:8.0f} utterances/s  ({dt / n * 1e3:.3f} ms per forward wall, host enqueue {t_enq / n * 1e3:.3f} ms per forward)", flush=True)
            out[(B, ns)] = n * B / dt
    return out


if __name__ == "__main__":
    main()
