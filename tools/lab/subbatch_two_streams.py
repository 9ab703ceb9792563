"""Concept test: a batch between steps as TWO concurrent sub-forwards (main = a step-friendly count on one stream, the rest on a
second stream) against the single forward.  Uses capacity mode (known T) for both so that no host wait serialises the enqueue."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smart_nar_fast_tts_amd.workload as wl
from smart_nar_fast_tts_amd.model import FastSpeech2Align
cfg = wl.model_config("ljspeech"); sd = wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=8.0)
m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval(); m.load_state_dict(sd)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B, Bm in ((9, 8), (10, 8), (12, 8), (17, 16), (20, 16), (24, 16), (5, 4)):
    sp, tx, ln, L = wl.synth_inputs(B, 128, seed=0)
    a = [dev(x) for x in (sp, tx, ln)]
    with torch.no_grad():
        ref = m(a[0], a[1], a[2], L)
        T = int(ref[0].shape[1])
        am = [x[:Bm].contiguous() for x in a]; asd = [x[Bm:].contiguous() for x in a]
        Tm = int(m(am[0], am[1], am[2], L)[0].shape[1]); Ts = int(m(asd[0], asd[1], asd[2], L)[0].shape[1])
        def single(): m(a[0], a[1], a[2], L, max_mel_len=T, async_status=True)
        def sync_single(): m(a[0], a[1], a[2], L)
        def split():
            with torch.cuda.stream(s1): m(am[0], am[1], am[2], L, max_mel_len=Tm, async_status=True)
            with torch.cuda.stream(s2): m(asd[0], asd[1], asd[2], L, max_mel_len=Ts, async_status=True)
        def main_only():
            with torch.cuda.stream(s1): m(am[0], am[1], am[2], L, max_mel_len=Tm, async_status=True)
        def side_only():
            with torch.cuda.stream(s2): m(asd[0], asd[1], asd[2], L, max_mel_len=Ts, async_status=True)
        print(f"B={B} = {Bm}+{B-Bm}: single sync {timeit(sync_single):.3f}  single async {timeit(single):.3f}  split on 2 streams {timeit(split):.3f}  (main alone {timeit(main_only):.3f}, side alone {timeit(side_only):.3f})", flush=True)
