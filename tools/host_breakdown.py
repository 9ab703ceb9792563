import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import smart_nar_fast_tts_amd.workload as wl
from smart_nar_fast_tts_amd.model import FastSpeech2Align
from smart_nar_fast_tts_amd import _lib
cfg = wl.model_config("ljspeech")
m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
m.load_state_dict(wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=8.0))
sp, tx, ln, L = wl.synth_inputs(1, 100, seed=0)
a = [torch.from_numpy(x).cuda() for x in (sp, tx, ln)]
lib = m._lib
T = {}
def wrap(name):
    f = getattr(lib, name)
    def g(*args):
        t0 = time.perf_counter(); r = f(*args); T.setdefault(name, []).append((t0, time.perf_counter())); return r
    return g
class L2:
    def __getattr__(self, n):
        return wrap(n) if n in ("ns_forward_durations", "ns_forward_durations_packed", "ns_forward_mel", "ns_forward_mel_packed") else getattr(lib, n)
m._lib = L2()
w0 = m._wait_phase1
def w(dev, pin_np=None):
    t0 = time.perf_counter(); w0(dev, pin_np); T.setdefault("wait", []).append((t0, time.perf_counter()))
m._wait_phase1 = w
rows = []
with torch.no_grad():
    for i in range(30):
        T.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
        o = m(a[0], a[1], a[2], L)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        d, wt, ml = T["ns_forward_durations"][0], T["wait"][0], (T.get("ns_forward_mel_packed") or T["ns_forward_mel"])[0]
        rows.append([(d[0]-t0), (d[1]-d[0]), (wt[0]-d[1]), (wt[1]-wt[0]), (ml[0]-wt[1]), (ml[1]-ml[0]), (t1-ml[1]), (t2-t1), (t2-t0)])
r = np.median(np.array(rows[5:]) * 1e6, axis=0)
print("us: pre %.1f | C phase1 enqueue %.1f | to wait %.1f | wait %.1f | python after wait %.1f | C phase2 enqueue %.1f | return %.1f | final sync %.1f | total %.1f" % tuple(r))
