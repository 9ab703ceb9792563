#!/usr/bin/env python3
"""The imported REFERENCE against itself, and against a float64 evaluation of itself — BUILD CONTAINER ONLY.

    python tools/reference_self_deviation.py [r06] [--pins pin_cfg2_b16 ...] [--no-f64]

(a) Self deviation.  The reference (``model.fastspeech2_align.FastSpeech2Align`` imported from the read-only checkout by the
    ``tests/golden/make_golden.py`` recipe) is run on the inputs of the committed BASELINE pins under changed
    fp32 summation orders — ``torch.set_num_threads(1)`` vs ``(8)``, ``torch.backends.mkldnn.flags(enabled=False)`` — and every
    variant is compared with the baseline evaluation (8 threads, mkldnn on: the one the fixtures hold) with the SAME
    classification the HIP path is held to (oracle/parity.py): duration flips, relative deviation of the two bucketized
    quantities (pitch free-running; energy with the pitch buckets pinned), bucket flips on/off edge, and the free-running
    PostNet max-abs.  This is the measurement behind DESIGN §2's statement that ``torch.bucketize``
    (model/modules.py:86-88,97-99) makes the free-running 1e-3 criterion unattainable for ANY evaluation whose summation
    order differs — the reference's own included — and behind the value of ``oracle/parity.py:EDGE_REL``.

(b) Float64 truth.  The same module cast with ``.double()`` (weights, tables and bins keep their fp32 values, only the
    arithmetic widens) is run with the bucket decisions pinned to the fp32 reference's (``p_targets`` / ``e_targets`` = the
    fp32 predictions, model/modules.py:82-84,93-95) and written, sub-sampled, to ``tests/golden/f64_cfg*.npz`` together with
    the statistics of |reference-fp32 - float64|.  ``tests/test_gpu_parity.py::test_accuracy_against_float64`` then holds the
    HIP path to 1.5x the reference's own distance from the truth, per quantity.

Writes ``profiles/<round>_reference_self_deviation.{md,json}``.  Nothing here runs on the GPU box; nothing from the reference
is copied (the fixtures hold numbers only).
"""
import argparse
import contextlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_golden as mg  # noqa: E402  (puts /root/reference on sys.path and registers the two import stubs)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import smart_nar_fast_tts_amd.workload as wl  # noqa: E402
from oracle import parity  # noqa: E402
from tests.util import load_golden, weights_for  # noqa: E402

PINS = {"pin_cfg1_single": "f64_cfg1", "pin_cfg2_b16": "f64_cfg2", "pin_cfg3_b128_sharded": None, "pin_cfg4_d512": "f64_cfg4",
        "pin_cfg5_longform": "f64_cfg5"}
# (label, threads, mkldnn enabled)
VARIANTS = [("threads=1", 1, True), ("mkldnn off, threads=8", 8, False), ("mkldnn off, threads=1", 1, False)]
STRIDE = 16


@contextlib.contextmanager
def evaluation(threads, mkldnn):
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        with torch.backends.mkldnn.flags(enabled=mkldnn):
            yield
    finally:
        torch.set_num_threads(old)


def forward(model, z, L, dtype=torch.float32, **kw):
    kw = {k: (torch.from_numpy(np.ascontiguousarray(v)).to(dtype) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    with torch.no_grad():
        out = model(torch.from_numpy(z["speakers"]), torch.from_numpy(z["texts"]), torch.from_numpy(z["in_src_lens"]), L, **kw)
    names = ["output", "postnet_output", "p_predictions", "e_predictions", "log_d_predictions", "d_rounded", "src_masks",
             "mel_masks", "src_lens", "mel_lens"]
    return {n: out[i].detach().numpy() for i, n in enumerate(names)}


def stats(a, b, sel=None, rel=False):
    """max / p99.9 / median of |a - b| (relative to max(|b|, 1) when rel) over sel."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    if rel:
        d = d / np.maximum(np.abs(b), 1.0)
    if sel is not None:
        d = d[sel]
    if d.size == 0:
        return {"max": 0.0, "p999": 0.0, "median": 0.0, "n": 0}
    return {"max": float(d.max()), "p999": float(np.quantile(d, 0.999)), "median": float(np.median(d)), "n": int(d.size)}


def compare(base, free, pinned_p, bins_p, bins_e):
    """One variant against the baseline evaluation, with the HIP path's own classification."""
    rec = {"duration_flips": int((base["d_rounded"] != free["d_rounded"]).sum()),
           "frame_counts_equal": bool(np.array_equal(base["mel_lens"], free["mel_lens"])),
           "log_d": stats(free["log_d_predictions"], base["log_d_predictions"], ~base["src_masks"])}
    if not rec["frame_counts_equal"]:
        return rec
    valid = ~base["mel_masks"]
    for key, got, ref, bins in (("pitch", free["p_predictions"], base["p_predictions"], bins_p),
                                ("energy", pinned_p["e_predictions"], base["e_predictions"], bins_e)):
        inr = parity.in_range(ref, bins, valid)
        fl = parity.classify_bucket_flips(got, ref, bins, valid)
        rec[key] = dict(stats(got, ref, inr, rel=True), in_range_frames=int(inr.sum()), bucket_flips=fl[0], off_edge=fl[1],
                        by_more_than_one=fl[2])
    rec["energy_free_running_bucket_flips"] = parity.classify_bucket_flips(free["e_predictions"], base["e_predictions"], bins_e, valid)[0]
    d = np.abs(free["postnet_output"].astype(np.float64) - base["postnet_output"])[valid]  # [frames, 80]
    rec["postnet_free_running"] = {"max_abs": float(d.max()), "frames_over_1e-3": int((d.max(axis=1) > 1e-3).sum()),
                                   "frames": int(valid.sum())}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("round", nargs="?", default="r06")
    ap.add_argument("--pins", nargs="+", default=list(PINS))
    ap.add_argument("--no-f64", action="store_true")
    ap.add_argument("--no-write-fixtures", action="store_true")
    args = ap.parse_args()
    report = {"round": args.round, "torch": torch.__version__, "host_cores": os.cpu_count(), "edge_rel_in_use": parity.EDGE_REL,
              "baseline_evaluation": "threads=8, mkldnn on (the evaluation tests/golden/pin_cfg*.npz hold)", "pins": {}}
    for pin in args.pins:
        meta, z = load_golden(pin)
        cfg, sd = weights_for(meta)
        L = int(meta["L"])
        model = mg.build_reference(cfg, sd)
        bins_p, bins_e = sd["variance_adaptor.pitch_bins"], sd["variance_adaptor.energy_bins"]
        t0 = time.time()
        with evaluation(8, True):
            base = forward(model, z, L)
        # the baseline evaluation IS the committed fixture (checked bit for bit: the recipe is reproducible here)
        assert np.array_equal(base["mel_lens"], z["mel_lens"]) and np.array_equal(base["p_predictions"], z["p_predictions"]), pin
        assert np.array_equal(base["postnet_output"][:, ::STRIDE], z["postnet_output_sub"]), pin
        rec = {"B": int(meta["B"]), "L": L, "T_pad": int(base["output"].shape[1]), "valid_frames": int((~base["mel_masks"]).sum()),
               "variants": {}}
        print(f"== {pin}: B {rec['B']} T_pad {rec['T_pad']} valid frames {rec['valid_frames']} (baseline {time.time() - t0:.1f} s)", flush=True)
        for label, threads, mk in VARIANTS:
            t0 = time.time()
            with evaluation(threads, mk):
                free = forward(model, z, L)
                pinned_p = forward(model, z, L, p_targets=base["p_predictions"]) if np.array_equal(free["mel_lens"], base["mel_lens"]) else None
            r = compare(base, free, pinned_p, bins_p, bins_e)
            r["bit_identical_to_baseline"] = bool(all(np.array_equal(free[k], base[k]) for k in ("output", "postnet_output", "p_predictions", "e_predictions", "log_d_predictions")))
            r["seconds"] = round(time.time() - t0, 1)
            rec["variants"][label] = r
            print(f"   {label}: {json.dumps(r)}", flush=True)
        if not args.no_f64:
            t0 = time.time()
            m64 = mg.build_reference(cfg, sd).double()
            with evaluation(8, True):
                # the fp32 baseline with ITS OWN decisions as targets (== its free run, since the targets are its predictions) and
                # the float64 evaluation taking the same decisions: what differs downstream is arithmetic only
                t64 = forward(m64, z, L, dtype=torch.float64, p_targets=base["p_predictions"], e_targets=base["e_predictions"])
            assert np.array_equal(t64["d_rounded"], base["d_rounded"].astype(np.float64)), (pin, "float64 durations differ from fp32's")
            assert np.array_equal(t64["mel_lens"], base["mel_lens"])
            valid = ~base["mel_masks"]
            sub = np.zeros_like(valid)
            sub[:, ::STRIDE] = True
            vs = (valid & sub)[:, ::STRIDE]
            f = {"log_d": stats(base["log_d_predictions"], t64["log_d_predictions"], ~base["src_masks"]),
                 "pitch_rel": stats(base["p_predictions"], t64["p_predictions"], parity.in_range(base["p_predictions"], bins_p, valid), rel=True),
                 "energy_rel": stats(base["e_predictions"], t64["e_predictions"], parity.in_range(base["e_predictions"], bins_e, valid), rel=True),
                 "pitch_abs": stats(base["p_predictions"], t64["p_predictions"], valid),
                 "energy_abs": stats(base["e_predictions"], t64["e_predictions"], valid),
                 "mel_sub": stats(base["output"][:, ::STRIDE], t64["output"][:, ::STRIDE], vs),
                 "postnet_sub": stats(base["postnet_output"][:, ::STRIDE], t64["postnet_output"][:, ::STRIDE], vs),
                 "mel_all_frames": stats(base["output"], t64["output"], valid),
                 "postnet_all_frames": stats(base["postnet_output"], t64["postnet_output"], valid),
                 "seconds": round(time.time() - t0, 1)}
            rec["reference_fp32_minus_float64"] = f
            print(f"   fp32 - f64: {json.dumps(f)}", flush=True)
            # every fp32 variant's distance from the truth too (pinned to the baseline's decisions): is any of them closer?
            if PINS.get(pin) and not args.no_write_fixtures:
                path = os.path.join(ROOT, "tests", "golden", PINS[pin] + ".npz")
                np.savez_compressed(
                    path, meta=np.array(json.dumps(dict(meta, source_pin=pin, frame_stride=STRIDE, arithmetic="float64",
                                                        buckets="pinned to the fp32 reference's predictions (the pin's p/e_predictions)"))),
                    p_predictions=t64["p_predictions"], e_predictions=t64["e_predictions"], log_d_predictions=t64["log_d_predictions"],
                    output_sub=t64["output"][:, ::STRIDE], postnet_output_sub=t64["postnet_output"][:, ::STRIDE],
                    ref32_stats=np.array(json.dumps(f)))
                print(f"   wrote {path} {os.path.getsize(path) / 1024:.0f} KiB", flush=True)
            del m64
        report["pins"][pin] = rec
        del model

    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    base = os.path.join(ROOT, "profiles", f"{args.round}_reference_self_deviation")
    json.dump(report, open(base + ".json", "w"), indent=1)
    open(base + ".md", "w").write(render(report))
    print(render(report))


def render(rep):
    L = [f"# The reference against itself, and against float64 — {rep['round']}", "",
         f"`tools/reference_self_deviation.py`, build container ({rep['host_cores']} cores, torch {rep['torch']}), the imported reference "
         "(`/root/reference/model/fastspeech2_align.py`) on the inputs of the committed BASELINE pins.", "",
         f"Baseline evaluation: {rep['baseline_evaluation']}; checked bit for bit against the fixtures before anything else runs.", "",
         "## (a) fp32 reference vs fp32 reference under a changed summation order", "",
         "Same classification as the HIP path's (`oracle/parity.py`): relative deviation = |variant - baseline| / max(|baseline|, 1) on the",
         "frames whose baseline value lies inside the bin range; pitch free-running, energy with the pitch buckets pinned to the baseline's;",
         f"a flip is *off edge* when the baseline value is not within `EDGE_REL` = {rep['edge_rel_in_use']:g} of a bin edge.", "",
         "| pin | variant | bit-identical | duration flips | pitch max rel | pitch p99.9 | pitch flips (off edge) | energy max rel | energy p99.9 | "
         "energy flips (off edge) | free-running PostNet max-abs | frames > 1e-3 |",
         "|---|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for pin, rec in rep["pins"].items():
        for label, r in rec["variants"].items():
            if "pitch" not in r:
                L.append(f"| {pin} | {label} | {r['bit_identical_to_baseline']} | {r['duration_flips']} | frame counts differ | | | | | | | |")
                continue
            p, e, pn = r["pitch"], r["energy"], r["postnet_free_running"]
            L.append(f"| {pin} | {label} | {r['bit_identical_to_baseline']} | {r['duration_flips']} | {p['max']:.2e} | {p['p999']:.2e} | "
                     f"{p['bucket_flips']} ({p['off_edge']}) | {e['max']:.2e} | {e['p999']:.2e} | {e['bucket_flips']} ({e['off_edge']}) | "
                     f"{pn['max_abs']:.2e} | {pn['frames_over_1e-3']} / {pn['frames']} |")
    if any("reference_fp32_minus_float64" in r for r in rep["pins"].values()):
        L += ["", "## (b) fp32 reference vs the same module in float64 (bucket decisions pinned to the fp32 run's)", "",
              "max / p99.9 of |fp32 - float64|; pitch / energy relative as above, log-duration and mel absolute (every valid frame).", "",
              "| pin | log_d max | pitch rel max | pitch rel p99.9 | energy rel max | energy rel p99.9 | mel max | mel p99.9 | PostNet mel max | PostNet p99.9 |",
              "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
        for pin, rec in rep["pins"].items():
            f = rec.get("reference_fp32_minus_float64")
            if f:
                L.append(f"| {pin} | {f['log_d']['max']:.2e} | {f['pitch_rel']['max']:.2e} | {f['pitch_rel']['p999']:.2e} | {f['energy_rel']['max']:.2e} | "
                         f"{f['energy_rel']['p999']:.2e} | {f['mel_all_frames']['max']:.2e} | {f['mel_all_frames']['p999']:.2e} | "
                         f"{f['postnet_all_frames']['max']:.2e} | {f['postnet_all_frames']['p999']:.2e} |")
    return "\n".join(L) + "\n"


if __name__ == "__main__":
    main()
