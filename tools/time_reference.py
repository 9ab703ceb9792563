#!/usr/bin/env python3
"""Times the REFERENCE's own CPU forward beside the build's CPU restatement — BUILD CONTAINER ONLY.

    python tools/time_reference.py [--configs cfg1_single cfg2_b16] [--runs 3] [--write]

Imports ``model.fastspeech2_align.FastSpeech2Align`` from the read-only checkout at /root/reference exactly like
``tests/golden/make_golden.py`` does (two import stubs, synthetic stats.json), loads the bench's seeded
weights (``workload.synth_state_dict``, the ones bench.py runs) and times ``forward()`` on the bench's inputs:

* as-is                       — what a user of the reference gets;
* table rebuild excluded      — the same forward with ``get_sinusoid_encoding_table`` (transformer/Models.py:10-30), which
                                the reference re-evaluates in pure Python on every call with T > max_seq_len
                                (transformer/Models.py:218-225), replaced by a cached copy: the number that is fair to
                                compare kernels against;
* oracle ("port")             — oracle/fs2_oracle.py on the same inputs and thread count: the figure bench.py can re-time
                                on the GPU box's host, where the reference's Python cannot travel.

``--write`` replaces the block between the BEGIN/END markers in BASELINE.md with the measured table.  Nothing here runs
on the GPU box; nothing from the reference is copied.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_golden as mg  # noqa: E402  (sets up sys.path for the reference and the import stubs)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import smart_nar_fast_tts_amd.workload as wl  # noqa: E402
from oracle import fs2_oracle as orc  # noqa: E402

BEGIN, END = "<!-- BEGIN time_reference.py -->", "<!-- END time_reference.py -->"


def timed(fn, runs):
    fn()  # warm-up
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), min(ts), max(ts), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="+", default=["cfg1_single", "cfg2_b16"])
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--write", action="store_true", help="rewrite the marked block of BASELINE.md")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)

    import transformer.Models as RM  # the reference module that owns the table builder

    rows = []
    for name in args.configs:
        cfg_name, B, L, fpp = wl.WORKLOADS[name]
        cfg = wl.model_config(cfg_name)
        sd = wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=fpp)
        model = mg.build_reference(cfg, sd)
        sp, tx, ln, Lmax = wl.synth_inputs(B, L, seed=0)
        inp = [torch.from_numpy(a) for a in (sp, tx, ln)]

        def ref_fwd():
            with torch.no_grad():
                return model(inp[0], inp[1], inp[2], Lmax)

        runs = args.runs if B > 1 else max(args.runs, 11)
        t_asis = timed(ref_fwd, runs)
        frames = int(t_asis[3][9].sum())
        T_pad = int(t_asis[3][0].shape[1])
        # the per-call table rebuild alone, and the forward with it served from a cache
        real_table = RM.get_sinusoid_encoding_table
        t0 = time.perf_counter()
        if T_pad > cfg["max_seq_len"]:
            real_table(T_pad, cfg["transformer"]["decoder_hidden"])
        t_table = time.perf_counter() - t0 if T_pad > cfg["max_seq_len"] else 0.0
        cache = {}

        def cached_table(n_position, d_hid, padding_idx=None):
            key = (n_position, d_hid, padding_idx)
            if key not in cache:
                cache[key] = real_table(n_position, d_hid, padding_idx)
            return cache[key]

        RM.get_sinusoid_encoding_table = cached_table
        try:
            t_notab = timed(ref_fwd, runs)
        finally:
            RM.get_sinusoid_encoding_table = real_table
        w = orc.to_torch_weights(sd)

        def orc_fwd():
            with torch.no_grad():
                return orc.forward(w, cfg, inp[0], inp[1], inp[2], Lmax)

        t_orc = timed(orc_fwd, runs)
        assert torch.equal(t_orc[3][9], t_asis[3][9]), "oracle and reference disagree on frame counts"
        err = float((t_orc[3][1] - t_asis[3][1]).abs().max())
        rows.append(dict(workload=name, B=B, L=L, T_pad=T_pad, valid_frames=frames, threads=args.threads, runs=runs,
                         ref_asis_s=t_asis[0], ref_asis_min_s=t_asis[1], ref_asis_max_s=t_asis[2],
                         table_rebuild_s=t_table, ref_no_table_s=t_notab[0], oracle_s=t_orc[0],
                         ref_asis_fps=frames / t_asis[0], ref_no_table_fps=frames / t_notab[0], oracle_fps=frames / t_orc[0],
                         oracle_vs_ref_postnet_max_abs=err))
        print(json.dumps(rows[-1]), flush=True)

    lines = [BEGIN,
             f"Regenerate with `python tools/time_reference.py --write` (build container only; torch {torch.__version__} CPU kernels, "
             f"{args.threads} threads on {os.cpu_count()} cores; median of the timed runs after one warm-up; bench weights seed 0).",
             "",
             "| Workload | valid frames (T_pad) | reference as-is | of which table rebuild | reference, table cached | oracle (port) | oracle vs reference, PostNet mel max-abs |",
             "|---|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        lines.append(
            f"| {r['workload']} (B={r['B']}, L={r['L']}) | {r['valid_frames']} ({r['T_pad']}) | "
            f"{r['ref_asis_s'] * 1e3:.1f} ms = {r['ref_asis_fps']:.0f} frames/s (min {r['ref_asis_min_s'] * 1e3:.1f}, max {r['ref_asis_max_s'] * 1e3:.1f}, n={r['runs']}) | "
            f"{r['table_rebuild_s'] * 1e3:.1f} ms | {r['ref_no_table_s'] * 1e3:.1f} ms = {r['ref_no_table_fps']:.0f} frames/s | "
            f"{r['oracle_s'] * 1e3:.1f} ms = {r['oracle_fps']:.0f} frames/s | {r['oracle_vs_ref_postnet_max_abs']:.1e} |")
    lines.append(END)
    block = "\n".join(lines)
    print(block)
    if args.write:
        p = os.path.join(ROOT, "BASELINE.md")
        s = open(p).read()
        if BEGIN in s and END in s:
            s = s[:s.index(BEGIN)] + block + s[s.index(END) + len(END):]
        else:
            s = s.rstrip("\n") + "\n\n## Reference CPU forward timed beside the port (tools/time_reference.py)\n\n" + block + "\n"
        open(p, "w").write(s)


if __name__ == "__main__":
    main()
