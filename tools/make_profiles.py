#!/usr/bin/env python3
"""Turn one round's rocprofv3 outputs (gpurun_out/rNN_{trace,fetch,write,mfma,lds}[_workload]/t_results.db +
rNN_bench*.json, written by tools/collect_profiles.sh on the GPU box) into the committed summaries under profiles/:

    python tools/make_profiles.py r02

* profiles/rNN_kernel_stats.md          per-kernel stats of `rocprofv3 --kernel-trace --stats -- python bench.py ...`
* profiles/rNN_pmc.md                   per-dispatch PMC averages (separate passes) for the hot kernels
* profiles/dominant_kernel_traffic.json what bench.py reports as roofline.traffic (HBM bytes per launch)
* profiles/rNN_bench.json               the bench line of the same build (un-profiled) and under the trace
"""
import json
import re
import subprocess
import sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = "gpurun_out"


def run(*a):
    return subprocess.run(["python", *a], capture_output=True, text=True).stdout


def pmc_rows(db):
    rows = {}
    for line in run("tools/rocpd_pmc.py", db).splitlines():
        m = re.match(r"\| `(.+?)` \| (\d+) \| (\d+) \| ([\d.]+) \| (\w+) \| ([\d.]+) \|", line)
        if m:
            rows[(m.group(1), int(m.group(2)), m.group(5))] = (float(m.group(6)), float(m.group(4)), int(m.group(3)))
    return rows


open(f"profiles/{ROUND}_kernel_stats.md", "w").write(run("tools/rocpd_stats.py", f"{G}/{ROUND}_trace/t_results.db"))
import os
import shutil
if os.path.exists(f"{G}/{ROUND}_rocprofv3_kernel_stats.csv"):  # rocprofv3 --stats' own summary file, verbatim
    shutil.copy(f"{G}/{ROUND}_rocprofv3_kernel_stats.csv", f"profiles/{ROUND}_rocprofv3_kernel_stats.csv")

bench = json.load(open(f"{G}/{ROUND}_bench.json"))
under = json.load(open(f"{G}/{ROUND}_bench_under_trace.json"))

# The dominant kernel's launches inside the TIMED region of the traced run (rocprofv3's table averages the warm-up
# launches in as well): the last `launches` dispatches of the biggest-grid k_conv_gemm, to set beside bench.py's
# HIP-event average of the same launches in the same process.
import sqlite3
con = sqlite3.connect(f"{G}/{ROUND}_trace/t_results.db")
top = con.execute("select name, grid_x / workgroup_x as wgs, sum(end - start) as tot from kernels where name like '%k_conv_gemm<%' "
                  "group by name, wgs order by tot desc limit 1").fetchone()
dom_name = re.sub(r"\(.*", "", top[0]).replace("void ", "").replace("ns::", "") if top else ""
rows = con.execute("select start, end, grid_x / workgroup_x as wgs from kernels where name = ? order by start", (top[0],)).fetchall() if top else []
if rows:
    big = max(r[2] for r in rows)
    durs = [(r[1] - r[0]) / 1e3 for r in rows if r[2] == big]
    # the timed region of the traced run = its last `steps` forwards (bench.py's events sample every 5th of them since round 4)
    n = max(1, len(durs) // max(1, (under["steps"] + under["warmup"]))) * int(under["steps"])
    timed = durs[-n:]
    under["rocprofv3_same_launches"] = {"kernel_workgroups": int(big), "launches": len(timed),
                                        "avg_us": round(sum(timed) / len(timed), 1),
                                        "hip_event_avg_us_in_bench": round(under["roofline"]["avg_launch_ms"] * 1e3, 1)}
    # ONE number per claim: the three averages this kernel can be quoted with, side by side, each with the fraction of the
    # fp32 MFMA peak it implies (algorithmic flops per launch / average duration / 157.3 TFLOP/s), and the launches in order,
    # forward by forward, so that the spread between them can be read off instead of guessed at
    cw = under["config"]["workload"]
    rows_dom = under["config"]["global_batch"] * int(re.search(r"T_pad (\d+)", cw).group(1))
    gflop = 2.0 * rows_dom * 9 * 256 * 1024 / 1e9
    per_fwd = max(1, len(durs) // max(1, (under["steps"] + under["warmup"])))
    fwd_avgs = [sum(durs[i:i + per_fwd]) / per_fwd for i in range(0, len(durs) - per_fwd + 1, per_fwd)]
    last = durs[-per_fwd:]
    frac = lambda us: gflop / us * 1e3 / 157.3  # GFLOP / us = PFLOP/s
    under["dominant_kernel_averages"] = {
        "gflop_per_launch": round(gflop, 2),
        "all_traced_launches": {"n": len(durs), "avg_us": round(sum(durs) / len(durs), 1), "frac": round(frac(sum(durs) / len(durs)), 4)},
        "timed_region_only": {"n": len(timed), "avg_us": round(sum(timed) / len(timed), 1), "frac": round(frac(sum(timed) / len(timed)), 4)},
        "last_forward": {"n": len(last), "avg_us": round(sum(last) / len(last), 1), "frac": round(frac(sum(last) / len(last)), 4)},
        "hip_events_in_bench_same_process": {"avg_us": round(under["roofline"]["avg_launch_ms"] * 1e3, 1), "frac": under["roofline"]["frac"]},
        "unprofiled_bench_hip_events": {"avg_us": round(bench["roofline"]["avg_launch_ms"] * 1e3, 1), "frac": bench["roofline"]["frac"]},
        "per_forward_avg_us_in_launch_order": [round(x, 1) for x in fwd_avgs],
        "min_us": round(min(durs), 1), "max_us": round(max(durs), 1)}
    a = under["dominant_kernel_averages"]
    with open(f"profiles/{ROUND}_kernel_stats.md", "a") as fh:
        fh.write(f"\n## Dominant kernel: one number per claim\n\n{int(big)}-workgroup `{dom_name}`, {gflop:.2f} GFLOP per launch (2 x rows x 9 x 256 x 1024, rows = B x T_pad = {rows_dom}); "
                 "fraction = GFLOP / average duration / 157.3 TFLOP/s.\n\n"
                 "| average over | launches | avg us | of peak |\n|---|---:|---:|---:|\n"
                 f"| every launch of the trace (warm-up forwards included) | {a['all_traced_launches']['n']} | {a['all_traced_launches']['avg_us']} | {a['all_traced_launches']['frac']:.3f} |\n"
                 f"| the timed region of the traced run (what bench.py's roofline covers) | {a['timed_region_only']['n']} | {a['timed_region_only']['avg_us']} | {a['timed_region_only']['frac']:.3f} |\n"
                 f"| the last forward of the trace | {a['last_forward']['n']} | {a['last_forward']['avg_us']} | {a['last_forward']['frac']:.3f} |\n"
                 f"| bench.py's own events in the same (profiled) process: it times every 5th forward of its timed region, under the trace the FIRST timed forward only | {under['roofline']['launches']} | {a['hip_events_in_bench_same_process']['avg_us']} | {a['hip_events_in_bench_same_process']['frac']:.3f} |\n"
                 f"| bench.py's events (on the dispatch packets), UNPROFILED run (`{ROUND}_bench.json`: `roofline.frac` on the driver's line; `roofline.frac_rocprof` is the second row) | {bench['roofline']['launches']} | {a['unprofiled_bench_hip_events']['avg_us']} | {a['unprofiled_bench_hip_events']['frac']:.3f} |\n\n"
                 f"Launch-order view, average per forward (us): {', '.join(str(x) for x in a['per_forward_avg_us_in_launch_order'])} "
                 f"(min {a['min_us']}, max {a['max_us']}).  The spread inside one trace is the order of the forwards, not noise: the first "
                 "forwards after the process starts run the same kernel slower and the figure settles over the following ones (the chip raises "
                 "its clock under sustained load; `GRBM_GUI_ACTIVE` / duration in the PMC pass gives the clock of the settled state), and the "
                 "profiler's own per-dispatch cost sits on top in the traced run (compare the last two rows).  DESIGN.md quotes the "
                 "unprofiled HIP-event figure and names it as such.\n")


def launches_per_forward(db):
    """kernel dispatches between two consecutive k_embed_pos launches (= one forward; the device-to-host copy of
    mel_lens is a blit kernel and is counted), from the last complete forward of the trace"""
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from kernels order by start").fetchall()]
    idx = [i for i, n in enumerate(names) if "k_embed_pos" in n]
    if len(idx) < 2:
        return None
    seg = names[idx[-2]:idx[-1]]
    return len(seg), sum(1 for n in seg if "rocclr" in n)


lp = launches_per_forward(f"{G}/{ROUND}_trace/t_results.db")
if lp:
    with open(f"profiles/{ROUND}_kernel_stats.md", "a") as fh:
        fh.write(f"\nLaunches per config-2 forward (dispatches from one `k_embed_pos` to the next, bench.py --no-extras under the "
                 f"trace): **{lp[0]}** ({lp[0] - lp[1]} kernels of the library + {lp[1]} runtime copy kernel for the mel_lens read).\n")
    under["launches_per_forward"] = lp[0]
json.dump({"bench": bench, "bench_under_kernel_trace": under}, open(f"profiles/{ROUND}_bench.json", "w"), indent=1)

# the other BASELINE configs (kernel trace only) and the opt-in bf16x3 mode
for fn in ("batch_size_sweep", "batch_size_sweep_plan0", "batch_size_sweep_ragged", "batch_size_sweep_ragged_grid", "fuzz",
           "ab_tile16", "ab_tile16_ragged", "phase1_packing_ab"):
    src = f"{G}/{ROUND}_{fn}.txt"
    if os.path.exists(src):
        txt = "".join(l for l in open(src) if "amdgpu.ids" not in l)
        open(f"profiles/{ROUND}_{fn}.txt", "w").write(txt)
for b in (9, 17, 20):  # per-launch sequences of batch sizes that sit between steps
    src = f"{G}/{ROUND}_tb_b{b}.seq.txt"
    if os.path.exists(src):
        shutil.copy(src, f"profiles/{ROUND}_launch_sequence_b{b}.txt")
for wl in ("cfg1_single", "cfg4_d512", "cfg5_longform", "cfg5_longform_gaussian", "bf16x3",
           "cfg2_b16_ragged_packed", "cfg2_b16_ragged_grid", "cfg5_longform_ragged_packed", "cfg5_longform_ragged_grid"):
    db = f"{G}/{ROUND}_trace_{wl}/t_results.db"
    if not os.path.exists(db):
        continue
    txt = run("tools/rocpd_stats.py", db)
    lpw = launches_per_forward(db)
    if lpw:
        txt += f"\nLaunches per forward: {lpw[0]} (incl. {lpw[1]} runtime copy kernel).\n"
    ub = f"{G}/{ROUND}_bench_under_trace_{wl}.json"
    if os.path.exists(ub) and os.path.getsize(ub) > 2:
        u = json.load(open(ub))
        txt += (f"\nbench.py under this trace: {u['value']:.0f} frames/s, {u['ms_per_step']:.3f} ms/step, dominant kernel "
                f"{u['roofline']['achieved']} TFLOP/s ({u['config']['workload'][:60]})\n")
    open(f"profiles/{ROUND}_kernel_stats_{wl}.md", "w").write(txt)
if os.path.exists(f"{G}/{ROUND}_bench_bf16x3.json") and os.path.getsize(f"{G}/{ROUND}_bench_bf16x3.json") > 2:
    shutil.copy(f"{G}/{ROUND}_bench_bf16x3.json", f"profiles/{ROUND}_bench_bf16x3.json")

# the dominant kernel = the most expensive (kernel, grid) of the trace
f = pmc_rows(f"{G}/{ROUND}_fetch/t_results.db")
w = pmc_rows(f"{G}/{ROUND}_write/t_results.db")
m = pmc_rows(f"{G}/{ROUND}_mfma/t_results.db")
dom = max((k for k in f if k[2] == "FETCH_SIZE" and "k_conv_gemm" in k[0]), key=lambda k: f[k][1] * f[k][2])
kname, wgs = dom[0], dom[1]
cfgw = bench["config"]["workload"]
T_pad = int(re.search(r"T_pad (\d+)", cfgw).group(1))
rows = bench["config"]["global_batch"] * T_pad
alg = (rows * 256 + 1024 * 2304 + rows * 1024) * 4
fetch_b = f[(kname, wgs, "FETCH_SIZE")][0] * 1024 * 2  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md §HBM)
write_b = w[(kname, wgs, "WRITE_SIZE")][0] * 1024
busy = m[(kname, wgs, "SQ_VALU_MFMA_BUSY_CYCLES")][0]
gui = m[(kname, wgs, "GRBM_GUI_ACTIVE")][0]
dur = m[(kname, wgs, "GRBM_GUI_ACTIVE")][1]
d = {
    "kernel": f"{kname} (decoder FFN w_1: Conv1d k=9 256->1024, bias+ReLU), {wgs} workgroups, rows B*T_pad = {rows}",
    # geometry of the measured launch: bench.py reports roofline.traffic only for a run whose dominant kernel matches it
    "rows": rows, "d_model": 256, "d_inner": 1024, "k": 9, "workgroups": int(wgs),
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / MFMA counters, one pass each, `python bench.py --steps 5 --warmup 2`; per-launch averages",
    "fetch_size_kb_raw": f[(kname, wgs, "FETCH_SIZE")][0], "write_size_kb_raw": w[(kname, wgs, "WRITE_SIZE")][0],
    "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
    "algorithmic_bytes_per_launch": alg,
    "note": "FETCH_SIZE counts the L2's fabric-side requests (Infinity-Cache hits included); the 9.4 MB weight matrix is fetched once per XCD (8x = 75 MB), the activation rows once",
    "mfma_util_pct": round(100 * busy / ((gui / 8) * 1024), 2), "clock_ghz_under_pmc": round(gui / 8 / (dur * 1e-6) / 1e9, 3),
    "mfma_flops_executed": m[(kname, wgs, "SQ_INSTS_VALU_MFMA_MOPS_F32")][0] * 512, "avg_duration_us_under_pmc": dur,
    # the same launches in the round's rocprofv3 kernel trace, timed region only (bench.py reports roofline.frac_rocprof from it)
    "rocprof_timed_avg_us": under.get("dominant_kernel_averages", {}).get("timed_region_only", {}).get("avg_us"),
    "rocprof_source": f"profiles/{ROUND}_kernel_stats.md (Dominant kernel: one number per claim)",
}
json.dump(d, open("profiles/dominant_kernel_traffic.json", "w"), indent=1)

# fused attention of the decoder stack: FETCH / WRITE per launch against its algorithmic bytes (Q, K, V read once, O written
# once = 4 * rows * d * 4 B), configs 2 and 5
att = {}
for suf, label in (("", "cfg2_b16"), ("_cfg5_longform", "cfg5_longform")):
    fdb, wdb = f"{G}/{ROUND}_fetch{suf}/t_results.db", f"{G}/{ROUND}_write{suf}/t_results.db"
    if not (os.path.exists(fdb) and os.path.exists(wdb)):
        continue
    fa, wa = pmc_rows(fdb), pmc_rows(wdb)
    ks = [k for k in fa if "k_attention<" in k[0] and k[2] == "FETCH_SIZE"]
    if not ks:
        continue
    k = max(ks, key=lambda k: fa[k][1] * fa[k][2])  # the decoder launches (longest total time)
    ub = f"{G}/{ROUND}_bench_under_trace{suf}.json"
    u = json.load(open(ub)) if os.path.exists(ub) and os.path.getsize(ub) > 2 else bench
    T = int(re.search(r"T_pad (\d+)", u["config"]["workload"]).group(1))
    rows_a = u["config"]["global_batch"] * T
    att[label] = {"kernel": k[0], "rows": rows_a, "T_pad": T, "avg_duration_us_under_pmc": round(fa[k][1], 1),
                  "fetch_bytes_corrected_x2": fa[k][0] * 2048, "write_bytes": wa[(k[0], k[1], "WRITE_SIZE")][0] * 1024,
                  "algorithmic_bytes_per_launch": 4 * rows_a * 256 * 4}
    att[label]["hbm_bytes_per_launch"] = att[label]["fetch_bytes_corrected_x2"] + att[label]["write_bytes"]
    att[label]["traffic_over_algorithmic"] = round(att[label]["hbm_bytes_per_launch"] / att[label]["algorithmic_bytes_per_launch"], 2)
if att:
    json.dump(att, open(f"profiles/{ROUND}_attention_traffic.json", "w"), indent=1)

with open(f"profiles/{ROUND}_pmc.md", "w") as fh:
    fh.write(f"# PMC passes for {ROUND} (rocprofv3 --pmc, one counter group per pass; per-dispatch averages, hot kernels only)\n")
    for tag in ("fetch", "write", "mfma", "lds", "fetch_cfg5_longform", "write_cfg5_longform", "mfma_cfg5_longform", "lds_cfg5_longform"):
        if not os.path.exists(f"{G}/{ROUND}_{tag}/t_results.db"):
            continue
        out = run("tools/rocpd_pmc.py", f"{G}/{ROUND}_{tag}/t_results.db").splitlines()
        keep = [l for l in out if l.startswith("| kernel") or l.startswith("|---")]
        body = [l for l in out if l.startswith("| `")]
        # hot kernels of the benchmark workload: large grids only
        body = [l for l in body if int(l.split("|")[2]) >= 250 or "k_attention<" in l]
        fh.write(f"\n## pass: {tag}\n\n" + "\n".join(keep + body[:40]) + "\n")
# SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES per hot kernel (what share of its waves' lifetime the matrix pipe was busy)
with open(f"profiles/{ROUND}_pmc.md", "a") as fh:
    fh.write("\n## derived: matrix-pipe busy share per kernel (mfma pass): SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES\n\n"
             "| kernel | workgroups | launches | avg us | MFMA busy / wave cycles | MFMA util % (busy / (GUI_ACTIVE/8 x 1024)) |\n|---|---:|---:|---:|---:|---:|\n")
    keys = sorted({(k[0], k[1]) for k in m if k[2] == "SQ_WAVE_CYCLES"}, key=lambda k: -m[(k[0], k[1], "SQ_WAVE_CYCLES")][1] * m[(k[0], k[1], "SQ_WAVE_CYCLES")][2])
    for kn, wg in keys[:24]:
        try:
            b_, w_, g_ = m[(kn, wg, "SQ_VALU_MFMA_BUSY_CYCLES")], m[(kn, wg, "SQ_WAVE_CYCLES")], m[(kn, wg, "GRBM_GUI_ACTIVE")]
        except KeyError:
            continue
        if w_[0] <= 0 or b_[0] <= 0:
            continue
        fh.write(f"| `{kn[:60]}` | {wg} | {w_[2]} | {w_[1]:.1f} | {b_[0] / w_[0]:.3f} | {100 * b_[0] / ((g_[0] / 8) * 1024):.1f} |\n")
print(json.dumps(d, indent=1))
print(bench["ms_per_step"], bench["value"], bench["roofline"])

# per-launch roofline tables (every kernel of one forward, matched to its operation and flops)
for wl in ("", "cfg1_single", "cfg4_d512", "cfg5_longform", "cfg5_longform_gaussian"):
    if os.path.exists(f"{G}/{ROUND}_trace{'_' + wl if wl else ''}/t_results.db"):
        subprocess.run(["python", "tools/roofline_table.py", ROUND] + ([wl] if wl else []))
