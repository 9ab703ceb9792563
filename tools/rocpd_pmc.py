#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite file (ROCm 7.2), grouped by kernel and grid.

    python tools/rocpd_pmc.py gpurun_out/r01_fetch/t_results.db [min_grid]
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("ns::", "")[:60]


def summarise(path, min_calls=1):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, grid_size_x, workgroup_size_x, counter_name, value, dispatch_id, duration "
                     "from counters_collection").fetchall()
    # a counter may appear once per hardware instance (XCD/SE): sum them per dispatch first
    per = defaultdict(float)
    dur = {}
    for k, g, w, cn, v, did, d in rows:
        per[(short(k), g // max(w, 1), cn, did)] += v
        dur[(short(k), g // max(w, 1), did)] = d
    agg = defaultdict(lambda: [0, 0.0])
    for (k, g, cn, did), v in per.items():
        a = agg[(k, g, cn)]
        a[0] += 1
        a[1] += v
    dagg = defaultdict(lambda: [0, 0.0])
    for (k, g, did), d in dur.items():
        a = dagg[(k, g)]
        a[0] += 1
        a[1] += d
    return agg, dagg


if __name__ == "__main__":
    agg, dagg = summarise(sys.argv[1])
    print(f"# per-dispatch counter averages from {sys.argv[1]}\n")
    print("| kernel | workgroups | calls | avg dur us | counter | avg per dispatch |")
    print("|---|---:|---:|---:|---|---:|")
    for (k, g, cn), a in sorted(agg.items(), key=lambda kv: (-dagg[(kv[0][0], kv[0][1])][1], kv[0][2])):
        d = dagg[(k, g)]
        if d[1] / 1e3 < 200:  # skip kernels with < 0.2 ms total
            continue
        print(f"| `{k}` | {g} | {a[0]} | {d[1] / d[0] / 1e3:.1f} | {cn} | {a[1] / a[0]:.1f} |")
