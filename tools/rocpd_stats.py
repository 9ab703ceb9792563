#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as a per-kernel stats table — the same
numbers `rocprofv3 --stats` reports — so the summary can be committed under profiles/ as text.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("ns::", "")[:90]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"# kernel stats from {path.split('/')[-1]} (durations in microseconds)\n")
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{short(n)}` | {a[0]} | {a[1]:.1f} | {a[1] / a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100 * a[1] / total:.1f} |")
    print(f"\ntotal kernel time {total / 1e3:.2f} ms over {sum(a[0] for a in agg.values())} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
