#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as a per-kernel stats table — the same
numbers `rocprofv3 --stats` reports, additionally split by launch geometry (workgroup count) so that one shape of
a templated kernel (e.g. the decoder's FFN k=9 GEMM) can be read off — and print it as markdown for profiles/.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("ns::", "")[:70]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, start, end from kernels").fetchall()
    agg = {}
    for n, gx, gy, gz, wx, wy, wz, s, e in rows:
        wgs = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
        a = agg.setdefault((short(n), wgs), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"# kernel stats from rocprofv3 --kernel-trace ({path.split('/')[-2]}/{path.split('/')[-1]}; durations in microseconds)\n")
    print("| kernel | workgroups | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for (n, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if a[1] / total < 0.001:
            continue
        print(f"| `{n}` | {g} | {a[0]} | {a[1]:.1f} | {a[1] / a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100 * a[1] / total:.1f} |")
    print(f"\ntotal kernel time {total / 1e3:.2f} ms over {sum(a[0] for a in agg.values())} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
