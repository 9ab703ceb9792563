"""Overlap of kernels from different HIP streams, from a rocprofv3 kernel trace (CSV) of tools/multi_stream_small.py.

    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/multi_stream_small.py --batches 9 --phonemes 128 --streams 3 --n 60
    python tools/inflight_overlap.py OUT [label]

Prints, over the last 60 % of the trace (the timed forwards): wall time covered by at least one kernel, by at least two, the sum
of kernel durations, and the same per kernel name for the launches that ran beside another stream's kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else root
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no kernel_trace.csv under {root}")
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    t_lo = rows[0][0] + int(0.4 * (rows[-1][1] - rows[0][0]))
    rows = [r for r in rows if r[0] >= t_lo]
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, cover = 0, ev[0][0], defaultdict(int)
    for t, d in ev:
        cover[min(depth, 3)] += t - last
        last = t
        depth += d
    span = rows[-1][1] - rows[0][0]
    ksum = sum(e - s for s, e, _, _ in rows)
    queues = sorted({q for _, _, _, q in rows})
    print(f"## {label}")
    print(f"{len(rows)} launches on {len(queues)} hardware queue(s) over {span / 1e6:.2f} ms; sum of kernel durations {ksum / 1e6:.2f} ms ({ksum / span:.2f} x the span)")
    print(f"time with no kernel running {cover[0] / span:.1%}, exactly one {cover[1] / span:.1%}, two {cover[2] / span:.1%}, three or more {cover[3] / span:.1%}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
