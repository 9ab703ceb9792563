#!/usr/bin/env python3
"""Print the ordered kernel sequence of the LAST forward in a rocprofv3 --kernel-trace CSV (gap = idle time before the
kernel on the device timeline).   python tools/trace_seq.py gpurun_out/tr1/t_kernel_trace.csv [first-kernel-substring]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_embed_pos"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
lo = starts[-1]
seq = rows[lo:]
t_prev = None
tot = gaps = 0.0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("ns::", "").replace("void ", "")
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    gap = (s - t_prev) / 1e3 if t_prev else 0.0
    print(f"{(e - s) / 1e3:8.2f} us  gap {gap:6.2f}  wgs {wg:6d}  lds {r.get('LDS_Block_Size', '?'):>6}  vgpr {r.get('VGPR_Count', '?'):>4}  {name}")
    tot += (e - s) / 1e3
    gaps += gap
    t_prev = e
print(f"kernels {len(seq)}  kernel time {tot:.1f} us  gaps {gaps:.1f} us  span {(int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3:.1f} us")
