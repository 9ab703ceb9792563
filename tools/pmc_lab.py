#!/usr/bin/env python3
"""Per-(kernel, grid) averages of the counters collected by tools/pmc_lab.sh.  python tools/pmc_lab.py [substr]"""
import csv
import re
import sys
from collections import defaultdict

sub = sys.argv[1] if len(sys.argv) > 1 else "k_conv_gemm"
agg = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in ("a", "b"):
    for r in csv.DictReader(open(f"gpurun_out/pmc_lab_{d}/t_counter_collection.csv")):
        if sub not in r["Kernel_Name"]:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("ns::", "").replace("void ", "")
        key = (name, int(r["Grid_Size"]) // int(r["Workgroup_Size"]), r["LDS_Block_Size"], r["VGPR_Count"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, c in agg.items():
    a = {k: sum(v) / len(v) for k, v in c.items()}
    print(f"\n{key[0]}  wgs {key[1]} lds {key[2]} vgpr {key[3]}  n={len(c['GRBM_GUI_ACTIVE'])}  avg dur {sum(dur[key]) / max(1, len(dur[key])):.1f} us")
    gui = a.get("GRBM_GUI_ACTIVE", 0)
    if gui:
        d_us = sum(dur[key]) / len(dur[key])
        print(f"   clock {gui / 8 / d_us / 1e3:.2f} GHz (GRBM_GUI_ACTIVE/8 XCDs / dur)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            print(f"   MFMA busy {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / ((gui / 8) * 1024):.1f} % of (cycles x 1024 SIMDs)")
    wc = a.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in a:
                print(f"   {k:22s} {100 * a[k] / wc:5.1f} % of SQ_WAVE_CYCLES")
        if "SQ_BUSY_CYCLES" in a:
            print(f"   SQ_BUSY_CYCLES {a['SQ_BUSY_CYCLES']:.3g}  SQ_WAVE_CYCLES {wc:.3g}  waves-in-flight avg {wc / a['SQ_BUSY_CYCLES'] * 4 / 1:.2f} (x4 quad-cycle units?)")
    for k in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVES"):
        if k in a:
            print(f"   {k:28s} {a[k]:.4g}")
