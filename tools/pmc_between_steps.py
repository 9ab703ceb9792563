#!/usr/bin/env python3
"""Matrix-pipe busy share per kernel at batch sizes BETWEEN the steps (B = 9, 17), from one rocprofv3 --pmc pass each:

    gpurun -- 'cd /tmp && for B in 9 17; do rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
        -d $GRAFT_REPO_ROOT/gpurun_out/r05_mfma_b$B -o t -- python $GRAFT_REPO_ROOT/bench.py --batch $B --steps 5 --warmup 2 --no-extras; done'
    python tools/pmc_between_steps.py r05 9 17 >> profiles/r05_pmc.md
"""
import sys

sys.path.insert(0, "tools")
from rocpd_pmc import summarise  # noqa: E402

rnd = sys.argv[1]
for B in sys.argv[2:]:
    agg, dagg = summarise(f"gpurun_out/{rnd}_mfma_b{B}/t_results.db")
    print(f"\n## derived: matrix-pipe busy share per kernel at B = {B} x 1010 rows (the 16-row tile family's launches), SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES\n")
    print("| kernel | workgroups | launches | avg us | MFMA busy / wave cycles | MFMA util % (busy / (GUI_ACTIVE/8 x 1024)) |\n|---|---:|---:|---:|---:|---:|")
    keys = sorted({(k, g) for (k, g, cn) in agg if cn == "SQ_WAVE_CYCLES"}, key=lambda kg: -dagg[kg][1])
    for k, g in keys[:16]:
        try:
            b_, w_, g_ = agg[(k, g, "SQ_VALU_MFMA_BUSY_CYCLES")], agg[(k, g, "SQ_WAVE_CYCLES")], agg[(k, g, "GRBM_GUI_ACTIVE")]
        except KeyError:
            continue
        if w_[1] <= 0 or b_[1] <= 0:
            continue
        d = dagg[(k, g)]
        print(f"| `{k}` | {g} | {w_[0]} | {d[1] / d[0] / 1e3:.1f} | {b_[1] / w_[1]:.3f} | {100 * (b_[1] / b_[0]) / ((g_[1] / g_[0] / 8) * 1024):.1f} |")
