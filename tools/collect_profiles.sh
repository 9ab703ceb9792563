#!/bin/bash
# Collect one round's rocprofv3 evidence ON THE GPU BOX (run through gpurun), then summarise it here with
# `python tools/make_profiles.py rNN`:
#
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
#
# One kernel-trace pass plus one PMC pass per counter group (never combined with a trace domain other than
# --kernel-trace; FETCH_SIZE and WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes).
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"

python "$ROOT/bench.py" > "$OUT/${R}_bench.log" 2>&1
tail -1 "$OUT/${R}_bench.log" > "$OUT/${R}_bench.json"

rocprofv3 --kernel-trace --stats --output-format rocpd csv -d "$OUT/${R}_trace" -o t -- $BENCH > "$OUT/${R}_trace.log" 2>&1
grep '^{' "$OUT/${R}_trace.log" | tail -1 > "$OUT/${R}_bench_under_trace.json"
# rocprofv3's own per-kernel summary, as it wrote it (committed next to the per-geometry table make_profiles.py derives)
find "$OUT/${R}_trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/${R}_rocprofv3_kernel_stats.csv" \;

rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/${R}_fetch" -o t -- $BENCH > "$OUT/${R}_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/${R}_write" -o t -- $BENCH > "$OUT/${R}_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  -d "$OUT/${R}_mfma" -o t -- $BENCH > "$OUT/${R}_mfma.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  -d "$OUT/${R}_lds" -o t -- $BENCH > "$OUT/${R}_lds.log" 2>&1
ls -la "$OUT"/${R}_*/ | head -40
cat "$OUT/${R}_bench.json"
