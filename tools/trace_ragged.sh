#!/bin/bash
# kernel trace of the ragged config-2 batch on packed rows (and on the grid): gpurun -- 'bash tools/trace_ragged.sh TAG'
set -u
TAG=${1:-rg}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for MODE in packed grid; do
  PK=1; [ "$MODE" = grid ] && PK=0
  NS_PACKED=$PK rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_$MODE" -o t -- python "$ROOT/bench.py" --ragged --steps 3 --warmup 2 --no-extras > "$OUT/${TAG}_$MODE.log" 2>&1
  python "$ROOT/tools/trace_seq.py" "$(find "$OUT/${TAG}_$MODE" -name 't_kernel_trace.csv' | head -1)" > "$OUT/${TAG}_$MODE.seq.txt" 2>&1
  tail -1 "$OUT/${TAG}_$MODE.seq.txt"
done
