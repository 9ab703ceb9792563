#!/bin/bash
# kernel traces of the forward at several batch sizes (launch planner work): gpurun -- 'bash tools/trace_batches.sh TAG 5 9 12 20'
set -u
TAG=${1:-tb}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_b$B" -o t -- python "$ROOT/bench.py" --batch "$B" --steps 3 --warmup 2 --no-extras > "$OUT/${TAG}_b$B.log" 2>&1
  python "$ROOT/tools/trace_seq.py" "$(find "$OUT/${TAG}_b$B" -name 't_kernel_trace.csv' | head -1)" > "$OUT/${TAG}_b$B.seq.txt" 2>&1
  tail -1 "$OUT/${TAG}_b$B.seq.txt"
done
