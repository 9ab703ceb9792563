#!/bin/bash
# Same-box A/B of two launch-plan settings over batch sizes: alternating processes (tools/batch_sweep.py), so that box-to-box and
# clock-state differences cancel.   tools/ab_sweep.sh <out-prefix> "<env A>" "<env B>" [batch list] [extra batch_sweep args]
#   tools/ab_sweep.sh gpurun_out/ab_tile16 "NS_TILE16=0" "NS_TILE16=1" 4,5,6,7,8,9,10,11,12
out=$1; A=$2; B=$3; list=${4:-4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32}; shift 3; if [ $# -gt 0 ]; then shift; fi
for rep in 1 2; do
  env $A python tools/batch_sweep.py --batches $list "$@" > ${out}_A${rep}.txt 2>&1
  env $B python tools/batch_sweep.py --batches $list "$@" > ${out}_B${rep}.txt 2>&1
done
python - "$out" "$A" "$B" <<'PY'
import re, sys
out, A, B = sys.argv[1:4]
def load(f):
    d = {}
    for l in open(f):
        m = re.match(r"B=\s*(\d+) T\s+(\d+) rows\s+(\d+):\s+([\d.]+) ms", l)
        if m: d[int(m.group(1))] = float(m.group(4))
    return d
a = [load(f"{out}_A{r}.txt") for r in (1, 2)]
b = [load(f"{out}_B{r}.txt") for r in (1, 2)]
print(f"# A = {A}   B = {B}   (ms per forward, best of two alternating runs each)")
for k in sorted(a[0]):
    ta, tb = min(x[k] for x in a if k in x), min(x[k] for x in b if k in x)
    print(f"B={k:2d}  A {ta:7.3f}  B {tb:7.3f}  B/A {100 * (tb / ta - 1):+6.1f} %")
PY
