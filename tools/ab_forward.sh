#!/bin/bash
# Same-box comparison of several builds of the library, the way the tile rules and epilogue changes of round 2 were
# settled (the back-to-back kernel lab got some of them wrong: it keeps weights and residual rows hotter than a forward does).
#
#   # here: build variants, e.g. hipcc ... -DNS_LAB_X -c gemm_conv.hip -o /tmp/x.o; link as scratch/libnarfs2_vX.so
#   gpurun -- 'bash tools/ab_forward.sh "scratch/libnarfs2_v0.so scratch/libnarfs2_vX.so" [bench.py args]'
#
# Each round runs bench.py --no-extras once per build, in the given order; 3-5 rounds separate 0.1 % of a config-2 step.
LIBS=$1; shift
ROUNDS=${AB_ROUNDS:-4}
DST=smart-nar_fast_tts_amd/csrc/libnarfs2.so
cp $DST /tmp/libnarfs2_keep.so
for i in $(seq $ROUNDS); do for lib in $LIBS; do
  cp $lib $DST
  python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline_by_kernel']
print('$lib', round(d['ms_per_step'], 3), 'ms/step   w_1', round(k['ffn_w1']['avg_launch_ms'] * 1e3, 1), 'us   attention',
      round(k['attention']['avg_launch_ms'] * 1e3, 1), 'us   postnet', round(k['postnet_mid']['avg_launch_ms'] * 1e3, 1), 'us')"
done; done
cp /tmp/libnarfs2_keep.so $DST
