#!/bin/bash
# Collect one round's rocprofv3 evidence ON THE GPU BOX (run through gpurun), then summarise it here with
# `python tools/make_profiles.py rNN`:
#
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r02'
#
# Per workload one kernel-trace pass; for config 2 and config 5 additionally one PMC pass per counter group (never
# combined with a trace domain other than --kernel-trace; FETCH_SIZE and WRITE_SIZE in separate passes as
# MI355X_MICROARCH.md prescribes).  bench.py runs with --no-extras under the profiler: the trace then holds exactly
# (warmup + steps) forwards of the workload, nothing else.
set -u
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

python "$ROOT/bench.py" > "$OUT/${R}_bench.log" 2>&1
tail -1 "$OUT/${R}_bench.log" > "$OUT/${R}_bench.json"

for WL in cfg2_b16 cfg1_single cfg4_d512 cfg5_longform cfg5_longform_gaussian; do
  TAG=${R}_trace; [ "$WL" != cfg2_b16 ] && TAG=${R}_trace_${WL}
  BENCH="python $ROOT/bench.py --workload $WL --steps 5 --warmup 2 --no-extras"
  rocprofv3 --kernel-trace --stats --output-format rocpd csv -d "$OUT/$TAG" -o t -- $BENCH > "$OUT/$TAG.log" 2>&1
  grep '^{' "$OUT/$TAG.log" | tail -1 > "$OUT/${TAG/_trace/_bench_under_trace}.json"
done
# variable-length batches: phase 2 on packed rows (include/nar_fs2.h ns_forward_mel_packed) and, for comparison, on the padded grid
for WL in cfg2_b16 cfg5_longform; do
  for MODE in packed grid; do
    TAG=${R}_trace_${WL}_ragged_${MODE}
    PK=1; [ "$MODE" = grid ] && PK=0
    NS_PACKED=$PK rocprofv3 --kernel-trace --stats --output-format rocpd csv -d "$OUT/$TAG" -o t -- python $ROOT/bench.py --workload $WL --ragged --steps 5 --warmup 2 --no-extras > "$OUT/$TAG.log" 2>&1
    grep '^{' "$OUT/$TAG.log" | tail -1 > "$OUT/${TAG/_trace/_bench_under_trace}.json"
  done
done
# the launch planner's acceptance sweep: forward time against the batch size, this build and (NS_PLAN=0) round 3's one-tile rules, same box
python "$ROOT/tools/batch_sweep.py" > "$OUT/${R}_batch_size_sweep.txt" 2> "$OUT/${R}_batch_size_sweep.err"
NS_PLAN=0 python "$ROOT/tools/batch_sweep.py" > "$OUT/${R}_batch_size_sweep_plan0.txt" 2>> "$OUT/${R}_batch_size_sweep.err"
python "$ROOT/tools/batch_sweep.py" --ragged --batches 4,8,12,16,24,32 > "$OUT/${R}_batch_size_sweep_ragged.txt" 2>> "$OUT/${R}_batch_size_sweep.err"
NS_PACKED=0 python "$ROOT/tools/batch_sweep.py" --ragged --batches 4,8,12,16,24,32 > "$OUT/${R}_batch_size_sweep_ragged_grid.txt" 2>> "$OUT/${R}_batch_size_sweep.err"
# per-launch view of batch sizes that sit between steps (B = 9, 17, 20)
bash "$ROOT/tools/trace_batches.sh" ${R}_tb 9 17 20 > "$OUT/${R}_tb.log" 2>&1
# the 16-row tile family off / on, alternating processes on this box (round 5): uniform batches B = 4 ... 32, ragged batches
( cd "$ROOT" && tools/ab_sweep.sh "$OUT/${R}_ab_tile16" "NS_TILE16=0" "NS_TILE16=1" > "$OUT/${R}_ab_tile16.txt" 2>&1 )
( cd "$ROOT" && tools/ab_sweep.sh "$OUT/${R}_ab_tile16_ragged" "NS_TILE16=0" "NS_TILE16=1" 4,8,9,12,16,20,24,32 --ragged > "$OUT/${R}_ab_tile16_ragged.txt" 2>&1 )
# phase 1 on packed phoneme rows against the grid (ragged batches, src_lens on the host)
python "$ROOT/tools/phase1_packing_ab.py" > "$OUT/${R}_phase1_packing_ab.txt" 2>&1
cd /tmp
# rocprofv3's own per-kernel summary of the config-2 run, as it wrote it
find "$OUT/${R}_trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/${R}_rocprofv3_kernel_stats.csv" \;

for WL in cfg2_b16 cfg5_longform; do
  SUF=""; [ "$WL" != cfg2_b16 ] && SUF=_${WL}
  BENCH="python $ROOT/bench.py --workload $WL --steps 5 --warmup 2 --no-extras"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/${R}_fetch$SUF" -o t -- $BENCH > "$OUT/${R}_fetch$SUF.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/${R}_write$SUF" -o t -- $BENCH > "$OUT/${R}_write$SUF.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/${R}_mfma$SUF" -o t -- $BENCH > "$OUT/${R}_mfma$SUF.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    -d "$OUT/${R}_lds$SUF" -o t -- $BENCH > "$OUT/${R}_lds$SUF.log" 2>&1
done
# single utterance: fabric fetch per launch (are the small-grid GEMMs re-streaming weights? tools/lab/README.md round 3)
BENCH="python $ROOT/bench.py --workload cfg1_single --steps 5 --warmup 2 --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/${R}_fetch_cfg1_single" -o t -- $BENCH > "$OUT/${R}_fetch_cfg1_single.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/${R}_write_cfg1_single" -o t -- $BENCH > "$OUT/${R}_write_cfg1_single.log" 2>&1
# the opt-in bf16x3 mode, for the record (never the headline)
python "$ROOT/bench.py" --matmul bf16x3 --no-cpu-baseline > "$OUT/${R}_bench_bf16x3.log" 2>&1
tail -1 "$OUT/${R}_bench_bf16x3.log" > "$OUT/${R}_bench_bf16x3.json"
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d "$OUT/${R}_trace_bf16x3" -o t -- python $ROOT/bench.py --matmul bf16x3 --steps 5 --warmup 2 --no-extras > "$OUT/${R}_trace_bf16x3.log" 2>&1
# the long randomized parity run (the -m gpu suite holds a time-boxed slice of it: tests/test_gpu_stress.py)
python "$ROOT/tests/fuzz_gpu.py" --iters 360 --seed 41 > "$OUT/${R}_fuzz.txt" 2>&1
python "$ROOT/tests/fuzz_gpu.py" --iters 40 --seed 42 --big >> "$OUT/${R}_fuzz.txt" 2>&1
# forwards in flight on several streams (capacity mode): independent B=1 requests, and the batch-size staircase as throughput
# (the committed profiles/rNN_multi_stream_*.txt are these outputs plus a "Reading:" paragraph)
python "$ROOT/tools/multi_stream_small.py" > "$OUT/${R}_multi_stream_small.raw.txt" 2>&1
python "$ROOT/tools/multi_stream_small.py" --batches 8,9,12,16,17,20 --phonemes 128 --streams 1,2,3 --n 100 > "$OUT/${R}_multi_stream_batches.raw.txt" 2>&1
ls -d "$OUT"/${R}_*/ | head -40
cat "$OUT/${R}_bench.json" | cut -c1-300
