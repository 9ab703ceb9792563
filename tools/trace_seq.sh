cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr1 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload cfg1_single --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tr1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tr2.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/tr1 $GRAFT_REPO_ROOT/gpurun_out/tr2
