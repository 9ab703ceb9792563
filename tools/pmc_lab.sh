#!/bin/bash
# PMC counters for a lab binary ON THE GPU BOX:  gpurun -- 'bash tools/pmc_lab.sh ./tools/lab/gemm_lab_ln 1'
# (separate passes; --kernel-trace only, as the pool requires).  CSVs land in gpurun_out/pmc_lab_*/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
BIN=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $ROOT/gpurun_out/pmc_lab_a -o t -- $ROOT/$BIN "$@" > $ROOT/gpurun_out/pmc_lab_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES \
  --output-format csv -d $ROOT/gpurun_out/pmc_lab_b -o t -- $ROOT/$BIN "$@" > $ROOT/gpurun_out/pmc_lab_b.log 2>&1
ls $ROOT/gpurun_out/pmc_lab_a $ROOT/gpurun_out/pmc_lab_b
