#!/bin/bash
# SQ-level counters (LDS / MFMA busy) of the GEMM micro-benchmarks, one rocprofv3 --pmc pass per counter group.
# usage (GPU box): bash tools/pmc_sq.sh [gemm_bench args]   -> gpurun_out/pmc_sq/passN/
set -e
OUT=$PWD/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/gemm_bench.py --rounds 2 "$@" > $OUT/pass$i.log 2>&1
done
