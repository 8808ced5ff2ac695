#!/bin/bash
# Hardware counters of the GEMM micro-benchmarks, one rocprofv3 --pmc pass per counter group (kernel trace only).
# usage (GPU box): [PMC_GROUPS="A B|C D"] bash tools/pmc_sq.sh [gemm_bench args]   -> gpurun_out/pmc_sq/passN/
set -e
OUT=$PWD/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
GROUPS_=${PMC_GROUPS:-"SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES|SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS|SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS GRBM_GUI_ACTIVE"}
i=0
IFS='|' read -ra GRPS <<< "$GROUPS_"
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/gemm_bench.py --rounds 2 "$@" > $OUT/pass$i.log 2>&1
done
