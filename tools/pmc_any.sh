#!/bin/bash
# rocprofv3 counter passes over an arbitrary python tool:  PMC_GROUPS="A B|C D" bash tools/pmc_any.sh tools/attn_bench.py [args]
set -e
OUT=$PWD/gpurun_out/pmc_any
rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
IFS='|' read -ra GRPS <<< "$PMC_GROUPS"
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $SCRIPT "$@" > $OUT/pass$i.log 2>&1
done
