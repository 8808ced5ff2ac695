#!/bin/bash
# Round evidence, collected on the GPU box:
#   bash tools/collect_profiles.sh                                   -> gpurun_out/evidence/        (the default workload, with the bench line)
#   bash tools/collect_profiles.sh TAG --config CFG --batch B        -> gpurun_out/evidence_TAG/    (another configs/*.yaml workload)
# Three rocprofv3 passes of the same bench command: kernel stats, FETCH_SIZE, WRITE_SIZE (counters in passes of their own).
set -e
ROOT=$PWD
TAG=${1:-}
if [ -n "$TAG" ]; then shift; OUT=$ROOT/gpurun_out/evidence_$TAG; else OUT=$ROOT/gpurun_out/evidence; fi
ARGS=$(echo "$*" | sed "s#configs/#$ROOT/configs/#g")   # rocprofv3 runs from /tmp: absolute config path
rm -rf $OUT; mkdir -p $OUT
python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py $ARGS --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers > $OUT/pmc_fetch.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers > $OUT/pmc_write.log 2>&1
cd $ROOT
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_hbm_traffic_per_launch.json 3 > $OUT/pmc_traffic.txt
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats/*/*kernel_trace.csv $OUT/pmc_fetch/*/*kernel_trace.csv $OUT/pmc_write/*/*kernel_trace.csv
