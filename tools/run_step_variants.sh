#!/bin/bash
# whole-step A/B: tools/run_step_variants.sh SPEC...   SPEC = LIB[,VAR=VAL]...  (LIB "full" = the product library,
# otherwise ssrl_vit_mae_jepa_amd/lib_dbg_LIB/ built with `make EXTRA=... OBJDIR=../build_dbg_LIB LIBDIR=../lib_dbg_LIB`)
set -e
mkdir -p gpurun_out
for spec in "$@"; do
  IFS=',' read -ra parts <<< "$spec"
  v=${parts[0]}
  envs=("${parts[@]:1}")
  if [ "$v" = full ]; then unset MAE_HIP_LIB; else export MAE_HIP_LIB=$PWD/ssrl_vit_mae_jepa_amd/lib_dbg_$v/libmae_hip.so; fi
  env "${envs[@]}" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$spec', round(d['ms_per_step'],3), 'ms', round(d['value']), 'img/s loss', round(d['final_loss'],6), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()})
" >> gpurun_out/step_variants.log
done
