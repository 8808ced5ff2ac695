#!/bin/bash
# kernel-phase ablation of the NT GEMM: the same bench against alternate builds of the library (see csrc/Makefile EXTRA=)
# usage: tools/run_dbg_variants.sh VARIANT...   ("full" = the product library)
set -e
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = full ]; then unset MAE_HIP_LIB; else export MAE_HIP_LIB=$PWD/ssrl_vit_mae_jepa_amd/lib_dbg_$v/libmae_hip.so; fi
  echo "=== variant $v" >> gpurun_out/dbg_variants.log
  timeout -k 10 300 python tools/gemm_bench.py --variants v2 --rounds 5 $GB_ARGS 2>&1 | grep -v amdgpu.ids >> gpurun_out/dbg_variants.log
done
