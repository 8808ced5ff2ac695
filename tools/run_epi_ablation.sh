#!/bin/bash
# epilogue ablation of the NT GEMM (fc1 + GELU forward): alternate builds of the library (csrc/Makefile EXTRA=-DMAE_DBG_EPI_*)
# usage: tools/run_epi_ablation.sh OUT VARIANT...   ("full" = the product library); GB_ARGS = extra gemm_bench.py arguments
out=$1; shift
: > $out
for v in "$@"; do
  if [ "$v" = full ]; then unset MAE_HIP_LIB; else export MAE_HIP_LIB=$PWD/ssrl_vit_mae_jepa_amd/lib_dbg_$v/libmae_hip.so; fi
  echo "=== variant $v" >> $out
  timeout -k 10 300 python tools/gemm_bench.py --rounds 7 $GB_ARGS 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
