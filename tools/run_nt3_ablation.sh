#!/bin/bash
# phase ablation of the NT GEMM K-loop (v2 and v3 side by side): alternate builds with the LDS-DMA and / or the epilogue compiled out
# usage: [VARIANTS="full nt3_s1 ..."] tools/run_nt3_ablation.sh OUT [gemm_bench args]   (builds: make EXTRA="-DMAE_DBG_NO_DMA ..." OBJDIR=../build_dbg_X LIBDIR=../lib_dbg_X)
out=$1; shift
: > $out
for v in ${VARIANTS:-full no_epi no_dma no_dma_epi}; do
  if [ "$v" = full ]; then unset MAE_HIP_LIB; else export MAE_HIP_LIB=$PWD/ssrl_vit_mae_jepa_amd/lib_dbg_$v/libmae_hip.so; fi
  echo "=== variant $v" >> $out
  timeout -k 10 300 python tools/gemm_bench.py --rounds 5 --variants v2,v3 "$@" 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
