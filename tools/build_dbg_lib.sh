#!/bin/bash
# Alternate builds of libmae_hip.so with a debug macro: tools/build_dbg_lib.sh vmcnt0 epi_nogelu ...  -> ssrl_vit_mae_jepa_amd/lib_dbg_<name>/
# (selected at run time with MAE_HIP_LIB=<path>; the product library is untouched)
set -e
cd "$(dirname "$0")/../ssrl_vit_mae_jepa_amd/csrc"
for v in "$@"; do
  up=$(echo "$v" | tr a-z A-Z)
  make -j"${JOBS:-8}" EXTRA=-DMAE_DBG_$up OBJDIR=../build_dbg_$v LIBDIR=../lib_dbg_$v > /dev/null
  echo "built lib_dbg_$v"
done
