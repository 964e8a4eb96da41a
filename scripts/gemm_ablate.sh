#!/bin/bash
# Timing-only ablation builds of the direct-to-LDS GEMM (WRONG results by construction): `make variant NAME=x VSRCS=gemm.hip EXTRA=-DMD_ABL_...`.
# usage: gemm_ablate.sh "N LAYOUT [REPS]" NAME...      ("base" = the main build); prints gemm_clock.py's two lines per variant
spec=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset MDHIP_LIB_VARIANT; else export MDHIP_LIB_VARIANT=$v; fi   # (the product file is never swapped)
  echo "== $v"
  timeout -k 10 120 python3 scripts/gemm_clock.py $spec 2>&1 | tail -2
done
