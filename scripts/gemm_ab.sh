#!/bin/bash
# A/B of libmdhip variants (scripts/ab/libmdhip_NAME.so, `make variant`) on the GEMM micro-benchmark, one box, two passes over the
# variants (A B C A B C) so that drift shows. usage: gemm_ab.sh "SHAPES" NAME...   ("base" = the main build)
shapes=$1; shift
for pass in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset MDHIP_LIB_VARIANT; else export MDHIP_LIB_VARIANT=$v; fi   # (the product file is never swapped)
    echo "== $v (pass $pass)"
    timeout -k 10 200 python3 scripts/gemm_bench.py $shapes 2>&1 | grep -v "^$"
  done
done
