#!/bin/bash
R=$(pwd)
for v in default wavePF2 wavePF8; do
  if [ $v = default ]; then unset PPASR_HIP_LIB; else export PPASR_HIP_LIB=$R/tools/_ts/lib_$v.so; fi
  echo "== $v"
  timeout 400 python tools/bench_ds2.py uni 2>&1 | grep "unidirectional" | cut -c1-260
done
