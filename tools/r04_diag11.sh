#!/bin/bash
R=$(pwd)
for rep in 1 2; do
for v in default attnNW8 attnNW2; do
  if [ $v = default ]; then unset PPASR_HIP_LIB; else export PPASR_HIP_LIB=$R/tools/_ts/lib_$v.so; fi
  for cfg in cfg5; do
    timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-pipeline --steps 40 --warmup 5 > $R/gpurun_out/r04j_${cfg}_$v.json 2> $R/gpurun_out/r04j_${cfg}_$v.err
    python - <<PY
import json
try:
    d = json.load(open("$R/gpurun_out/r04j_${cfg}_$v.json")); c = d["roofline"]["classes"]
    print("$cfg $v rep$rep", d["value"], d["ms_per_step"], {k: (c[k].get("ms_per_step"), c[k].get("frac")) for k in c if "attention" in k})
except Exception as e: print("$cfg $v FAILED", e)
PY
  done
done
done
