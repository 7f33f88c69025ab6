#!/bin/bash
# round-4 call 3: full GPU suite (all failures), attention key-split variants, conv2 tile table, cfg5 / cfg4 lines
R=$(pwd)
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q > $R/gpurun_out/r04_gpu_tests3.log 2>&1; tail -8 $R/gpurun_out/r04_gpu_tests3.log
for ns in 0 1 2 4; do
  if [ $ns = 0 ]; then unset PPASR_ATTN_SPLIT; else export PPASR_ATTN_SPLIT=$ns; fi
  for cfg in cfg5 cfg4; do
    timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-pipeline --steps 40 --warmup 5 > $R/gpurun_out/r04c_${cfg}_ns$ns.json 2> $R/gpurun_out/r04c_${cfg}_ns$ns.err
    python - <<PY
import json
try:
    d = json.load(open("$R/gpurun_out/r04c_${cfg}_ns$ns.json")); c = d["roofline"]["classes"]
    print("$cfg ns=$ns", d["value"], d["ms_per_step"], {k: c[k].get("ms_per_step") for k in c if "attention" in k or k == "conv2"})
except Exception as e: print("$cfg ns=$ns FAILED", e)
PY
  done
done
unset PPASR_ATTN_SPLIT
for cfg in cfg5 cfg4 cfg2; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline > $R/gpurun_out/r04c_${cfg}.json 2> $R/gpurun_out/r04c_${cfg}.err
  python - <<PY
import json
try:
    d = json.load(open("$R/gpurun_out/r04c_${cfg}.json")); print("$cfg", d["value"], d["ms_per_step"])
    for k, c in d["roofline"]["classes"].items(): print("   ", k, c.get("ms_per_step"), c.get("frac"))
except Exception as e: print("$cfg FAILED", e)
PY
done
