#!/bin/bash
R=$(pwd)
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $R/gpurun_out/r04_gpu_tests14.log 2>&1; tail -3 $R/gpurun_out/r04_gpu_tests14.log
for cfg in cfg5 cfg4; do
  for rep in 1 2; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --steps 80 --warmup 5 > $R/gpurun_out/r04m_$cfg.json 2> $R/gpurun_out/r04m_$cfg.err
  python - <<PY
import json
try:
    d = json.load(open("$R/gpurun_out/r04m_$cfg.json")); print("$cfg rep$rep", d["value"], d["ms_per_step"], d["config"].get("serial"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["whole_path_tflops_per_gpu"])
except Exception as e: print("$cfg FAILED", e)
PY
  done
done
