#!/bin/bash
R=$(pwd)
for thr in 0 256 128 1024; do
  if [ $thr = 0 ]; then unset PPASR_BEAM_THREADS; else export PPASR_BEAM_THREADS=$thr; fi
  timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-pipeline --steps 30 --warmup 4 > $R/gpurun_out/r04k_cfg5_t$thr.json 2> $R/gpurun_out/r04k_cfg5_t$thr.err
  python - <<PY
import json
try:
    d = json.load(open("$R/gpurun_out/r04k_cfg5_t$thr.json")); c = d["roofline"]["classes"]
    print("cfg5 beam threads $thr", d["value"], d["ms_per_step"], {k: c[k].get("ms_per_step") for k in c if "beam" in k or "prune" in k})
except Exception as e: print("cfg5 $thr FAILED", e)
PY
done
