#!/bin/bash
# round-4 call 5: attention defaults (32-key sub-blocks, 4 waves / SIMD, early K' requests, boustrophedon XCD map)
R=$(pwd)
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $R/gpurun_out/r04_gpu_tests5.log 2>&1; tail -4 $R/gpurun_out/r04_gpu_tests5.log
for cfg in cfg5 cfg4; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --steps 60 --warmup 5 > $R/gpurun_out/r04e_$cfg.json 2> $R/gpurun_out/r04e_$cfg.err
  python - <<PY
import json
try:
    d = json.load(open("$R/gpurun_out/r04e_$cfg.json")); print("$cfg", d["value"], d["ms_per_step"], d["config"].get("serial"))
    for k, c in d["roofline"]["classes"].items(): print("   ", k, c.get("ms_per_step"), c.get("frac"))
except Exception as e: print("$cfg FAILED", e)
PY
done
PPASR_ATTN_FUSE_MIN_BLOCKS=100000 timeout 300 python bench.py --config cfg2 --no-cpu-baseline --steps 60 --warmup 5 > $R/gpurun_out/r04e_cfg2two.json 2> $R/gpurun_out/r04e_cfg2two.err
python - <<PY
import json
d = json.load(open("$R/gpurun_out/r04e_cfg2two.json")); c = d["roofline"]["classes"]
print("cfg2 two-kernel route", d["value"], d["ms_per_step"], {k: (c[k].get("ms_per_step"), c[k].get("frac")) for k in c if "attention" in k or "out_glu" in k})
PY
timeout 300 python tools/bench_ds2.py > $R/gpurun_out/r04e_ds2.txt 2>&1; tail -30 $R/gpurun_out/r04e_ds2.txt
