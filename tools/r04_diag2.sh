#!/bin/bash
# round-4 call 2: full GPU suite on the new attention kernel, cfg5 / cfg4 A/B (PPASR_ATTN_LEGACY), row-block tuning
R=$(pwd)
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r04_gpu_tests2.log 2>&1; tail -5 $R/gpurun_out/r04_gpu_tests2.log
for v in legacy new; do
  if [ $v = legacy ]; then export PPASR_ATTN_LEGACY=1; else unset PPASR_ATTN_LEGACY; fi
  for cfg in cfg5 cfg4; do
    timeout 300 python bench.py --config $cfg --no-cpu-baseline --steps 60 --warmup 5 > $R/gpurun_out/r04b_${cfg}_$v.json 2> $R/gpurun_out/r04b_${cfg}_$v.err
    timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-pipeline --steps 60 --warmup 5 > $R/gpurun_out/r04b_${cfg}_${v}_np.json 2>> $R/gpurun_out/r04b_${cfg}_$v.err
    python - <<PY
import json
for f in ("r04b_${cfg}_$v.json", "r04b_${cfg}_${v}_np.json"):
    try:
        d = json.load(open("$R/gpurun_out/" + f)); print(f, d["value"], d["ms_per_step"])
        if f.endswith("_np.json"):
            for k, c in d["roofline"]["classes"].items(): print("   ", k, c.get("ms_per_step"), c.get("frac"))
    except Exception as e: print(f, "FAILED", e)
PY
  done
done
unset PPASR_ATTN_LEGACY
timeout 400 python tools/r04_tune_rows.py > $R/gpurun_out/r04_tune_rows.txt 2>&1; cat $R/gpurun_out/r04_tune_rows.txt
