#!/bin/bash
# fused front end: bit-identity test, full GPU suite, benches with PPASR_CONV12 on / off on one box
python -m pytest tests/test_front_fused_gpu.py tests/test_row_block_gpu.py -q -x 2>&1 | tail -6
python -m pytest tests -m gpu -q -x --deselect tests/test_row_block_gpu.py --deselect tests/test_front_fused_gpu.py 2>&1 | tail -5
for c in cfg2 cfg4 cfg5; do
 for f in 1 0; do
  PPASR_CONV12=$f python bench.py --config $c --no-cpu-baseline --steps 60 > gpurun_out/r04p_${c}_$f.json 2> gpurun_out/r04p_${c}_$f.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04p_${c}_$f.json"))
c=d["roofline"]["classes"]
print("$c conv12=$f", d["value"], d["ms_per_step"], (d["config"].get("serial") or {}).get("ms_per_step"), {k: round(v["ms_per_step"],4) for k,v in c.items() if k in ("conv2","k_conv1","k_conv12","dense")}, [k for k in c if "conv" in k])
PY
 done
done
