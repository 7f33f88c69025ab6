#!/bin/bash
python -m pytest tests/test_ragged_gpu.py tests/test_row_block_gpu.py tests/test_squeezeformer_gpu.py -q -x 2>&1 | tail -3
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --config cfg5 --no-cpu-baseline --steps 80 > gpurun_out/r04w_$label.json 2> gpurun_out/r04w_$label.err || tail -3 gpurun_out/r04w_$label.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04w_$label.json"))
print("$label", d["value"], d["ms_per_step"], "serial", (d["config"].get("serial") or {}).get("ms_per_step"), (d["config"].get("serial") or {}).get("value"))
PY
}
run table1 X=1
run padded1 PPASR_BLOCK_TABLE=0
run table2 X=1
run padded2 PPASR_BLOCK_TABLE=0
