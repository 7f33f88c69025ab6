#!/bin/bash
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --config cfg5 --no-cpu-baseline --steps 80 > gpurun_out/r04x_$label.json 2> gpurun_out/r04x_$label.err || tail -3 gpurun_out/r04x_$label.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04x_$label.json"))
print("$label", d["value"], d["ms_per_step"], "serial", (d["config"].get("serial") or {}).get("ms_per_step"), (d["config"].get("serial") or {}).get("value"))
PY
}
run table1 X=1
run padded1 PPASR_BLOCK_TABLE=0
run table2 X=1
