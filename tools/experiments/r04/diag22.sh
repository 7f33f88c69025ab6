#!/bin/bash
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --config cfg5 --no-cpu-baseline --steps 60 > gpurun_out/r04t_$label.json 2> gpurun_out/r04t_$label.err || tail -3 gpurun_out/r04t_$label.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04t_$label.json"))
print("$label", d["value"], d["ms_per_step"], "serial", (d["config"].get("serial") or {}).get("ms_per_step"))
PY
}
run plain PPASR_CU_PARTITION=0
run se2pair PPASR_CU_PARTITION=16 PPASR_CU_DEC_BITS=16-23,48-55
run se3pair PPASR_CU_PARTITION=16 PPASR_CU_DEC_BITS=24-31,56-63
run se0pair PPASR_CU_PARTITION=16 PPASR_CU_DEC_BITS=0-7,32-39
run se23 PPASR_CU_PARTITION=16 PPASR_CU_DEC_BITS=16-31
