#!/bin/bash
# cfg5: 8-tap chunks in the 32-row tail's depthwise conv (161 registers: can share a CU with the beam search) vs 16 (217)
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --config cfg5 --no-cpu-baseline --steps 80 > gpurun_out/r04q_$label.json 2> gpurun_out/r04q_$label.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04q_$label.json"))
c=d["roofline"]["classes"]
print("$label", d["value"], d["ms_per_step"], "serial", (d["config"].get("serial") or {}).get("ms_per_step"), " tail %.3f mid %.3f" % (c["k_sq_tail<31>"]["ms_per_step"], c["k_sq_mid"]["ms_per_step"]))
PY
}
for r in 1 2 3; do
run base$r X=1
run tc8_$r PPASR_HIP_LIB=tools/_ts/lib_tc8.so
done
