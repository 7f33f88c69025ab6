#!/bin/bash
# cfg5 pipelined: beam search at 64 registers (co-resident with the 96-register 16-wave layer kernels?) x W16 on/off
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --config cfg5 --no-cpu-baseline --steps 60 > gpurun_out/r04o_$label.json 2> gpurun_out/r04o_$label.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04o_$label.json"))
c=d["roofline"]["classes"]
print("$label", d["value"], d["ms_per_step"], "serial", (d["config"].get("serial") or {}).get("ms_per_step"), " tail %.3f mid %.3f conv2 %.3f beam %.3f" % (c["k_sq_tail<31>"]["ms_per_step"], c["k_sq_mid"]["ms_per_step"], c["conv2"]["ms_per_step"], c["k_ctc_beam<512>"]["ms_per_step"]))
PY
}
run base X=1
run w16 PPASR_W16=1
run beam64 PPASR_HIP_LIB=tools/_ts/lib_beam64.so
run beam64_w16 PPASR_HIP_LIB=tools/_ts/lib_beam64.so PPASR_W16=1
run base2 X=1
