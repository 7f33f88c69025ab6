#!/bin/bash
# cfg5 pipelined: decoder stream on its own CUs (hipExtStreamCreateWithCUMask) vs plain streams
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --config cfg5 --no-cpu-baseline --steps 80 > gpurun_out/r04r_$label.json 2> gpurun_out/r04r_$label.err || tail -3 gpurun_out/r04r_$label.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04r_$label.json"))
c=d["roofline"]["classes"]
print("$label", d["value"], d["ms_per_step"], "serial", (d["config"].get("serial") or {}).get("ms_per_step"), " beam %.3f tail %.3f" % (c["k_ctc_beam<512>"]["ms_per_step"], c["k_sq_tail<31>"]["ms_per_step"]))
PY
}
python -m pytest tests/test_ragged_gpu.py -q -x 2>&1 | tail -2
for r in 1 2; do
run plain$r PPASR_CU_PARTITION=0
run auto$r X=1
run p16_$r PPASR_CU_PARTITION=16
run p8_$r PPASR_CU_PARTITION=8
run p32_$r PPASR_CU_PARTITION=32
done
