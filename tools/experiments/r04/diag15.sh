#!/bin/bash
# 16-wave kernels: row-block tests, full GPU suite, benches of the three Conformer-family configs
python -m pytest tests/test_row_block_gpu.py -q -x 2>&1 | tail -15
python -m pytest tests -m gpu -q -x --deselect tests/test_row_block_gpu.py 2>&1 | tail -5
for c in cfg2 cfg4 cfg5; do
  python bench.py --config $c --no-cpu-baseline --steps 60 > gpurun_out/r04n_$c.json 2> gpurun_out/r04n_$c.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04n_$c.json"))
print("$c", d["value"], d["ms_per_step"], (d["config"].get("serial") or {}).get("ms_per_step"), d["roofline"]["kernel"], d["roofline"]["frac"])
for k,v in sorted(d["roofline"]["classes"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:9]:
    print(f"   {k:26s} {v['ms_per_step']:.4f} ms x{v['launches_per_step']:.0f} frac {v.get('frac')}")
PY
done
