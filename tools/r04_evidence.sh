#!/bin/bash
TAG=${1:-r04c}
python -m pytest tests -m gpu -q 2>&1 | tail -3
bash tools/collect_evidence.sh $TAG cfg2 cfg4 cfg5 cfg1 2>&1 | grep -E "^configs|^dominant" 
python tools/bench_ds2.py > gpurun_out/${TAG}_ds2_shapes.txt 2> gpurun_out/${TAG}_ds2_shapes.err
