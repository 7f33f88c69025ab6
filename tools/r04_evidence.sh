#!/bin/bash
TAG=${1:-r04b}
bash tools/collect_evidence.sh $TAG cfg2 cfg4 cfg5 cfg1 2>&1 | tail -70
python tools/bench_ds2.py > gpurun_out/${TAG}_ds2_shapes.txt 2> gpurun_out/${TAG}_ds2_shapes.err
timeout 120 tools/cu_mask_probe > gpurun_out/${TAG}_cu_mask_probe.txt 2>&1
