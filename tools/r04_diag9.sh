#!/bin/bash
R=$(pwd)
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_row_block_gpu.py -q -x -s > $R/gpurun_out/r04_rowblock_tests9.log 2>&1; grep -E "rows|passed|failed|Error" $R/gpurun_out/r04_rowblock_tests9.log | tail -12
timeout 900 python -m pytest tests -m gpu -q -x > $R/gpurun_out/r04_gpu_tests9.log 2>&1; tail -3 $R/gpurun_out/r04_gpu_tests9.log
timeout 500 python tools/bench_batch_sizes.py conformer efficient_conformer > $R/gpurun_out/r04h_batch_sizes.txt 2>&1; cat $R/gpurun_out/r04h_batch_sizes.txt
