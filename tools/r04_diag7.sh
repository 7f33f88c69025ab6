#!/bin/bash
R=$(pwd)
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_deepspeech2_gpu.py tests/test_ref_pin_gpu.py -q -x -k "deepspeech2 or ds2" > $R/gpurun_out/r04_ds2_tests7.log 2>&1; tail -4 $R/gpurun_out/r04_ds2_tests7.log
timeout 400 python tools/bench_ds2.py > $R/gpurun_out/r04g_ds2.txt 2>&1; cut -c1-330 $R/gpurun_out/r04g_ds2.txt | tail -6
