#!/bin/bash
# round-4 call 6: DeepSpeech2 wavefront with row tiles / GRU: tests + shapes; cfg4 gap analysis
R=$(pwd)
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_deepspeech2_gpu.py tests/test_ref_pin_gpu.py -q -x -k "deepspeech2 or ds2" > $R/gpurun_out/r04_ds2_tests.log 2>&1; tail -6 $R/gpurun_out/r04_ds2_tests.log
timeout 400 python tools/bench_ds2.py > $R/gpurun_out/r04f_ds2.txt 2>&1; cut -c1-330 $R/gpurun_out/r04f_ds2.txt | tail -8
cd /tmp && export TMPDIR=/tmp
for mode in serial pipe; do
  extra=""; [ $mode = serial ] && extra="--no-pipeline"
  timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_cfg4$mode -o cfg4 -- python $R/bench.py --config cfg4 $extra --no-cpu-baseline --steps 30 --warmup 3 > $R/gpurun_out/kt_cfg4$mode.log 2>&1
  db=$(find $R/gpurun_out/kt_cfg4$mode -name '*_results.db' | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_gaps.py $db > $R/gpurun_out/r04f_gaps_cfg4$mode.txt
  rm -rf $R/gpurun_out/kt_cfg4$mode
  echo "== cfg4 $mode"; cat $R/gpurun_out/r04f_gaps_cfg4$mode.txt | cut -c1-160
done
