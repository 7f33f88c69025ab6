#!/bin/bash
# rocprofv3 kernel traces of the non-headline BASELINE configs (tools/bench_configs.py), one run per config.
#   usage (GPU box, repo root): bash tools/trace_configs.sh r03a
tag=${1:-r03x}
R=$(pwd)
mkdir -p $R/gpurun_out
python tools/bench_configs.py > $R/gpurun_out/bench_configs_$tag.txt 2> $R/gpurun_out/bench_configs_$tag.err
cd /tmp && export TMPDIR=/tmp
for c in cfg1 cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_${c}_$tag -o $c -- python $R/tools/bench_configs.py $c > $R/gpurun_out/kt_${c}_$tag.log 2>&1
  db=$(find $R/gpurun_out/kt_${c}_$tag -name '*_results.db' | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $R/gpurun_out/kt_${c}_$tag.txt
  rm -rf $R/gpurun_out/kt_${c}_$tag
done
cat $R/gpurun_out/bench_configs_$tag.txt
