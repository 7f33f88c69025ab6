#!/bin/bash
# Round-5 evidence on the GPU box (from the repo root): bash tools/r05_evidence.sh r05a
TAG=${1:-r05a}
R=$(pwd)
mkdir -p gpurun_out profiles
bash tools/collect_evidence.sh $TAG cfg2 cfg4 cfg5 cfg1 2>&1 | grep -E "^configs|^dominant"
# the fp16 x3 mode's own line + the kernel trace of the same command
python bench.py --gemm f16x3 > gpurun_out/${TAG}_bench_f16x3.json 2> gpurun_out/${TAG}_bench_f16x3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_f16x3 -o ${TAG}_f16x3 -- python $R/bench.py --gemm f16x3 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_f16x3.log 2>&1
cd $R
db=$(find gpurun_out/prof_${TAG}_f16x3 -name '*_results.db' 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db > gpurun_out/${TAG}_f16x3_kernel_trace.txt
rm -rf gpurun_out/prof_${TAG}_f16x3
cp gpurun_out/${TAG}_bench_f16x3.json gpurun_out/${TAG}_f16x3_kernel_trace.txt profiles/ 2>/dev/null
python tools/bench_ds2.py > gpurun_out/${TAG}_ds2_shapes.txt 2> gpurun_out/${TAG}_ds2_shapes.err
python tools/bench_beam.py > gpurun_out/${TAG}_beam.txt 2> gpurun_out/${TAG}_beam.err
python tools/bench_streams.py > gpurun_out/${TAG}_streams.txt 2> gpurun_out/${TAG}_streams.err
python tools/phase_ts.py --h3 > gpurun_out/${TAG}_phase_ts_h3.txt 2>&1
cp gpurun_out/${TAG}_ds2_shapes.txt gpurun_out/${TAG}_beam.txt gpurun_out/${TAG}_streams.txt gpurun_out/${TAG}_phase_ts_h3.txt profiles/ 2>/dev/null
cp profiles/${TAG}* profiles/hbm_traffic*.json gpurun_out/ 2>/dev/null
python tools/show_bench.py gpurun_out/${TAG}_bench_f16x3.json 2>/dev/null | head -12
true
