#!/bin/bash
# Round-6 evidence on the GPU box (from the repo root): bash tools/r06_evidence.sh r06a
TAG=${1:-r06a}
R=$(pwd)
mkdir -p gpurun_out profiles
bash tools/collect_evidence.sh $TAG cfg2 cfg4 cfg5 cfg1 2>&1 | grep -E "^configs|^dominant"
python tools/bench_ds2.py > gpurun_out/${TAG}_ds2_shapes.txt 2> gpurun_out/${TAG}_ds2_shapes.err
python tools/bench_beam.py flat trained > gpurun_out/${TAG}_beam.txt 2> gpurun_out/${TAG}_beam.err
python tools/bench_streams.py > gpurun_out/${TAG}_streams.txt 2> gpurun_out/${TAG}_streams.err
# the beam search alone under rocprofv3 (prune pre-pass + search kernels of tools/bench_beam.py's first lines)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_beam -o ${TAG}_beam -- python $R/tools/experiments/r06/quick_beam.py 10 100 300 > $R/gpurun_out/prof_${TAG}_beam.log 2>&1
cd $R
db=$(find gpurun_out/prof_${TAG}_beam -name '*_results.db' 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db > gpurun_out/${TAG}_beam_kernel_trace.txt
rm -rf gpurun_out/prof_${TAG}_beam
# one streaming session's chunk under rocprofv3: launches, busy / wall time per chunk, per-kernel averages
cd /tmp
SESS=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_chunk -o chunk --output-format csv -- python $R/tools/experiments/r06/stream_ab.py > $R/gpurun_out/prof_${TAG}_chunk.log 2>&1
cd $R
csv=$(find gpurun_out/prof_${TAG}_chunk -name '*kernel_trace.csv' 2>/dev/null | head -1)
[ -n "$csv" ] && python tools/experiments/r06/trace_chunks.py $csv > gpurun_out/${TAG}_stream_chunk_trace.txt
rm -rf gpurun_out/prof_${TAG}_chunk
for i in 1 2 3; do python tools/experiments/r06/stream_ab.py; done > gpurun_out/${TAG}_stream_ab.txt 2>/dev/null
cp gpurun_out/${TAG}_stream_chunk_trace.txt gpurun_out/${TAG}_stream_ab.txt profiles/ 2>/dev/null
# every compiled kernel launched by the GPU suite (tools/kernel_coverage.py --list was run on the build box)
PPASR_KCOV=gpurun_out/kcov.tsv python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gpu_suite.txt 2>&1
python tools/kernel_coverage.py --report gpurun_out/kcov.tsv > gpurun_out/${TAG}_kernel_coverage.txt 2>&1
cp gpurun_out/${TAG}_ds2_shapes.txt gpurun_out/${TAG}_beam.txt gpurun_out/${TAG}_streams.txt gpurun_out/${TAG}_beam_kernel_trace.txt gpurun_out/${TAG}_kernel_coverage.txt profiles/ 2>/dev/null
cp profiles/${TAG}* profiles/hbm_traffic*.json gpurun_out/ 2>/dev/null
true
