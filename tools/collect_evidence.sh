#!/bin/bash
# Round evidence: bench line + rocprofv3 kernel-trace summary of the same command + PMC passes.
#   usage (on the GPU box, from the repo root): bash tools/collect_evidence.sh r01e
tag=${1:-r01x}
R=$(pwd)
mkdir -p $R/gpurun_out
python bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma_$tag -o $tag -- python $R/tools/profile_step.py --steps 2 > $R/gpurun_out/pmc_mfma_$tag.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$tag -o $tag -- python $R/tools/profile_step.py --steps 2 > $R/gpurun_out/pmc_fetch_$tag.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$tag -o $tag -- python $R/tools/profile_step.py --steps 2 > $R/gpurun_out/pmc_write_$tag.log 2>&1
cd $R
for d in prof pmc_mfma pmc_fetch pmc_write; do
  db=$(find gpurun_out/${d}_$tag -name '*_results.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > gpurun_out/${d}_$tag.txt
done
cat gpurun_out/bench_$tag.json
