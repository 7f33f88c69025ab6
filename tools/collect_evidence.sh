#!/bin/bash
# Round evidence: bench line + rocprofv3 kernel-trace summary of the same command + PMC passes (each --pmc set in its
# own run, with --kernel-trace only) + a sustained-load bench record.  Summaries land in gpurun_out/ AND in profiles/
# (profiles/hbm_traffic.json is regenerated and stamped with the digest of the kernel sources).
#   usage (on the GPU box, from the repo root): bash tools/collect_evidence.sh r02a
tag=${1:-r02x}
R=$(pwd)
mkdir -p $R/gpurun_out $R/profiles
python bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err
python bench.py --steps 1500 --warmup 20 --no-cpu-baseline > $R/gpurun_out/bench_sustained_$tag.json 2>> $R/gpurun_out/bench_$tag.err
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
cp gpurun_out/prof_$tag.txt profiles/${tag}_kernel_trace_bench.txt
cp gpurun_out/pmc_mfma_$tag.txt profiles/${tag}_pmc_mfma_busy.txt
{ echo "# FETCH_SIZE and WRITE_SIZE collected in two separate rocprofv3 --pmc passes (tools/collect_evidence.sh); unit KiB per dispatch.";
  echo "# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of wide coalesced reads -> HBM read bytes = 2 x FETCH_SIZE.";
  grep "| FETCH_SIZE |" gpurun_out/pmc_fetch_$tag.txt; grep "| WRITE_SIZE |" gpurun_out/pmc_write_$tag.txt; } > profiles/${tag}_pmc_hbm_traffic.txt
python tools/make_hbm_traffic.py gpurun_out/pmc_fetch_$tag.txt gpurun_out/pmc_write_$tag.txt $tag > gpurun_out/hbm_traffic_$tag.json
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
# the bench line again, now that the traffic evidence belongs to this build
python bench.py > $R/gpurun_out/bench_$tag.json 2>> $R/gpurun_out/bench_$tag.err
cp gpurun_out/bench_$tag.json profiles/${tag}_bench.json
cp gpurun_out/bench_sustained_$tag.json profiles/${tag}_bench_sustained.json
cp profiles/${tag}_*.txt profiles/${tag}_*.json gpurun_out/ 2>/dev/null
cat gpurun_out/bench_$tag.json
