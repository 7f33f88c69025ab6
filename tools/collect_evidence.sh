#!/bin/bash
# Round evidence for every BASELINE config, on the sources as they are: bench line + rocprofv3 kernel-trace summary of the
# same command + PMC passes (each --pmc set in its own run, with --kernel-trace only) + (cfg2) a sustained-load record.
# Summaries land in gpurun_out/ AND in profiles/ (profiles/hbm_traffic[_cfgN].json regenerated, stamped with the digest
# of the kernel sources).
#   usage (on the GPU box, from the repo root): bash tools/collect_evidence.sh r03a [cfg2 cfg4 cfg5 cfg1]
tag=${1:-r03x}
shift
cfgs=${@:-cfg2 cfg4 cfg5 cfg1}
R=$(pwd)
mkdir -p $R/gpurun_out $R/profiles
for cfg in $cfgs; do
  if [ $cfg = cfg2 ]; then sfx=""; else sfx="_$cfg"; fi
  pre=${tag}${sfx}
  python bench.py --config $cfg > $R/gpurun_out/bench_$pre.json 2> $R/gpurun_out/bench_$pre.err
  if [ $cfg = cfg2 ]; then
    python bench.py --steps 1500 --warmup 20 --no-cpu-baseline > $R/gpurun_out/bench_sustained_$tag.json 2>> $R/gpurun_out/bench_$pre.err
  fi
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$pre -o $pre -- python $R/bench.py --config $cfg --no-cpu-baseline > $R/gpurun_out/prof_$pre.log 2>&1
  if [ $cfg = cfg2 ]; then
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma_$pre -o $pre -- python $R/bench.py --config $cfg --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_mfma_$pre.log 2>&1
  fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$pre -o $pre -- python $R/bench.py --config $cfg --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_fetch_$pre.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$pre -o $pre -- python $R/bench.py --config $cfg --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_write_$pre.log 2>&1
  cd $R
  for d in prof pmc_mfma pmc_fetch pmc_write; do
    db=$(find gpurun_out/${d}_$pre -name '*_results.db' 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py $db > gpurun_out/${d}_$pre.txt
    rm -rf gpurun_out/${d}_$pre
  done
  cp gpurun_out/prof_$pre.txt profiles/${pre}_kernel_trace_bench.txt
  [ -f gpurun_out/pmc_mfma_$pre.txt ] && cp gpurun_out/pmc_mfma_$pre.txt profiles/${pre}_pmc_mfma_busy.txt
  { echo "# FETCH_SIZE and WRITE_SIZE collected in two separate rocprofv3 --pmc passes (tools/collect_evidence.sh $tag $cfg); unit KiB per dispatch.";
    echo "# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of wide coalesced reads -> HBM read bytes = 2 x FETCH_SIZE.";
    grep "| FETCH_SIZE |" gpurun_out/pmc_fetch_$pre.txt; grep "| WRITE_SIZE |" gpurun_out/pmc_write_$pre.txt; } > profiles/${pre}_pmc_hbm_traffic.txt
  python tools/make_hbm_traffic.py gpurun_out/pmc_fetch_$pre.txt gpurun_out/pmc_write_$pre.txt $tag $cfg gpurun_out/prof_$pre.txt > gpurun_out/hbm_traffic_$pre.json
  # the bench line again, now that the traffic evidence belongs to this build
  python bench.py --config $cfg > $R/gpurun_out/bench_$pre.json 2>> $R/gpurun_out/bench_$pre.err
  cp gpurun_out/bench_$pre.json profiles/${pre}_bench.json
  [ $cfg = cfg2 ] && cp gpurun_out/bench_sustained_$tag.json profiles/${tag}_bench_sustained.json
  python tools/show_bench.py gpurun_out/bench_$pre.json 2>/dev/null | head -40
done
cp profiles/${tag}*.txt profiles/${tag}*.json profiles/hbm_traffic*.json gpurun_out/ 2>/dev/null
true
