#!/bin/bash
# round-4 diagnostics: new row-block tests, GPU suite on the new build, the 16-row GEMM microbenchmark, per-dispatch
# timelines of cfg5 (round-3 kernels vs new) / cfg4 (serial), cfg5 bench lines
R=$(pwd)
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_row_block_gpu.py -x -q -s > $R/gpurun_out/r04_rowblock_tests.log 2>&1; tail -15 $R/gpurun_out/r04_rowblock_tests.log
timeout 120 tools/mb_rb16 > $R/gpurun_out/r04_mb_rb16.txt 2>&1; cat $R/gpurun_out/r04_mb_rb16.txt
for v in legacy new; do
  if [ $v = legacy ]; then export PPASR_SQ_LEGACY=1; else unset PPASR_SQ_LEGACY; fi
  timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 60 --warmup 5 > $R/gpurun_out/r04_cfg5_$v.json 2> $R/gpurun_out/r04_cfg5_$v.err
  timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-pipeline --steps 60 --warmup 5 > $R/gpurun_out/r04_cfg5_${v}_np.json 2>> $R/gpurun_out/r04_cfg5_$v.err
  python - <<PY
import json
for f in ("r04_cfg5_$v.json", "r04_cfg5_${v}_np.json"):
    try:
        d = json.load(open("$R/gpurun_out/" + f)); print(f, d["value"], d["ms_per_step"])
        if f.endswith("_np.json"):
            for k, c in d["roofline"]["classes"].items(): print("   ", k, c.get("ms_per_step"), c.get("frac"))
    except Exception as e: print(f, "FAILED", e)
PY
done
unset PPASR_SQ_LEGACY
cd /tmp && export TMPDIR=/tmp
for cfg in cfg5legacy cfg5 cfg4; do
  c=${cfg%legacy}
  if [ $cfg = cfg5legacy ]; then export PPASR_SQ_LEGACY=1; else unset PPASR_SQ_LEGACY; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$cfg -o $cfg -- python $R/bench.py --config $c --no-pipeline --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/kt_$cfg.log 2>&1
  db=$(find $R/gpurun_out/kt_$cfg -name '*_results.db' | head -1)
  if [ -n "$db" ]; then
    python $R/tools/rocpd_timeline.py $db 140 > $R/gpurun_out/r04_timeline_$cfg.txt
    python $R/tools/rocpd_summary.py $db > $R/gpurun_out/r04_summary_$cfg.txt
  fi
  rm -rf $R/gpurun_out/kt_$cfg
done
unset PPASR_SQ_LEGACY
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r04_gpu_tests.log 2>&1; tail -5 $R/gpurun_out/r04_gpu_tests.log
