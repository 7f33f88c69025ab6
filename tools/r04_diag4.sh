#!/bin/bash
# round-4 call 4: attention kernel variants (tools/_ts/lib_attn*.so) on cfg5 / cfg4 / the two-kernel route of cfg2, timeline
R=$(pwd)
mkdir -p $R/gpurun_out
timeout 300 python -m pytest tests/test_row_block_gpu.py -x -q > $R/gpurun_out/r04_rowblock_tests4.log 2>&1; tail -3 $R/gpurun_out/r04_rowblock_tests4.log
for v in default attnB attnE attnG; do
  if [ $v = default ]; then unset PPASR_HIP_LIB; else export PPASR_HIP_LIB=$R/tools/_ts/lib_$v.so; fi
  timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-pipeline --steps 40 --warmup 5 > $R/gpurun_out/r04d_cfg5_$v.json 2> $R/gpurun_out/r04d_cfg5_$v.err
  PPASR_ATTN_FUSE_MIN_BLOCKS=100000 timeout 300 python bench.py --config cfg2 --no-cpu-baseline --steps 60 --warmup 5 > $R/gpurun_out/r04d_cfg2two_$v.json 2> $R/gpurun_out/r04d_cfg2two_$v.err
  python - <<PY
import json
for f in ("r04d_cfg5_$v.json", "r04d_cfg2two_$v.json"):
    try:
        d = json.load(open("$R/gpurun_out/" + f)); c = d["roofline"]["classes"]
        print(f, d["value"], d["ms_per_step"], {k: (c[k].get("ms_per_step"), c[k].get("frac")) for k in c if "attention" in k or "out_glu" in k})
    except Exception as e: print(f, "FAILED", e)
PY
done
unset PPASR_HIP_LIB
cd /tmp && export TMPDIR=/tmp
for cfg in cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$cfg -o $cfg -- python $R/bench.py --config $cfg --no-pipeline --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/kt_$cfg.log 2>&1
  db=$(find $R/gpurun_out/kt_$cfg -name '*_results.db' | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_timeline.py $db 140 > $R/gpurun_out/r04d_timeline_$cfg.txt
  rm -rf $R/gpurun_out/kt_$cfg
done
cd $R
grep "k_attention" gpurun_out/r04d_timeline_cfg5.txt | tail -12 | awk '{print $6}' | tr '\n' ' '; echo
