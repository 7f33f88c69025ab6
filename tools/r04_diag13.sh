#!/bin/bash
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for mode in pipe serial; do
  extra=""; [ $mode = serial ] && extra="--no-pipeline"
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_cfg5$mode -o cfg5 -- python $R/bench.py --config cfg5 $extra --no-cpu-baseline --steps 40 --warmup 3 > $R/gpurun_out/kt_cfg5$mode.log 2>&1
  db=$(find $R/gpurun_out/kt_cfg5$mode -name '*_results.db' | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db | cut -c1-60,200-400 | sed 's/(.*) | / | /' > $R/gpurun_out/r04l_summary_cfg5$mode.txt
  rm -rf $R/gpurun_out/kt_cfg5$mode
  tail -1 $R/gpurun_out/kt_cfg5$mode.log | cut -c1-200
done
cd $R
python - <<'PY'
import re
def load(f):
    out={}
    for l in open(f):
        p=[x.strip() for x in l.split("|")]
        if len(p)>=5 and not l.startswith("#"):
            try: out[p[0][:48]]=(int(p[-4]),float(p[-2]))
            except Exception: pass
    return out
a=load("gpurun_out/r04l_summary_cfg5pipe.txt"); b=load("gpurun_out/r04l_summary_cfg5serial.txt")
for k in sorted(a, key=lambda k:-a[k][0]*a[k][1])[:16]:
    if k in b: print(f"{k:50s} calls {a[k][0]:5d}  pipelined avg {a[k][1]:9.2f} us   serial avg {b[k][1]:9.2f} us   x{a[k][1]/b[k][1]:.2f}")
PY
