#!/bin/bash
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --config cfg5 --no-cpu-baseline --steps 60 > gpurun_out/r04s_$label.json 2> gpurun_out/r04s_$label.err || tail -3 gpurun_out/r04s_$label.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04s_$label.json"))
print("$label", d["value"], d["ms_per_step"], "serial", (d["config"].get("serial") or {}).get("ms_per_step"))
PY
}
run plain PPASR_CU_PARTITION=0
run allbits PPASR_CU_PARTITION=-1
run p16 PPASR_CU_PARTITION=16
cd /tmp && export TMPDIR=/tmp
PPASR_CU_PARTITION=16 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/kt_p16 -o p16 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 30 --warmup 3 > /dev/null 2>&1
db=$(find $GRAFT_REPO_ROOT/gpurun_out/kt_p16 -name '*_results.db' | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db | awk -F'|' '{n=$1; sub(/\(.*/,"",n); printf "%-60s |%s|%s|%s\n", substr(n,1,60), $2, $3, $4}' | head -14
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kt_p16
