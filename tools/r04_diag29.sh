#!/bin/bash
python -m pytest tests/test_block_table_gpu.py -q 2>&1 | tail -1
for v in base wpf4 wpf4o4 wpf1 base; do
echo "== $v"
if [ $v = base ]; then python tools/bench_ds2.py uni 2>/dev/null; else PPASR_HIP_LIB=tools/_ts/lib_$v.so python tools/bench_ds2.py uni 2>/dev/null; fi | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); k=[v for n,v in d['kernels'].items() if 'k_lstm_wave' in n][0]
    print(d['B'], d['ms'], d['audio_s_per_s'], 'wave avg us', k['avg_us'])
"
done
