#!/bin/bash
for rt in 0 1 2 4; do
echo "== PPASR_WAVE_RT=$rt"
if [ $rt = 0 ]; then python tools/bench_ds2.py uni 2>/dev/null; else PPASR_WAVE_RT=$rt python tools/bench_ds2.py uni 2>/dev/null; fi | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); k=[v for n,v in d['kernels'].items() if 'k_lstm_wave' in n][0]
    print(d['B'], d['ms'], d['audio_s_per_s'], 'wave avg us', k['avg_us'])
"
done
