#!/bin/bash
python -m pytest tests -m gpu -q -x -k "deepspeech or ds2 or generic or general or shapes or width" 2>&1 | tail -3
python tools/bench_ds2.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(d['model'], d['B'], d['ms'], d['audio_s_per_s'], {n[:30]:(v['ms_per_step'], v.get('frac')) for n,v in list(d['kernels'].items())[:3]})
"
python tools/bench_shapes.py 2>/dev/null | tail -8
