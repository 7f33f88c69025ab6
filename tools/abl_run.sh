#!/bin/bash
# kernel experiments: run the bench profile leg against alternative builds under build/abl_*/
for d in "" build/abl_*; do
  if [ -n "$d" ]; then export PPASR_HIP_LIB=$PWD/$d/libppasr_hip.so; fi
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 > /tmp/b.json 2>/tmp/b.err
  echo "== ${d:-base}: $(python tools/show_bench.py /tmp/b.json | grep -E 'attn|value' | tr '\n' ' ')"
done
