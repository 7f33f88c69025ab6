#!/bin/bash
echo "=== 16 waves (k_conv_ffn_t<kW16>) ==="; python tools/phase_ts.py --t 2>&1 | tail -30
echo "=== 8 waves (k_conv_ffn) ==="; PPASR_W16=0 python tools/phase_ts.py 2>&1 | sed -n 1,28p
