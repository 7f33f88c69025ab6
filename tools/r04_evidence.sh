#!/bin/bash
python -m pytest tests -m gpu -q 2>&1 | tail -4
bash tools/collect_evidence.sh r04a cfg2 cfg4 cfg5 cfg1 2>&1 | tail -60
