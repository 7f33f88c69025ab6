#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: build libppasr_hip with extra compile flags into tools/_ts/lib_NAME.so
# (A/B kernel experiments on ONE box: PPASR_HIP_LIB=tools/_ts/lib_NAME.so python bench.py ...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_ts
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
g.build_lib(lib='tools/_ts/lib_' + sys.argv[1] + '.so', extra_flags=sys.argv[2:], obj_dir='build/obj_' + sys.argv[1], verbose=False)
" "$name" "$@"
echo built tools/_ts/lib_${name}.so
