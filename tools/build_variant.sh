#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: build libppasr_hip with extra compile flags into tools/_ts/lib_NAME.so
# (A/B kernel experiments on ONE box: PPASR_HIP_LIB=tools/_ts/lib_NAME.so python bench.py ...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_ts
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude "$@" -o tools/_ts/lib_${name}.so ppasr_amd/csrc/*.hip
echo built tools/_ts/lib_${name}.so
