#!/bin/bash
# tools/variant.sh NAME [-DFLAG=..]...  builds pbrt-v3-iile_amd/lib/variants/libiile_gpu_NAME.so
# (A/B experiments: IILE_GPU_LIB=<that .so> python bench.py ...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/pbrt-v3-iile_amd/lib/variants
cd $R/pbrt-v3-iile_amd/csrc
hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wno-unused-function "$@" \
  -shared -o ../lib/variants/libiile_gpu_$N.so device/api.hip device/kernels.hip
