#!/bin/bash
# Build an alternative libiile_gpu.so for kernel A/B runs:  tools/build_variant.sh NAME "TU [TU..]" "-DFLAG=.. [...]"
#   -> pbrt-v3-iile_amd/lib/variants/libiile_gpu_NAME.so  (select it with IILE_GPU_LIB=<path>; bench.py records the override)
# Only the listed translation units (names of csrc/device/*.hip without the suffix) are recompiled with the extra flags.
set -e
NAME=$1; TUS=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/pbrt-v3-iile_amd/csrc
F="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
mkdir -p /tmp/variants/$NAME $R/pbrt-v3-iile_amd/lib/variants
make -s -C $C -j8 gpu
OBJS=""
for tu in $(cd $C/device && ls *.hip | sed s/.hip//); do
  if [[ " $TUS " == *" $tu "* ]]; then
    hipcc $F $FLAGS -c $C/device/$tu.hip -o /tmp/variants/$NAME/$tu.o &
    OBJS="$OBJS /tmp/variants/$NAME/$tu.o"
  else
    OBJS="$OBJS $C/device/$tu.o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -o $R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$NAME.so $OBJS
echo built $R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$NAME.so
