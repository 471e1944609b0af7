#!/bin/bash
# Builds libmic_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OBJS=""
for f in mic_core igemm gemm_ln pool decode_ops search_ops swin drivers; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ mic_common.h -nt build/$f.o ] || [ ../../include/mic_hip.h -nt build/$f.o ]; then
    mkdir -p build
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
  fi
  OBJS="$OBJS build/$f.o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmic_hip.so $OBJS
echo "built $(pwd)/libmic_hip.so"
