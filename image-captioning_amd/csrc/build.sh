#!/bin/bash
# Builds libmic_hip.so for gfx950 (cross-compiles without a GPU).  Every source is compiled in
# parallel; the script fails if any compile fails, so a stale object is never linked.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OBJS=""
PIDS=""
mkdir -p build
for f in mic_core igemm igemm8 fused_block stem_pool gemm_ln pool decode_ops attn_fused decode_persist search_ops swin drivers noalloc; do
  [ -f $f.hip ] || continue
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ mic_common.h -nt build/$f.o ] || [ decode_shared.h -nt build/$f.o ] || [ igemm_shared.h -nt build/$f.o ] || [ search_shared.h -nt build/$f.o ] || [ ../../include/mic_hip.h -nt build/$f.o ]; then
    # compile to a temporary name: an interrupted or failed compile leaves no object behind
    ( $HIPCC $FLAGS -c $f.hip -o build/$f.o.tmp && mv build/$f.o.tmp build/$f.o ) &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS build/$f.o"
done
FAIL=0
for p in $PIDS; do
  wait $p || FAIL=1
done
if [ $FAIL -ne 0 ]; then
  echo "build.sh: a hipcc compile failed" >&2
  exit 1
fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmic_hip.so $OBJS
echo "built $(pwd)/libmic_hip.so"
