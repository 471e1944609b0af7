#!/bin/bash
# Times the fused bottleneck launch with phases cut out (MIC_FUSED_DEBUG bits; results of those runs are wrong on purpose).
for d in 0 1 30 31 2 4 8 16 6 24; do
  MIC_FUSED_DEBUG=$d timeout -k 10 100 python tools/fused_sweep.py 2>&1 | grep -v amdgpu.ids || exit 1
done
