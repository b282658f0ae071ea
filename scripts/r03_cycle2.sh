#!/bin/bash
mkdir -p gpurun_out
for u in 1024 1536 2048 3072 4096 8192 16384; do echo "--- MIC_RS_UNIT_PX=$u"; MIC_RS_UNIT_PX=$u python scripts/time_c5.py 2>&1 | grep -v amdgpu.ids | grep -E "iter 0|placements \(soft|layers"; done
