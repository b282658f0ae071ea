#!/bin/bash
# Round 5: cold resample timings (every run behind mic_plan_invalidate), lane vs marching kernel
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
out=gpurun_out/r05_cycle4
rm -rf $out && mkdir -p $out
for r in 1 2; do
  echo "-- marching"; MIC_RS_LANE=0 python3 scripts/time_resample_cold.py 2>&1 | grep -v amdgpu.ids | tee -a $out/march.txt
  echo "-- lane"; python3 scripts/time_resample_cold.py 2>&1 | grep -v amdgpu.ids | tee -a $out/lane.txt
  echo "-- lane, chunk 10000"; MIC_RS_LANE_CHUNK=10000 python3 scripts/time_resample_cold.py 2>&1 | grep -v amdgpu.ids | tee -a $out/lane_10000.txt
done
