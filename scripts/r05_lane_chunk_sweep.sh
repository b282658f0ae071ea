#!/bin/bash
# Round 5: the lane kernel's chunk size for calls below one round of 4096 wave slots (MIC_RS_LANE_CHUNK, model cycles per slot)
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
out=gpurun_out/r05_lane_chunk
rm -rf $out && mkdir -p $out
for pass in 1 2; do for c in 15000 8000 11000 20000; do
  echo "== MIC_RS_LANE_CHUNK=$c (pass $pass)" | tee -a $out/sweep.txt
  MIC_RS_LANE_CHUNK=$c timeout -k 10 200 python3 scripts/time_resample_cold.py 2>&1 | grep -v "^C3\|^16-canvas" | tee -a $out/sweep.txt
done; done
