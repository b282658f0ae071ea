#!/bin/bash
# Round 5: chunk sizes by dispatch generation (MIC_LANE_AGE=w0,w1,w2,w3; needs profiles/r05_lane_age.patch in BOTH builds:
#   scripts/build_variant.sh age -p profiles/r05_lane_age.patch; ... probe -p profiles/r05_lane_age.patch -p profiles/r05_lane_stage_probe.patch)
# -- events on the patched build (MIC_LIB), stage stamps on the probe build
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
out=gpurun_out/r05_lane_age
rm -rf $out && mkdir -p $out
for pass in 1 2; do for w in "" "1.2,1.05,0.95,0.8" "1.4,1.1,0.85,0.65" "1.6,1.15,0.8,0.45"; do
  echo "== MIC_LANE_AGE=$w (pass $pass)" | tee -a $out/sweep.txt
  MIC_LANE_AGE=$w timeout -k 10 200 python3 scripts/time_resample_cold.py 2>&1 | grep "^C3\|^16-canvas" | tee -a $out/sweep.txt
done; done
for w in "" "1.4,1.1,0.85,0.65"; do
  echo "== probe MIC_LANE_AGE=$w" | tee -a $out/probe.txt
  MIC_LANE_AGE=$w MIC_LIB=$PWD/build/var_probe.bin timeout -k 10 300 python3 scripts/lane_stage_probe.py 2>&1 | grep -A11 "C3 placements (soft)" | tee -a $out/probe.txt
done
