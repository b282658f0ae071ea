#!/bin/bash
# Round 5 soak: randomised differential run against the oracle, default routing, every qualifying layer forced through the lane
# kernel, and forced through the marching kernel
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
mkdir -p gpurun_out/r05_soak
( echo "== default routing"; timeout -k 10 400 python3 scripts/soak.py 240 5001 2>&1 | tail -2
  echo "== MIC_RS_LANE_MIN_SLOTS=0 (lane kernel for every qualifying layer)"; MIC_RS_LANE_MIN_SLOTS=0 timeout -k 10 400 python3 scripts/soak.py 240 5002 2>&1 | tail -2
  echo "== MIC_RS_LANE=0 MIC_RS_MARCH_MIN_UNITS=0 (marching kernel)"; MIC_RS_LANE=0 MIC_RS_MARCH_MIN_UNITS=0 timeout -k 10 300 python3 scripts/soak.py 120 5003 2>&1 | tail -2 ) | tee gpurun_out/r05_soak/soak.txt
