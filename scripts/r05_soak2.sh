#!/bin/bash
# Round 5, second seeds: default routing and every qualifying layer through the lane kernel with two-tile pieces forced as well
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
mkdir -p gpurun_out/r05_soak2
( echo "== default routing, seed 6001"; timeout -k 10 500 python3 scripts/soak.py 330 6001 2>&1 | tail -2
  echo "== MIC_RS_LANE_MIN_SLOTS=0, seed 6002"; MIC_RS_LANE_MIN_SLOTS=0 timeout -k 10 400 python3 scripts/soak.py 240 6002 2>&1 | tail -2
  echo "== MIC_RS_LANE_MIN_SLOTS=0 MIC_RS_LANE_SPLIT=0,0 (two x-tiles per piece everywhere), seed 6003"; MIC_RS_LANE_MIN_SLOTS=0 MIC_RS_LANE_SPLIT=0,0 timeout -k 10 300 python3 scripts/soak.py 150 6003 2>&1 | tail -2
  echo "== MIC_RS_LANE=0 MIC_RS_TILE_SMALL_PX=0 (tile kernel, 64 x 64 tiles everywhere), seed 6004"; MIC_RS_LANE=0 MIC_RS_TILE_SMALL_PX=0 timeout -k 10 300 python3 scripts/soak.py 150 6004 2>&1 | tail -2 ) | tee gpurun_out/r05_soak2/soak.txt
