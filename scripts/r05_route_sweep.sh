#!/bin/bash
# Round 5: which resample kernel for calls below one round -- lane (shipped from 256 slots), tile (forced), marching (forced)
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
out=gpurun_out/r05_route
rm -rf $out && mkdir -p $out
for pass in 1 2; do
  for route in "lane:" "tile:MIC_RS_LANE=0 MIC_RS_MARCH_MIN_UNITS=100000000" "march:MIC_RS_LANE=0 MIC_RS_MARCH_MIN_UNITS=0"; do
    name=${route%%:*}; envs=${route#*:}
    echo "== $name (pass $pass)" | tee -a $out/sweep.txt
    env $envs timeout -k 10 200 python3 scripts/time_resample_cold.py 2>&1 | grep -v "amdgpu.ids" | tee -a $out/sweep.txt
  done
done
