#!/bin/bash
# The lane kernel's chunks dealt XCD by XCD (shipped) against dealt in launch order (MIC_RS_LANE_XCDMAP=0), in the library:
# rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE + L2 hit / miss counters of the C3 placements canvas
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
export TMPDIR=/tmp MIC_ITERS=12 MIC_ALPHA=soft
out=gpurun_out/r05_xcdmap
rm -rf $out && mkdir -p $out
for m in 1 0; do
  export MIC_RS_LANE_XCDMAP=$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$m -- python3 scripts/prof_placements.py > $out/kt$m.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch$m -- python3 scripts/prof_placements.py > $out/fetch$m.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write$m -- python3 scripts/prof_placements.py > $out/write$m.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc$m -- python3 scripts/prof_placements.py > $out/tcc$m.log 2>&1
  echo "== MIC_RS_LANE_XCDMAP=$m"
  cut -d, -f1-4 $out/kt$m/*/*kernel_stats.csv | grep lane
  for d in fetch write tcc; do python3 scripts/pmc_summary.py $out/$d$m | grep -A3 "resample_lane" | grep -v "^mic"; done
done | tee $out/summary.txt
