#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof_r05b; rm -rf $out; mkdir -p $out
for a in soft binary; do
  MIC_ALPHA=$a MIC_ITERS=12 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$a -- python3 scripts/prof_placements.py > $out/$a.log 2>&1 || { echo FAILED $a; tail -3 $out/$a.log; exit 1; }
  cp $out/$a/*/*kernel_stats.csv $out/placements_$a.kernel_stats.csv
  echo "== $a"; cut -d, -f1-4 $out/placements_$a.kernel_stats.csv | head -5
done
