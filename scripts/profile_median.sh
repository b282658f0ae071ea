#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
# Kernel-trace durations of the median kernel per case (tuning knobs: MIC_MEDIAN_DBG, MIC_MEDIAN_BLOCKS).
export TMPDIR=/tmp  # (already in the repo copy: line 2)
mkdir -p gpurun_out
for c in 3840x2160:noise 3840x2160:flat 7680x4320:noise 7680x4320:sprinkle 492x492:sprinkle; do
  export MIC_CASE=$c
  rm -rf gpurun_out/prof_med
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_med -- python3 scripts/prof_median.py > gpurun_out/prof_med.log 2>&1 || { tail -5 gpurun_out/prof_med.log; exit 1; }
  python3 - "$c" <<'PY'
import csv, glob, os, sys
for f in glob.glob("gpurun_out/prof_med/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "median" in r["Name"]:
            print(f"{sys.argv[1]:22s} dbg={os.environ.get('MIC_MEDIAN_DBG','0')} blocks={os.environ.get('MIC_MEDIAN_BLOCKS','-')}: avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:7.1f}")
PY
done
