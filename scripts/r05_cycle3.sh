#!/bin/bash
# Round 5: lane kernel in the library -- full GPU tests, C5 / placements timings lane vs marching, planarize trace
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_cycle3
rm -rf $out && mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do
  echo "-- lane";  python3 scripts/time_c5.py 2>&1 | grep -v amdgpu.ids | tee -a $out/time_c5_lane.txt
  echo "-- march"; MIC_RS_LANE=0 python3 scripts/time_c5.py 2>&1 | grep -v amdgpu.ids | tee -a $out/time_c5_march.txt
done
MIC_ITERS=8 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 scripts/prof_placements.py > $out/kt.log 2>&1; cut -d, -f1-4 $out/kt/*/*kernel_stats.csv | grep -i "resample\|planar"
