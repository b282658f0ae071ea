#!/bin/bash
# Round 5: the lane kernel's DEEP form (two band register sets, three waves per SIMD) for launches of <= 3072 slots against the
# one-set form (MIC_RS_LANE_DEEP=0)
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
out=gpurun_out/r05_lane_deep
rm -rf $out && mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_march.py tests/test_gpu_resident_layers.py tests/test_gpu_multi_atlas.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -2 $out/pytest.log; [ $rc -eq 0 ] || exit 1
for pass in 1 2; do for d in 0 3072; do
  echo "== MIC_RS_LANE_DEEP=$d (pass $pass)" | tee -a $out/sweep.txt
  MIC_RS_LANE_DEEP=$d timeout -k 10 200 python3 scripts/time_resample_cold.py 2>&1 | grep -v "amdgpu.ids\|^C3\|^16-canvas" | tee -a $out/sweep.txt
done; done
