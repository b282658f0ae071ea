#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
# FETCH_SIZE / TCC_EA0_RDREQ calibration per access shape (scripts/calib_fetch.hip) and the same counters on the
# resample launch (scripts/prof_placements.py): which correction applies to the band loader's 64-byte row segments.
export TMPDIR=/tmp  # (already in the repo copy: line 2)
out=gpurun_out/prof_fetch_calib
rm -rf $out && mkdir -p $out
for c in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum" WRITE_SIZE; do
  tag=$(echo $c | tr ' ' '+')
  rocprofv3 --pmc $c --output-format csv -d $out/calib_$tag -- ./scripts/calib_fetch.bin > $out/calib_$tag.log 2>&1 || { echo "FAILED calib $c"; tail -3 $out/calib_$tag.log; }
  MIC_ITERS=8 rocprofv3 --pmc $c --output-format csv -d $out/rs_$tag -- python3 scripts/prof_placements.py > $out/rs_$tag.log 2>&1 || { echo "FAILED rs $c"; tail -3 $out/rs_$tag.log; }
done
for d in $out/calib_* $out/rs_*; do [ -d $d ] && { echo "== $d"; python3 scripts/pmc_summary.py $d | grep -v "^ *$" ; }; done > $out/summary.txt 2>&1
grep -E "^==|wide1k|seg|dword|resample_march|planarize|FETCH|RDREQ|TCC_|WRITE" $out/summary.txt | grep -v flush | head -120
