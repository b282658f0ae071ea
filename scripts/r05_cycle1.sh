#!/bin/bash
# Round 5, first GPU pass: GPU tests on the fixed single-job launch, then the single-canvas cases by rocprofv3 kernel
# trace for 1 / 2 / 4 pages per workgroup (builds alternating) -> gpurun_out/r05_cycle1/
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_cycle1
rm -rf $out && mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
one() {  # one <label> <lib or ""> <round>
  label=$1; lib=$2; rnd=$3
  d=$out/$label.$rnd
  MIC_CASES_JSON=$PWD/$out/$label.$rnd.cases.json MIC_LIB=$lib MIC_ITERS=40 timeout -k 10 300 \
    rocprofv3 --kernel-trace --output-format csv -d $d -- python3 scripts/prof_single5.py > $out/$label.$rnd.log 2>&1 || { echo "FAILED $label"; tail -5 $out/$label.$rnd.log; return 1; }
  python3 scripts/trace_cases.py $d/*/*kernel_trace.csv $out/$label.$rnd.cases.json "$label.$rnd" | tee -a $out/summary.txt
}
for rnd in 1 2; do
  one ppw4 "" $rnd && one ppw1 $PWD/build/var_ppw1.bin $rnd && one ppw2 $PWD/build/var_ppw2.bin $rnd || exit 1
done
