#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r03_tests.log; [ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/profile_round3.sh > gpurun_out/r03_profile.log 2>&1; tail -60 gpurun_out/r03_profile.log
