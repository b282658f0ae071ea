#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r03_tests.log; [ $rc -ne 0 ] && exit $rc
python scripts/prof_dropin4k.py 2>&1 | tail -12
