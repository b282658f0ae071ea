#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r03_tests.log; [ $rc -ne 0 ] && exit $rc
for v in "A=1" "MIC_DIRECT_HOST_MAX=0" "MIC_LAYER_ARGS=0" "MIC_LAYER_ARGS=0 MIC_DIRECT_HOST_MAX=0"; do echo "--- $v"; env $v python scripts/prof_c1.py 2>&1 | grep -E "whole_call|r03_composite_one|device_only|composite_device_plus"; done
for v in "A=1" "MIC_LAYER_ARGS=0"; do echo "--- single canvas $v"; env $v python scripts/time_single.py 2>&1 | tail -6; done
