#!/bin/bash
mkdir -p gpurun_out
for v in "MIC_MEDIAN_EXP=0" "MIC_MEDIAN_EXP=8"; do echo "--- median $v"; env $v python scripts/time_median.py 2>&1 | grep -E "noise" ; done > gpurun_out/r03_median4.txt 2>&1; cat gpurun_out/r03_median4.txt
MIC_MEDIAN_EXP=8 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "median" 2>&1 | tail -3
