#!/bin/bash
mkdir -p gpurun_out
python scripts/prof_run_layouts_save.py > gpurun_out/r03_run_layouts_prof.txt 2>&1; head -60 gpurun_out/r03_run_layouts_prof.txt
timeout -k 10 330 python scripts/soak.py 240 30301 > gpurun_out/soak_r03a.log 2>&1; tail -3 gpurun_out/soak_r03a.log
MIC_RS_MARCH_MIN_UNITS=0 timeout -k 10 200 python scripts/soak.py 150 30302 > gpurun_out/soak_r03b.log 2>&1; tail -3 gpurun_out/soak_r03b.log
