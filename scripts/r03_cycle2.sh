#!/bin/bash
mkdir -p gpurun_out
MIC_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 4 --workload c4 --steps 4 --warmup 1 > gpurun_out/r03_rehearsal4.json 2> gpurun_out/r03_rehearsal4.err; echo "rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/r03_rehearsal4.json')); print(r['ranks'], r['backend'], r['config']['canvases_per_step_total'], r['config']['canvases_per_step_per_gpu'], r['per_rank']['canvas_sizes'], r['value'])"
MIC_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 4 --steps 4 --warmup 1 --batch 4 > gpurun_out/r03_rehearsal4w.json 2>> gpurun_out/r03_rehearsal4.err; echo "rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/r03_rehearsal4w.json')); print(r['ranks'], r['scaling'], r['value'], r['c4_strong']['canvases_per_rank'], r['c4_strong']['value'], [d['rank'] for d in r['devices']])"
tail -3 gpurun_out/r03_rehearsal4.err
