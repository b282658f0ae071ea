#!/bin/bash
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('   ms_per_step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms'])"; }
echo "baseline"; run; run
for v in SALU_64 SALU_128 VALU_64 VALU_128; do echo "$v"; MIC_LIB=$PWD/scripts/libmic_exp_$v.bin run; done
