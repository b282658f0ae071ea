#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_c_abi.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('   ms_per_step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms'], 'c4', r['c4_strong']['ms_per_step'] if 'c4_strong' in r else None)"; }
for i in 1 2 3; do echo "new"; run; echo "prev"; MIC_LIB=$PWD/scripts/libmic_prev.bin run; done
