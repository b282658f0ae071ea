#!/bin/bash
# A/B on the bench's own legs: a previous build (scripts/var_old.bin: build the commit to compare against with scripts/build_variant.sh from a
# checkout of it) vs the tree's library
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c_abi.py tests/test_gpu_march.py -m gpu -x -q > gpurun_out/cyc_tests.log 2>&1
tail -1 gpurun_out/cyc_tests.log
grep -q "failed\|error" gpurun_out/cyc_tests.log && exit 1
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('   soft', r['placements_mode_lanczos']['resample_ms'], 'binary', r['placements_mode_lanczos_binary_cutouts']['resample_ms'], 'batch16/canvas', r['placements_mode_lanczos_batch']['resample_ms_per_canvas'])"; }
for i in 1 2; do echo new; run; echo old; MIC_LIB=$PWD/scripts/var_old.bin run; done
