#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_next_rows.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python scripts/prof_run_layouts_save.py 2>&1 | head -14
python scripts/png_bench.py 2>&1 | tail -14
