#!/bin/bash
python -m pytest tests/test_next_rows.py -m gpu -x -q 2>&1 | tail -2
python scripts/time_run_layouts.py 2>&1 | grep -v amdgpu.ids
python scripts/time_run_layouts.py 2>&1 | grep -v amdgpu.ids
