#!/bin/bash
# ad-hoc GPU cycle (rewritten per experiment): GPU parity tests, then whatever is being measured
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/cyc_tests.log 2>&1
tail -2 gpurun_out/cyc_tests.log
