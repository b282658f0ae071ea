#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
# One GPU iteration: parity tests, microbench, kernel trace + PMC of the C3 batch composite.
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 300 python scripts/microbench.py 2>&1 | grep -v amdgpu.ids | grep -v -E "Traceback|File|plan.run|raise|ValueError|fn\(\)|report\(|timeit" | head -14
export TMPDIR=/tmp  # (already in the repo copy: line 2)
rm -rf gpurun_out/prof_kt2 gpurun_out/prof_pmc2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt2 -- python3 scripts/prof_composite.py > gpurun_out/prof_kt2.log 2>&1
cat gpurun_out/prof_kt2/*/*kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/prof_pmc2 -- python3 scripts/prof_composite.py > gpurun_out/prof_pmc2.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/prof_pmc2
