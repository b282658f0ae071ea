#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
export TMPDIR=/tmp  # (already in the repo copy: line 2)
for r in ${RATIOS:-2 3}; do
  rm -rf gpurun_out/pc4p
  MIC_RATIO=$r rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/pc4p -- python3 scripts/prof_c4.py > gpurun_out/pc4p.log 2>&1
  echo "== ratio $r alpha ${MIC_ALPHA:-binary}"; python3 scripts/pmc_summary.py gpurun_out/pc4p | grep -A9 composite
done
