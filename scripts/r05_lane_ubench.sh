#!/bin/bash
# Round 5: the lane resample kernel's timing harness -> gpurun_out/r05_lane/
#   scripts/r05_lane_ubench.sh "<variant> <args...>" ...     each run twice
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_lane
mkdir -p $out
for spec in "$@"; do
  set -- $spec; v=$1; shift
  for r in 1 2; do timeout -k 5 60 build/ubl_$v.bin "$@" | grep -v checksum | sed "s/^/$v $*: /" | tee -a $out/summary.txt || exit 1; done
done
if [ -n "${PMC:-}" ]; then
  pm() { n=$1; shift; timeout -k 10 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$n -- build/ubl_base.bin $PMC > $out/$n.log 2>&1 || { echo "FAILED $n"; tail -3 $out/$n.log; }; }
  rm -rf $out/pmc1 $out/pmc2 $out/pmc3
  pm pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
  pm pmc2 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
  for d in pmc1 pmc2; do python3 scripts/pmc_summary.py $out/$d | grep -A10 "resample_lane" | tee -a $out/pmc_summary.txt; done
fi
