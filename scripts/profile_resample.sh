#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
# rocprofv3 view of the marching resample kernel on the C3 placements workload (32 LANCZOS layers per canvas).
# usage: scripts/profile_resample.sh [outdir]   (MIC_ALPHA=soft|binary)
export TMPDIR=/tmp  # (already in the repo copy: line 2)
out=${1:-gpurun_out/prof_rs}
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 scripts/prof_placements.py > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
cut -d, -f1-4 $out/kt/*/*kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pmc1 -- python3 scripts/prof_placements.py > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc2 -- python3 scripts/prof_placements.py > $out/pmc2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/pmc3 -- python3 scripts/prof_placements.py > $out/pmc3.log 2>&1
for d in pmc1 pmc2 pmc3; do python3 scripts/pmc_summary.py $out/$d | grep -A12 "resample\|planarize"; done
