#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r03_sq; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $out/p1 -- python3 scripts/prof_placements.py > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $out/p2 -- python3 scripts/prof_placements.py > $out/p2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $out/b1 -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/b1.log 2>&1
for d in p1 p2; do python3 scripts/pmc_summary.py $out/$d; done > $out/placements_pmc_summary.txt 2>&1
python3 scripts/pmc_summary.py $out/b1 > $out/bench_pmc_summary.txt 2>&1
grep -A9 "resample_march\|composite_kernel" $out/placements_pmc_summary.txt | head -60; grep -A9 "composite_kernel" $out/bench_pmc_summary.txt | head -24
