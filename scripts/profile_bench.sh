#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
# rocprofv3 evidence for bench.py's roofline line: kernel-trace stats, then HBM byte counters in
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
# Usage (on the GPU box, from the repo root): scripts/profile_bench.sh <tag>
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp  # (already in the repo copy: line 2)
args="bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $args > $out/kt.log 2>&1
cp $out/kt/*/*kernel_stats.csv $out/kernel_stats.csv
grep -h '^{' $out/kt.log > $out/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $args > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $args > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -- python3 $args > $out/pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_tcc -- python3 $args > $out/pmc_tcc.log 2>&1
for d in pmc_fetch pmc_write pmc_sq pmc_tcc; do echo "== $d"; python3 scripts/pmc_summary.py $out/$d; done > $out/pmc_summary.txt 2>&1
cat $out/kernel_stats.csv; cat $out/pmc_summary.txt; cat $out/bench_under_rocprof.json
