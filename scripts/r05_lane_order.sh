#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_order; rm -rf $out; mkdir -p $out
# (MIC_LANE_ORDER=0: chunks dealt in the order of the cut; 1, the default since: by layer, rows, columns for whole-round launches;
# the first run of this script used a variant build that also took row quanta of 4 and 8 tiles: level with 1)
for p in 1 2; do for o in 0 1; do echo "== MIC_LANE_ORDER=$o"; MIC_LANE_ORDER=$o python3 scripts/time_resample_cold.py 2>&1 | grep "^C3\|^16-canv\|^12 layers"; done; done | tee $out/times.txt
for o in 0 1; do
  MIC_LANE_ORDER=$o MIC_ITERS=12 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$o -- python3 scripts/prof_placements.py > $out/fetch_$o.log 2>&1
  python3 scripts/pmc_summary.py $out/fetch_$o 2>&1 | grep -i "lane\|composite" | head -4 | sed "s/^/order=$o /"
done | tee $out/fetch.txt
