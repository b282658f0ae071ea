#!/bin/bash
# Round 5: the lane resample kernel in the library: GPU tests, then rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE (separate
# passes) + instruction counters of the C3 placements canvas (soft and binary alpha), lane and marching kernel side by side
# -> gpurun_out/r05_lane_prof/
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_lane_prof
rm -rf $out && mkdir -p $out
if [ "${TESTS:-1}" = 1 ]; then
  timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest.log
  [ $rc -eq 0 ] || exit 1
fi
kt() {  # kt <name> <env...>: kernel trace + stats of prof_placements.py
  name=$1; shift
  env "$@" MIC_ITERS=16 true
  ( export "$@" MIC_ITERS=16; rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/kt -- python3 scripts/prof_placements.py > $out/$name.kt.log 2>&1 ) || { echo "FAILED kt $name"; tail -5 $out/$name.kt.log; return 1; }
  cp $out/$name/kt/*/*kernel_stats.csv $out/$name.kernel_stats.csv
  echo "== $name"; cut -d, -f1-4 $out/$name.kernel_stats.csv | grep -i "resample\|composite\|planar"
}
pmc() {  # pmc <name> <pass> <counters...> with the env of the caller
  name=$1; pass=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d $out/$name/$pass -- python3 scripts/prof_placements.py > $out/$name.$pass.log 2>&1 || { echo "FAILED $name $pass"; tail -3 $out/$name.$pass.log; }
}
kt lane_soft MIC_ALPHA=soft
kt march_soft MIC_ALPHA=soft MIC_RS_LANE=0
kt lane_binary MIC_ALPHA=binary
kt march_binary MIC_ALPHA=binary MIC_RS_LANE=0
export MIC_ITERS=12 MIC_ALPHA=soft
pmc lane_soft fetch FETCH_SIZE
pmc lane_soft write WRITE_SIZE
pmc lane_soft pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
pmc lane_soft pmc2 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
python3 scripts/traffic_json.py resample_lane_kernel $out/lane_soft/fetch $out/lane_soft/write $out/lane_soft.kernel_stats.csv 107305248 $out/lane_resample_traffic.json "32 LANCZOS layers of the C3 placements canvas (soft alpha) through the lane kernel: 53.0 MB of cutouts in, 54.3 MB of resampled layers out"
for d in pmc1 pmc2 fetch write; do python3 scripts/pmc_summary.py $out/lane_soft/$d | grep -A10 "resample_lane\|planarize_tiled" | tee -a $out/lane_soft_pmc_summary.txt; done
cat $out/lane_resample_traffic.json
