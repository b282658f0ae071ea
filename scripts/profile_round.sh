#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
# rocprofv3 evidence of a round: kernel traces and HBM byte counters (FETCH_SIZE and WRITE_SIZE in separate --pmc
# passes: they do not fit one pass on gfx950) for (1) bench.py's headline batch, (2) its cold-inputs leg, (3) the
# placements-mode canvas (resample + composite), (4) single-canvas launches.  Usage: scripts/profile_round.sh r02
tag=${1:-r02}
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp  # (already in the repo copy: line 2)
run() {  # run <name> <python args...>: kernel trace + FETCH + WRITE passes
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/kt -- python3 "$@" > $out/$name.kt.log 2>&1 || { echo "FAILED kt $name"; tail -5 $out/$name.kt.log; return 1; }
  cp $out/$name/kt/*/*kernel_stats.csv $out/$name.kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$name/fetch -- python3 "$@" > $out/$name.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$name/write -- python3 "$@" > $out/$name.write.log 2>&1
  echo "== $name"; cut -d, -f1-4 $out/$name.kernel_stats.csv
}
run bench bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras && grep -h '^{' $out/bench.kt.log > $out/bench_under_rocprof.json
python3 scripts/traffic_json.py composite_kernel $out/bench/fetch $out/bench/write $out/bench.kernel_stats.csv 786809664 $out/hbm_traffic.json "bench.py headline batch: 16 canvases, one shared 16 MB atlas"
run cold scripts/prof_cold.py
python3 scripts/traffic_json.py composite_kernel $out/cold/fetch $out/cold/write $out/cold.kernel_stats.csv 786809664 $out/cold_traffic.json "cold inputs: an atlas per canvas, two input sets and three output sets rotating"
MIC_ITERS=12 run placements scripts/prof_placements.py
python3 scripts/traffic_json.py resample_march_kernel $out/placements/fetch $out/placements/write $out/placements.kernel_stats.csv 107305248 $out/resample_traffic.json "32 LANCZOS layers of the C3 placements canvas (soft alpha): 53.0 MB of cutouts in, 54.3 MB of resampled layers out"
python3 scripts/traffic_json.py composite_kernel $out/placements/fetch $out/placements/write $out/placements.kernel_stats.csv 85370000 $out/placements_composite_traffic.json "composite of the 32 resampled layers onto one 4K canvas"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $out/placements/sq1 -- python3 scripts/prof_placements.py > $out/placements.sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $out/placements/sq2 -- python3 scripts/prof_placements.py > $out/placements.sq2.log 2>&1
for d in sq1 sq2; do python3 scripts/pmc_summary.py $out/placements/$d; done > $out/placements_pmc_summary.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/single/kt -- python3 scripts/prof_single.py > $out/single.kt.log 2>&1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$out/single/kt/*/*kernel_trace.csv")[0])))
by = {}
for r in rows:
    if "composite_kernel" in r["Kernel_Name"]:
        by.setdefault(r["Grid_Size_X"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$out/single_canvas_kernel_trace.txt", "w") as f:
    for g, v in by.items():
        v = v[len(v) // 4:]
        f.write(f"composite_kernel<..., one job> grid.x {g}: {len(v)} launches, mean {sum(v) / len(v) / 1e3:.2f} us, min {min(v) / 1e3:.2f} us, max {max(v) / 1e3:.2f} us\n")
print(open("$out/single_canvas_kernel_trace.txt").read())
PY
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $out/bench/sq -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $out/bench.sq.log 2>&1
python3 scripts/pmc_summary.py $out/bench/sq > $out/bench_pmc_summary.txt 2>&1
cat $out/hbm_traffic.json $out/cold_traffic.json $out/resample_traffic.json $out/placements_composite_traffic.json; cat $out/bench_pmc_summary.txt | head -30
