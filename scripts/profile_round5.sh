#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
# rocprofv3 evidence of round 5 -> gpurun_out/prof_r05 (copied into profiles/r05_* afterwards): kernel traces + HBM byte
# counters (FETCH_SIZE, WRITE_SIZE in separate --pmc passes) for bench.py's headline batch, the C4 strong-scaling leg,
# the placements canvas (resample + composite), single-canvas launches, C5 at 8K and the contact sheet.
tag=r05
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp  # (already in the repo copy: line 2)
kt() {  # kt <name> <python args...>: kernel trace + stats
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/kt -- python3 "$@" > $out/$name.kt.log 2>&1 || { echo "FAILED kt $name"; tail -5 $out/$name.kt.log; return 1; }
  cp $out/$name/kt/*/*kernel_stats.csv $out/$name.kernel_stats.csv
  echo "== $name"; cut -d, -f1-4 $out/$name.kernel_stats.csv | head -12
}
pmc() {  # pmc <name> <python args...>: FETCH_SIZE and WRITE_SIZE passes
  name=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$name/fetch -- python3 "$@" > $out/$name.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$name/write -- python3 "$@" > $out/$name.write.log 2>&1
}
kt bench bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras && grep -h '^{' $out/bench.kt.log > $out/bench_under_rocprof.json
pmc bench bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras
python3 scripts/traffic_json.py composite_kernel $out/bench/fetch $out/bench/write $out/bench.kernel_stats.csv 786809664 $out/hbm_traffic.json "bench.py headline batch: 16 canvases, one shared 16 MB atlas (the c4_strong leg's 64-canvas launches are in the same trace)"
kt c4 bench.py --workload c4 --steps 30 --warmup 5
MIC_ITERS=12 kt placements scripts/prof_placements.py
MIC_ITERS=12 MIC_WARM=1 kt placements_warm scripts/prof_placements.py
MIC_ITERS=12 pmc placements scripts/prof_placements.py
python3 scripts/traffic_json.py resample_lane_kernel $out/placements/fetch $out/placements/write $out/placements.kernel_stats.csv 107305248 $out/resample_traffic.json "32 LANCZOS layers of the C3 placements canvas (soft alpha): 53.0 MB of cutouts in, 54.3 MB of resampled layers out"
python3 scripts/traffic_json.py composite_kernel $out/placements/fetch $out/placements/write $out/placements.kernel_stats.csv 85370000 $out/placements_composite_traffic.json "composite of the 32 resampled layers onto one 4K canvas"
kt single scripts/prof_single.py
kt c5 scripts/prof_c5.py
pmc c5 scripts/prof_c5.py
python3 - <<PY
import csv, glob
def trace(name, pats):
    rows = list(csv.DictReader(open(glob.glob("$out/%s/kt/*/*kernel_trace.csv" % name)[0])))
    by = {}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if any(p in k for p in pats):
            by.setdefault((k, r["Grid_Size_X"], r.get("Grid_Size_Y", "")), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines = []
    for (k, gx, gy), v in sorted(by.items()):
        v = v[len(v) // 4:] or v
        lines.append(f"{k} grid {gx}x{gy}: {len(v)} launches, mean {sum(v) / len(v) / 1e3:.2f} us, min {min(v) / 1e3:.2f} us, max {max(v) / 1e3:.2f} us")
    return lines
with open("$out/single_canvas_kernel_trace.txt", "w") as f:
    f.write("\n".join(trace("single", ["composite_kernel"])) + "\n")
with open("$out/c5_kernel_trace.txt", "w") as f:
    f.write("# C5 (audio_book at 7680x4320): median colour, contact sheet (tile resample + composite), 4 composites of x8 LANCZOS upscales\n")
    f.write("\n".join(trace("c5", ["composite_kernel", "resample", "median", "planarize"])) + "\n")
print(open("$out/single_canvas_kernel_trace.txt").read()); print(open("$out/c5_kernel_trace.txt").read())
PY
python3 scripts/pmc_summary.py $out/c5/fetch > $out/c5_fetch_summary.txt 2>&1; python3 scripts/pmc_summary.py $out/c5/write > $out/c5_write_summary.txt 2>&1
cat $out/hbm_traffic.json $out/resample_traffic.json $out/placements_composite_traffic.json | head -100
