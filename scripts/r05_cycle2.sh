#!/bin/bash
# Round 5: GPU tests, the default bench line (verification keys), C5 / sheet kernel trace -> gpurun_out/r05_cycle2/
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_cycle2
rm -rf $out && mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
r = json.loads([l for l in open("$out/bench.json") if l.startswith("{")][0])
print("value", r["value"], "frac", r["roofline"]["frac"], "verified", r["roofline"].get("verified"), r["roofline"].get("verification"))
print("c4", r["c4_strong"]["value"], r["c4_strong"].get("verified"), r["c4_strong"].get("verified_canvases"))
PY
MIC_ITERS=12 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -- python3 scripts/prof_c5.py > $out/c5.log 2>&1 || { echo FAILED c5; tail -5 $out/c5.log; }
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$out/c5/*/*kernel_trace.csv")[0])))
by = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if any(p in k for p in ["composite_kernel", "resample", "median", "planarize"]):
        by.setdefault((k, r["Grid_Size_X"], r["Workgroup_Size_X"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$out/c5_kernel_trace.txt", "w") as f:
    f.write("# C5 (audio_book at 7680x4320): median colour, contact sheet (tile resample + composite), 4 composites of x8 LANCZOS upscales\n")
    for (k, gx, wx), v in sorted(by.items()):
        v = v[len(v) // 4:] or v
        f.write(f"{k} grid {gx} ({int(gx)//int(wx)} x {wx}): {len(v)} launches, mean {sum(v) / len(v) / 1e3:.2f} us, min {min(v) / 1e3:.2f} us, max {max(v) / 1e3:.2f} us\n")
print(open("$out/c5_kernel_trace.txt").read())
PY
