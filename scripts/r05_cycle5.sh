#!/bin/bash
# Round 5: one-wave workgroups for small single-canvas launches (tests + traces), the split cold-start stages
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_cycle5
rm -rf $out && mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit 1
for r in 1 2 3; do python3 bench.py --cold-start-child 2>/dev/null | tail -1 > $out/cold_$r.json; python3 -c "
import json; d=json.load(open('$out/cold_$r.json')); print({k:v for k,v in d.items() if k!='pillow_numpy'})"; done
MIC_CASES_JSON=$PWD/$out/single.cases.json MIC_ITERS=40 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/single -- python3 scripts/prof_single5.py > $out/single.log 2>&1 || { echo FAILED single; tail -5 $out/single.log; }
python3 scripts/trace_cases.py $out/single/*/*kernel_trace.csv $out/single.cases.json r05 | tee $out/single_summary.txt
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
    f.write("# C5 (audio_book at 7680x4320): median colour, contact sheet (tile resample + composite), 4 composites of x8 LANCZOS upscales (cold: resampled in every run)\n")
    for (k, gx, wx), v in sorted(by.items()):
        v = v[len(v) // 4:] or v
        f.write(f"{k} grid {gx} ({int(gx)//int(wx)} x {wx}): {len(v)} launches, mean {sum(v) / len(v) / 1e3:.2f} us, min {min(v) / 1e3:.2f} us, max {max(v) / 1e3:.2f} us\n")
print(open("$out/c5_kernel_trace.txt").read())
PY
