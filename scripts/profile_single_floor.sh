#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace) of single-canvas composite launches against the page-writer floor
# (build/ubench_page.bin) and, when build/var_cabl1.bin / var_cabl2.bin (scripts/build_variant.sh with the sed edits of profiles/r05_single_canvas_floor.txt) exist, the stage-skipping builds.
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof_single_floor
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/floor -- build/ubench_page.bin > $out/floor.log 2>&1
for v in full cabl1 cabl2; do
  if [ $v = full ]; then unset MIC_LIB; elif [ -f build/var_$v.bin ]; then export MIC_LIB=$PWD/build/var_$v.bin; else continue; fi
  rocprofv3 --kernel-trace --output-format csv -d $out/$v -- python3 scripts/prof_single.py > $out/$v.log 2>&1
done
unset MIC_LIB
python3 - <<PY
import csv, glob
for name in ("floor", "full", "cabl1", "cabl2"):
    g = glob.glob("$out/%s/*/*kernel_trace.csv" % name)
    if not g: continue
    by = {}
    for r in csv.DictReader(open(g[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "page_kernel" in k or "composite_kernel" in k:
            by.setdefault((k, r["Grid_Size_X"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("==", name)
    for (k, gx), v in sorted(by.items(), key=lambda kv: (int(kv[0][1]), kv[0][0])):
        v = sorted(v[len(v) // 4:] or v)
        print(f"  {k} grid {gx}: n {len(v)} median {v[len(v) // 2] / 1e3:.2f} us min {v[0] / 1e3:.2f} mean {sum(v) / len(v) / 1e3:.2f}")
PY
