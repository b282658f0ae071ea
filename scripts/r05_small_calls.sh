#!/bin/bash
# Round 5: small LANCZOS calls -- the tile kernel's entries in the kernel arguments (MIC_RS_TILE_ARGS=1, shipped) against
# staged + uploaded (=0): wall of the reference-sized call and the contact sheet, the kernels' durations by rocprofv3.
# (The same script measured the eager form of profiles/r05_tile_eager.patch with MIC_RS_TILE_EAGER=0/512.)
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_small_calls
rm -rf $out && mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_march.py tests/test_next_rows.py tests/test_gpu_resident_layers.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -2 $out/pytest.log; [ $rc -eq 0 ] || exit 1
for pass in 1 2; do for e in 0 1; do
  MIC_RS_TILE_ARGS=$e python3 scripts/time_c1_cold.py 2>&1 | tail -1 | sed "s/^/args=$e  /" | tee -a $out/wall.txt
done; done
for e in 0 1; do
  MIC_RS_TILE_ARGS=$e timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$e -- python3 scripts/time_c1_cold.py > $out/trace_$e.log 2>&1 || { echo FAILED trace $e; tail -5 $out/trace_$e.log; exit 1; }
  python3 - <<PY | tee -a $out/kernels.txt
import csv, glob, re
rows = list(csv.DictReader(open(glob.glob("$out/trace_$e/*/*kernel_trace.csv")[0])))
by = {}
for r in rows:
    m = re.search(r"(resample_\w+(<[^>]*>)?|copyBuffer|composite_kernel)", r["Kernel_Name"])
    if m:
        k = m.group(1)
        by.setdefault((k, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"])), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, gx, gy), v in sorted(by.items()):
    v = sorted(v)
    print(f"args=$e  {k} grid {gx} x {gy}: {len(v)} launches, median {v[len(v)//2] / 1e3:.2f} us, min {v[0] / 1e3:.2f} us")
PY
done
