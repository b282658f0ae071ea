#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
# Kernel-only roofline fraction of the composite kernel per C4 aspect ratio (rocprofv3 kernel trace).
export TMPDIR=/tmp  # (already in the repo copy: line 2)
for r in 0 1 2 3; do
  rm -rf gpurun_out/pc4
  MIC_RATIO=$r rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc4 -- python3 scripts/prof_c4.py > gpurun_out/pc4.log 2>&1 || { tail -5 gpurun_out/pc4.log; exit 1; }
  python3 - <<'PY'
import csv, glob, re
log = open("gpurun_out/pc4.log").read()
m = re.search(r"ratio (\S+) \((\d+), (\d+)\) alg_bytes (\d+)", log)
for f in glob.glob("gpurun_out/pc4/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "composite_kernel" in r["Name"]:
            ns = float(r["AverageNs"])
            print(f"{m.group(1):6s} {m.group(2)}x{m.group(3)} {r['Name'][:48]:48s} avg {ns/1e3:7.1f} us  frac {int(m.group(4))/ns/8000:.3f}")
PY
done
