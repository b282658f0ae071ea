#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT names the copy of the repo on the GPU box}" || exit 1
# HBM-side bytes of the resample kernel on the placements workload (separate --pmc passes; FETCH_SIZE is
# halved on gfx950 for wide reads -- doubled below as MI355X_MICROARCH.md prescribes).
export TMPDIR=/tmp  # (already in the repo copy: line 2)
out=gpurun_out/prof_rs_traffic
rm -rf $out && mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 scripts/prof_placements.py > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 scripts/prof_placements.py > $out/w.log 2>&1
python3 - <<'PY'
import re, subprocess
def val(d, name):
    t = subprocess.run(["python3", "scripts/pmc_summary.py", d], capture_output=True, text=True).stdout
    m = re.search(r"resample_mfma_kernel[^\n]*\n\s+" + name + r"\s+([0-9.]+)", t)
    return float(m.group(1))
f, w = val("gpurun_out/prof_rs_traffic/f", "FETCH_SIZE"), val("gpurun_out/prof_rs_traffic/w", "WRITE_SIZE")
log = open("gpurun_out/prof_rs_traffic/f.log").read()
m = re.search(r"'source_pixels': (\d+)", log)
print(f"resample_mfma_kernel per launch: FETCH_SIZE {f:.0f} KB (x2 = {2*f*1024/1e6:.1f} MB read), WRITE_SIZE {w:.0f} KB ({w*1024/1e6:.1f} MB written)")
print("stats:", re.search(r"\{.*\}", log).group(0))
PY
