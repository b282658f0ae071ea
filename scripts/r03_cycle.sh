#!/bin/bash
# one GPU iteration of round 3: GPU tests, the bench line at the driver's flags, PNG writer and median timings
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r03_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { tail -20 gpurun_out/r03_bench.err; exit 1; }
python - <<'PY'
import json
r = json.load(open("gpurun_out/r03_bench.json"))
for k in ("value", "ms_per_step", "c4_strong", "single_canvas", "c5_end_to_end", "contact_sheet", "run_layouts", "c1_bundle_dropin", "pil_dropin", "median_noise", "cpu_baseline", "cpu_baseline_all_cores", "placements_mode_lanczos", "placements_mode_lanczos_batch"):
    print(k, json.dumps(r.get(k))[:1500])
print("roofline", json.dumps(r["roofline"])[:600])
PY
python scripts/png_bench.py > gpurun_out/r03_png.txt 2>&1; cat gpurun_out/r03_png.txt
python scripts/time_median.py > gpurun_out/r03_median.txt 2>&1; cat gpurun_out/r03_median.txt
MIC_MEDIAN_TWO_LAUNCHES=1 python scripts/time_median.py > gpurun_out/r03_median_two.txt 2>&1; echo "--- two launches"; cat gpurun_out/r03_median_two.txt
