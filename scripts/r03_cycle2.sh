#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r03_tests.log; [ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_flags.json 2> gpurun_out/r03_bench.err || { tail -20 gpurun_out/r03_bench.err; exit 1; }
python - <<'PY'
import json
r = json.load(open("gpurun_out/r03_bench_driver_flags.json"))
for k in ("value", "ms_per_step", "ranks", "backend", "c1_bundle_dropin", "run_layouts", "cpu_baseline_all_cores"):
    print(k, json.dumps(r.get(k))[:900])
print("roofline", r["roofline"]["frac"], r["roofline"]["kernel_ms"], r["roofline"]["kernel_ms_rocprof"], r["roofline"]["frac_rocprof"])
print("c4_strong", r["c4_strong"]["value"], r["c4_strong"]["roofline_rank0"])
PY
python scripts/prof_c1.py > /dev/null 2>&1; cp gpurun_out/c1_breakdown.json gpurun_out/r03_c1_breakdown.json; python -c "
import json; d=json.load(open('gpurun_out/r03_c1_breakdown.json')); print({k:d[k] for k in ('whole_call_us','r03_rows_solid_scan_us','r03_composite_one_enqueue_plus_wait_us','r03_device_only_composite_one_us')})"
python scripts/prof_run_layouts_save.py 2>&1 | head -3
