#!/bin/bash
# final verification of a round: GPU tests, smoke, soaks (default routing and forced marching), bench at the driver's flags
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_tests_final.log 2>&1; rc=$?; tail -3 gpurun_out/r03_tests_final.log; [ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 280 python scripts/soak.py 200 30321 > gpurun_out/soak_r03_final_a.log 2>&1; tail -1 gpurun_out/soak_r03_final_a.log
MIC_RS_MARCH_MIN_UNITS=0 timeout -k 10 280 python scripts/soak.py 200 30322 > gpurun_out/soak_r03_final_b.log 2>&1; tail -1 gpurun_out/soak_r03_final_b.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_flags.json 2> gpurun_out/r03_bench.err || { tail -20 gpurun_out/r03_bench.err; exit 1; }
python -c "
import json; r=json.load(open('gpurun_out/r03_bench_driver_flags.json'))
print(r['value'], r['roofline']['frac'], r['roofline']['frac_rocprof'], r['c4_strong']['value'], r['c1_bundle_dropin']['identity_scale'], r['run_layouts']['ms_with_png_artifacts'], r['pil_dropin'])"
