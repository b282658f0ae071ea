#!/bin/bash
# one default bench line per call (a fresh box each): the box-to-box spread of the headline
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
mkdir -p gpurun_out/r05_box
n=$(date +%s)
python3 bench.py --no-extras 2>/dev/null | tail -1 > gpurun_out/r05_box/bench_$n.json
python3 -c "
import json; d=json.load(open('gpurun_out/r05_box/bench_$n.json')); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'verified', r.get('verified'), 'cpu', d.get('cpu_baseline',{}).get('value'))"
