#!/bin/bash
# Round 5: the round-end sequence on one box -- GPU tests, smoke(), the default bench line
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_final
rm -rf $out && mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit 1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { echo FAILED smoke; tail -5 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
python3 bench.py > $out/bench.json 2> $out/bench.err || { echo FAILED bench; tail -5 $out/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric','value','unit','ms_per_step')}, d['roofline'], d.get('c4_strong',{}).get('verified'))"
