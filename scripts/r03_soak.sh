#!/bin/bash
# Differential soaks of the final library: HIP path vs the oracle on random composites / PIL drop-ins / medians / resizes.
mkdir -p gpurun_out
( timeout -k 10 330 python scripts/soak.py 270 20261004 > gpurun_out/soak_a.log 2>&1 ) &&
( MIC_RS_MARCH_MIN_UNITS=0 timeout -k 10 330 python scripts/soak.py 270 777 > gpurun_out/soak_b.log 2>&1 ) &&
( timeout -k 10 330 python scripts/soak.py 270 31337 > gpurun_out/soak_c.log 2>&1 ) &&
( MIC_RS_MARCH_MIN_UNITS=0 MIC_LAYER_ARGS=0 timeout -k 10 330 python scripts/soak.py 270 424242 > gpurun_out/soak_d.log 2>&1 )
rc=$?
tail -n 1 gpurun_out/soak_a.log gpurun_out/soak_b.log gpurun_out/soak_c.log gpurun_out/soak_d.log
exit $rc
