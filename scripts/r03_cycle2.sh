#!/bin/bash
for i in 1 2; do for u in 0 6500 8000 9500 10600 12000; do MIC_N=100 MIC_ALPHAS=soft,binary MIC_RS_UNIT_PX=$u timeout -k 10 120 python scripts/time_placements.py || exit 1; done; done
