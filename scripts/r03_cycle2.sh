#!/bin/bash
for b in 512 384 320 256 192 128; do for t in 0 1; do echo "--- max blocks $b two_launches $t"; MIC_MEDIAN_MAX_BLOCKS=$b MIC_MEDIAN_TWO_LAUNCHES=$t python scripts/time_median.py 2>&1 | grep -E "^(1080p|4k|8k) +noise"; done; done
