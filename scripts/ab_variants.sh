#!/bin/bash
# A/B of compile-time variants of libmic.so on the GPU box: rebuilds there (hipcc is on the box).
for v in "-DMIC_HOT_WAVES=5" "-DMIC_HOT_WAVES=6" "-DMIC_HOT_WAVES=7" "-DMIC_HOT_WAVES=8"; do
  MIC_EXTRA_CFLAGS="$v" python -m image_transformation_amd.build --force > /dev/null 2>&1
  echo "== variant [$v]"
  timeout -k 10 300 python scripts/microbench.py 2>&1 | grep -E "composite C3 flex binary|composite 0 layers \(fill\)"
  timeout -k 10 300 python scripts/microbench.py 2>&1 | grep -E "composite C3 flex binary"
done
MIC_EXTRA_CFLAGS="" python -m image_transformation_amd.build --force > /dev/null 2>&1
