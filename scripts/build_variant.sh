#!/bin/bash
# Build an experimental variant of libmic.so next to the scripts (git-ignored *.bin; it travels to the GPU box):
#   scripts/build_variant.sh NAME [extra hipcc flags...]   ->  scripts/var_NAME.bin   (use with MIC_LIB=$PWD/scripts/var_NAME.bin)
set -e
name=$1; shift
cd "$(dirname "$0")/../image_transformation_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -mllvm -amdgpu-mfma-vgpr-form "$@" -o ../../scripts/var_$name.bin mic_api.hip kernels_composite.hip kernels_resample.hip \
  kernels_resample_tile.hip kernels_median.hip kernels_overlay.hip resample_coeffs.cpp flex_place.cpp png_encode.cpp png_decode.cpp
echo built scripts/var_$name.bin
