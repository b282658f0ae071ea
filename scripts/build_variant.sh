#!/bin/bash
# Build an experimental variant of libmic.so (git-ignored build/var_NAME.bin; it travels to the GPU box):
#   scripts/build_variant.sh NAME [-e 'sed expr' FILE]... [-p PATCH]... [extra hipcc flags...]
#       ->  build/var_NAME.bin   (use with MIC_LIB=$PWD/build/var_NAME.bin)
# The sources are copied to a scratch directory first; -e edits one copied file with sed, -p applies a patch
# (paths relative to the repo root) to the copy.  Intermediates stay under /tmp: nothing but the .bin reaches the tree.
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d /tmp/mic_var_XXXXXX)
trap 'rm -rf "$tmp"' EXIT
mkdir -p "$tmp/image_transformation_amd" "$tmp/include" "$root/build"
cp -r "$root/image_transformation_amd/csrc" "$tmp/image_transformation_amd/csrc"
cp "$root/include/mic.h" "$tmp/include/"
flags=()
while [ $# -gt 0 ]; do
  case "$1" in
    -e) sed -i -e "$2" "$tmp/image_transformation_amd/csrc/$3"; shift 3;;
    -p) (cd "$tmp" && patch -p1 -s < "$root/$2"); shift 2;;
    *) flags+=("$1"); shift;;
  esac
done
cd "$tmp/image_transformation_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -mllvm -amdgpu-mfma-vgpr-form "${flags[@]}" -o "$root/build/var_$name.bin" mic_api.hip kernels_composite.hip kernels_resample.hip \
  kernels_resample_lane.hip kernels_resample_tile.hip kernels_median.hip kernels_overlay.hip resample_coeffs.cpp flex_place.cpp png_encode.cpp png_decode.cpp
echo "built build/var_$name.bin"
