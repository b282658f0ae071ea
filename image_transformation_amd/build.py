"""Build libmic.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m image_transformation_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmic.so")
SOURCES = ["mic_api.hip", "kernels_composite.hip", "kernels_resample.hip", "kernels_resample_tile.hip", "kernels_median.hip",
           "kernels_overlay.hip",
           "resample_coeffs.cpp", "flex_place.cpp"]
HEADERS = ["mic_internal.h", "resample_coeffs.h", "flex_place.h", os.path.join("..", "..", "include", "mic.h")]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
           # MFMA accumulators in VGPRs: the resample kernel's epilogues read every accumulator with
           # VALU instructions, which cannot address AGPRs (one v_accvgpr_read per value otherwise)
           "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-o", LIB] + os.environ.get("MIC_EXTRA_CFLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
