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
SOURCES = ["mic_api.hip", "kernels_composite.hip", "kernels_resample.hip", "kernels_resample_lane.hip", "kernels_resample_tile.hip", "kernels_median.hip",
           "kernels_overlay.hip",
           "resample_coeffs.cpp", "flex_place.cpp", "png_encode.cpp", "png_decode.cpp"]
HEADERS = ["mic_internal.h", "lane_unit.h", "lane_partition.h", "resample_mfma.h", "resample_coeffs.h", "flex_place.h", "png_encode.h", "png_decode.h", "png_checksum.h", "host_pool.h", os.path.join("..", "..", "include", "mic.h")]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


DIGEST = LIB + ".digest"  # hex SHA-256 of what the library was built from (travels with the .so)


def _flags() -> list:
    return ["-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
            # MFMA accumulators in VGPRs: the resample kernel's epilogues read every accumulator with
            # VALU instructions, which cannot address AGPRs (one v_accvgpr_read per value otherwise)
            "-mllvm", "-amdgpu-mfma-vgpr-form"] + os.environ.get("MIC_EXTRA_CFLAGS", "").split()


def _source_digest() -> str:
    import hashlib
    h = hashlib.sha256()
    h.update((ARCH + " " + " ".join(_flags())).encode())
    for rel in SOURCES + HEADERS:
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read())
    return h.hexdigest()


def _stale() -> bool:
    """Is libmic.so missing or built from other sources / flags than the tree holds?  Decided by content (the digest
    written next to the library), not by file times: a copy of the tree need not keep them in order."""
    if not os.path.exists(LIB):
        return True
    if not all(os.path.exists(os.path.join(CSRC, rel)) for rel in SOURCES + HEADERS):
        return False  # a source-less (installed) tree: nothing to compare with, the library is what there is
    try:
        with open(DIGEST, "r", encoding="ascii") as f:
            return f.read().strip() != _source_digest()
    except OSError:
        pass
    t = os.path.getmtime(LIB)  # a library from before the digest existed: file times
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    import fcntl
    # one builder at a time (the ranks of a multi-GPU job import the package together); the others wait, find the
    # library fresh and return.  The library is written under another name and renamed into place: a process that
    # has the old one mapped keeps it, nobody ever sees a half-written file.
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return LIB
        digest = _source_digest()
        tmp = f"{LIB}.tmp.{os.getpid()}"
        cmd = [hipcc(), f"--offload-arch={ARCH}"] + _flags() + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        with open(DIGEST + ".tmp", "w", encoding="ascii") as f:
            f.write(digest + "\n")
        os.replace(DIGEST + ".tmp", DIGEST)
    return LIB


def check_isa(verbose: bool = False) -> None:
    """Build-time canary of the v_ashr_pk_u8_i32 workaround (kernels_resample.hip: clip8), on the CPU: compile the
    resample sources to gfx950 assembly and look at what the compiler made of them.  hipcc 7.2 fuses "shift, clamp to
    0..255, pack" into v_ashr_pk_u8_i32 and then treats the untouched upper half of its destination as zero; clip8's
    empty asm keeps the two-pass kernels on v_med3_i32 instead, and the MFMA kernels only use the instruction through
    its builtin with the result cut to 16 bits.  If an update of the compiler brings the fused form back into the
    two-pass kernels, or drops the instruction from the MFMA epilogues (their measured cost model), this raises --
    the device-side known-answer test (mic_selftest, run by smoke()) is the check of the arithmetic itself."""
    import re
    import tempfile
    flags = [f for f in _flags() if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "resample.s")
        subprocess.check_call([hipcc(), f"--offload-arch={ARCH}"] + flags + ["-S", "--cuda-device-only", "-o", out,
                                                                               os.path.join(CSRC, "kernels_resample.hip")],
                              stderr=subprocess.DEVNULL)
        with open(out, encoding="utf-8", errors="replace") as f:
            text = f.read()
    bodies = {}
    for m in re.finditer(r"^(_ZN3mic\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    two_pass = [b for n, b in bodies.items() if "resample_h_kernel" in n or "resample_v_kernel" in n]
    march = [b for n, b in bodies.items() if "resample_march_kernel" in n]
    if len(two_pass) != 2 or len(march) != 1:
        raise RuntimeError(f"check_isa: expected resample_h/v/march kernels in the assembly, found {sorted(bodies)}")
    for b in two_pass:
        if "v_ashr_pk_u8_i32" in b or "v_med3_i32" not in b:
            raise RuntimeError("check_isa: the two-pass resample kernels no longer clamp with v_med3_i32 -- the compiler's own "
                               "v_ashr_pk_u8_i32 fusion (which ORs a stale upper half, ROCm 7.2) is back: see clip8")
    if march[0].count("v_ashr_pk_u8_i32") < 8 or march[0].count("v_ashr_pk_i8_i32") < 8:
        raise RuntimeError("check_isa: the marching kernel's epilogues lost their v_ashr_pk_{u8,i8}_i32 (clip8x4)")
    # the lane kernel (round 5) uses the same helpers (resample_mfma.h); it must not spill either: its register budget
    # (118 of the 128 VGPRs that four waves per SIMD allow) is what its structure was chosen for
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "lane.s")
        subprocess.check_call([hipcc(), f"--offload-arch={ARCH}"] + flags + ["-S", "--cuda-device-only", "-o", out,
                                                                               os.path.join(CSRC, "kernels_resample_lane.hip")],
                              stderr=subprocess.DEVNULL)
        with open(out, encoding="utf-8", errors="replace") as f:
            text = f.read()
    m = re.search(r"^(_ZN3mic20resample_lane_kernel\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M)
    if not m:
        raise RuntimeError("check_isa: resample_lane_kernel not found in the assembly")
    lane = m.group(2)
    # (six bodies: one or two x-tiles x general / keeps width / keeps height: 288 + 192 + 192 MFMAs)
    if lane.count("v_ashr_pk_u8_i32") < 48 or lane.count("v_ashr_pk_i8_i32") < 48 or lane.count("v_mfma_i32_16x16x64_i8") < 600:
        raise RuntimeError("check_isa: the lane kernel lost its MFMA chains / v_ashr_pk_{u8,i8}_i32 epilogues")
    if "scratch_" in lane:
        raise RuntimeError("check_isa: the lane kernel spills to scratch (its band loop then waits on scratch traffic: 62 us measured)")
    if verbose:
        print("check_isa ok: two-pass kernels clamp with v_med3_i32, MFMA epilogues use v_ashr_pk_{u8,i8}_i32 via the builtin, "
              "the lane kernel does not spill")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
