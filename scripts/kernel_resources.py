"""Registers, LDS and the compiler's occupancy figure of every kernel in libmic.so, from hipcc's own assembly
metadata (the evidence behind the occupancy statements in DESIGN.md; rocprofv3's VGPR_Count column prints half the
allocated vector registers on gfx950, e.g. 32 for the composite kernel's 63).
    python scripts/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_transformation_amd import build as b

def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return names

print("# " + " ".join(["hipcc", f"--offload-arch={b.ARCH}"] + b._flags()))
print(f"{'kernel':100s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'LDS(static)':>11s} {'scratch':>7s} {'waves/SIMD':>10s}")
with tempfile.TemporaryDirectory() as td:
    for src in b.SOURCES:
        if not src.endswith(".hip"):
            continue
        s_path = os.path.join(td, src + ".s")
        subprocess.check_call([b.hipcc(), f"--offload-arch={b.ARCH}"] + [f for f in b._flags() if f not in ("-shared", "-fPIC")] +
                              ["-S", "--cuda-device-only", "-o", s_path, os.path.join(b.CSRC, src)], stderr=subprocess.DEVNULL)
        text = open(s_path).read()
        rows = []
        for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M):
            name = m.group(1)
            blk = text[m.end():m.end() + 4000]
            def num(key):
                mm = re.search(r"; " + key + r": (\d+)", blk)
                return int(mm.group(1)) if mm else -1
            rows.append((name, num("NumVgprs"), num("NumAgprs"), num("TotalNumSgprs"), num("LDSByteSize"), num("ScratchSize"), num("Occupancy")))
        for nm, row in zip(demangle([r[0] for r in rows]), rows):
            print(f"{nm[:100]:100s} {row[1]:5d} {row[2]:5d} {row[3]:5d} {row[4]:11d} {row[5]:7d} {row[6]:10d}")
print("# resample_march_kernel / resample_tile_kernel take dynamic LDS on top (mic_api.hip: rs_march_lds_bytes, rs_tile_lds_bytes);")
print("# with the 28.7 KB of the C3 placements call the marching kernel runs 5 workgroups (20 waves) per CU.")
