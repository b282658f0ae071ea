"""The host-side C++ of libmic that eats untrusted text (the Flex JSON parser/placer behind mic_render /
mic_flex_place) and builds the resample tables, compiled with g++ -fsanitize=address,undefined and
driven with every fixture tree plus thousands of byte-level mutations of them (truncations, flips,
insertions of structural characters, deep nesting).  No crash, no sanitizer report.  CPU only."""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "image_transformation_amd", "csrc")


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("san") / "host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I", CSRC, os.path.join(ROOT, "tests", "native", "host_sanitize_main.cpp"),
           os.path.join(CSRC, "flex_place.cpp"), os.path.join(CSRC, "resample_coeffs.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr.lower():
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def _pack(cases):
    buf = [struct.pack("<I", len(cases))]
    for text, sizes, (W, H) in cases:
        buf.append(struct.pack("<I", len(text)))
        buf.append(text)
        buf.append(struct.pack("<I", len(sizes)))
        for oid, (w, h) in sizes.items():
            buf.append(struct.pack("<iii", oid, w, h))
        buf.append(struct.pack("<ii", W, H))
    return b"".join(buf)


def test_flex_parser_and_tables_under_asan_ubsan(binary, golden_dir):
    with open(os.path.join(golden_dir, "flex.json"), encoding="utf-8") as f:
        g = json.load(f)
    rng = np.random.default_rng(4096)
    cases = []
    base = []
    for case in g["cases"] + g["kat"]:
        sizes = {int(k): tuple(v) for k, v in case["sizes"].items()}
        text = json.dumps(case["layout"]).encode()
        base.append((text, sizes, tuple(case["canvas"])))
    cases += base
    structural = b'{}[]",:\\-0123456789.eEtfn \n'
    for _ in range(4000):
        text, sizes, canvas = base[int(rng.integers(0, len(base)))]
        b = bytearray(text)
        for _m in range(int(rng.integers(1, 6))):
            kind = int(rng.integers(0, 5))
            pos = int(rng.integers(0, max(1, len(b))))
            if kind == 0 and b:
                del b[pos:pos + int(rng.integers(1, 12))]
            elif kind == 1 and b:
                b[pos] = int(rng.integers(0, 256))
            elif kind == 2:
                b[pos:pos] = bytes([structural[int(rng.integers(0, len(structural)))]]) * int(rng.integers(1, 4))
            elif kind == 3:
                b = b[:pos]
            else:
                b[pos:pos] = str(int(rng.integers(-2 ** 40, 2 ** 40))).encode()
        cases.append((bytes(b), sizes, canvas))
    sq = {1: (230, 62), 2: (357, 207), 3: (257, 137), 4: (131, 32)}
    deep = b'{"root":' + b'{"type":"flex","direction":"row","children":[' * 3000 + b'{"object_id":1}' + b"]}" * 3000 + b"}"
    cases += [(b"", sq, (9, 9)), (b"{", sq, (9, 9)), (b'{"root":', sq, (9, 9)), (b"\xff\xfe\x00", sq, (9, 9)),
              (b'{"root":{"children":[' + b'{"object_id":1},' * 5000 + b'{"object_id":2}]}}', sq, (492, 492)),
              (deep, sq, (492, 492)), (b'{"root":{"gap_px":99999999999999999999999,"children":[]}}', sq, (5, 5)),
              (b'{"root":{"padding_px":-0,"children":[{"object_id":"0004"}]}}', sq, (5, 5))]
    # round 3: int() coercion of a container's gap_px / padding_px (floats, exponents, numeric strings, bools) -- the
    # extremes of strtod / the double -> integer cast and of the string parser, then mutated like everything else
    coerce = [b"3.9", b"-3.9", b"1e2", b"1e308", b"-1e308", b"1e999", b"1e-999", b"0.0000000000000000001e19", b"16777215.99",
              b"16777216.0", b"-16777216.5", b"9007199254740993", b"123456789012345678901234567890.5", b'"12"', b'" +7 "',
              b'"-"', b'""', b'"   "', b'"+0000000001"', b'"1234567890"', b'"12\n"', b'"\t3"', b'"1_2"', b'"0x10"', b"true",
              b"false", b"null", b"-0.0", b"2.5e-1", b"1E+2", b"1e+", b"1.", b".5", b"-", b"1e"]
    extra = []
    for v in coerce:
        for key in (b"gap_px", b"padding_px"):
            extra.append((b'{"root":{"type":"flex","direction":"column","' + key + b'":' + v +
                          b',"children":[{"object_id":1},{"' + key + b'":' + v + b',"children":[{"object_id":"2"}]}]}}', sq, (492, 492)))
    cases += extra
    for _ in range(600):
        text, sizes, canvas = extra[int(rng.integers(0, len(extra)))]
        b = bytearray(text)
        pos = int(rng.integers(0, len(b)))
        b[pos:pos] = bytes([structural[int(rng.integers(0, len(structural)))]]) * int(rng.integers(1, 3))
        cases.append((bytes(b), sizes, canvas))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([binary], input=_pack(cases), capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    out = r.stdout.decode()
    assert out.startswith("ok=") and b"ERROR" not in r.stderr and b"runtime error" not in r.stderr, r.stderr[-3000:]
    ok = int(out.split()[0].split("=")[1])
    assert ok >= 240  # the fixture trees themselves are still placed


def test_png_encoder_under_asan_ubsan(tmp_path):
    """csrc/png_encode.cpp (the artifacts' encoder: an unchecked bit writer sized by deflate_bound, pooled scratch, worker
    threads) compiled with -fsanitize=address,undefined and driven with noise / flat / periodic / striped images of the
    size classes around its chunkings, every level and thread count; two runs produce the same bytes."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    out = str(tmp_path / "png_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-DMIC_PNG_EXACT_ALLOC", "-I", CSRC, os.path.join(ROOT, "tests", "native", "png_sanitize_main.cpp"),
           os.path.join(CSRC, "png_encode.cpp"), "-lpthread", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr.lower():
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    os.makedirs(tmp_path / "files", exist_ok=True)
    runs = [subprocess.run([out, str(tmp_path / "files")], capture_output=True, env=env, timeout=900) for _ in range(2)]
    for r in runs:
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
        assert r.stdout.startswith(b"ok=") and b"ERROR" not in r.stderr and b"runtime error" not in r.stderr, r.stderr[-3000:]
    assert runs[0].stdout == runs[1].stdout  # deterministic bytes, whatever the thread scheduling
    from PIL import Image
    assert len(list((tmp_path / "files").glob("a*.png"))) == 24 and Image.open(tmp_path / "files" / "a5.png").size == (300, 200)


def test_png_huffman_codes_are_complete_and_limited(tmp_path):
    """huff_lengths (the deflate code builder of the PNG writer) on 200 000 adversarial frequency sets, under
    ASan / UBSan: complete codes, 15 / 7-bit limits, more frequent symbols never longer."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    out = str(tmp_path / "png_huffman")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", CSRC,
           os.path.join(ROOT, "tests", "native", "png_huffman_main.cpp"), "-lpthread", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr.lower():
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([out], capture_output=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and r.stdout.strip() == b"bad=0 monotonic_violations=0", (r.stdout, r.stderr[-2000:])


def test_host_pool_under_tsan(tmp_path):
    """csrc/host_pool.h (the parked threads that build a call's axis tables) under ThreadSanitizer: four caller threads
    run 1 200 loops of 1-97 parts on the one pool at once; every part runs exactly once, a throwing part makes run()
    return false, no race is reported; a fork()ed child (which has the pool object but none of its threads) still completes
    its loops."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "host_pool_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", CSRC, os.path.join(ROOT, "tests", "native", "host_pool_main.cpp"),
           "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in (r.stderr or "").lower() and "cannot find" in r.stderr.lower():
        pytest.skip("libtsan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert b"ThreadSanitizer" not in r.stderr and b"data race" not in r.stderr, r.stderr[-3000:]
    assert b"bad=0" in r.stdout and b"loops=1200" in r.stdout and b"fork_child=ok" in r.stdout, r.stdout


def test_lane_partition_under_asan_ubsan(tmp_path):
    """csrc/lane_partition.h (the host-side cut of a lane-kernel launch into equal-cost chunks, dealt XCD by XCD; round 5) on 60
    random launches through the real lane-form axis tables: every tile of output rows of every strip emitted exactly once,
    every record reachable from exactly one wave slot, bands / table addresses / window bounds as the tables say."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "lane_partition_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", CSRC,
           os.path.join(ROOT, "tests", "native", "lane_partition_main.cpp"), os.path.join(CSRC, "resample_coeffs.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr.lower():
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert b"launches=60" in r.stdout and b" ok" in r.stdout, r.stdout
