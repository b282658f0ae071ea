"""image_transformation_amd.build: staleness by content digest, one builder at a time, atomic replacement.
Runs with a stand-in compiler (HIPCC) and a library path of its own: nothing here touches the real libmic.so."""
import os
import stat
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    from image_transformation_amd import build as b
    b.LIB = sys.argv[1]
    b.DIGEST = b.LIB + ".digest"
    print(b.build(force="force" in sys.argv[2:]))
""" % ROOT)


def _fake_hipcc(tmp_path):
    log = tmp_path / "compiles.log"
    cc = tmp_path / "fake_hipcc"
    cc.write_text(textwrap.dedent(f"""\
        #!/bin/sh
        # stand-in for hipcc: finds "-o <path>", takes its time, writes the output
        out=""
        while [ $# -gt 0 ]; do
          if [ "$1" = "-o" ]; then out="$2"; shift; fi
          shift
        done
        sleep 0.7
        echo "$out" >> {log}
        echo built > "$out"
        """))
    cc.chmod(cc.stat().st_mode | stat.S_IXUSR)
    return str(cc), log


def test_concurrent_importers_build_once(tmp_path):
    cc, log = _fake_hipcc(tmp_path)
    lib = str(tmp_path / "libmic_test.so")
    env = dict(os.environ, HIPCC=cc)
    procs = [subprocess.Popen([sys.executable, "-c", DRIVER, lib], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for _ in range(4)]
    for p in procs:
        out, err = p.communicate(timeout=60)
        assert p.returncode == 0, err.decode()
        assert out.decode().strip() == lib
    assert len(log.read_text().splitlines()) == 1      # the ranks of a job: one compiles, the others wait and reuse
    assert open(lib).read() == "built\n"
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]  # written under another name, renamed into place
    assert len(open(lib + ".digest").read().strip()) == 64


def test_staleness_is_by_content_not_by_file_time(tmp_path):
    sys.path.insert(0, ROOT)
    from image_transformation_amd import build as b
    cc, log = _fake_hipcc(tmp_path)
    lib = str(tmp_path / "libmic_test.so")
    env = dict(os.environ, HIPCC=cc)
    subprocess.check_call([sys.executable, "-c", DRIVER, lib], env=env, stdout=subprocess.DEVNULL)
    old = (b.LIB, b.DIGEST)
    try:
        b.LIB, b.DIGEST = lib, lib + ".digest"
        assert not b._stale()
        os.utime(lib, (1, 1))                      # a copy that lost its file times: still fresh
        assert not b._stale()
        with open(b.DIGEST, "w") as f:             # built from other sources: stale whatever the times say
            f.write("0" * 64 + "\n")
        os.utime(lib, None)
        assert b._stale()
        os.remove(b.DIGEST)                        # a library from before the digest existed: file times decide
        os.utime(lib, (1, 1))
        assert b._stale()
    finally:
        b.LIB, b.DIGEST = old
    # other flags are other contents
    subprocess.check_call([sys.executable, "-c", DRIVER, lib], env=env, stdout=subprocess.DEVNULL)
    n = len(log.read_text().splitlines())
    subprocess.check_call([sys.executable, "-c", DRIVER, lib], env=dict(env, MIC_EXTRA_CFLAGS="-DX=1"), stdout=subprocess.DEVNULL)
    assert len(log.read_text().splitlines()) == n + 1


def test_isa_canary_of_the_ashr_pk_workaround():
    """build.check_isa (run by __graft_entry__.build()): the compiled resample kernels still have the shape the
    v_ashr_pk_u8_i32 workaround relies on (no GPU needed: hipcc -S)."""
    from image_transformation_amd import build
    build.check_isa()
