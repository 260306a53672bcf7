"""Sanitizer leg: the host code that consumes untrusted files - gzip containers, deflate streams cut
at guessed block starts, table text - and the host fit, rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer
(-fno-sanitize-recover: the first report aborts) and driven with random and mutated inputs by tests/native/io_sanitize.cpp.
CPU only; the same three sources are part of libfithic_mi355x.so."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fithic_amd", "csrc")
FLAGS = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
         "-ffp-contract=off", "-pthread", "-I", CSRC]


def test_reader_gunzip_and_host_fit_under_asan_and_ubsan(tmp_path):
    srcs = [os.path.join(CSRC, "fhx_io.cpp"), os.path.join(CSRC, "fhx_gunzip.cpp"), os.path.join(CSRC, "fhx_host.cpp"),
            os.path.join(ROOT, "tests", "native", "io_sanitize.cpp")]
    objs, procs = [], []
    for src in srcs:                                  # the four translation units in parallel
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen(["g++"] + FLAGS + ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, out
    exe = str(tmp_path / "io_sanitize")
    r = subprocess.run(["g++"] + FLAGS + objs + ["-lz", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=1:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    for seed in (11, 12):
        r = subprocess.run([exe, str(scratch), "300", str(seed)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        assert "0 check failures" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
