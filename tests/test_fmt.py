"""The two host-checkable halves of the GPU-side significances writer (fhx_write_significances_device), compiled with g++ and
run on the CPU:

  * fhx_fmt.hpp - "%e" / "%f" with exact integer arithmetic (the function bodies the format kernel runs) against the C
    library on millions of random and adversarial doubles (the reference's rows: fithic/fithic.py:1202, :1212);
  * fhx_deflate.hpp - tokeniser, sinks, code maps, CRC combine, package-merge codes and block header (the function bodies
    the deflate kernels run), rows encoded out of order at scanned bit offsets, every member inflated by zlib (which
    verifies CRC-32 and ISIZE).
The kernels themselves (offset scans, LDS staging, atomics) are covered by the GPU tests (test_gpu_writer.py)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, src, libs=()):
    exe = os.path.join(str(tmp_path), os.path.splitext(os.path.basename(src))[0])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "fithic_amd", "csrc"), os.path.join(ROOT, src), "-o", exe,
                           *libs])
    return exe


def test_integer_printf_e_and_f_equal_the_c_library(tmp_path):
    exe = _build(tmp_path, "tests/native/fmt_check.cpp")
    r = subprocess.run([exe, "2000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 differences" in r.stdout, r.stdout


def test_row_deflate_inflates_with_zlib(tmp_path):
    exe = _build(tmp_path, "tests/native/deflate_check.cpp", ["-lz"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 differences" in r.stdout, r.stdout
