#!/usr/bin/env python3
"""Device gzip decoder on a contacts file of N members (2^18 rows each, as the library's writer cuts them): wall time of
fhx_debug_inflate_file (upload + gz_inflate + gz_crc + copy back), best of 3, and the text checked against zlib.
    python profiles/inflate_bench.py [members]"""
import gzip
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fithic_amd import _capi

members = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = members << 18
rng = np.random.default_rng(1)
names = ["chr%d" % k for k in range(1, 23)]
c1 = np.sort(rng.integers(0, 22, n)).astype(np.int32)
m1 = np.sort(rng.integers(0, 50_000, n)).astype(np.int32) * 5000 + 2500
m2 = (m1 + rng.integers(4, 400, n).astype(np.int32) * 5000).astype(np.int32)
cnt = rng.integers(1, 300, n).astype(np.int32)
path = "/dev/shm/inflate_bench.gz"
_capi.host_write_contacts(path, names, c1, m1.astype(np.int32), c1, m2, cnt, gzip_level=1)
with gzip.open(path, "rb") as f:
    want = f.read()
ctx = _capi.Context(0)
best = 1e9
for _ in range(3):
    t0 = time.time()
    got = ctx.debug_inflate_file(path, len(want) + 64)
    best = min(best, time.time() - t0)
print("%d members, %.1f MB gz -> %.1f MB text: %.1f ms (upload + inflate + CRC + copy back), text %s zlib's"
      % (members, os.path.getsize(path) / 1e6, len(want) / 1e6, best * 1e3, "EQUALS" if got == want else "DIFFERS FROM"))
ctx.close()
os.remove(path)
