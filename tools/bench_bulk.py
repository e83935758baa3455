"""Bulk entry points of the host layer that are not part of a BASELINE configuration: key issuing for bsw / aw11 (ac17: tools/bench_keygen.py) and
lsw::encrypt, one call at a time vs packed.  One JSON line per entry point."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rabe_amd import hostlib as hl                 # noqa: E402
from rabe_amd.schemes import aw11, bsw, lsw        # noqa: E402

host = hl.Host(0)


def best_of(fn, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def single_rate(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return reps / (time.perf_counter() - t0)


# ---- bsw::keygen, 100 attributes
pk, msk = bsw.setup(host)
attrs = ["a%d" % i for i in range(100)]
out = {"entry": "rabe_bsw_keygen_packed", "attributes": 100, "single_call_per_s": round(single_rate(lambda: bsw.keygen(host, pk, msk, attrs), 10), 1)}
for n in (4096, 32768):
    it = np.zeros(n, dtype=np.uint32)
    buf, _ = bsw.keygen_packed(host, pk, msk, [attrs], it)
    buf = np.empty(buf.size, dtype=np.uint8)
    out["packed_%d_per_s" % n] = round(n / best_of(lambda: bsw.keygen_packed(host, pk, msk, [attrs], it, out=buf)), 1)
print(json.dumps(out), flush=True)

# ---- aw11::keygen, one authority x 20 attributes
gk = aw11.setup(host)
attrs = ["A%d" % i for i in range(20)]
apk, amsk = aw11.authgen(host, gk, attrs)
cnt = [0]


def one_aw11():
    cnt[0] += 1
    aw11.keygen(host, gk, amsk, "u%d" % cnt[0], attrs)


out = {"entry": "rabe_aw11_keygen_packed", "attributes": 20, "single_call_per_s": round(single_rate(one_aw11, 10), 1)}
for n in (4096, 65536):
    gids = ["user%07d" % i for i in range(n)]
    it = np.zeros(n, dtype=np.uint32)
    buf, _ = aw11.keygen_packed(host, gk, amsk, gids, [attrs], it)
    buf = np.empty(buf.size, dtype=np.uint8)
    out["packed_%d_per_s" % n] = round(n / best_of(lambda: aw11.keygen_packed(host, gk, amsk, gids, [attrs], it, out=buf)), 1)
print(json.dumps(out), flush=True)

# ---- lsw::encrypt, 200 attributes
pk, msk = lsw.setup(host)
attrs = ["a%d" % i for i in range(200)]
pt = b"dance like no one's watching, encrypt like everyone is!"
out = {"entry": "rabe_lsw_encrypt_packed", "attributes": 200, "single_call_per_s": round(single_rate(lambda: lsw.encrypt(host, pk, attrs, pt), 5), 1)}
for n in (2048, 16384):
    it = np.zeros(n, dtype=np.uint32)
    blob = np.frombuffer(pt * n, dtype=np.uint8)
    off = (np.arange(n + 1) * len(pt)).astype(np.uint64)
    buf, _ = lsw.encrypt_packed(host, pk, [attrs], it, blob, off)
    buf = np.empty(buf.size, dtype=np.uint8)
    out["packed_%d_per_s" % n] = round(n / best_of(lambda: lsw.encrypt_packed(host, pk, [attrs], it, blob, off, out=buf)), 1)
print(json.dumps(out), flush=True)
host.close()
