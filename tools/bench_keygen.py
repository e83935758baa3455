"""AC17 cp_keygen through the host layer: one key per call (rabe_ac17_cp_keygen) vs rabe_ac17_cp_keygen_packed (n keys per call)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rabe_amd import hostlib as hl          # noqa: E402
from rabe_amd.schemes import ac17           # noqa: E402

host = hl.Host(0)
pk, msk = ac17.setup(host)
attrs = ["a%d" % (i + 1) for i in range(50)]
ac17.cp_keygen(host, msk, attrs)
t0 = time.perf_counter()
for _ in range(20):
    ac17.cp_keygen(host, msk, attrs)
single = 20 / (time.perf_counter() - t0)
out = {"attributes": 50, "single_call_keys_per_s": round(single, 1)}
for n in (4096, 65536):
    it = np.zeros(n, dtype=np.uint32)
    buf, _ = ac17.cp_keygen_packed(host, msk, [attrs], it)
    buf = np.empty(buf.size, dtype=np.uint8)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        blob, off = ac17.cp_keygen_packed(host, msk, [attrs], it, out=buf)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["packed_%d_keys_per_s" % n] = round(n / best, 1)
    out["packed_%d_record_bytes" % n] = int(blob.size)
print(json.dumps(out))
host.close()
