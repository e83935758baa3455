"""AC17 packed encrypt + decrypt of n items (50 attributes) through the host layer, unchunked vs pipelined (RABE_PACKED_LANES / RABE_PACKED_CHUNK):
where does cutting a batch into chunks on several engine lanes pay?   usage: python tools/bench_packed_pipeline.py N [chunk lanes]..."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rabe_amd import hostlib as hl          # noqa: E402
from rabe_amd import hostprep as hp         # noqa: E402
from rabe_amd.schemes import ac17           # noqa: E402


def main():
    n = int(sys.argv[1])
    legs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(2, len(sys.argv) - 1, 2)] or [(1 << 30, 1), (n // 4, 3)]
    host = hl.Host(0)
    attrs = ["a%d" % (i + 1) for i in range(50)]
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, attrs)
    import random
    prnd = random.Random(2)
    trees = [hp.random_binary_tree(attrs, prnd) for _ in range(16)]
    pols = [hp.to_json(t) for t in trees]
    pts = [b"dance like no one's watching, encrypt like everyone is!" + i.to_bytes(4, "little") for i in range(n)]
    item_pol = np.arange(n, dtype=np.uint32) % len(pols)
    pt_blob = b"".join(pts)
    pt_off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
    pt_np = np.frombuffer(pt_blob, dtype=np.uint8)
    ct_buf, _ = ac17.cp_encrypt_packed(host, pk, pols, item_pol, pt_np, pt_off)
    ct_buf = np.zeros(ct_buf.size, dtype=np.uint8)
    pt_buf = np.zeros(ct_buf.size, dtype=np.uint8)
    for chunk, lanes in legs:
        os.environ["RABE_PACKED_CHUNK"] = str(chunk)
        os.environ["RABE_PACKED_LANES"] = str(lanes)
        best = None
        for rep in range(4):
            t0 = time.perf_counter()
            ct_blob, ct_off = ac17.cp_encrypt_packed(host, pk, pols, item_pol, pt_np, pt_off, out=ct_buf)
            t1 = time.perf_counter()
            out_blob, out_off, status = ac17.cp_decrypt_packed(host, sk, ct_blob, ct_off, out=pt_buf)
            t2 = time.perf_counter()
            ok = out_blob.tobytes() == pt_blob and not status.any()
            out_blob, out_off, status = ac17.cp_decrypt_packed(host, sk, ct_blob, ct_off, out=pt_buf, trusted=True)
            t3 = time.perf_counter()
            if rep and (best is None or t2 - t0 < best[0]):
                best = (t2 - t0, t1 - t0, t2 - t1, t3 - t2)
        print(json.dumps({"n": n, "chunk": chunk, "lanes": lanes, "ops_per_s": round(n / best[0]), "encrypt_ms": round(1e3 * best[1], 1),
                          "decrypt_ms": round(1e3 * best[2], 1), "decrypt_trusted_ms": round(1e3 * best[3], 1), "ok": bool(ok)}), flush=True)
    host.close()


main()
