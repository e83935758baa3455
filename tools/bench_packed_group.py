"""AC17 packed encrypt + decrypt of n items (50 attributes) through ONE engine and through a device GROUP (default: GPU 0 listed twice =
two blocks side by side on one GPU; `--devices 0,1,..` for real multi-GPU nodes): does running the blocks' copies beside each other's
kernels pay?   usage: python tools/bench_packed_group.py N [--devices 0,0] [--reps 4] [--only group|single]
Under `rocprofv3 --kernel-trace --memory-copy-trace` the last call's timeline is what tools/timeline.py prints."""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("--devices", default="0,0")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    n = a.n
    devs = [int(x) for x in a.devices.split(",")]
    import random
    attrs = ["a%d" % (i + 1) for i in range(50)]
    prnd = random.Random(2)
    trees = [hp.random_binary_tree(attrs, prnd) for _ in range(16)]
    pols = [hp.to_json(t) for t in trees]
    pts = [b"dance like no one's watching, encrypt like everyone is!" + i.to_bytes(4, "little") for i in range(n)]
    item_pol = np.arange(n, dtype=np.uint32) % len(pols)
    pt_blob = b"".join(pts)
    pt_off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
    pt_np = np.frombuffer(pt_blob, dtype=np.uint8)
    first = hl.Host(0)
    pk, msk = ac17.setup(first)
    sk = ac17.cp_keygen(first, msk, attrs)
    first.close()
    for name, mk in (("single", lambda: hl.Host(0)), ("group", lambda: hl.Host(devices=devs))):
        if a.only and a.only != name:
            continue
        host = mk()
        ct_buf, _ = ac17.cp_encrypt_packed(host, pk, pols, item_pol, pt_np, pt_off)
        ct_buf = np.zeros(ct_buf.size, dtype=np.uint8)
        pt_buf = np.zeros(ct_buf.size, dtype=np.uint8)
        best = None
        for rep in range(a.reps):
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
        print(json.dumps({"host": name, "devices": devs if name == "group" else [0], "n": n, "ops_per_s_checked": round(n / best[0]),
                          "ops_per_s_trusted": round(n / (best[1] + best[3])), "encrypt_ms": round(1e3 * best[1], 1), "decrypt_ms": round(1e3 * best[2], 1),
                          "decrypt_trusted_ms": round(1e3 * best[3], 1), "ok": bool(ok)}), flush=True)
        host.close()


main()
