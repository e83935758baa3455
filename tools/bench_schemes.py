#!/usr/bin/env python3
"""Throughput of the bsw / lsw / aw11 batch entry points (BASELINE configs 3-5 at a reduced batch) through the host
layer, one GPU.  Not the judged metric (bench.py is config 2); prints one JSON line per config.
usage: python tools/bench_schemes.py [--batch 256]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rabe_amd import hostlib as hl  # noqa: E402
from rabe_amd.schemes import aw11, bsw, ghw11, lsw  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--only", default="")
ap.add_argument("--rounds", type=int, default=2, help="timed repetitions per config: the first meets cold fixed-base tables of the key elements, later ones warm ones")
args = ap.parse_args()
B = args.batch
PT = b"dance like no one's watching, encrypt like everyone is!"
host = hl.Host(0)
if os.environ.get("RABE_FIXED_BASE_MIN"):
    host.set_fixed_base_min(int(os.environ["RABE_FIXED_BASE_MIN"]))


def leaf(a):
    return '{"name": "%s"}' % a


def nest(ns):
    if len(ns) == 1:
        return leaf(ns[0])
    h = len(ns) // 2
    return '{"name": "and", "children": [%s, %s]}' % (nest(ns[:h]), nest(ns[h:]))


def report(name, n_ops, secs, extra):
    print(json.dumps({"config": name, "batch": B, "ops_per_s": round(n_ops / secs, 1), "seconds": round(secs, 3), **extra}), flush=True)


if args.only in ("", "bsw"):
    attrs = ["b%d" % i for i in range(100)]
    flat = '{"name": "and", "children": [%s]}' % ", ".join(leaf(a) for a in attrs)
    pk, msk = bsw.setup(host)
    sk = bsw.keygen(host, pk, msk, attrs)
    bsw.decrypt_batch(host, [sk] * 2, bsw.encrypt_batch(host, pk, [flat] * 2, hl.JSON_POLICY, [PT] * 2))   # warm-up (tables)
    for rd in range(args.rounds):
        t0 = time.perf_counter()
        cts = bsw.encrypt_batch(host, pk, [flat] * B, hl.JSON_POLICY, [PT] * B)
        t1 = time.perf_counter()
        pts = bsw.decrypt_batch(host, [sk] * B, cts)
        t2 = time.perf_counter()
        assert pts == [PT] * B
        report("3: BSW CP-ABE, 100-leaf AND tree (201 pairings/item)", B, t2 - t0, {"round": rd, "encrypt_s": round(t1 - t0, 3), "decrypt_s": round(t2 - t1, 3)})

if args.only in ("", "lsw"):
    attrs = ["c%d" % i for i in range(200)]
    policy = '{"name": "and", "children": [%s]}' % ", ".join(leaf(a) for a in attrs)
    pk, msk = lsw.setup(host)
    ct = lsw.encrypt(host, pk, attrs, PT)
    lsw.decrypt_batch(host, lsw.keygen_batch(host, pk, msk, [policy] * 2, hl.JSON_POLICY), [ct] * 2)
    for rd in range(args.rounds):
        t0 = time.perf_counter()
        sks = lsw.keygen_batch(host, pk, msk, [policy] * B, hl.JSON_POLICY)
        t1 = time.perf_counter()
        pts = lsw.decrypt_batch(host, sks, [ct] * B)
        t2 = time.perf_counter()
        assert pts == [PT] * B
        report("4: LSW KP-ABE keygen+decrypt, 200 attributes (400 pairings/item)", B, t2 - t0, {"round": rd, "keygen_s": round(t1 - t0, 3), "decrypt_s": round(t2 - t1, 3)})

if args.only in ("", "aw11"):
    gk = aw11.setup(host)
    auth, names = [], []
    for a in range(10):
        n = ["AUTH%dX%d" % (a, k) for k in range(20)]
        names += n
        auth.append(aw11.authgen(host, gk, n))
    policy = nest(names)
    sk = aw11.keygen(host, gk, auth[0][1], "alice", names[:20])
    for a in range(1, 10):
        for n in names[20 * a:20 * a + 20]:
            aw11.add_to_attribute(host, gk, auth[a][1], n, sk)
    pks = [p for p, _ in auth]
    aw11.decrypt_batch(host, gk, [sk] * 2, aw11.encrypt_batch(host, gk, pks, [policy] * 2, hl.JSON_POLICY, [PT] * 2))
    for rd in range(args.rounds):
        t0 = time.perf_counter()
        cts = aw11.encrypt_batch(host, gk, pks, [policy] * B, hl.JSON_POLICY, [PT] * B)
        t1 = time.perf_counter()
        pts = aw11.decrypt_batch(host, gk, [sk] * B, cts)
        t2 = time.perf_counter()
        assert pts == [PT] * B
        report("5: AW11, 10 authorities x 20 attributes (400 pairings/item)", B, t2 - t0, {"round": rd, "encrypt_s": round(t1 - t0, 3), "decrypt_s": round(t2 - t1, 3)})
if args.only in ("", "ghw11"):
    attrs = ["g%d" % i for i in range(50)]
    policy = nest(attrs)
    pk, msk = ghw11.setup(host)
    tk, rk = ghw11.tkgen(host, ghw11.keygen(host, pk, msk, attrs))
    n_ct = min(B, 64)                       # encrypt has no batch entry point: a few ciphertexts, repeated
    cts = [ghw11.encrypt(host, pk, policy, hl.JSON_POLICY, PT) for _ in range(n_ct)]
    items = [cts[i % n_ct] for i in range(B)]
    ghw11.transform_batch(host, items[:2], [tk] * 2)
    t0 = time.perf_counter()
    tcts = ghw11.transform_batch(host, items, [tk] * B)
    t1 = time.perf_counter()
    assert ghw11.decrypt_out(host, tcts[-1], rk, items[-1]) == PT
    report("8f-1: GHW11 transform (outsourced decryption), 50-attribute AND policy (52 pairings/item)", B, t1 - t0, {"transform_s": round(t1 - t0, 3)})
host.close()
