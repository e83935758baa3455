"""Device-level BSW CP-ABE path (rhip_bsw_encrypt_batch / rhip_bsw_decrypt_batch, include/rabe_hip.h) against the oracle:
every ciphertext element byte for byte on the same tape, and the decrypted Gt of a batch that mixes policies (k-ary
AND / OR gates), keys (one of them not satisfying one policy's OR branch first) and the prepared / unprepared key paths."""
import random

import pytest

from oracle import bn254 as bn
from oracle import policy as pol
from oracle import schemes as sch
from oracle.tape import ListRng, SeededRng
from rabe_amd import Engine
from rabe_amd import engine as E
from rabe_amd import hostprep as hp

pytestmark = pytest.mark.gpu

T1 = ("and", [("leaf", "A"), ("or", [("leaf", "B"), ("leaf", "C")]), ("leaf", "D")])            # 3-ary AND over an OR
T2 = ("or", [("and", [("leaf", "A"), ("leaf", "B")]), ("and", [("leaf", "C"), ("leaf", "D"), ("leaf", "E")])])
T3 = ("and", [("leaf", "E"), ("and", [("leaf", "A"), ("and", [("leaf", "B"), ("leaf", "C")])])])  # nested binary ANDs: coefficients 2, -1 ...


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def world():
    rng = SeededRng(31)
    pk, msk = sch.bsw_setup(rng)
    sk_all = sch.bsw_keygen(pk, msk, ["A", "B", "C", "D", "E"], rng)
    sk_acde = sch.bsw_keygen(pk, msk, ["E", "D", "C", "A"], rng)      # no B: T1 takes the OR's second child, T2 its second AND
    return pk, msk, [sk_all, sk_acde]


def le(x):
    return hp.fr_le(x)


def test_bsw_device_batch_matches_oracle(eng, world):
    pk, _msk, sks = world
    trees = [T1, T2, T3]
    tt = hp.TreeTables(trees)
    dtt = E.DevTreeTables(eng, tt)
    dpk = E.BswPk(eng, bn.g1_to_le(pk["g1"]), bn.g2_to_le(pk["g2"]), bn.g1_to_le(pk["h"]), bn.gt_to_le(pk["e_gg_alpha"]))
    rnd = random.Random(7)
    items = [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 0), (0, 0)]          # (policy, key)
    n = len(items)
    # ---- encrypt on the device with explicit randomness, and the oracle on the same tape
    secrets = [rnd.randrange(1, bn.R) for _ in range(n)]
    rhos = [rnd.randrange(1, bn.R) for _ in range(n)]
    msgs = [bn.gt_pow(pk["e_gg_alpha"], r) for r in rhos]
    coefs = [[rnd.randrange(bn.R) for _ in range(tt.n_coef(p))] for p, _ in items]
    leaf_off, coef_off = [0], [0]
    for (p, _), c in zip(items, coefs):
        leaf_off.append(leaf_off[-1] + tt.n_leaves(p))
        coef_off.append(coef_off[-1] + len(c))
    total = leaf_off[-1]
    d_c, d_cp, d_g1, d_g2 = eng.alloc(n * 64), eng.alloc(n * 384), eng.alloc(total * 64), eng.alloc(total * 128)
    d_leaf_off = eng.upload_u32(leaf_off)
    E.bsw_encrypt_dev(eng, dpk, n, total, d_leaf_off, eng.upload_u32([tt.first_leaf[p] for p, _ in items]),
                      eng.upload_u32([tt.first_gate[p] for p, _ in items]), dtt, eng.upload(b"".join(le(s) for s in secrets)),
                      eng.upload(b"".join(le(x) for c in coefs for x in c) or bytes(32)), eng.upload_u32(coef_off[:-1]),
                      eng.upload(b"".join(bn.gt_to_le(m) for m in msgs)), d_c, d_cp, d_g1, d_g2)
    cts = []
    for i, (p, _) in enumerate(items):
        ct = sch.bsw_encrypt(pk, hp.to_json(trees[p]), pol.JSON, ListRng([secrets[i]] + coefs[i]), msgs[i])
        cts.append(ct)
    assert eng.download(d_c) == b"".join(bn.g1_to_le(ct["c"]) for ct in cts)
    assert eng.download(d_cp) == b"".join(bn.gt_to_le(ct["c_p"]) for ct in cts)
    assert eng.download(d_g1) == b"".join(bn.g1_to_le(y["g1"]) for ct in cts for y in ct["c_y"])
    assert eng.download(d_g2) == b"".join(bn.g2_to_le(y["g2"]) for ct in cts for y in ct["c_y"])
    # ---- decrypt on the device (selection tables per (policy, key)), both key paths
    key_attrs = [[d["string"] for d in sk["d_j"]] for sk in sks]
    sel_ct, sel_sk, sel_z, sel_start, pair_off = [], [], [], [], [0]
    for p, k in items:
        ok, idx = hp.pruned_leaf_indices(key_attrs[k], trees[p])
        assert ok
        z = hp.leaf_coefficients(trees[p])
        names = tt.flat[p]["names"]
        sel_start.append(len(sel_ct))
        for y in idx:
            sel_ct.append(y)
            sel_sk.append(key_attrs[k].index(names[y]))
            sel_z.append(z[y])
        pair_off.append(pair_off[-1] + 2 * len(idx) + 1)
    attr_off = [0]
    for sk in sks:
        attr_off.append(attr_off[-1] + len(sk["d_j"]))
    d_sk_d = eng.upload(b"".join(bn.g2_to_le(sk["d"]) for sk in sks))
    d_sk_g1 = eng.upload(b"".join(bn.g1_to_le(d["g1"]) for sk in sks for d in sk["d_j"]))
    d_sk_g2 = eng.upload(b"".join(bn.g2_to_le(d["g2"]) for sk in sks for d in sk["d_j"]))
    want = b"".join(bn.gt_to_le(sch.bsw_decrypt(sks[k], cts[i])) for i, (_, k) in enumerate(items))
    assert want == b"".join(bn.gt_to_le(m) for m in msgs)
    lines = E.BswSkLines(eng, len(sks), attr_off[-1], d_sk_d, d_sk_g2)
    for sk_lines in (None, lines):
        d_out = eng.alloc(n * 384)
        E.bsw_decrypt_dev(eng, n, max(b - a for a, b in zip(pair_off, pair_off[1:])), pair_off[-1], eng.upload_u32(pair_off),
                          eng.upload_u32(sel_start), eng.upload_u32(sel_ct), eng.upload_u32(sel_sk), eng.upload(b"".join(le(z) for z in sel_z)),
                          d_c, d_cp, d_g1, d_g2, d_leaf_off, d_sk_d, d_sk_g1, d_sk_g2, eng.upload_u32(attr_off),
                          eng.upload_u32([k for _, k in items]), sk_lines, d_out)
        assert eng.download(d_out) == want, "prepared" if sk_lines else "unprepared"
    lines.destroy()
    dpk.destroy()


def test_bsw_device_decrypt_many_chunks(eng, world):
    """one item whose pairs spread over several lanes (chunks), and a zero coefficient / infinity argument is skipped"""
    pk, msk, _ = world
    names = ["n%d" % i for i in range(9)]
    tree = ("and", [("leaf", x) for x in names])                   # flat 9-ary AND: full-size Lagrange coefficients, 19 pairs
    rng = SeededRng(5)
    sk = sch.bsw_keygen(pk, msk, names, rng)
    msg = bn.gt_pow(pk["e_gg_alpha"], 777)
    ct = sch.bsw_encrypt(pk, hp.to_json(tree), pol.JSON, rng, msg)
    assert sch.bsw_decrypt(sk, ct) == msg
    n = 3
    z = hp.leaf_coefficients(tree)
    d_out = eng.alloc(n * 384)
    E.bsw_decrypt_dev(eng, n, 19, 19 * n, eng.upload_u32([0, 19, 38, 57]), eng.upload_u32([0, 0, 0]), eng.upload_u32(list(range(9))),
                      eng.upload_u32(list(range(9))), eng.upload(b"".join(le(x) for x in z)),
                      eng.upload(bn.g1_to_le(ct["c"]) * n), eng.upload(bn.gt_to_le(ct["c_p"]) * n),
                      eng.upload(b"".join(bn.g1_to_le(y["g1"]) for y in ct["c_y"]) * n), eng.upload(b"".join(bn.g2_to_le(y["g2"]) for y in ct["c_y"]) * n),
                      eng.upload_u32([0, 9, 18, 27]), eng.upload(bn.g2_to_le(sk["d"])), eng.upload(b"".join(bn.g1_to_le(d["g1"]) for d in sk["d_j"])),
                      eng.upload(b"".join(bn.g2_to_le(d["g2"]) for d in sk["d_j"])), eng.upload_u32([0, 9]), eng.upload_u32([0] * n), None, d_out)
    assert eng.download(d_out) == bn.gt_to_le(msg) * n
    # the one-key entry point: the three items share their nine selection entries, so the key's scaled Dj.g1 are computed once per entry
    # (9 entries, 27 key-side pairs) -- with walking and with prepared key lines
    d_skd, d_g2s = eng.upload(bn.g2_to_le(sk["d"])), eng.upload(b"".join(bn.g2_to_le(d["g2"]) for d in sk["d_j"]))
    lines = E.BswSkLines(eng, 1, 9, d_skd, d_g2s) if hasattr(E, "BswSkLines") else None
    for ln in ([None, lines] if lines else [None]):
        d_out = eng.alloc(n * 384)
        E.bsw_decrypt_one_sk_dev(eng, n, 19, 19 * n, 9, eng.upload_u32([0, 19, 38, 57]), eng.upload_u32([0, 0, 0]), eng.upload_u32(list(range(9))),
                                 eng.upload_u32(list(range(9))), eng.upload(b"".join(le(x) for x in z)),
                                 eng.upload(bn.g1_to_le(ct["c"]) * n), eng.upload(bn.gt_to_le(ct["c_p"]) * n),
                                 eng.upload(b"".join(bn.g1_to_le(y["g1"]) for y in ct["c_y"]) * n), eng.upload(b"".join(bn.g2_to_le(y["g2"]) for y in ct["c_y"]) * n),
                                 eng.upload_u32([0, 9, 18, 27]), d_skd, eng.upload(b"".join(bn.g1_to_le(d["g1"]) for d in sk["d_j"])), d_g2s,
                                 eng.upload_u32([0, 9]), ln, d_out)
        assert eng.download(d_out) == bn.gt_to_le(msg) * n
