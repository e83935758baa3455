"""Device-level LSW (keygen + decrypt) and AW11 (encrypt + decrypt) paths (include/rabe_hip.h) against the oracle on the
same explicit randomness: every produced element byte for byte, and the decrypted Gt of batches that mix policies."""
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


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


def le(x):
    return hp.fr_le(x)


def offsets(counts):
    out = [0]
    for c in counts:
        out.append(out[-1] + c)
    return out


# ---------------------------------------------------------------------------------------------------------------- LSW
L1 = ("and", [("leaf", "A"), ("leaf", "B"), ("or", [("leaf", "C"), ("leaf", "D")])])
L2 = ("or", [("and", [("leaf", "D"), ("leaf", "E")]), ("leaf", "A")])
L3 = ("and", [("leaf", x) for x in "ABCDE"])                            # flat 5-ary AND: full-size coefficients


def test_lsw_device_keygen_decrypt_match_oracle(eng):
    rng = SeededRng(41)
    pk, msk = sch.lsw_setup(rng)
    trees = [L1, L2, L3]
    tt = hp.TreeTables(trees)
    dtt = E.DevTreeTables(eng, tt)
    dpk = E.LswPk(eng, bn.g1_to_le(pk["g1"]), bn.g2_to_le(pk["g2"]))
    ct_attrs = [["B", "A", "D", "E", "C"], ["E", "D", "A"]]
    msgs = [bn.gt_pow(pk["e_gg_alpha"], 1234567), bn.gt_pow(pk["e_gg_alpha"], 7654321)]
    cts = [sch.lsw_encrypt(pk, a, rng, m) for a, m in zip(ct_attrs, msgs)]
    items = [(0, 0), (1, 0), (2, 0), (1, 1), (0, 0), (1, 1)]              # (key policy, ciphertext)
    n = len(items)
    rnd = random.Random(11)
    coefs = [[rnd.randrange(bn.R) for _ in range(tt.n_coef(p))] for p, _ in items]
    rands = [[rnd.randrange(1, bn.R) for _ in range(tt.n_leaves(p))] for p, _ in items]
    leaf_off = offsets([tt.n_leaves(p) for p, _ in items])
    coef_off = offsets([len(c) for c in coefs])
    total = leaf_off[-1]
    d_d1, d_d2 = eng.alloc(total * 64), eng.alloc(total * 128)
    d_leaf_off = eng.upload_u32(leaf_off)
    E.lsw_keygen_dev(eng, dpk, n, total, d_leaf_off, eng.upload_u32([tt.first_leaf[p] for p, _ in items]),
                     eng.upload_u32([tt.first_gate[p] for p, _ in items]), dtt, eng.upload(le(msk["alpha1"]) + le(msk["alpha2"])),
                     eng.upload(b"".join(le(x) for c in coefs for x in c) or bytes(32)), eng.upload_u32(coef_off[:-1]),
                     eng.upload(b"".join(le(x) for r in rands for x in r)), d_d1, d_d2)
    sks = [sch.lsw_keygen(pk, msk, hp.to_json(trees[p]), pol.JSON, ListRng(coefs[i] + rands[i])) for i, (p, _) in enumerate(items)]
    assert eng.download(d_d1) == b"".join(bn.g1_to_le(row[1]) for sk in sks for row in sk["dj"])
    assert eng.download(d_d2) == b"".join(bn.g2_to_le(row[2]) for sk in sks for row in sk["dj"])
    # ---- decrypt: the keys are the device's own output rows (item i's key = rows [leaf_off[i], leaf_off[i+1]))
    sel_sk, sel_ct, sel_z, sel_start, pair_off = [], [], [], [], [0]
    for p, c in items:
        ok, idx = hp.pruned_leaf_indices(ct_attrs[c], trees[p])
        assert ok
        z = hp.leaf_coefficients(trees[p])
        names = tt.flat[p]["names"]
        sel_start.append(len(sel_sk))
        for y in idx:
            sel_sk.append(y)
            sel_ct.append(ct_attrs[c].index(names[y]))
            sel_z.append(z[y])
        pair_off.append(pair_off[-1] + len(idx) + 1)
    want = b"".join(bn.gt_to_le(sch.lsw_decrypt(sks[i], cts[c])) for i, (_, c) in enumerate(items))
    assert want == b"".join(bn.gt_to_le(msgs[c]) for _, c in items)
    d_e2 = eng.upload(b"".join(bn.g2_to_le(ct["e2"]) for ct in cts))
    lines = E.G2Lines(eng, len(cts), d_e2)
    for e2_lines in (None, lines):
        d_out = eng.alloc(n * 384)
        E.lsw_decrypt_dev(eng, n, max(b - a for a, b in zip(pair_off, pair_off[1:])), pair_off[-1], len(sel_sk), eng.upload_u32(pair_off),
                          eng.upload_u32(sel_start), eng.upload_u32(sel_sk), eng.upload_u32(sel_ct), eng.upload(b"".join(le(z) for z in sel_z)),
                          eng.upload(b"".join(bn.gt_to_le(cts[c]["e1"]) for _, c in items)), d_e2,
                          eng.upload(b"".join(bn.g1_to_le(row[1]) for ct in cts for row in ct["ej"])),
                          eng.upload_u32(offsets([len(a) for a in ct_attrs])), eng.upload_u32([c for _, c in items]), d_d1, d_d2, d_leaf_off, None,
                          e2_lines, d_out)
        assert eng.download(d_out) == want, "prepared e2" if e2_lines else "walking e2"
    # ---- the one-ciphertext entry point (n keys against ciphertext 0): selection entries SHARED by the items of one policy, so the scaled
    # ciphertext rows are computed once per entry; the same bytes as the oracle / the general entry point
    its = [i for i, (_, c) in enumerate(items) if c == 0] * 3                       # items 0, 1, 2, 4 repeated: policies 0, 1, 2, 0
    shared_start, sel_sk1, sel_ct1, sel_z1 = {}, [], [], []
    for i in its:
        p = items[i][0]
        if p not in shared_start:
            shared_start[p] = len(sel_sk1)
            a, b = sel_start[i], (sel_start[i + 1] if i + 1 < n else len(sel_sk))
            sel_sk1 += sel_sk[a:b]
            sel_ct1 += sel_ct[a:b]
            sel_z1 += sel_z[a:b]
    po1 = [0]
    for i in its:
        po1.append(po1[-1] + pair_off[i + 1] - pair_off[i])
    assert len(sel_sk1) < po1[-1] - len(its)                                            # fewer entries than scaled pairs: the shared path runs
    d_e2_0 = eng.upload(bn.g2_to_le(cts[0]["e2"]))
    lines0 = E.G2Lines(eng, 1, d_e2_0)
    d_out = eng.alloc(len(its) * 384)
    E.lsw_decrypt_one_ct_dev(eng, len(its), max(b - a for a, b in zip(po1, po1[1:])), po1[-1], len(sel_sk1), eng.upload_u32(po1),
                             eng.upload_u32([shared_start[items[i][0]] for i in its]), eng.upload_u32(sel_sk1), eng.upload_u32(sel_ct1),
                             eng.upload(b"".join(le(z) for z in sel_z1)), eng.upload(bn.gt_to_le(cts[0]["e1"]) * len(its)), d_e2_0,
                             eng.upload(b"".join(bn.g1_to_le(row[1]) for row in cts[0]["ej"])), d_d1, d_d2, d_leaf_off, eng.upload_u32(its), lines0, d_out)
    assert eng.download(d_out) == bn.gt_to_le(msgs[0]) * len(its)
    # the same entry point with one entry list PER ITEM (nothing shared: as many entries as scaled pairs): the per-pair scaling path
    own = [i for i, (_, c) in enumerate(items) if c == 0]
    st2, sk2, ct2, z2, po2 = [], [], [], [], [0]
    for i in own:
        a, b = sel_start[i], (sel_start[i + 1] if i + 1 < n else len(sel_sk))
        st2.append(len(sk2))
        sk2 += sel_sk[a:b]
        ct2 += sel_ct[a:b]
        z2 += sel_z[a:b]
        po2.append(po2[-1] + pair_off[i + 1] - pair_off[i])
    assert len(sk2) == po2[-1] - len(own)
    d_out = eng.alloc(len(own) * 384)
    E.lsw_decrypt_one_ct_dev(eng, len(own), max(b - a for a, b in zip(po2, po2[1:])), po2[-1], len(sk2), eng.upload_u32(po2), eng.upload_u32(st2),
                             eng.upload_u32(sk2), eng.upload_u32(ct2), eng.upload(b"".join(le(z) for z in z2)), eng.upload(bn.gt_to_le(cts[0]["e1"]) * len(own)),
                             d_e2_0, eng.upload(b"".join(bn.g1_to_le(row[1]) for row in cts[0]["ej"])), d_d1, d_d2, d_leaf_off, eng.upload_u32(own), None,
                             d_out)
    assert eng.download(d_out) == bn.gt_to_le(msgs[0]) * len(own)
    lines0.destroy()
    lines.destroy()
    dpk.destroy()


def test_lsw_device_keygen_with_negative_leaves_matches_oracle(eng):
    """lsw/mod.rs:137-146: a "!x" leaf gets (d3, d4, d5) from the share, b, b^2 and the master key's h_g1; positive rows of the same
    key keep (d1, d2).  Device (rhip_lsw_keygen_batch_signed) against the oracle on the same tape, byte for byte."""
    rng = SeededRng(47)
    pk, msk = sch.lsw_setup(rng)
    N1 = ("and", [("leaf", "A"), ("leaf", "!B"), ("or", [("leaf", "!C"), ("leaf", "D")])])
    N2 = ("or", [("and", [("leaf", "!D"), ("leaf", "E")]), ("leaf", "!A")])
    trees = [N1, N2, L3]
    tt = hp.TreeTables(trees)
    dtt = E.DevTreeTables(eng, tt)
    dpk = E.LswPk(eng, bn.g1_to_le(pk["g1"]), bn.g2_to_le(pk["g2"]))
    items = [0, 1, 2, 0, 1]
    n = len(items)
    rnd = random.Random(17)
    coefs = [[rnd.randrange(bn.R) for _ in range(tt.n_coef(p))] for p in items]
    rands = [[rnd.randrange(1, bn.R) for _ in range(tt.n_leaves(p))] for p in items]
    leaf_off = offsets([tt.n_leaves(p) for p in items])
    coef_off = offsets([len(c) for c in coefs])
    total = leaf_off[-1]
    d = [eng.alloc(total * 64), eng.alloc(total * 128), eng.alloc(total * 64), eng.alloc(total * 64), eng.alloc(total * 64)]
    leaf_neg = [1 if nm.startswith("!") else 0 for f in tt.flat for nm in f["names"]]
    E.lsw_keygen_signed_dev(eng, dpk, n, total, eng.upload_u32(leaf_off), eng.upload_u32([tt.first_leaf[p] for p in items]),
                            eng.upload_u32([tt.first_gate[p] for p in items]), dtt, eng.upload_u32(leaf_neg),
                            eng.upload(le(msk["alpha1"]) + le(msk["alpha2"])), eng.upload(le(msk["b"])), bn.g1_to_le(msk["h_g1"]),
                            eng.upload(b"".join(le(x) for c in coefs for x in c) or bytes(32)), eng.upload_u32(coef_off[:-1]),
                            eng.upload(b"".join(le(x) for r in rands for x in r)), *d)
    sks = [sch.lsw_keygen(pk, msk, hp.to_json(trees[p]), pol.JSON, ListRng(coefs[i] + rands[i])) for i, p in enumerate(items)]
    rows = [row for sk in sks for row in sk["dj"]]
    assert any(r[1] is None for r in rows) and any(r[3] is None for r in rows)
    g1 = lambda p: bn.g1_to_le(p)              # None (the struct's unused slot) is the identity: zeros
    assert eng.download(d[0]) == b"".join(g1(r[1]) for r in rows)
    assert eng.download(d[1]) == b"".join(bn.g2_to_le(r[2]) for r in rows)
    assert eng.download(d[2]) == b"".join(g1(r[3]) for r in rows)
    assert eng.download(d[3]) == b"".join(g1(r[4]) for r in rows)
    assert eng.download(d[4]) == b"".join(g1(r[5]) for r in rows)
    dpk.destroy()


# ---------------------------------------------------------------------------------------------------------------- AW11
W1 = ("and", [("leaf", "A"), ("and", [("leaf", "D"), ("or", [("leaf", "B"), ("leaf", "C")])])])
W2 = ("or", [("and", [("leaf", "E"), ("leaf", "A")]), ("and", [("leaf", "C"), ("leaf", "D")])])
W3 = ("and", [("and", [("leaf", "A"), ("leaf", "B")]), ("and", [("leaf", "C"), ("and", [("leaf", "D"), ("leaf", "E")])])])


@pytest.mark.parametrize("attr_bits", [None, "8", "9", "14"], ids=["default-signed-10", "plain-8", "signed-9", "signed-14"])
def test_aw11_device_encrypt_decrypt_match_oracle(eng, attr_bits, monkeypatch):
    """the same bytes whatever window form the per-attribute tables take (read by rhip_aw11_pk_create: RABE_AW11_ATTR_BITS)"""
    if attr_bits is None:
        monkeypatch.delenv("RABE_AW11_ATTR_BITS", raising=False)
    else:
        monkeypatch.setenv("RABE_AW11_ATTR_BITS", attr_bits)
    rng = SeededRng(43)
    gk = sch.aw11_setup(rng)
    auth = [sch.aw11_authgen(gk, ["A", "B", "C"], rng), sch.aw11_authgen(gk, ["D", "E"], rng)]
    pks = [a[0] for a in auth]
    all_attr = [t for pk in pks for t in pk["attr"]]                         # (name, egg_alpha, g2_y)
    names = [t[0] for t in all_attr]
    dpk = E.Aw11Pk(eng, bn.g1_to_le(gk["g1"]), bn.g2_to_le(gk["g2"]), [bn.gt_to_le(t[1]) for t in all_attr], [bn.g2_to_le(t[2]) for t in all_attr])
    trees = [W1, W2, W3]
    tt = hp.TreeTables(trees)
    dtt = E.DevTreeTables(eng, tt)
    d_leaf_attr = eng.upload_u32([names.index(nm) for f in tt.flat for nm in f["names"]])

    def make_key(gid, attrs):
        sk = {"gid": gid, "attr": []}
        for a in attrs:
            msk = auth[0][1] if a in "ABC" else auth[1][1]
            sk["attr"] += sch.aw11_keygen(gk, msk, gid, [a])["attr"]
        return sk
    sks = [make_key("alice", ["A", "B", "C", "D", "E"]), make_key("bob", ["E", "D", "C", "A"])]
    items = [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 0)]                  # (policy, key)
    n = len(items)
    rnd = random.Random(13)
    s = [rnd.randrange(1, bn.R) for _ in range(n)]
    egg = bn.pairing(gk["g1"], gk["g2"])
    msgs = [bn.gt_pow(egg, rnd.randrange(1, bn.R)) for _ in range(n)]
    coefs = [[rnd.randrange(bn.R) for _ in range(2 * tt.n_coef(p))] for p, _ in items]
    rands = [[rnd.randrange(1, bn.R) for _ in range(tt.n_leaves(p))] for p, _ in items]
    row_off = offsets([tt.n_leaves(p) for p, _ in items])
    coef_off = offsets([len(c) for c in coefs])
    total = row_off[-1]
    d_c0, d_c1, d_c2, d_c3 = eng.alloc(n * 384), eng.alloc(total * 384), eng.alloc(total * 128), eng.alloc(total * 128)
    d_row_off = eng.upload_u32(row_off)
    E.aw11_encrypt_dev(eng, dpk, n, total, d_row_off, eng.upload_u32([tt.first_leaf[p] for p, _ in items]),
                       eng.upload_u32([tt.first_gate[p] for p, _ in items]), eng.upload_u32([tt.n_coef(p) for p, _ in items]), dtt, d_leaf_attr,
                       eng.upload(b"".join(le(x) for x in s)), eng.upload(b"".join(le(x) for c in coefs for x in c) or bytes(32)),
                       eng.upload_u32(coef_off[:-1]), eng.upload(b"".join(le(x) for r in rands for x in r)),
                       eng.upload(b"".join(bn.gt_to_le(m) for m in msgs)), d_c0, d_c1, d_c2, d_c3)
    cts = [sch.aw11_encrypt(gk, pks, hp.to_json(trees[p]), pol.JSON, ListRng([s[i]] + coefs[i] + rands[i]), msgs[i]) for i, (p, _) in enumerate(items)]
    assert eng.download(d_c0) == b"".join(bn.gt_to_le(ct["c_0"]) for ct in cts)
    assert eng.download(d_c2) == b"".join(bn.g2_to_le(r[2]) for ct in cts for r in ct["c"])
    assert eng.download(d_c3) == b"".join(bn.g2_to_le(r[3]) for ct in cts for r in ct["c"])
    assert eng.download(d_c1) == b"".join(bn.gt_to_le(r[1]) for ct in cts for r in ct["c"])
    # ---- decrypt
    key_attrs = [[a[0] for a in sk["attr"]] for sk in sks]
    sel_ct, sel_sk, sel_z, sel_start, pair_off = [], [], [], [], [0]
    for p, k in items:
        ok, idx = hp.pruned_leaf_indices(key_attrs[k], trees[p])
        assert ok
        z = hp.leaf_coefficients(trees[p])
        nm = tt.flat[p]["names"]
        sel_start.append(len(sel_ct))
        for y in idx:
            sel_ct.append(y)
            sel_sk.append(key_attrs[k].index(nm[y]))
            sel_z.append(z[y])
        pair_off.append(pair_off[-1] + len(idx) + 1)
    want = b"".join(bn.gt_to_le(sch.aw11_decrypt(gk, sks[k], cts[i])) for i, (_, k) in enumerate(items))
    assert want == b"".join(bn.gt_to_le(m) for m in msgs)
    d_out = eng.alloc(n * 384)
    E.aw11_decrypt_dev(eng, n, max(b - a for a, b in zip(pair_off, pair_off[1:])), pair_off[-1], len(sel_ct), eng.upload_u32(pair_off),
                       eng.upload_u32(sel_start), eng.upload_u32(sel_ct), eng.upload_u32(sel_sk), eng.upload(b"".join(le(z) for z in sel_z)),
                       d_c0, d_c1, d_c2, d_c3, d_row_off,
                       eng.upload(b"".join(bn.g1_to_le(sch.sha3_hash_g1(gk["g1"], sk["gid"])) for sk in sks)),
                       eng.upload(b"".join(bn.g1_to_le(a[1]) for sk in sks for a in sk["attr"])), eng.upload_u32(offsets([len(sk["attr"]) for sk in sks])),
                       eng.upload_u32([k for _, k in items]), d_out)
    assert eng.download(d_out) == want
    dpk.destroy()
