"""Byte-for-byte parity at BASELINE.json's full attribute counts (not just round trips): the reference-order C
restatement (oracle/c/rabe_ref.c through oracle/cport.py, itself pinned to the Python oracle at small sizes in
tests/test_oracle_c.py) is fast enough to check a slice of the real workloads:
  config 2  16 items of bench.py's batch shape: AC17, 50 attributes, 16 random binary AND/OR policies, 26-bit windows for
            g, prepared key -- every ciphertext element and the decrypted Gt
  config 3  one BSW item at 100 leaves (flat 100-ary AND: full-size Lagrange coefficients, 201 pairings) and one at the
            AND-of-ORs shape, prepared and unprepared key"""
import random

import pytest

from oracle import bn254 as bn
from oracle import cport
from oracle import policy as pol
from rabe_amd import hostprep as hp

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not cport.available(), reason="oracle/c not built")]


def le(x):
    return hp.fr_le(x)


@pytest.fixture(scope="module")
def eng():
    from rabe_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def test_config2_sixteen_items_against_reference_order(eng):
    from rabe_amd import engine as E
    rnd = random.Random(2)
    attrs = ["a%d" % (i + 1) for i in range(50)]
    trees = [hp.random_binary_tree(attrs, rnd) for _ in range(16)]

    def rfr():
        return rnd.randrange(1, bn.R)
    # ---- ac17::setup / cp_keygen in reference order over the C primitives (ac17/mod.rs:141-264)
    g, h = cport.g1_mul(bn.G1_GEN, rfr()), cport.g2_mul(bn.G2_GEN, rfr())
    a, b, k = [rfr(), rfr()], [rfr(), rfr()], [rfr(), rfr(), rfr()]
    h_a = [cport.g2_mul(h, a[0]), cport.g2_mul(h, a[1]), h]
    e_gh = cport.pairing(g, h)
    e_gh_ka = [cport.gt_pow(e_gh, (k[i] * a[i] + k[2]) % bn.R) for i in range(2)]
    g_k = [cport.g1_mul(g, x) for x in k]
    r = [rfr(), rfr()]
    br = [b[0] * r[0] % bn.R, b[1] * r[1] % bn.R, (r[0] + r[1]) % bn.R]
    k_0 = [cport.g2_mul(h, x) for x in br]
    sk_k = []
    for y in attrs:
        sigma = rfr()
        row = []
        for t in range(2):
            acc = None
            for l in range(3):
                term = cport.g1_mul(cport.g1_mul(g, cport.hash_fr(y + str(l) + str(t))), br[l] * bn.fr_inv(a[t]) % bn.R)
                acc = term if acc is None else bn.g1_add(acc, term)
            row.append(bn.g1_add(acc, cport.g1_mul(g, sigma * bn.fr_inv(a[t]) % bn.R)))
        row.append(cport.g1_mul(g, (-sigma) % bn.R))
        sk_k.append(row)
    sigma_p = rfr()
    k_p = []
    for t in range(2):
        acc = g_k[t]
        for l in range(3):
            acc = bn.g1_add(acc, cport.g1_mul(cport.g1_mul(g, cport.hash_fr("01" + str(l) + str(t))), br[l] * bn.fr_inv(a[t]) % bn.R))
        k_p.append(bn.g1_add(acc, cport.g1_mul(g, sigma_p * bn.fr_inv(a[t]) % bn.R)))
    k_p.append(bn.g1_add(g_k[2], cport.g1_mul(g, (-sigma_p) % bn.R)))
    # ---- the engine: tables with 26-bit windows (bench.py's default), prepared key
    dpk = E.Ac17Pk(eng, bn.g1_to_le(g), [bn.g2_to_le(x) for x in h_a], [bn.gt_to_le(x) for x in e_gh_ka])
    dpk.set_g_window(26)
    dk0 = eng.upload(b"".join(bn.g2_to_le(x) for x in k_0))
    dk = eng.upload(b"".join(bn.g1_to_le(p) for row in sk_k for p in row))
    dkp = eng.upload(b"".join(bn.g1_to_le(p) for p in k_p))
    lines = E.Ac17SkLines(eng, 1, dk0)
    tables, sels, rows, A_off = [], [], [], [0]
    for t in trees:
        pi, A, _ = hp.ac17_policy_table(t)
        ok, ct_sel, sk_sel = hp.ac17_decrypt_selection(attrs, pi, t)
        assert ok
        tables.append(A)
        sels.append((ct_sel, sk_sel, pi))
        rows.append(len(pi))
        A_off.append(A_off[-1] + len(pi))
    n = 16
    row_off = [0]
    for x in rows:
        row_off.append(row_off[-1] + x)
    s = [(rfr(), rfr()) for _ in range(n)]
    msgs = [cport.gt_pow(e_gh, rfr()) for _ in range(n)]
    dc0, dc, dcp, dout = eng.alloc(n * 3 * 128), eng.alloc(row_off[-1] * 192), eng.alloc(n * 384), eng.alloc(n * 384)
    d_row_off = eng.upload_u32(row_off)
    E.ac17_encrypt_dev(eng, dpk, n, eng.upload(b"".join(tables)), eng.upload_u32(A_off[:-1]), d_row_off, row_off[-1],
                       eng.upload(b"".join(le(x) + le(y) for x, y in s)), eng.upload(b"".join(bn.gt_to_le(m) for m in msgs)), dc0, dc, dcp)
    ct_sel_all, sk_sel_all, cso, sso = [], [], [0], [0]
    for cs, ss, _ in sels:
        ct_sel_all += cs
        sk_sel_all += ss
        cso.append(len(ct_sel_all))
        sso.append(len(sk_sel_all))
    E.ac17_decrypt_prepared_dev(eng, n, dc0, dc, d_row_off, dcp, lines, dk, eng.upload_u32([0, 50]), dkp, eng.upload_u32([0] * n),
                                eng.upload_u32(ct_sel_all), eng.upload_u32(cso), eng.upload_u32(sk_sel_all), eng.upload_u32(sso), dout)
    got_c0, got_c, got_cp, got_out = eng.download(dc0), eng.download(dc), eng.download(dcp), eng.download(dout)
    # ---- the reference-order restatement on the same randomness
    pkb = (bn.g1_to_le(g), b"".join(bn.g2_to_le(x) for x in h_a), b"".join(bn.gt_to_le(x) for x in e_gh_ka))
    for i, t in enumerate(trees):
        pi, c0, c, cp = cport.ac17_cp_encrypt_raw(pkb, hp.to_json(t), pol.JSON, s[i][0], s[i][1], bn.gt_to_le(msgs[i]))
        assert pi == sels[i][2]
        assert got_c0[i * 384:(i + 1) * 384] == c0, i
        assert got_c[row_off[i] * 192:row_off[i + 1] * 192] == c, i
        assert got_cp[i * 384:(i + 1) * 384] == cp, i
        out = cport.ac17_cp_decrypt_raw(c0, c, cp, b"".join(bn.g2_to_le(x) for x in k_0), b"".join(bn.g1_to_le(p) for row in sk_k for p in row),
                                        b"".join(bn.g1_to_le(p) for p in k_p), sels[i][0], sels[i][1])
        assert got_out[i * 384:(i + 1) * 384] == out == bn.gt_to_le(msgs[i]), i
    lines.destroy()
    dpk.destroy()


@pytest.mark.parametrize("shape", ["flat", "mixed"])
def test_config3_bsw_hundred_leaves_against_reference_order(eng, shape):
    from rabe_amd import engine as E
    rnd = random.Random(3)
    rng = cport._Rnd(33)
    names = ["b%d" % i for i in range(100)]
    g1, g2 = cport._g1m(bn.g1_to_le(bn.G1_GEN), rng.fr()), cport._g2m(bn.g2_to_le(bn.G2_GEN), rng.fr())
    beta, alpha = rng.fr(), rng.fr()
    g2_alpha = cport._g2m(g2, alpha)
    pk = {"g1": g1, "g2": g2, "h": cport._g1m(g1, beta), "e_gg_alpha": cport._pair(g1, g2_alpha)}
    r = rng.fr()
    g2_r = cport._g2m(g2, r)
    sk = {"d": cport._g2m(cport._g2a(g2_alpha, g2_r), bn.fr_inv(beta)), "d_j": []}
    for j in names:
        r_j = rng.fr()
        sk["d_j"].append((j, cport._g1m(g1, r_j), cport._g2a(g2_r, cport._g2m(cport._g2m(g2, cport.hash_fr(j)), r_j))))
    leaves = [("leaf", x) for x in names]
    tree = ("and", leaves) if shape == "flat" else ("and", [("or", [leaves[2 * i], leaves[2 * i + 1]]) for i in range(50)])
    policy = hp.to_json(tree)
    tt = hp.TreeTables([tree])
    secret = rnd.randrange(1, bn.R)
    coefs = [rnd.randrange(bn.R) for _ in range(tt.n_coef(0))]
    msg = cport._gtp(pk["e_gg_alpha"], rnd.randrange(1, bn.R))
    from oracle.tape import ListRng
    ct = cport.bsw_encrypt_raw(pk, policy, ListRng([secret] + coefs), msg)                 # reference order
    want = cport.bsw_decrypt_raw(sk, ct)
    assert want == msg
    dpk = E.BswPk(eng, g1, g2, pk["h"], pk["e_gg_alpha"])
    dtt = E.DevTreeTables(eng, tt)
    d_c, d_cp, d_g1, d_g2 = eng.alloc(64), eng.alloc(384), eng.alloc(100 * 64), eng.alloc(100 * 128)
    d_leaf_off = eng.upload_u32([0, 100])
    E.bsw_encrypt_dev(eng, dpk, 1, 100, d_leaf_off, eng.upload_u32([0]), eng.upload_u32([0]), dtt, eng.upload(le(secret)),
                      eng.upload(b"".join(le(x) for x in coefs)), eng.upload_u32([0]), eng.upload(msg), d_c, d_cp, d_g1, d_g2)
    assert eng.download(d_c) == ct["c"] and eng.download(d_cp) == ct["c_p"]
    assert eng.download(d_g1) == b"".join(y[1] for y in ct["c_y"])
    assert eng.download(d_g2) == b"".join(y[2] for y in ct["c_y"])
    ok, idx = hp.pruned_leaf_indices(names, tree)
    assert ok
    z = hp.leaf_coefficients(tree)
    leaf_names = tt.flat[0]["names"]
    d_sk_d = eng.upload(sk["d"])
    d_sk_g1, d_sk_g2 = eng.upload(b"".join(d[1] for d in sk["d_j"])), eng.upload(b"".join(d[2] for d in sk["d_j"]))
    lines = E.BswSkLines(eng, 1, 100, d_sk_d, d_sk_g2)
    for sk_lines in (lines, None):
        d_out = eng.alloc(384)
        E.bsw_decrypt_dev(eng, 1, 2 * len(idx) + 1, 2 * len(idx) + 1, eng.upload_u32([0, 2 * len(idx) + 1]), eng.upload_u32([0]), eng.upload_u32(idx),
                          eng.upload_u32([names.index(leaf_names[y]) for y in idx]), eng.upload(b"".join(le(z[y]) for y in idx)), d_c, d_cp, d_g1,
                          d_g2, d_leaf_off, d_sk_d, d_sk_g1, d_sk_g2, eng.upload_u32([0, 100]), eng.upload_u32([0]), sk_lines, d_out)
        assert eng.download(d_out) == want
    lines.destroy()
    dpk.destroy()


def _offsets(counts):
    out = [0]
    for c in counts:
        out.append(out[-1] + c)
    return out


def test_config4_lsw_two_hundred_leaves_against_reference_order(eng):
    """BASELINE config 4 at its full size, one item: lsw::keygen under a flat 200-leaf AND (199 coefficient draws, 200 share
    randoms) and lsw::decrypt of a 200-attribute ciphertext, device against the reference-order C port byte for byte."""
    from rabe_amd import engine as E
    from oracle.tape import ListRng
    rnd = random.Random(4)
    rng = cport._Rnd(44)
    names = ["c%d" % i for i in range(200)]
    g1, g2 = cport._g1m(bn.g1_to_le(bn.G1_GEN), rng.fr()), cport._g2m(bn.g2_to_le(bn.G2_GEN), rng.fr())
    msk = {"alpha1": rng.fr(), "alpha2": rng.fr()}
    pk = {"g1": g1, "g2": g2, "e_gg_alpha": cport._gtp(cport._pair(g1, g2), msk["alpha1"] * msk["alpha2"] % bn.R)}
    secret = rng.fr()
    msg = cport._gtp(pk["e_gg_alpha"], rng.fr())
    ct = {"e1": cport._gtm(cport._gtp(pk["e_gg_alpha"], secret), msg), "e2": cport._g2m(g2, secret),
          "ej": [(a, cport._g1m(cport._g1m(g1, cport.hash_fr(a)), secret)) for a in names]}
    tree = ("and", [("leaf", x) for x in names])
    policy = hp.to_json(tree)
    tt = hp.TreeTables([tree])
    assert tt.n_leaves(0) == 200 and tt.n_coef(0) == 199
    coefs = [rnd.randrange(bn.R) for _ in range(199)]
    rands = [rnd.randrange(1, bn.R) for _ in range(200)]
    sk = cport.lsw_keygen_raw(pk, msk, policy, ListRng(coefs + rands))                     # reference order
    want = cport.lsw_decrypt_raw(sk, ct)
    assert want == msg
    dpk = E.LswPk(eng, g1, g2)
    dtt = E.DevTreeTables(eng, tt)
    d_d1, d_d2 = eng.alloc(200 * 64), eng.alloc(200 * 128)
    d_leaf_off = eng.upload_u32([0, 200])
    E.lsw_keygen_dev(eng, dpk, 1, 200, d_leaf_off, eng.upload_u32([tt.first_leaf[0]]), eng.upload_u32([tt.first_gate[0]]), dtt,
                     eng.upload(le(msk["alpha1"]) + le(msk["alpha2"])), eng.upload(b"".join(le(x) for x in coefs)), eng.upload_u32([0]),
                     eng.upload(b"".join(le(x) for x in rands)), d_d1, d_d2)
    # leaf order of the flattened tree = DFS order = the reference's share order
    leaf_names = tt.flat[0]["names"]
    assert leaf_names == [row[0] for row in sk["dj"]]
    assert eng.download(d_d1) == b"".join(row[1] for row in sk["dj"])
    assert eng.download(d_d2) == b"".join(row[2] for row in sk["dj"])
    ok, idx = hp.pruned_leaf_indices(names, tree)
    assert ok and len(idx) == 200
    z = hp.leaf_coefficients(tree)
    d_e2 = eng.upload(ct["e2"])
    lines = E.G2Lines(eng, 1, d_e2)
    for e2_lines in (None, lines):
        d_out = eng.alloc(384)
        E.lsw_decrypt_dev(eng, 1, 201, 201, 200, eng.upload_u32([0, 201]), eng.upload_u32([0]), eng.upload_u32(idx),
                          eng.upload_u32([names.index(leaf_names[y]) for y in idx]), eng.upload(b"".join(le(z[y]) for y in idx)),
                          eng.upload(ct["e1"]), d_e2, eng.upload(b"".join(row[1] for row in ct["ej"])), eng.upload_u32([0, 200]),
                          eng.upload_u32([0]), d_d1, d_d2, d_leaf_off, None, e2_lines, d_out)
        assert eng.download(d_out) == want, "prepared e2" if e2_lines else "walking e2"
    lines.destroy()
    dpk.destroy()


def test_config5_aw11_ten_by_twenty_against_reference_order(eng):
    """BASELINE config 5 at its full size, one item: aw11::encrypt under binary ANDs over all 10 x 20 attributes and
    aw11::decrypt with a key holding all of them, device against the reference-order C port byte for byte."""
    from rabe_amd import engine as E
    from oracle.tape import ListRng
    rnd = random.Random(5)
    rng = cport._Rnd(55)
    gk = {"g1": cport._g1m(bn.g1_to_le(bn.G1_GEN), rng.fr()), "g2": cport._g2m(bn.g2_to_le(bn.G2_GEN), rng.fr())}
    egg = cport._pair(gk["g1"], gk["g2"])
    names = ["AUTH%dX%d" % (i // 20, i % 20) for i in range(200)]
    hg = cport.hash_fr("alice")
    pk_attr, sk = {}, {"gid": "alice", "attr": []}
    for nm in names:
        a, y = rng.fr(), rng.fr()
        pk_attr[nm] = (cport._gtp(egg, a), cport._g2m(gk["g2"], y))
        sk["attr"].append((nm, cport._g1m(gk["g1"], (a + hg * y) % bn.R)))

    def nest(nodes):
        return nodes[0] if len(nodes) == 1 else ("and", [nest(nodes[:len(nodes) // 2]), nest(nodes[len(nodes) // 2:])])
    tree = nest([("leaf", x) for x in names])
    policy = hp.to_json(tree)
    tt = hp.TreeTables([tree])
    assert tt.n_leaves(0) == 200
    nc = tt.n_coef(0)
    s = rnd.randrange(1, bn.R)
    coefs = [rnd.randrange(bn.R) for _ in range(2 * nc)]
    rands = [rnd.randrange(1, bn.R) for _ in range(200)]
    msg = cport._gtp(egg, rnd.randrange(1, bn.R))
    ct = cport.aw11_encrypt_raw(gk, pk_attr, policy, ListRng([s] + coefs + rands), msg)       # reference order
    want = cport.aw11_decrypt_raw(gk, sk, ct)
    assert want == msg
    dpk = E.Aw11Pk(eng, gk["g1"], gk["g2"], [pk_attr[nm][0] for nm in names], [pk_attr[nm][1] for nm in names])
    dtt = E.DevTreeTables(eng, tt)
    leaf_names = tt.flat[0]["names"]
    d_leaf_attr = eng.upload_u32([names.index(nm) for nm in leaf_names])
    d_c0, d_c1, d_c2, d_c3 = eng.alloc(384), eng.alloc(200 * 384), eng.alloc(200 * 128), eng.alloc(200 * 128)
    d_row_off = eng.upload_u32([0, 200])
    E.aw11_encrypt_dev(eng, dpk, 1, 200, d_row_off, eng.upload_u32([tt.first_leaf[0]]), eng.upload_u32([tt.first_gate[0]]), eng.upload_u32([nc]), dtt,
                       d_leaf_attr, eng.upload(le(s)), eng.upload(b"".join(le(x) for x in coefs)), eng.upload_u32([0]),
                       eng.upload(b"".join(le(x) for x in rands)), eng.upload(msg), d_c0, d_c1, d_c2, d_c3)
    assert len(ct["c"]) == 200
    assert eng.download(d_c0) == ct["c_0"]
    assert eng.download(d_c1) == b"".join(r[1] for r in ct["c"])
    assert eng.download(d_c2) == b"".join(r[2] for r in ct["c"])
    assert eng.download(d_c3) == b"".join(r[3] for r in ct["c"])
    key_attrs = [a[0] for a in sk["attr"]]
    ok, idx = hp.pruned_leaf_indices(key_attrs, tree)
    assert ok and len(idx) == 200
    z = hp.leaf_coefficients(tree)
    d_out = eng.alloc(384)
    E.aw11_decrypt_dev(eng, 1, 201, 201, 200, eng.upload_u32([0, 201]), eng.upload_u32([0]), eng.upload_u32(idx),
                       eng.upload_u32([key_attrs.index(leaf_names[y]) for y in idx]), eng.upload(b"".join(le(z[y]) for y in idx)),
                       d_c0, d_c1, d_c2, d_c3, d_row_off, eng.upload(cport._g1m(gk["g1"], hg)), eng.upload(b"".join(a[1] for a in sk["attr"])),
                       eng.upload_u32([0, 200]), eng.upload_u32([0]), d_out)
    assert eng.download(d_out) == want
    dpk.destroy()
