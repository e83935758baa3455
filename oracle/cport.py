"""ctypes front-end of oracle/c/rabe_ref.c -- TEST ORACLE / CPU BASELINE, not product code.

The C file runs the reference's group loops in the reference's operation order; the string work
(policy parse, MSP, pruning) is done here with oracle/policy.py and handed over as numbers."""
import ctypes
import os
import random

from . import bn254 as bn
from . import cbuild
from . import policy as pol

_LIB = None


def available():
    try:
        lib()
        return True
    except Exception:
        return False


def lib():
    global _LIB
    if _LIB is None:
        path = cbuild.LIB if os.path.exists(cbuild.LIB) else cbuild.build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _buf(n):
    return ctypes.create_string_buffer(n)


def g1_mul(p, k):
    o = _buf(64); lib().rref_g1_mul(bn.g1_to_le(p), bn.fr_to_le(k), o); return bn.g1_from_le(o.raw)


def g2_mul(p, k):
    o = _buf(128); lib().rref_g2_mul(bn.g2_to_le(p), bn.fr_to_le(k), o); return bn.g2_from_le(o.raw)


def pairing(p, q):
    o = _buf(384); lib().rref_pairing(bn.g1_to_le(p), bn.g2_to_le(q), o); return bn.gt_from_le(o.raw)


def gt_pow(a, k):
    o = _buf(384); lib().rref_gt_pow(bn.gt_to_le(a), bn.fr_to_le(k), o); return bn.gt_from_le(o.raw)


def hash_fr(label):
    o = _buf(32); b = label.encode("utf-8"); lib().rref_hash_fr(b, ctypes.c_size_t(len(b)), o); return int.from_bytes(o.raw, "little")


def ac17_cp_encrypt_raw(pk_bytes, policy, language, s0, s1, msg_bytes):
    """pk_bytes = (g, h_a[3] joined, e_gh_ka[2] joined) canonical.  Returns (pi, c0, c, cp) as bytes."""
    tree = pol.parse(policy, language)
    m, pi, _c = pol.calculate_msp(tree)
    n_rows, n_cols = len(m), len(m[0])
    flat = (ctypes.c_int8 * (n_rows * n_cols))(*[x for row in m for x in row])
    stride = max(len(x.encode()) for x in pi) + 1
    labels = b"".join(x.encode().ljust(stride, b"\0") for x in pi)
    c0, c, cp = _buf(384), _buf(n_rows * 192), _buf(384)
    lib().rref_ac17_cp_encrypt(pk_bytes[0], pk_bytes[1], pk_bytes[2], n_rows, n_cols, flat, labels, stride,
                               bn.fr_to_le(s0) + bn.fr_to_le(s1), msg_bytes, c0, c, cp)
    return pi, c0.raw, c.raw, cp.raw


def ac17_cp_decrypt_raw(ct_c0, ct_c, ct_cp, sk_k0, sk_k, sk_kp, ct_sel, sk_sel):
    out = _buf(384)
    a1 = (ctypes.c_uint32 * max(1, len(ct_sel)))(*ct_sel)
    a2 = (ctypes.c_uint32 * max(1, len(sk_sel)))(*sk_sel)
    lib().rref_ac17_cp_decrypt(ct_c0, ct_c, ct_cp, sk_k0, sk_k, sk_kp, a1, len(ct_sel), a2, len(sk_sel), out)
    return out.raw


def ac17_encdec(policy, n_attrs, n_items, seed=0):
    """CPU-baseline workload: n_items x (cp_encrypt + cp_decrypt) at n_attrs attributes in reference order.
    Key material is synthetic (random points), which does not change the operation count."""
    rnd = random.Random(seed)

    def rfr():
        return rnd.randrange(1, bn.R)
    g = g1_mul(bn.G1_GEN, rfr())
    h = g2_mul(bn.G2_GEN, rfr())
    e = pairing(g, h)
    pk = (bn.g1_to_le(g), b"".join(bn.g2_to_le(g2_mul(h, rfr())) for _ in range(3)), b"".join(bn.gt_to_le(gt_pow(e, rfr())) for _ in range(2)))
    attrs = ["a%d" % (i + 1) for i in range(n_attrs)]
    sk_k0 = b"".join(bn.g2_to_le(g2_mul(h, rfr())) for _ in range(3))
    sk_k = b"".join(bn.g1_to_le(g1_mul(g, rfr())) for _ in range(3 * n_attrs))
    sk_kp = b"".join(bn.g1_to_le(g1_mul(g, rfr())) for _ in range(3))
    tree = pol.parse(policy, pol.JSON)
    ok, lst = pol.calc_pruned(attrs, tree)
    assert ok
    msg = bn.gt_to_le(e)
    outs = []
    import time
    seconds = 0.0
    for _ in range(n_items):
        s0, s1 = rfr(), rfr()
        t0 = time.perf_counter()
        pi, c0, c, cp = ac17_cp_encrypt_raw(pk, policy, pol.JSON, s0, s1, msg)
        ct_sel, sk_sel = [], []
        for name, _nc in lst:
            ct_sel += [i for i, n in enumerate(pi) if n == name]
            sk_sel += [i for i, n in enumerate(attrs) if n == name]
        outs.append(ac17_cp_decrypt_raw(c0, c, cp, sk_k0, sk_k, sk_kp, ct_sel, sk_sel))
        seconds += time.perf_counter() - t0
    return outs, seconds


# ---------------------------------------------------------------------------------------------------------------------
# bsw / lsw / aw11 in the reference's operation order over the C primitives (canonical bytes in, canonical bytes out):
# the CPU baselines of bench.py --config 3 / 4 / 5.  Every loop below follows the cited reference lines statement by
# statement (same number of G*Fr, pairing, Gt::pow, inverse calls); string / Fr work goes through oracle/policy.py.
def _g1m(p, k):
    o = _buf(64); lib().rref_g1_mul(p, bn.fr_to_le(k % bn.R), o); return o.raw


def _g2m(p, k):
    o = _buf(128); lib().rref_g2_mul(p, bn.fr_to_le(k % bn.R), o); return o.raw


def _g1a(p, q):
    o = _buf(64); lib().rref_g1_add(p, q, o); return o.raw


def _g2a(p, q):
    o = _buf(128); lib().rref_g2_add(p, q, o); return o.raw


def _pair(p, q):
    o = _buf(384); lib().rref_pairing(p, q, o); return o.raw


def _gtp(a, k):
    o = _buf(384); lib().rref_gt_pow(a, bn.fr_to_le(k % bn.R), o); return o.raw


def _gtm(a, b):
    o = _buf(384); lib().rref_gt_mul(a, b, o); return o.raw


def _gti(a):
    o = _buf(384); lib().rref_gt_inv(a, o); return o.raw


GT_ONE_LE = (1).to_bytes(32, "little") + bytes(352)


def _tree(kind, names, binary=False):
    def nest(nodes):
        if len(nodes) == 1:
            return nodes[0]
        return '{"name": "and", "children": [%s, %s]}' % (nest(nodes[:len(nodes) // 2]), nest(nodes[len(nodes) // 2:]))
    leaves = ['{"name": "%s"}' % x for x in names]
    if kind == "mixed":
        ors = ['{"name": "or", "children": [%s, %s]}' % (leaves[2 * i], leaves[2 * i + 1]) for i in range(len(leaves) // 2)]
        return nest(ors) if binary else '{"name": "and", "children": [%s]}' % ", ".join(ors)
    if kind == "flat" and not binary:
        return '{"name": "and", "children": [%s]}' % ", ".join(leaves)
    return nest(leaves)


class _Rnd:
    def __init__(self, seed):
        self.r = random.Random(seed)

    def fr(self):
        return self.r.randrange(1, bn.R)


def bsw_encrypt_raw(pk, policy, rng, msg):
    """bsw/mod.rs:217-251; pk = dict of canonical bytes"""
    secret = rng.fr()
    tree = pol.parse(policy, pol.JSON)
    shares = pol.gen_shares_policy(secret, tree, rng)
    c = _g1m(pk["h"], secret)
    c_p = _gtm(_gtp(pk["e_gg_alpha"], secret), msg)
    c_y = []
    for node, val in shares:
        j = pol.remove_index(node)
        c_y.append((node, _g1m(pk["g1"], val), _g2m(_g2m(pk["g2"], hash_fr(j)), val)))
    return {"policy": policy, "c": c, "c_p": c_p, "c_y": c_y}


def bsw_decrypt_raw(sk, ct):
    """bsw/mod.rs:260-318; sk = {"d": bytes, "d_j": [(name, g1, g2)]}"""
    attr = [v[0] for v in sk["d_j"]]
    tree = pol.parse(ct["policy"], pol.JSON)
    assert pol.traverse_policy(attr, tree)
    ok, pruned = pol.calc_pruned(attr, tree)
    assert ok
    z = pol.calc_coefficients(tree, 1)
    a = GT_ONE_LE
    for k, j in pruned:
        c_y = next((x for x in ct["c_y"] if x[0] == j), None)
        d_j = next((x for x in sk["d_j"] if x[0] == k), None)
        if c_y is None or d_j is None:
            continue
        for zname, zval in z:
            if zname == j:
                t = _gtm(_pair(c_y[1], d_j[2]), _gti(_pair(d_j[1], c_y[2])))
                a = _gtm(a, _gtp(t, zval))
    return _gtm(ct["c_p"], _gti(_gtm(_pair(ct["c"], sk["d"]), _gti(a))))


def bsw_encdec(n_attrs, n_items, tree="flat", seed=0):
    """n_items x (bsw::encrypt + bsw::decrypt) at n_attrs leaves; returns the seconds spent inside the two functions"""
    import time
    rng = _Rnd(seed)
    g1, g2 = _g1m(bn.g1_to_le(bn.G1_GEN), rng.fr()), _g2m(bn.g2_to_le(bn.G2_GEN), rng.fr())
    beta, alpha = rng.fr(), rng.fr()
    g2_alpha = _g2m(g2, alpha)
    pk = {"g1": g1, "g2": g2, "h": _g1m(g1, beta), "e_gg_alpha": _pair(g1, g2_alpha)}
    names = ["b%d" % i for i in range(n_attrs)]
    r = rng.fr()
    g2_r = _g2m(g2, r)
    sk = {"d": _g2m(_g2a(g2_alpha, g2_r), bn.fr_inv(beta)), "d_j": []}
    for j in names:                                                   # bsw::keygen :125-152 (input generation, untimed)
        r_j = rng.fr()
        sk["d_j"].append((j, _g1m(g1, r_j), _g2a(g2_r, _g2m(g2, hash_fr(j) * r_j))))
    policy = _tree(tree, names)
    seconds = 0.0
    for _ in range(n_items):
        msg = _gtp(pk["e_gg_alpha"], rng.fr())
        t0 = time.perf_counter()
        ct = bsw_encrypt_raw(pk, policy, rng, msg)
        out = bsw_decrypt_raw(sk, ct)
        seconds += time.perf_counter() - t0
        assert out == msg
    return seconds


def lsw_keygen_raw(pk, msk, policy, rng):
    """lsw/mod.rs:121-170, positive leaves"""
    tree = pol.parse(policy, pol.JSON)
    shares = pol.gen_shares_policy(msk["alpha1"], tree, rng)
    dj = []
    for share_str, share_value in shares:
        striped = pol.remove_index(share_str)
        rnd_ = rng.fr()
        share_hash = _g1m(pk["g1"], hash_fr(striped))
        dj.append((striped, _g1a(_g1m(pk["g1"], msk["alpha2"] * share_value), _g1m(share_hash, rnd_)), _g2m(pk["g2"], rnd_)))
    return {"policy": policy, "dj": dj}


def lsw_decrypt_raw(sk, ct):
    """lsw/mod.rs:228-290, positive leaves"""
    attr = [a[0] for a in ct["ej"]]
    tree = pol.parse(sk["policy"], pol.JSON)
    ok, lst = pol.calc_pruned(attr, tree)
    assert ok
    prod_t = GT_ONE_LE
    coeff_list = pol.calc_coefficients(tree, 1)
    for name, name_col in lst:
        sk_attr = next(a for a in sk["dj"] if a[0] == name)
        ct_attr = next(a for a in ct["ej"] if a[0] == name)
        coeff = next(c for c in coeff_list if c[0] == name_col)
        z_y = _gtm(_pair(sk_attr[1], ct["e2"]), _gti(_pair(ct_attr[1], sk_attr[2])))
        prod_t = _gtm(prod_t, _gtp(z_y, coeff[1]))
    return _gtm(ct["e1"], _gti(prod_t))


def lsw_keygen_dec(n_attrs, n_items, tree="flat", seed=0):
    import time
    rng = _Rnd(seed)
    g1, g2 = _g1m(bn.g1_to_le(bn.G1_GEN), rng.fr()), _g2m(bn.g2_to_le(bn.G2_GEN), rng.fr())
    msk = {"alpha1": rng.fr(), "alpha2": rng.fr()}
    pk = {"g1": g1, "g2": g2, "e_gg_alpha": _gtp(_pair(g1, g2), msk["alpha1"] * msk["alpha2"])}
    names = ["c%d" % i for i in range(n_attrs)]
    secret = rng.fr()                                                  # lsw::encrypt :180-219 (only what decrypt reads; untimed)
    msg = _gtp(pk["e_gg_alpha"], rng.fr())
    ct = {"e1": _gtm(_gtp(pk["e_gg_alpha"], secret), msg), "e2": _g2m(g2, secret),
          "ej": [(a, _g1m(_g1m(g1, hash_fr(a)), secret)) for a in names]}
    policy = _tree(tree, names)
    seconds = 0.0
    for _ in range(n_items):
        t0 = time.perf_counter()
        sk = lsw_keygen_raw(pk, msk, policy, rng)
        out = lsw_decrypt_raw(sk, ct)
        seconds += time.perf_counter() - t0
        assert out == msg
    return seconds


def aw11_encrypt_raw(gk, pk_attr, policy, rng, msg):
    """aw11/mod.rs:241-289; pk_attr = {NAME: (egg_alpha, g2_y)}"""
    tree = pol.parse(policy, pol.JSON)
    pol.calculate_msp(tree)
    s = rng.fr()
    s_shares = pol.gen_shares_policy(s, tree, rng)
    w_shares = pol.gen_shares_policy(0, tree, rng)
    c_0 = _gtm(msg, _gtp(_pair(gk["g1"], gk["g2"]), s))
    c = []
    for i, (attr_name, attr_share) in enumerate(s_shares):
        r_x = rng.fr()
        pk_a = pk_attr.get(pol.remove_index(attr_name.upper()))
        if pk_a is None:
            continue
        c.append((attr_name.upper(), _gtm(_gtp(_pair(gk["g1"], gk["g2"]), attr_share), _gtp(pk_a[0], r_x)), _g2m(gk["g2"], r_x),
                  _g2a(_g2m(pk_a[1], r_x), _g2m(gk["g2"], w_shares[i][1]))))
    return {"policy": policy, "c_0": c_0, "c": c}


def aw11_decrypt_raw(gk, sk, ct):
    """aw11/mod.rs:298-366; sk = {"gid": str, "attr": [(NAME, K)]}"""
    str_attr = [a[0] for a in sk["attr"]]
    tree = pol.parse(ct["policy"], pol.JSON)
    assert pol.traverse_policy(str_attr, tree)
    ok, lst = pol.calc_pruned(str_attr, tree)
    coeff_list = pol.calc_coefficients(tree, 1)
    assert ok
    h = _g1m(gk["g1"], hash_fr(sk["gid"]))
    egg_s = GT_ONE_LE
    for name, name_col in lst:
        sk_attr = next(a for a in sk["attr"] if a[0] == name)
        ct_attr = next(a for a in ct["c"] if a[0] == name_col)
        num = _gtm(ct_attr[1], _pair(h, ct_attr[3]))
        dem = _pair(sk_attr[1], ct_attr[2])
        coeff = next(c[1] for c in coeff_list if c[0] == name_col)
        egg_s = _gtm(egg_s, _gtp(_gtm(num, _gti(dem)), coeff))
    return _gtm(ct["c_0"], _gti(egg_s))


def aw11_encdec(n_attrs, n_items, tree="nested", seed=0):
    import time
    rng = _Rnd(seed)
    gk = {"g1": _g1m(bn.g1_to_le(bn.G1_GEN), rng.fr()), "g2": _g2m(bn.g2_to_le(bn.G2_GEN), rng.fr())}
    egg = _pair(gk["g1"], gk["g2"])
    per = max(1, n_attrs // 10)
    names = ["AUTH%dX%d" % (i // per, i % per) for i in range(n_attrs)]
    hg = hash_fr("alice")
    pk_attr, sk = {}, {"gid": "alice", "attr": []}
    for nm in names:                                                   # authgen :121-151 and keygen :165-231 (input generation, untimed)
        a, y = rng.fr(), rng.fr()
        pk_attr[nm] = (_gtp(egg, a), _g2m(gk["g2"], y))
        sk["attr"].append((nm, _g1m(gk["g1"], a + hg * y)))
    policy = _tree(tree, names, binary=True)
    seconds = 0.0
    for _ in range(n_items):
        msg = _gtp(egg, rng.fr())
        t0 = time.perf_counter()
        ct = aw11_encrypt_raw(gk, pk_attr, policy, rng, msg)
        out = aw11_decrypt_raw(gk, sk, ct)
        seconds += time.perf_counter() - t0
        assert out == msg
    return seconds
