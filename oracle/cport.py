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
