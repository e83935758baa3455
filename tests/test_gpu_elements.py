"""GPU parity, Level E: every element operation of the C ABI against the Python oracle, bit-exact."""
import hashlib
import random

import pytest

from oracle import bn254 as bn

pytestmark = pytest.mark.gpu

RND = random.Random(777)


@pytest.fixture(scope="module")
def eng():
    from rabe_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def le(x):
    return int(x).to_bytes(32, "little")


def test_device_is_gfx950(eng):
    n_cu, name = eng.device_info()
    assert "gfx950" in name and n_cu >= 200


def test_fr_ops(eng):
    n = 300
    a = [RND.randrange(bn.R) for _ in range(n)]
    b = [RND.randrange(bn.R) for _ in range(n)]
    a[0], b[0], a[1], b[1], a[2] = 0, 0, bn.R - 1, bn.R - 1, 1
    A, B = [le(x) for x in a], [le(x) for x in b]
    assert eng.fr_op(0, A, B) == [le((x + y) % bn.R) for x, y in zip(a, b)]
    assert eng.fr_op(1, A, B) == [le((x - y) % bn.R) for x, y in zip(a, b)]
    assert eng.fr_op(2, A, B) == [le((x * y) % bn.R) for x, y in zip(a, b)]
    assert eng.fr_op(3, A) == [le((-x) % bn.R) for x in a]
    assert eng.fr_op(4, A[:40]) == [le(pow(x, bn.R - 2, bn.R)) for x in a[:40]]


def test_fr_from_digest(eng):
    labels = ["A00", "B21", "0110", "attribute-with-long-name", ""]
    digs = [hashlib.sha3_256(s.encode()).digest() for s in labels] + [b"\xff" * 32, b"\x00" * 32]
    got = eng.fr_from_be32_reduce(digs)
    assert got == [le(bn.fr_from_be32_reduce(d)) for d in digs]


def test_g1_ops(eng):
    ks = [RND.randrange(1, bn.R) for _ in range(6)]
    pts = [bn.g1_mul(bn.G1_GEN, k) for k in ks]
    P = [bn.g1_to_le(p) for p in pts]
    # add incl. doubling, inverse, infinity on either side
    a = P + [P[0], P[0], bytes(64), P[1], bytes(64)]
    b = P[1:] + P[:1] + [P[0], bn.g1_to_le(bn.g1_neg(pts[0])), P[2], bytes(64), bytes(64)]
    want = [bn.g1_to_le(bn.g1_add(bn.g1_from_le(x), bn.g1_from_le(y))) for x, y in zip(a, b)]
    assert eng.g1_add(a, b) == want
    assert eng.g1_neg(P + [bytes(64)]) == [bn.g1_to_le(bn.g1_neg(p)) for p in pts] + [bytes(64)]
    sc = [RND.randrange(bn.R) for _ in range(6)]
    sc[0], sc[1], sc[2] = 0, 1, bn.R - 1
    assert eng.g1_mul(P, [le(k) for k in sc]) == [bn.g1_to_le(bn.g1_mul(p, k)) for p, k in zip(pts, sc)]
    assert eng.g1_on_curve(P + [bytes(64), le(1) + le(3)]) == [1] * 6 + [1, 0]


def test_g2_ops(eng):
    ks = [RND.randrange(1, bn.R) for _ in range(4)]
    pts = [bn.g2_mul(bn.G2_GEN, k) for k in ks]
    Q = [bn.g2_to_le(p) for p in pts]
    a = Q + [Q[0], Q[0], bytes(128)]
    b = Q[1:] + Q[:1] + [Q[0], bn.g2_to_le(bn.g2_neg(pts[0])), Q[2]]
    want = [bn.g2_to_le(bn.g2_add(bn.g2_from_le(x), bn.g2_from_le(y))) for x, y in zip(a, b)]
    assert eng.g2_add(a, b) == want
    sc = [RND.randrange(bn.R) for _ in range(4)]
    sc[0] = 0
    assert eng.g2_mul(Q, [le(k) for k in sc]) == [bn.g2_to_le(bn.g2_mul(p, k)) for p, k in zip(pts, sc)]
    assert eng.g2_neg(Q) == [bn.g2_to_le(bn.g2_neg(p)) for p in pts]
    assert eng.g2_on_curve(Q + [bytes(128)]) == [1] * 5


def test_pairing_and_gt(eng):
    k1, k2 = RND.randrange(1, bn.R), RND.randrange(1, bn.R)
    p, q = bn.g1_mul(bn.G1_GEN, k1), bn.g2_mul(bn.G2_GEN, k2)
    e_gen = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    want = bn.gt_pow(e_gen, k1 * k2 % bn.R)          # bilinearity: avoids a second slow oracle pairing
    got = eng.pairing([bn.g1_to_le(bn.G1_GEN), bn.g1_to_le(p), bytes(64)],
                      [bn.g2_to_le(bn.G2_GEN), bn.g2_to_le(q), bn.g2_to_le(q)])
    assert got[0] == bn.gt_to_le(e_gen)
    assert got[1] == bn.gt_to_le(want)
    assert got[2] == bn.gt_to_le(bn.GT_ONE)
    # Gt mul / inv / pow
    a, b = e_gen, want
    assert eng.gt_mul([bn.gt_to_le(a)], [bn.gt_to_le(b)]) == [bn.gt_to_le(bn.gt_mul(a, b))]
    assert eng.gt_inv([bn.gt_to_le(b)]) == [bn.gt_to_le(bn.gt_inv(b))]
    k = RND.randrange(bn.R)
    assert eng.gt_pow([bn.gt_to_le(a), bn.gt_to_le(a)], [le(k), le(0)]) == [bn.gt_to_le(bn.gt_pow(a, k)), bn.gt_to_le(bn.GT_ONE)]
    # product of pairings with one final exponentiation: e(P,Q) * e(-P,Q) = 1 ; e(aG,H)e(bG,H) = e(G,H)^(a+b)
    prod = eng.pairing_product([0, 2, 4, 4],
                               [bn.g1_to_le(p), bn.g1_to_le(bn.g1_neg(p)), bn.g1_to_le(bn.g1_mul(bn.G1_GEN, 5)), bn.g1_to_le(bn.g1_mul(bn.G1_GEN, 7))],
                               [bn.g2_to_le(q), bn.g2_to_le(q), bn.g2_to_le(bn.G2_GEN), bn.g2_to_le(bn.G2_GEN)])
    assert prod[0] == bn.gt_to_le(bn.GT_ONE)
    assert prod[1] == bn.gt_to_le(bn.gt_pow(e_gen, 12))
    assert prod[2] == bn.gt_to_le(bn.GT_ONE)          # empty product


def test_fixed_base_tables(eng):
    kb = RND.randrange(1, bn.R)
    base1 = bn.g1_mul(bn.G1_GEN, kb)
    base2 = bn.g2_mul(bn.G2_GEN, kb)
    sc = [RND.randrange(bn.R) for _ in range(5)] + [0, 1, 255, 256, bn.R - 1, (1 << 248)]
    K = [le(k) for k in sc]
    t1 = eng.g1_table(bn.g1_to_le(base1))
    assert t1.mul(K) == [bn.g1_to_le(bn.g1_mul(base1, k)) for k in sc]
    t2 = eng.g2_table(bn.g2_to_le(base2))
    assert t2.mul(K) == [bn.g2_to_le(bn.g2_mul(base2, k)) for k in sc]
    e_gen = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    tt = eng.gt_table(bn.gt_to_le(e_gen))
    assert tt.mul(K[:7]) == [bn.gt_to_le(bn.gt_pow(e_gen, k)) for k in sc[:7]]
    t1.destroy(); t2.destroy(); tt.destroy()


def test_non_canonical_scalars_are_reduced_not_trusted(eng):
    """the raw C ABI accepts any rhip_fr word; the chains below (3k in 256 bits for the NAF, windows, GLV) need k < r,
    so the load reduces -- k + r, k + 5r and 2^256 - 1 give the group's answer for k mod r."""
    base = bn.g1_mul(bn.G1_GEN, 987654321)
    q2 = bn.g2_mul(bn.G2_GEN, 123456789)
    ks = [5, bn.R - 1, 1 << 253]
    raw = [k + bn.R for k in ks] + [7 + 5 * bn.R, (1 << 256) - 1, bn.R]
    assert all(x < 1 << 256 for x in raw)
    got = eng.g1_mul([bn.g1_to_le(base)] * len(raw), [le(x) for x in raw])
    assert got == [bn.g1_to_le(bn.g1_mul(base, x % bn.R)) for x in raw]
    got = eng.g2_mul([bn.g2_to_le(q2)] * len(raw), [le(x) for x in raw])
    assert got == [bn.g2_to_le(bn.g2_mul(q2, x % bn.R)) for x in raw]
    e = bn.pairing(base, q2)
    got = eng.gt_pow([bn.gt_to_le(e)] * 3, [le(x) for x in raw[:3]])
    assert got == [bn.gt_to_le(bn.gt_pow(e, x % bn.R)) for x in raw[:3]]


def test_membership_checks_reject_non_canonical_coordinates(eng):
    p = bn.g1_mul(bn.G1_GEN, 31337)
    good = bn.g1_to_le(p)
    x, y = p
    assert x + bn.P < 1 << 256
    alias = (x + bn.P).to_bytes(32, "little") + y.to_bytes(32, "little")          # the same residue, a second encoding
    d = eng.upload(good + alias)
    out = eng.alloc(8)
    from rabe_amd.engine import _sz
    eng._check(eng.lib.rhip_g1_on_curve(eng.ctx, _sz(2), d.ptr, out.ptr))
    import struct
    assert struct.unpack("<2I", eng.download(out)) == (1, 0)
