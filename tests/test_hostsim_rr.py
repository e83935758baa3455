"""The reduced-radix field core (rabe_amd/csrc/bn254/fp29.h, pairing29.h: 9 signed 29-bit limbs, lazily reduced sums with the
bounds carried in the types) executed on the CPU and compared bit-for-bit with the 8 x 32-bit core of the same headers and with
the Python big-int oracle.  The host build asserts every limb / value bound at run time as well (RB29_CHECK): a violated bound
aborts the test process.  The same comparisons run on the device in tests/test_gpu_rr.py."""
import ctypes
import random

import pytest

from oracle import bn254 as bn
from tests.hostsim import build as hs_build

try:
    HS = hs_build.load()
except Exception as e:  # pragma: no cover
    HS = None

pytestmark = pytest.mark.skipif(HS is None, reason="hostsim library could not be built")

RND = random.Random(20260929)
EDGE = [0, 1, 2, bn.P - 1, bn.P - 2, (bn.P - 1) // 2, (bn.P + 1) // 2, 1 << 253, (1 << 29) - 1, 1 << 29, (1 << 232) - 1, 1 << 232,
        sum(((1 << 28)) << (29 * k) for k in range(8)) % bn.P, sum(((1 << 29) - 1) << (29 * k) for k in range(8)) % bn.P]


def buf(n):
    return (ctypes.c_uint32 * (n // 4))()


def b2c(b):
    return (ctypes.c_uint32 * (len(b) // 4)).from_buffer_copy(b)


def le(x):
    return int(x).to_bytes(32, "little")


def fp2_le(a):
    return le(a[0]) + le(a[1])


def call(name, *ins, out=64):
    o = buf(out)
    getattr(HS, name)(*[b2c(x) for x in ins], o)
    return bytes(o)


def vals():
    return EDGE + [RND.randrange(bn.P) for _ in range(60)]


def test_round_trip_and_half():
    inv2 = pow(2, bn.P - 2, bn.P)
    for a in vals():
        assert call("hs_rr_roundtrip", le(a), out=32) == le(a)
        assert call("hs_rr_fp_half", le(a), out=32) == le(a * inv2 % bn.P)


def test_fp2_operations_match_the_oracle_and_the_8x32_core():
    vs = vals()
    for i in range(len(vs)):
        a = (vs[i], vs[(7 * i + 3) % len(vs)])
        b = (vs[(5 * i + 1) % len(vs)], vs[(11 * i + 2) % len(vs)])
        want = ((a[0] * b[0] - a[1] * b[1]) % bn.P, (a[0] * b[1] + a[1] * b[0]) % bn.P)
        got = call("hs_rr_fp2_mul", fp2_le(a), fp2_le(b))
        assert got == fp2_le(want) == call("hs_fp2_mul", fp2_le(a), fp2_le(b))
        assert call("hs_rr_fp2_sqr", fp2_le(a)) == fp2_le(((a[0] * a[0] - a[1] * a[1]) % bn.P, 2 * a[0] * a[1] % bn.P))
        assert call("hs_rr_fp2_mul_xi", fp2_le(a)) == fp2_le(((9 * a[0] - a[1]) % bn.P, (9 * a[1] + a[0]) % bn.P))
        # ((a + b)(a - b) - 3 a b) / 2 + xi (2 a b - (a + b)(a - b)): unreduced sums of sums, half, norm, norm_lin9 in one chain
        f2 = lambda x, y: ((x[0] * y[0] - x[1] * y[1]) % bn.P, (x[0] * y[1] + x[1] * y[0]) % bn.P)
        m = f2(((a[0] + b[0]) % bn.P, (a[1] + b[1]) % bn.P), ((a[0] - b[0]) % bn.P, (a[1] - b[1]) % bn.P))
        n = f2(a, b)
        inv2 = pow(2, bn.P - 2, bn.P)
        r = ((m[0] - 3 * n[0]) * inv2 % bn.P, (m[1] - 3 * n[1]) * inv2 % bn.P)
        y = ((2 * n[0] - m[0]) % bn.P, (2 * n[1] - m[1]) % bn.P)
        want = ((r[0] + 9 * y[0] - y[1]) % bn.P, (r[1] + 9 * y[1] + y[0]) % bn.P)
        assert call("hs_rr_fp2_mix", fp2_le(a), fp2_le(b)) == fp2_le(want)


@pytest.mark.parametrize("n,kinds", [(1, [0]), (1, [1]), (2, [0, 0]), (2, [1, 0]), (3, [0, 0, 1]), (5, [1, 0, 1, 0, 1]), (4, [0, 2, 1, 0]), (6, [1, 1, 1, 0, 0, 0])])
def test_miller_loop_multi_same_value_and_same_running_points(n, kinds):
    """pairing29.h: miller_loop_multi against pairing.h's.  The points the walking pairs end on are the same field elements, and so is the
    Miller value itself (before any final exponentiation) when every pair walks.  A PREPARED pair replays its lines divided by their
    y-coefficient (an Fq2 factor per line, removed by the final exponentiation): there the two Miller values differ and their final
    exponentiations -- the pairing product the oracle computes -- are the same bytes."""
    ks = [(RND.randrange(1, bn.R), RND.randrange(1, bn.R)) for _ in range(n)]
    p = b"".join(bn.g1_to_le(bn.g1_mul(bn.G1_GEN, a)) for a, _ in ks)
    q = b"".join(bn.g2_to_le(bn.g2_mul(bn.G2_GEN, b)) for _, b in ks)
    kk = (ctypes.c_int * n)(*kinds)
    prepared = any(k == 1 for k in kinds)

    def same(o1, o2):
        if not prepared:
            assert bytes(o1) == bytes(o2)
            return
        assert bytes(o1) != bytes(o2)
        e1, e2, e3 = buf(384), buf(384), buf(384)
        HS.hs_final_exp_ws(o1, e1)
        HS.hs_final_exp_ws(o2, e2)
        HS.hs_rr_final_exp(o2, e3)
        assert bytes(e1) == bytes(e2) == bytes(e3)
        return bytes(e1)

    o1, o2, t1, t2 = buf(384), buf(384), buf(384 * n), buf(384 * n)
    HS.hs_miller_multi(n, kk, b2c(p), b2c(q), o1, t1)
    HS.hs_rr_miller_multi(n, kk, b2c(p), b2c(q), o2, t2)
    e = same(o1, o2)
    assert bytes(t1) == bytes(t2)
    if e is not None and 2 not in kinds:
        acc = bn.GT_ONE
        for a, b in ks:
            acc = bn.gt_mul(acc, bn.gt_pow(bn.pairing(bn.G1_GEN, bn.G2_GEN), a * b % bn.R))
        assert e == bn.gt_to_le(acc)
    if n >= 2:      # an argument at infinity contributes 1
        p2 = bytes(64) + p[64:]
        HS.hs_miller_multi(n, kk, b2c(p2), b2c(q), o1, None)
        HS.hs_rr_miller_multi(n, kk, b2c(p2), b2c(q), o2, None)
        if kinds[0] == 1 and sum(1 for k in kinds if k == 1) == 1:
            prepared = False          # the only prepared pair is the skipped one
        same(o1, o2)


def _rand_fp12():
    return bn.fp12_from_coeffs([RND.randrange(bn.P) for _ in range(12)])


def test_fp12_operations_of_the_final_exponentiation():
    o1, o2 = buf(384), buf(384)
    for _ in range(4):
        a, b = _rand_fp12(), _rand_fp12()
        HS.hs_rr_fp12_mul(b2c(bn.gt_to_le(a)), b2c(bn.gt_to_le(b)), o1)          # a * conj(b)
        assert bytes(o1) == bn.gt_to_le(bn.fp12_mul(a, bn.fp12_conj(b)))
        for k in (1, 2, 3):
            HS.hs_rr_fp12_frob(b2c(bn.gt_to_le(a)), k, o1)
            HS.hs_fp12_frob(b2c(bn.gt_to_le(a)), k, o2)
            assert bytes(o1) == bytes(o2)
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    for k in (1, 2, RND.randrange(bn.R)):
        g = bn.gt_pow(e, k)
        HS.hs_rr_cyclotomic_sqr(b2c(bn.gt_to_le(g)), o1)
        assert bytes(o1) == bn.gt_to_le(bn.fp12_mul(g, g))


def test_final_exponentiation_same_bytes():
    """pairing29.h: final_exponentiation_ws against pairing.h's on Miller values and on arbitrary Fq12 elements"""
    o1, o2 = buf(384), buf(384)
    for _ in range(3):
        f = _rand_fp12()
        HS.hs_final_exp_ws(b2c(bn.gt_to_le(f)), o1)
        HS.hs_rr_final_exp(b2c(bn.gt_to_le(f)), o2)
        assert bytes(o1) == bytes(o2)


@pytest.mark.parametrize("n,kinds", [(1, [0]), (1, [1]), (2, [0, 0]), (3, [0, 0, 0]), (2, [1, 0]), (3, [0, 1, 0]), (5, [1, 0, 1, 0, 1]), (4, [0, 2, 1, 0]),
                                     (6, [1, 1, 1, 0, 0, 0])])
def test_miller_loop_on_two_lanes_same_value_and_same_running_points(n, kinds):
    """pairing29p.h: miller_loop_pair -- one unit on TWO lanes (the device kernel k_miller_pair_rr; here two host threads in lock step, every
    shared access bracketed by barriers) -- against pairing29.h's one-lane loop: the Miller value is the SAME field element (prepared lines are
    unit-y lines in both), and so are the points the walking pairs end on, for even and odd numbers of walking pairs (lane 1 idle in the
    last round), prepared and walking pairs mixed, and a skipped pair.  The host build checks every limb / value bound at run time."""
    ks = [(RND.randrange(1, bn.R), RND.randrange(1, bn.R)) for _ in range(n)]
    p = b"".join(bn.g1_to_le(bn.g1_mul(bn.G1_GEN, a)) for a, _ in ks)
    q = b"".join(bn.g2_to_le(bn.g2_mul(bn.G2_GEN, b)) for _, b in ks)
    kk = (ctypes.c_int * n)(*kinds)
    o1, o2, t1, t2 = buf(384), buf(384), buf(384 * n), buf(384 * n)
    HS.hs_rr_miller_multi(n, kk, b2c(p), b2c(q), o1, t1)
    HS.hs_rr_miller_pair(n, kk, b2c(p), b2c(q), o2, t2)
    assert bytes(o1) == bytes(o2)
    assert bytes(t1) == bytes(t2)
    if n >= 2:      # an argument at infinity: the pair is skipped, the lanes' shares of the walking pairs shift
        p2 = bytes(64) + p[64:]
        HS.hs_rr_miller_multi(n, kk, b2c(p2), b2c(q), o1, None)
        HS.hs_rr_miller_pair(n, kk, b2c(p2), b2c(q), o2, None)
        assert bytes(o1) == bytes(o2)
