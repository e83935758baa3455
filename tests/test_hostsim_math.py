"""The engine's BN254 math (rabe_amd/csrc/bn254/*.h -- the exact functions the HIP kernels call)
executed on the CPU and compared bit-for-bit with the Python big-int oracle.  No GPU needed; the
same comparisons run on the device in tests/test_gpu_elements.py."""
import ctypes
import random

import pytest

from oracle import bn254 as bn
from tests.hostsim import build as hs_build

try:
    HS = hs_build.load()
except Exception as e:  # pragma: no cover
    HS = None
    _ERR = e

pytestmark = pytest.mark.skipif(HS is None, reason="hostsim library could not be built")

RND = random.Random(20260928)


def buf(n):
    return (ctypes.c_uint32 * (n // 4))()


def b2c(b):
    return (ctypes.c_uint32 * (len(b) // 4)).from_buffer_copy(b)


def call(name, *ins, out=32):
    o = buf(out)
    args = [b2c(x) if isinstance(x, (bytes, bytearray)) else x for x in ins]
    getattr(HS, name)(*args, o)
    return bytes(o)


def le(x, mod=None):
    return int(x).to_bytes(32, "little")


def fp2_le(a):
    return le(a[0]) + le(a[1])


def rand_fp():
    return RND.randrange(bn.P)


def rand_fp12():
    return bn.fp12_from_coeffs([rand_fp() for _ in range(12)])


EDGE = [0, 1, 2, bn.P - 1, bn.P - 2, (1 << 253), (1 << 32) - 1, 1 << 32]


def test_fp_ops():
    vals = EDGE + [rand_fp() for _ in range(40)]
    for a in vals:
        for b in (vals[0], vals[3], rand_fp(), rand_fp()):
            assert call("hs_fp_mul", le(a), le(b)) == le(a * b % bn.P)
            assert call("hs_fp_add", le(a), le(b)) == le((a + b) % bn.P)
            assert call("hs_fp_sub", le(a), le(b)) == le((a - b) % bn.P)
        assert call("hs_fp_neg", le(a)) == le((-a) % bn.P)
    for a in [1, 2, bn.P - 1] + [rand_fp() for _ in range(5)]:
        assert call("hs_fp_inv", le(a)) == le(pow(a, bn.P - 2, bn.P))
    assert call("hs_fp_inv", le(0)) == le(0)
    # the inversion is a binary extended Euclid on the Montgomery residue x = a R: residues that are tiny, powers of two,
    # all-ones patterns and p - small drive its shortest / longest step sequences
    rinv = pow(1 << 256, -1, bn.P)
    residues = [1, 2, 3, 4, 1 << 31, 1 << 32, 1 << 128, 1 << 253, (1 << 253) - 1, (1 << 253) + 1, bn.P - 1, bn.P - 2, (bn.P + 1) // 2,
                (bn.P - 1) // 2, 0x55555555 * ((1 << 248) // 0xFFFFFFFF)] + [RND.randrange(1, bn.P) for _ in range(150)]
    for x in residues:
        a = x * rinv % bn.P
        assert call("hs_fp_inv", le(a)) == le(pow(a, bn.P - 2, bn.P))


def test_fr_ops():
    for _ in range(20):
        a, b = RND.randrange(bn.R), RND.randrange(bn.R)
        assert call("hs_fr_mul", le(a), le(b)) == le(a * b % bn.R)
    rinv = pow(1 << 256, -1, bn.R)
    for x in [1, 2, 3, 1 << 253, bn.R - 1, (bn.R + 1) // 2] + [RND.randrange(1, bn.R) for _ in range(60)]:
        a = x * rinv % bn.R
        assert call("hs_fr_inv", le(a)) == le(pow(a, bn.R - 2, bn.R))
    assert call("hs_fr_inv", le(0)) == le(0)
    # Fr::from_slice on an arbitrary 256-bit digest (reduction mod r)
    for x in [(1 << 256) - 1, bn.R, bn.R + 5, RND.getrandbits(256), 0]:
        assert call("hs_fr_reduce256", le(x)) == le(x % bn.R)


def test_fp2_ops():
    for _ in range(20):
        a = (rand_fp(), rand_fp())
        b = (rand_fp(), rand_fp())
        assert call("hs_fp2_mul", fp2_le(a), fp2_le(b), out=64) == fp2_le(bn.fp2_mul(a, b))
        assert call("hs_fp2_sqr", fp2_le(a), out=64) == fp2_le(bn.fp2_sqr(a))
        assert call("hs_fp2_mul_xi", fp2_le(a), out=64) == fp2_le(bn.fp2_mul_xi(a))
    a = (rand_fp(), rand_fp())
    assert call("hs_fp2_inv", fp2_le(a), out=64) == fp2_le(bn.fp2_inv(a))


def test_fp12_ops():
    for _ in range(6):
        a, b = rand_fp12(), rand_fp12()
        assert call("hs_fp12_mul", bn.gt_to_le(a), bn.gt_to_le(b), out=384) == bn.gt_to_le(bn.fp12_mul(a, b))
        assert call("hs_fp12_sqr", bn.gt_to_le(a), out=384) == bn.gt_to_le(bn.fp12_sqr(a))
        for k in (1, 2, 3):
            assert call("hs_fp12_frob", bn.gt_to_le(a), k, out=384) == bn.gt_to_le(bn.fp12_pow(a, bn.P ** k))
        l0, l1, l3 = [(rand_fp(), rand_fp()) for _ in range(3)]
        line = ((l0, bn.FP2_ZERO, bn.FP2_ZERO), (l1, l3, bn.FP2_ZERO))
        assert call("hs_fp12_mul_by_line", bn.gt_to_le(a), fp2_le(l0), fp2_le(l1), fp2_le(l3), out=384) == \
            bn.gt_to_le(bn.fp12_mul(a, line))
    a = rand_fp12()
    assert call("hs_fp12_inv", bn.gt_to_le(a), out=384) == bn.gt_to_le(bn.fp12_inv(a))


def test_cyclotomic_sqr():
    # an element of the cyclotomic subgroup: x^((p^6-1)(p^2+1))
    x = bn.fp12_pow(rand_fp12(), bn.FE_EASY)
    assert call("hs_fp12_cyclotomic_sqr", bn.gt_to_le(x), out=384) == bn.gt_to_le(bn.fp12_sqr(x))


def test_g1_g2_ops():
    for _ in range(4):
        k1, k2 = RND.randrange(1, bn.R), RND.randrange(1, bn.R)
        p1, p2 = bn.g1_mul(bn.G1_GEN, k1), bn.g1_mul(bn.G1_GEN, k2)
        assert call("hs_g1_add", bn.g1_to_le(p1), bn.g1_to_le(p2), out=64) == bn.g1_to_le(bn.g1_add(p1, p2))
        assert call("hs_g1_add_jac", bn.g1_to_le(p1), bn.g1_to_le(p2), le(rand_fp()), out=64) == bn.g1_to_le(bn.g1_add(p1, p2))
        assert call("hs_g1_madd_inl", bn.g1_to_le(p1), bn.g1_to_le(p2), le(rand_fp()), out=64) == bn.g1_to_le(bn.g1_add(p1, p2))
        assert call("hs_g1_mul", bn.g1_to_le(p1), le(k2), out=64) == bn.g1_to_le(bn.g1_mul(p1, k2))
        q1, q2 = bn.g2_mul(bn.G2_GEN, k1), bn.g2_mul(bn.G2_GEN, k2)
        assert call("hs_g2_add", bn.g2_to_le(q1), bn.g2_to_le(q2), out=128) == bn.g2_to_le(bn.g2_add(q1, q2))
        assert call("hs_g2_mul", bn.g2_to_le(q1), le(k2), out=128) == bn.g2_to_le(bn.g2_mul(q1, k2))
    p1 = bn.g1_mul(bn.G1_GEN, 5)
    # special cases: P + P, P + (-P), P + 0, 0 + P, k = 0, k = r
    assert call("hs_g1_add", bn.g1_to_le(p1), bn.g1_to_le(p1), out=64) == bn.g1_to_le(bn.g1_mul(bn.G1_GEN, 10))
    assert call("hs_g1_add", bn.g1_to_le(p1), bn.g1_to_le(bn.g1_neg(p1)), out=64) == bytes(64)
    assert call("hs_g1_add", bn.g1_to_le(p1), bytes(64), out=64) == bn.g1_to_le(p1)
    assert call("hs_g1_add", bytes(64), bn.g1_to_le(p1), out=64) == bn.g1_to_le(p1)
    assert call("hs_g1_add_jac", bn.g1_to_le(p1), bn.g1_to_le(p1), le(7), out=64) == bn.g1_to_le(bn.g1_mul(bn.G1_GEN, 10))
    # the table kernels' expanded mixed addition: the same special cases
    assert call("hs_g1_madd_inl", bn.g1_to_le(p1), bn.g1_to_le(p1), le(7), out=64) == bn.g1_to_le(bn.g1_mul(bn.G1_GEN, 10))
    assert call("hs_g1_madd_inl", bn.g1_to_le(p1), bn.g1_to_le(bn.g1_neg(p1)), le(7), out=64) == bytes(64)
    assert call("hs_g1_madd_inl", bn.g1_to_le(p1), bytes(64), le(7), out=64) == bn.g1_to_le(p1)
    assert call("hs_g1_madd_inl", bytes(64), bn.g1_to_le(p1), le(7), out=64) == bn.g1_to_le(p1)
    assert call("hs_g1_mul", bn.g1_to_le(p1), le(0), out=64) == bytes(64)
    assert call("hs_g1_mul", bn.g1_to_le(p1), le(bn.R), out=64) == bytes(64)
    assert HS.hs_g1_on_curve(b2c(bn.g1_to_le(p1))) == 1
    assert HS.hs_g1_on_curve(b2c(le(1) + le(3))) == 0
    assert HS.hs_g2_on_curve(b2c(bn.g2_to_le(bn.G2_GEN))) == 1


def test_pairing_matches_oracle():
    k1, k2 = RND.randrange(1, bn.R), RND.randrange(1, bn.R)
    p = bn.g1_mul(bn.G1_GEN, k1)
    q = bn.g2_mul(bn.G2_GEN, k2)
    want = bn.pairing(p, q)
    got = call("hs_pairing", bn.g1_to_le(p), bn.g2_to_le(q), out=384)
    assert got == bn.gt_to_le(want)
    # Miller values differ by subfield factors but agree after the final exponentiation
    m = call("hs_miller", bn.g1_to_le(p), bn.g2_to_le(q), out=384)
    assert bn.final_exponentiation(bn.gt_from_le(m)) == want
    assert call("hs_final_exp", m, out=384) == bn.gt_to_le(want)
    # Jacobian P (no inversion) path
    assert call("hs_pairing_jac", bn.g1_to_le(p), le(rand_fp()), bn.g2_to_le(q), out=384) == bn.gt_to_le(want)
    # infinity inputs -> 1
    assert call("hs_pairing", bytes(64), bn.g2_to_le(q), out=384) == bn.gt_to_le(bn.GT_ONE)
    assert call("hs_pairing", bn.g1_to_le(p), bytes(128), out=384) == bn.gt_to_le(bn.GT_ONE)
    # Gt pow
    k = RND.randrange(bn.R)
    assert call("hs_gt_pow", bn.gt_to_le(want), le(k), out=384) == bn.gt_to_le(bn.gt_pow(want, k))


def test_three_lane_cooperative_pairing_equals_single_lane():
    """coop3.h: the Miller value and the pairing computed by a triple of lanes (three host threads, all-gather
    emulated) are bit-identical in all three lanes and equal to the one-lane code / the oracle."""
    k1, k2 = RND.randrange(1, bn.R), RND.randrange(1, bn.R)
    p = bn.g1_mul(bn.G1_GEN, k1)
    q = bn.g2_mul(bn.G2_GEN, k2)
    P, Q = b2c(bn.g1_to_le(p)), b2c(bn.g2_to_le(q))
    out = buf(384)
    assert HS.hs_c3_pairing(P, None, Q, 0, out) == 1
    assert bytes(out) == call("hs_miller", bn.g1_to_le(p), bn.g2_to_le(q), out=384)      # same Miller value, not just same pairing
    assert HS.hs_c3_pairing(P, None, Q, 1, out) == 1
    assert bytes(out) == bn.gt_to_le(bn.gt_pow(bn.pairing(bn.G1_GEN, bn.G2_GEN), k1 * k2 % bn.R))
    z = b2c(le(rand_fp()))
    assert HS.hs_c3_pairing(P, z, Q, 1, out) == 1                                         # Jacobian P path
    assert bytes(out) == bn.gt_to_le(bn.gt_pow(bn.pairing(bn.G1_GEN, bn.G2_GEN), k1 * k2 % bn.R))
    assert HS.hs_c3_pairing(b2c(bytes(64)), None, Q, 1, out) == 1
    assert bytes(out) == bn.gt_to_le(bn.GT_ONE)


def test_prepared_g2_pairing():
    k1, k2 = RND.randrange(1, bn.R), RND.randrange(1, bn.R)
    p, q = bn.g1_mul(bn.G1_GEN, k1), bn.g2_mul(bn.G2_GEN, k2)
    want = bn.gt_to_le(bn.gt_pow(bn.pairing(bn.G1_GEN, bn.G2_GEN), k1 * k2 % bn.R))
    assert call("hs_pairing_prepared", bn.g1_to_le(p), bn.g2_to_le(q), out=384) == want
    assert call("hs_pairing_prepared", bytes(64), bn.g2_to_le(q), out=384) == bn.gt_to_le(bn.GT_ONE)
    assert call("hs_pairing_prepared", bn.g1_to_le(p), bytes(128), out=384) == bn.gt_to_le(bn.GT_ONE)


@pytest.mark.parametrize("fn", ["hs_pairing_pair", "hs_pairing_pair_parked"])
def test_paired_miller_loop(fn):
    ks = [RND.randrange(1, bn.R) for _ in range(4)]
    pa, pb = bn.g1_mul(bn.G1_GEN, ks[0]), bn.g1_mul(bn.G1_GEN, ks[2])
    qa, qb = bn.g2_mul(bn.G2_GEN, ks[1]), bn.g2_mul(bn.G2_GEN, ks[3])
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    z64, z128 = bytes(64), bytes(128)
    enc = lambda p, q: (bn.g1_to_le(p), bn.g2_to_le(q))
    assert call(fn, *enc(pa, qa), *enc(pb, qb), out=384) == bn.gt_to_le(bn.gt_pow(e, (ks[0] * ks[1] + ks[2] * ks[3]) % bn.R))
    assert call(fn, z64, bn.g2_to_le(qa), *enc(pb, qb), out=384) == bn.gt_to_le(bn.gt_pow(e, ks[2] * ks[3] % bn.R))
    assert call(fn, bn.g1_to_le(pa), z128, *enc(pb, qb), out=384) == bn.gt_to_le(bn.gt_pow(e, ks[2] * ks[3] % bn.R))
    assert call(fn, *enc(pa, qa), z64, bn.g2_to_le(qb), out=384) == bn.gt_to_le(bn.gt_pow(e, ks[0] * ks[1] % bn.R))
    assert call(fn, *enc(pa, qa), bn.g1_to_le(pb), z128, out=384) == bn.gt_to_le(bn.gt_pow(e, ks[0] * ks[1] % bn.R))
    assert call(fn, z64, z128, z64, z128, out=384) == bn.gt_to_le(bn.GT_ONE)


def test_final_exponentiation_over_workspace_slots():
    k1, k2 = RND.randrange(1, bn.R), RND.randrange(1, bn.R)
    p, q = bn.g1_mul(bn.G1_GEN, k1), bn.g2_mul(bn.G2_GEN, k2)
    m = call("hs_miller", bn.g1_to_le(p), bn.g2_to_le(q), out=384)
    assert call("hs_final_exp_ws", m, out=384) == call("hs_final_exp", m, out=384) == bn.gt_to_le(bn.pairing(p, q))


def test_gt_pow_signed_window_equals_binary():
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    f = bn.gt_to_le(bn.gt_pow(e, RND.randrange(1, bn.R)))
    for k in [0, 1, 7, 8, 9, 15, 16, 0x88888888, bn.R - 1, bn.R, (1 << 256) - 1, RND.randrange(bn.R), RND.randrange(bn.R),
              int("8" * 64, 16), int("7" * 64, 16), int("f" * 63, 16)]:
        kb = (k % (1 << 256)).to_bytes(32, "little")
        assert call("hs_gt_pow_window", f, kb, out=384) == call("hs_gt_pow", f, kb, out=384)
    k = RND.randrange(bn.R)
    assert call("hs_gt_pow_window", f, le(k), out=384) == bn.gt_to_le(bn.gt_pow(bn.gt_from_le(f), k))


def test_naf_scalar_multiplication_equals_oracle():
    """jac_mul_naf (curve.h): the engine's variable-base multiplication over the NAF of k"""
    p = bn.g1_mul(bn.G1_GEN, RND.randrange(1, bn.R))
    q = bn.g2_mul(bn.G2_GEN, RND.randrange(1, bn.R))
    for k in [0, 1, 2, 3, 5, 7, 0xFFFFFFFF, 1 << 32, (1 << 64) - 1, bn.R - 1, bn.R - 2, (bn.R - 1) // 2, int("a" * 63, 16) % bn.R,
              int("5" * 63, 16) % bn.R] + [RND.randrange(bn.R) for _ in range(6)]:
        assert call("hs_g1_mul_naf", bn.g1_to_le(p), le(k), out=64) == bn.g1_to_le(bn.g1_mul(p, k)), hex(k)
    for k in [0, 1, 3, bn.R - 1, RND.randrange(bn.R), RND.randrange(bn.R)]:
        assert call("hs_g2_mul_naf", bn.g2_to_le(q), le(k), out=128) == bn.g2_to_le(bn.g2_mul(q, k)), hex(k)
    assert call("hs_g1_mul_naf", bytes(64), le(5), out=64) == bytes(64)


def test_glv_decomposition_and_multiplication():
    """curve.h: glv_decompose / jac_mul_glv_g1 -- k = k1 + k2 lambda (mod r) with short k1, k2, and the same point as the plain chain"""
    import re
    text = open("rabe_amd/csrc/bn254/constants.h").read()
    lam = sum(int(x.rstrip("u"), 16) << (32 * i) for i, x in enumerate(re.search(r"RB_GLV_LAMBDA \{([^}]*)\}", text).group(1).replace(" ", "").split(",")))
    beta_m = sum(int(x.rstrip("u"), 16) << (32 * i) for i, x in enumerate(re.search(r"RB_GLV_BETA \{([^}]*)\}", text).group(1).replace(" ", "").split(",")))
    beta = beta_m * pow(1 << 256, -1, bn.P) % bn.P
    assert pow(lam, 3, bn.R) == 1 and lam != 1 and pow(beta, 3, bn.P) == 1 and beta != 1
    p = bn.g1_mul(bn.G1_GEN, RND.randrange(1, bn.R))
    assert (beta * p[0] % bn.P, p[1]) == bn.g1_mul(p, lam)                    # phi = [lambda]
    ks = [0, 1, 2, 3, lam, lam - 1, lam + 1, bn.R - 1, bn.R - 2, bn.R, bn.R + 5, (1 << 256) - 1, 1 << 255, (1 << 254) - 1, (bn.R - 1) // 2,
          int("a" * 63, 16), int("5" * 64, 16), (1 << 128) - 1, 1 << 128, (1 << 127) + 12345] + [RND.randrange(bn.R) for _ in range(60)]
    for k in ks:
        o = buf(72)
        HS.hs_glv_decompose(b2c(le(k)), o)
        w = list(o)
        k1 = sum(x << (32 * i) for i, x in enumerate(w[:8])) * (-1 if w[16] else 1)
        k2 = sum(x << (32 * i) for i, x in enumerate(w[8:16])) * (-1 if w[17] else 1)
        assert (k1 + k2 * lam - k) % bn.R == 0, hex(k)
        assert abs(k1) < 1 << 130 and abs(k2) < 1 << 130, (hex(k), k1.bit_length(), k2.bit_length())
        if k < 1 << 60:
            assert (k1, k2) == (k, 0)
        want = bn.g1_to_le(bn.g1_mul(p, k % bn.R))
        assert call("hs_g1_mul_glv", bn.g1_to_le(p), le(k), out=64) == want, hex(k)
    for k in [x for x in ks[:20] if x < bn.R]:            # the plain chain takes canonical scalars only (3k must fit 256 bits)
        assert call("hs_g1_mul_naf_plain", bn.g1_to_le(p), le(k), out=64) == bn.g1_to_le(bn.g1_mul(p, k % bn.R)), hex(k)


def test_shared_doubling_msm_equals_sum_of_products():
    """jac_msm_naf (curve.h): Straus over the NAFs -- lsw's sum of c_y * D1_y, aw11's sum of c_x * C3_x"""
    n = 7
    ps = [bn.g1_mul(bn.G1_GEN, RND.randrange(1, bn.R)) for _ in range(n)]
    ks = [RND.randrange(bn.R) for _ in range(n - 3)] + [0, 1, bn.R - 1]
    ps[2] = ps[1]                                   # equal bases: the doubling case inside the mixed addition
    ks[2] = ks[1]
    want = None
    for p, k in zip(ps, ks):
        want = bn.g1_add(want, bn.g1_mul(p, k)) if want is not None else bn.g1_mul(p, k)
    o = buf(64)
    HS.hs_g1_msm(n, b2c(b"".join(bn.g1_to_le(p) for p in ps)), b2c(b"".join(le(k) for k in ks)), o)
    assert bytes(o) == bn.g1_to_le(want)
    # cancelling terms: k P + (r - k) P = infinity
    HS.hs_g1_msm(2, b2c(bn.g1_to_le(ps[0]) * 2), b2c(le(ks[0]) + le(bn.R - ks[0])), o)
    assert bytes(o) == bytes(64)
    qs = [bn.g2_mul(bn.G2_GEN, RND.randrange(1, bn.R)) for _ in range(3)]
    k2 = [RND.randrange(bn.R) for _ in range(3)]
    want = None
    for q, k in zip(qs, k2):
        want = bn.g2_add(want, bn.g2_mul(q, k)) if want is not None else bn.g2_mul(q, k)
    o2 = buf(128)
    HS.hs_g2_msm(3, b2c(b"".join(bn.g2_to_le(q) for q in qs)), b2c(b"".join(le(k) for k in k2)), o2)
    assert bytes(o2) == bn.g2_to_le(want)


@pytest.mark.parametrize("n,kinds", [(1, [0]), (1, [1]), (2, [1, 0]), (3, [0, 0, 1]), (5, [1, 0, 1, 0, 1]), (4, [0, 2, 1, 0])])
def test_multi_pairing_on_one_accumulator(n, kinds):
    """miller_loop_multi (pairing.h): n pairings on one accumulator, walking / prepared / skipped pairs mixed"""
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    ks = [(RND.randrange(1, bn.R), RND.randrange(1, bn.R)) for _ in range(n)]
    p = b"".join(bn.g1_to_le(bn.g1_mul(bn.G1_GEN, a)) for a, _ in ks)
    q = b"".join(bn.g2_to_le(bn.g2_mul(bn.G2_GEN, b)) for _, b in ks)
    exp = sum(a * b for (a, b), kd in zip(ks, kinds) if kd != 2) % bn.R
    o = buf(384)
    HS.hs_pairing_multi(n, (ctypes.c_int * n)(*kinds), b2c(p), b2c(q), o)
    assert bytes(o) == bn.gt_to_le(bn.gt_pow(e, exp))
    if n >= 2:      # an argument at infinity contributes 1
        p2 = bytes(64) + p[64:]
        HS.hs_pairing_multi(n, (ctypes.c_int * n)(*kinds), b2c(p2), b2c(q), o)
        exp2 = sum(a * b for i, ((a, b), kd) in enumerate(zip(ks, kinds)) if kd != 2 and i != 0) % bn.R
        assert bytes(o) == bn.gt_to_le(bn.gt_pow(e, exp2))
