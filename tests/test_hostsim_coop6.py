"""Six-lane cooperative Fq12 arithmetic (rabe_amd/csrc/bn254/coop6.h -- the functions the k_*_c6 kernels call), run on the CPU by six
host threads per group (tests/hostsim) and compared bit for bit with the one-lane functions of the same headers and with the Python
big-int oracle.  The same comparisons run on the device in tests/test_gpu_coop6.py."""
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


def buf(n):
    return (ctypes.c_uint32 * (n // 4))()


def b2c(b):
    return (ctypes.c_uint32 * (len(b) // 4)).from_buffer_copy(b)


def c6(op, a, b=None):
    o = buf(384)
    HS.hs_c6_op(op, b2c(bn.gt_to_le(a)), b2c(bn.gt_to_le(b)) if b is not None else None, o)
    return bytes(o)


def rand_fp():
    return RND.randrange(bn.P)


def rand_fp12():
    return bn.fp12_from_coeffs([rand_fp() for _ in range(12)])


def edge_fp12(v):
    return bn.fp12_from_coeffs([v] * 12)


def test_c6_mul_sqr_against_oracle():
    vals = [rand_fp12() for _ in range(6)] + [edge_fp12(bn.P - 1), edge_fp12(0), edge_fp12(1), bn.FP12_ONE]
    for a in vals:
        for b in (vals[0], vals[6], vals[9]):
            assert c6(0, a, b) == bn.gt_to_le(bn.fp12_mul(a, b))
        assert c6(1, a) == bn.gt_to_le(bn.fp12_sqr(a))


def test_c6_extreme_limbs():
    """operands whose MONTGOMERY limbs are extreme (0xFFFFFFFF runs, p - 1): the unreduced six-product sums are at their bounds"""
    rinv = pow(1 << 256, -1, bn.P)
    res = [bn.P - 1, (1 << 253) - 1, 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16ffffffff % bn.P, (1 << 253) + (1 << 252) - 1]
    for x in res:
        a = edge_fp12(x * rinv % bn.P)
        assert c6(0, a, a) == bn.gt_to_le(bn.fp12_mul(a, a))
        assert c6(1, a) == bn.gt_to_le(bn.fp12_sqr(a))


def test_c6_line_and_frobenius():
    for _ in range(4):
        a = rand_fp12()
        l0, l1, l3 = [(rand_fp(), rand_fp()) for _ in range(3)]
        line = ((l0, bn.FP2_ZERO, bn.FP2_ZERO), (l1, l3, bn.FP2_ZERO))
        carrier = ((l0, l1, l3), (bn.FP2_ZERO, bn.FP2_ZERO, bn.FP2_ZERO))
        assert c6(3, a, carrier) == bn.gt_to_le(bn.fp12_mul(a, line))
        for k in (1, 2, 3):
            assert c6(5 + k, a) == bn.gt_to_le(bn.fp12_pow(a, bn.P ** k))


def test_c6_cyclotomic_and_final_exponentiation():
    x = bn.fp12_pow(rand_fp12(), bn.FE_EASY)
    assert c6(2, x) == bn.gt_to_le(bn.fp12_sqr(x))
    assert c6(5, x) == bn.gt_to_le(bn.fp12_pow(x, bn.U))
    f = rand_fp12()
    assert c6(4, f) == bn.gt_to_le(bn.final_exponentiation(f))


@pytest.mark.parametrize("n,kinds", [(1, [0]), (1, [1]), (2, [1, 0]), (6, [0, 0, 0, 1, 1, 1]), (7, [1, 0, 1, 0, 1, 0, 0]), (4, [0, 2, 1, 0]),
                                     (13, [0, 1] * 6 + [0])])
def test_c6_multi_pairing(n, kinds):
    """c6_miller_loop_multi + c6_final_exponentiation == miller_loop_multi + final_exponentiation (one lane) == the oracle"""
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    ks = [(RND.randrange(1, bn.R), RND.randrange(1, bn.R)) for _ in range(n)]
    p = b"".join(bn.g1_to_le(bn.g1_mul(bn.G1_GEN, a)) for a, _ in ks)
    q = b"".join(bn.g2_to_le(bn.g2_mul(bn.G2_GEN, b)) for _, b in ks)
    exp = sum(a * b for (a, b), kd in zip(ks, kinds) if kd != 2) % bn.R
    o, o1 = buf(384), buf(384)
    HS.hs_c6_pairing_multi(n, (ctypes.c_int * n)(*kinds), b2c(p), b2c(q), o)
    HS.hs_pairing_multi(n, (ctypes.c_int * n)(*kinds), b2c(p), b2c(q), o1)
    assert bytes(o) == bytes(o1)
    assert bytes(o) == bn.gt_to_le(bn.gt_pow(e, exp))
    if n >= 2:      # an argument at infinity contributes 1
        p2 = bytes(64) + p[64:]
        HS.hs_c6_pairing_multi(n, (ctypes.c_int * n)(*kinds), b2c(p2), b2c(q), o)
        exp2 = sum(a * b for i, ((a, b), kd) in enumerate(zip(ks, kinds)) if kd != 2 and i != 0) % bn.R
        assert bytes(o) == bn.gt_to_le(bn.gt_pow(e, exp2))


def test_selftest_digests_host_build_equals_exact_integer_generator():
    """bn254/selftest.h (what rhip_ctx_create runs on every SIMD) compiled for the host gives the digests tools/gen_selftest.py
    computed with exact integers -- two independent implementations of the 64 lanes' known answers"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("gen_selftest", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_selftest.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    HS.hs_selftest_digest.restype = ctypes.c_uint
    HS.hs_selftest_expected.restype = ctypes.c_uint
    for lane in range(64):
        want = gen.digest(lane)
        assert HS.hs_selftest_expected(lane) == want, "regenerate selftest_gen.h"
        assert HS.hs_selftest_digest(lane) == want, lane


def test_c6_gt_membership():
    """c6_gt_is_member (k_gt_is_member_c6): members of Gt pass; a random Fq12, a cyclotomic element of the wrong order and zero do not"""
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    for g in (e, bn.gt_pow(e, 12345678901234567890), bn.FP12_ONE):
        assert HS.hs_c6_gt_is_member(b2c(bn.gt_to_le(g))) == 1
    assert HS.hs_c6_gt_is_member(b2c(bn.gt_to_le(rand_fp12()))) == 0
    cyc = bn.fp12_pow(rand_fp12(), bn.FE_EASY)              # in the cyclotomic subgroup, order divides (p^4 - p^2 + 1): almost never r
    assert HS.hs_c6_gt_is_member(b2c(bn.gt_to_le(cyc))) == 0
    assert HS.hs_c6_gt_is_member(b2c(bytes(384))) == 0
