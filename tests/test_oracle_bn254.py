"""Pins oracle/bn254.py against the public alt_bn128 facts available without the reference:
EIP-196 doubling vector, group orders, bilinearity,
non-degeneracy, Frobenius == x^p, canonical encoding round trips."""
import random

from oracle import bn254 as bn


def test_curve_constants():
    assert bn.P.bit_length() == 254 and bn.R.bit_length() == 254
    assert bn.ATE_LOOP.bit_length() == 65
    assert bn.ec_is_on_curve(bn.FP, bn.G1_GEN)
    assert bn.ec_is_on_curve(bn.FP2, bn.G2_GEN)


def test_eip196_double():
    d = bn.g1_mul(bn.G1_GEN, 2)
    assert d == (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
                 0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)
    assert bn.g1_add(bn.G1_GEN, bn.G1_GEN) == d


def test_group_orders():
    assert bn.ec_mul(bn.FP, bn.G1_GEN, bn.R - 1) == bn.g1_neg(bn.G1_GEN)
    assert bn.g2_add(bn.ec_mul(bn.FP2, bn.G2_GEN, bn.R - 1), bn.G2_GEN) is None


def test_field_tower():
    rnd = random.Random(5)
    x = bn.fp12_from_coeffs([rnd.randrange(bn.P) for _ in range(12)])
    y = bn.fp12_from_coeffs([rnd.randrange(bn.P) for _ in range(12)])
    assert bn.fp12_mul(x, bn.fp12_inv(x)) == bn.FP12_ONE
    assert bn.fp12_mul(x, y) == bn.fp12_mul(y, x)
    assert bn.fp12_frobenius(x) == bn.fp12_pow(x, bn.P)
    assert bn.gt_from_le(bn.gt_to_le(x)) == x
    # w^2 = v, v^3 = xi
    w = (bn.FP6_ZERO, bn.FP6_ONE)
    v = ((bn.FP2_ZERO, bn.FP2_ONE, bn.FP2_ZERO), bn.FP6_ZERO)
    assert bn.fp12_mul(w, w) == v
    assert bn.fp12_mul(v, bn.fp12_mul(v, v)) == ((bn.XI, bn.FP2_ZERO, bn.FP2_ZERO), bn.FP6_ZERO)


def test_pairing_bilinear_nondegenerate():
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    assert e != bn.FP12_ONE
    assert bn.fp12_pow(e, bn.R) == bn.FP12_ONE
    a, b = 0x1234567890abcdef1234567890abcdef, 0xfedcba9876543210fedcba987654321
    assert bn.pairing(bn.g1_mul(bn.G1_GEN, a), bn.g2_mul(bn.G2_GEN, b)) == bn.fp12_pow(e, a * b % bn.R)
    assert bn.pairing(None, bn.G2_GEN) == bn.FP12_ONE
    assert bn.pairing(bn.g1_neg(bn.G1_GEN), bn.G2_GEN) == bn.fp12_inv(e)


def test_final_exponent_lineage():
    # pairing (libff/zcash-bn hard-part chain) is the exact optimal-ate pairing raised to 2z(6z^2+3z+1)
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    ex = bn.pairing_exact(bn.G1_GEN, bn.G2_GEN)
    assert e == bn.fp12_pow(ex, bn.FE_MULTIPLE % bn.R)
    assert e != ex and e != bn.FP12_ONE
