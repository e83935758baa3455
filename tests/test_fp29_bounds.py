"""Exact-integer checks of the bounds the reduced-radix field core (rabe_amd/csrc/bn254/fp29.h) states in its types and comments: what a
64-bit column can take, that the results of multiplications stay below 1.5 p, that the quotient estimate of norm() leaves |x| <= 0.51 p,
that the conversions stay inside 256 bits.  Pure Python integers / fractions -- the constants are parsed from the headers themselves."""
import os
import re
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
U = (1 << 28) + 4096                     # RB29_U: the unit of the limb bound
HALF = 1 << 28


def _consts():
    text = open(os.path.join(ROOT, "rabe_amd", "csrc", "bn254", "constants29.h")).read()
    out = {}
    for m in re.finditer(r"#define (RB29_\w+) \{ ([^}]*) \}", text):
        out[m.group(1)] = [int(x, 0) for x in m.group(2).split(",")]
    out["RB29_QK"] = int(re.search(r"#define RB29_QK (\d+)", text).group(1))
    out["RB29_PINV"] = int(re.search(r"#define RB29_PINV (0x[0-9a-f]+)u", text).group(1), 16)
    return out


C = _consts()


def val(l):
    return sum(int(x) << (29 * k) for k, x in enumerate(l))


def test_constants_are_what_the_header_says():
    assert val(C["RB29_P"]) == P and all(0 <= x < 1 << 29 for x in C["RB29_P"])
    assert val(C["RB29_PBAL"]) == P and all(abs(x) <= HALF for x in C["RB29_PBAL"][:8]) and C["RB29_PBAL"][0] & 1
    assert (P * C["RB29_PINV"] + 1) % (1 << 29) == 0
    r = 1 << 261
    for name, want in (("RB29_ONE", r % P), ("RB29_C266", (1 << 266) % P), ("RB29_C256", (1 << 256) % P)):
        assert val(C[name]) == want and all(abs(x) <= HALF for x in C[name][:8]), name
    assert C["RB29_QK"] == round(Fraction((1 << 44) * (1 << 232), P))


def test_a_column_holds_what_the_static_asserts_allow():
    """fp29.h: mul / mac2 / dot3 accept limb bounds with sum(La Lb) <= 10 per column set (an Fq2 dot product: 2 x the Fq2-level sum <= 10):
    nine products of limbs <= La U, Lb U per schoolbook product, plus the reduction's nine m p_j (m < 2^29), the rounding offset and the
    carry that comes in from the column below"""
    worst_products = 9 * 10 * U * U
    reduction = sum(((1 << 29) - 1) * pj for pj in C["RB29_P"])          # at most all nine p limbs meet one column; exact: sum over j
    carry_in = (1 << 63) >> 29                                             # whatever the column below held, shifted
    assert worst_products + reduction + HALF + carry_in < 1 << 63
    # (p's limbs are small on average, so 12 would still fit and 13 would not: the asserted 10 leaves two units of slack)
    assert 9 * 12 * U * U + reduction + HALF + carry_in < 1 << 63 <= 9 * 13 * U * U + reduction + HALF + carry_in


def test_results_of_multiplications_stay_below_one_and_a_half_p():
    """value bounds: |x| <= V 1.5 p; a reduction returns |T| / 2^261 + p at most; the static asserts allow sum(Va Vb) <= 36"""
    r = 1 << 261
    t_max = 36 * Fraction(3 * P, 2) ** 2
    assert t_max / r + P <= Fraction(3 * P, 2)
    assert 38 * Fraction(3 * P, 2) ** 2 / r + P > Fraction(3 * P, 2)          # (the margin is small: 37 would still hold, 38 would not)
    # the top limb of a stored value is far below the limb bound: |x| <= 1.5 p  =>  |l[8]| <= 1.5 p / 2^232 + 1
    assert Fraction(3 * P, 2) / (1 << 232) + 1 < U


def test_quotient_estimate_of_norm():
    """norm() / norm_lin9(): q = (l8 * QK + 2^43) >> 44 from the top limb alone.  x = l8 2^232 + low with |low| <= L U (2^232 - 1) / (2^29 - 1)
    (limbs 0..7 within L U, L <= 7).  Claim: |x - q p| <= 0.51 p for every top limb a value of up to 70 p can have."""
    qk = C["RB29_QK"]
    low_max = 7 * U * ((1 << 232) - 1) // ((1 << 29) - 1)
    worst = Fraction(0)
    step = 1 << 232
    tops = set()
    for k in range(-70, 71):                      # around every multiple and half-multiple of p, where the rounding flips
        for half in (0, 1):
            centre = (2 * k + half) * P // (2 * step)
            tops.update(range(centre - 2, centre + 3))
    tops.update((0, 1, -1, 70 * P // step, -(70 * P // step)))
    for l8 in tops:
        q = (l8 * qk + (1 << 43)) >> 44
        for low in (-low_max, 0, low_max):
            x = l8 * step + low
            worst = max(worst, Fraction(abs(x - q * P), P))
    assert worst <= Fraction(51, 100), float(worst)


def test_conversions_stay_inside_256_bits():
    """to_fp: v = a C256 / 2^261 with |a| <= 1.5 p, C256 < p  =>  |v| <= p + 1.5 p p / 2^261 < 1.01 p; v + 2 p lies in (0, 2^256) and needs at
    most three subtractions of p.  from_fp: the unpacked limbs are < 2^29 <= 2 U."""
    r = 1 << 261
    v_max = Fraction(3 * P, 2) * P / r + P
    assert v_max < Fraction(101, 100) * P
    assert 2 * P - v_max > 0 and 2 * P + v_max < 1 << 256 and 2 * P + v_max < 4 * P
    assert (1 << 29) <= 2 * U


def test_limb_sums_fit_int32():
    """add / sub allow L <= 7, norm() L <= 6 (it adds the rounding offset 2^28 first)"""
    assert 7 * U < 1 << 31 and 6 * U + HALF < 1 << 31 and 8 * U >= 1 << 31
