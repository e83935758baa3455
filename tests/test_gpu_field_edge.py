"""GPU: the field multiplication on adversarial limb patterns, bit-exact against the Python oracle.

The device multiplication (rabe_amd/csrc/bn254/fp.h) skips the carry capture for products that provably cannot overflow
their 64-bit column (tests/test_mac_plan.py proves the bounds with exact integers).  Random field elements almost never
come near those bounds, so this test drives the kernels with values whose *Montgomery representation* -- what the limbs in
the registers actually are -- is extreme: limbs of 0xFFFFFFFF, the modulus' own limbs +-1, p - 1, single saturated limbs.
Covered: the single-chain product (`rhip_fr_op` over Fr; Fp through the inversion inside `rhip_gt_inv`), the two-chain
product and the lazy Fq2 product with its two-chain reduction (`rhip_gt_mul`, `rhip_gt_inv`)."""
import random

import pytest

from oracle import bn254 as bn

pytestmark = pytest.mark.gpu

RND = random.Random(0xED6E)
W = 0xFFFFFFFF
RINV_P = pow(1 << 256, -1, bn.P)
RINV_R = pow(1 << 256, -1, bn.R)


@pytest.fixture(scope="module")
def eng():
    from rabe_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def le(x):
    return int(x).to_bytes(32, "little")


def mont_patterns(mod, n_random):
    """integers < mod to be used as MONTGOMERY representations"""
    pl = [(mod >> (32 * i)) & W for i in range(8)]
    top = pl[7]
    out = [mod - 1, mod - 2, 1, 0, (1 << 224) - 1, ((top - 1) << 224) | ((1 << 224) - 1), (top << 224) | ((1 << 192) - 1)]
    for i in range(8):
        out.append(W << (32 * i) if i < 7 else (top - 1) << 224)          # one saturated limb
        out.append(((1 << 256) - 1) ^ (W << (32 * i)))                    # all but one
    choices = lambda i: [0, 1, W, W - 1, 0x80000000, pl[i], (pl[i] - 1) & W, (pl[i] + 1) & W, RND.getrandbits(32)]
    for _ in range(n_random):
        out.append(sum(RND.choice(choices(i)) << (32 * i) for i in range(8)))
    return [x % mod for x in out]


def canon(m, rinv, mod):
    """the canonical value whose Montgomery form m*... is exactly `m`"""
    return (m * rinv) % mod


def test_fr_mul_on_extreme_limbs(eng):
    ms = mont_patterns(bn.R, 120)
    xs = [canon(m, RINV_R, bn.R) for m in ms]
    a = [x for x in xs for _ in range(4)]
    b = [RND.choice(xs) for _ in a]
    for i in range(len(xs)):
        b[4 * i] = xs[i]                                                   # squares
    got = eng.fr_op(2, [le(x) for x in a], [le(x) for x in b])
    assert got == [le(x * y % bn.R) for x, y in zip(a, b)]
    # the reduction of arbitrary 256-bit strings (second operand unbounded)
    digs = [bytes([0xFF] * 32), bytes([0xFF] * 4 + [0] * 28), bytes([0] * 28 + [0xFF] * 4)] + [RND.randbytes(32) for _ in range(30)]
    assert eng.fr_from_be32_reduce(digs) == [le(bn.fr_from_be32_reduce(d)) for d in digs]


def fp12_of(ms):
    return bn.fp12_from_coeffs([canon(m, RINV_P, bn.P) for m in ms])


def test_gt_mul_inv_on_extreme_limbs(eng):
    ms = mont_patterns(bn.P, 200)
    elems = []
    for i in range(0, len(ms) - 12, 3):                                    # sliding windows: every pattern meets every slot
        elems.append(fp12_of(ms[i:i + 12]))
    for m in ms[:24]:                                                      # the same pattern in all twelve coefficients
        elems.append(fp12_of([m] * 12))
    elems = [e for e in elems if e != bn.FP12_ZERO]
    A = [bn.gt_to_le(e) for e in elems]
    B = A[1:] + A[:1]
    got = eng.gt_mul(A, B)
    want = [bn.gt_to_le(bn.fp12_mul(x, y)) for x, y in zip(elems, elems[1:] + elems[:1])]
    assert got == want
    sq = eng.gt_mul(A, A)
    assert sq == [bn.gt_to_le(bn.fp12_mul(x, x)) for x in elems]
    # inversion: the Fq single-chain product and squaring (Fermat chain), the two-chain product, Fq2 / Fq6 inverses
    inv_in = [e for e in elems[:40] if bn.fp12_mul(e, bn.fp12_conj(e)) != bn.FP12_ZERO]
    got_inv = eng.gt_inv([bn.gt_to_le(e) for e in inv_in])
    for e, g in zip(inv_in, got_inv):
        try:
            want_inv = bn.fp12_inv(e)
        except Exception:
            continue                                                       # not invertible
        assert g == bn.gt_to_le(want_inv)


def test_g1_mul_glv_on_extreme_scalars(eng):
    """`rhip_g1_mul` runs the GLV chain (curve.h: jac_mul_glv_g1): any 256-bit scalar, also >= r, gives k mod r times the point"""
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    p = bn.g1_mul(bn.G1_GEN, RND.randrange(1, bn.R))
    ks = [0, 1, 2, lam, lam + 1, lam - 1, bn.R - 1, bn.R, bn.R + 5, (1 << 256) - 1, 1 << 255, (1 << 254) - 1, (1 << 128) - 1, 1 << 128, 1 << 127,
          int("a" * 64, 16), int("5" * 64, 16)] + [RND.randrange(1 << 256) for _ in range(40)]
    got = eng.g1_mul([bn.g1_to_le(p)] * len(ks), [le(k) for k in ks])
    assert got == [bn.g1_to_le(bn.g1_mul(p, k % bn.R)) for k in ks]
    assert eng.g1_mul([bytes(64)], [le(7)]) == [bytes(64)]


def test_fr_add_sub_neg_on_extreme_limbs(eng):
    """the carry chains of add / sub / neg (fp.h: asm statements without the compiler's wait states) on saturated limb patterns"""
    ms = mont_patterns(bn.R, 150)
    xs = [canon(m, RINV_R, bn.R) for m in ms] + [0, 1, bn.R - 1, bn.R - 2, (bn.R - 1) // 2, (bn.R + 1) // 2]
    a = [x for x in xs for _ in range(3)]
    b = [RND.choice(xs) for _ in a]
    for i in range(len(xs)):
        b[3 * i] = xs[i]                                                   # a + a, a - a
        b[3 * i + 1] = (bn.R - xs[i]) % bn.R                               # a + (-a) = 0
    A, B = [le(x) for x in a], [le(x) for x in b]
    assert eng.fr_op(0, A, B) == [le((x + y) % bn.R) for x, y in zip(a, b)]
    assert eng.fr_op(1, A, B) == [le((x - y) % bn.R) for x, y in zip(a, b)]
    assert eng.fr_op(3, A) == [le((-x) % bn.R) for x in a]


def test_fr_inv_on_extreme_residues(eng):
    """the field inversion (fp.h: inv -- Kaliski's almost-inverse, a binary extended Euclid on the Montgomery residue, then the
    power-of-two correction) on residues that drive its shortest / longest step sequences: tiny values, powers of two, r - small,
    saturated limbs.  Every lane has its own step sequence here (divergent); the block inversion of the kernels runs it uniformly."""
    ms = [m for m in mont_patterns(bn.R, 200) if m] + [2, 3, 4, 1 << 31, 1 << 32, 1 << 128, 1 << 253, (1 << 253) - 1, (1 << 253) + 1,
                                                        (bn.R - 1) // 2, (bn.R + 1) // 2, bn.R - 3]
    xs = [canon(m, RINV_R, bn.R) for m in ms]
    assert eng.fr_op(4, [le(x) for x in xs]) == [le(pow(x, bn.R - 2, bn.R)) for x in xs]
    assert eng.fr_op(4, [le(0)]) == [le(0)]


def test_mul_by_xi_on_extreme_values(eng):
    """fp2_mul_xi / fp2_add_mul_xi (tower.h: x + 9 y -+ z reduced through a quotient estimate): an Fq12 product whose only non-zero
    coefficients are a.c0.a2 = Y and b.c0.a1 = 1 is xi * Y in c0.a0; with b.c0.a1 = 1 and a.c0.a0 = X, b.c0.a0 = 1 the same slot
    carries X + ... paths of the Karatsuba form -- all compared with the oracle's Fq12 product, on values at the edges of [0, p)."""
    edge = [0, 1, 2, bn.P - 1, bn.P - 2, (bn.P - 1) // 9, (bn.P - 1) // 9 + 1, (2 * bn.P) // 9, (bn.P + 8) // 9, bn.P // 2, bn.P // 2 + 1,
            (1 << 253), (1 << 224) - 1] + [canon(m, RINV_P, bn.P) for m in mont_patterns(bn.P, 30)[:40]]
    A, B, want = [], [], []
    for i in range(300):
        y = (RND.choice(edge), RND.choice(edge))
        x = (RND.choice(edge), RND.choice(edge))
        z = (RND.choice(edge), RND.choice(edge))
        # a = x + y v^2 + z v (c0 only), b = 1 + v: the product's c0.a0 = x + xi y, c0.a1 = x + z, c0.a2 = z + y -- and squared terms below
        a = ((x, z, y), bn.FP6_ZERO)
        b = ((bn.FP2_ONE, bn.FP2_ONE, bn.FP2_ZERO), bn.FP6_ZERO)
        A.append(bn.gt_to_le(a)); B.append(bn.gt_to_le(b)); want.append(bn.gt_to_le(bn.fp12_mul(a, b)))
        # a full element times itself: every fused slot of fp6_mul / fp12_mul sees extreme operands
        f = bn.fp12_from_coeffs([RND.choice(edge) for _ in range(12)])
        A.append(bn.gt_to_le(f)); B.append(bn.gt_to_le(f)); want.append(bn.gt_to_le(bn.fp12_mul(f, f)))
    assert eng.gt_mul(A, B) == want
