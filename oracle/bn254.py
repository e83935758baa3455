"""BN254 (alt_bn128) big-integer arithmetic -- TEST ORACLE, not product code.

This module is test infrastructure.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product path (rabe_amd/) never does.

PARITY UNPINNED: the arithmetic the reference uses lives in the external crate
`rabe-bn = "0.4.23"` (/root/reference/Cargo.toml:33), which is absent from
/root/reference, and the reference holds no known-answer vectors for group or pairing
values.  This file restates the *published* algorithms of that
crate's lineage (zcash `bn`: BN254 with the tower Fq2=Fq[u]/(u^2+1),
Fq6=Fq2[v]/(v^3-(9+u)), Fq12=Fq6[w]/(w^2-v); G1: y^2=x^3+3, generator (1,2);
G2 on the D-type sextic twist y^2=x^3+3/(9+u); optimal-ate pairing whose final exponent is
the libff/zcash-bn one -- see FINAL_EXP below) in the most literal way available: affine
chord-and-tangent formulas, a Miller loop written over full Fq12 elements through the
untwist map, and a final exponentiation that is one generic square-and-multiply by the
integer exponent.  It is
pinned against the public alt_bn128 vectors (EIP-196 point doubling, generator orders)
and against bilinearity / non-degeneracy, see tests/test_oracle_bn254.py.

Everything here is canonical-integer arithmetic (no Montgomery form): two correct
implementations of the same maps necessarily agree bit for bit on canonical outputs.

Reference call sites this stands in for: `use rabe_bn::{Group, Gt, G1, G2, Fr, pairing}`
at src/schemes/ac17/mod.rs:42, bsw/mod.rs:23, lsw/mod.rs:23, aw11/mod.rs:27,
utils/hash/mod.rs:1, utils/secretsharing/mod.rs:1, utils/tools/mod.rs:1.
"""

# ----------------------------------------------------------------------------- parameters
U = 4965661367192848881
P = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
R = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
assert P == 21888242871839275222246405745257275088696311157297823662689037894645226208583
assert R == 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE_LOOP = 6 * U + 2          # 65 bits
B1 = 3                        # G1: y^2 = x^3 + 3

# ----------------------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
# elements are tuples (c0, c1) = c0 + c1*u

FP2_ZERO = (0, 0)
FP2_ONE = (1, 0)
XI = (9, 1)                   # non-residue 9+u


def fp2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def fp2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def fp2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def fp2_sqr(a):
    return fp2_mul(a, a)


def fp2_scalar(a, k):
    return ((a[0] * k) % P, (a[1] * k) % P)


def fp2_conj(a):
    return (a[0], (-a[1]) % P)


def fp2_inv(a):
    n = (a[0] * a[0] + a[1] * a[1]) % P
    ni = pow(n, P - 2, P)
    return ((a[0] * ni) % P, ((-a[1]) * ni) % P)


def fp2_mul_xi(a):
    # (c0 + c1 u)(9 + u) = 9c0 - c1 + (c0 + 9c1) u
    return ((9 * a[0] - a[1]) % P, (a[0] + 9 * a[1]) % P)


def fp2_pow(a, e):
    r = FP2_ONE
    b = a
    while e:
        if e & 1:
            r = fp2_mul(r, b)
        b = fp2_sqr(b)
        e >>= 1
    return r


B2 = fp2_mul((3, 0), fp2_inv(XI))      # G2 (twist): y^2 = x^3 + 3/(9+u)

# ----------------------------------------------------------------------------- Fp6 = Fp2[v]/(v^3 - xi)
# elements are tuples (a0, a1, a2) of Fp2

FP6_ZERO = (FP2_ZERO, FP2_ZERO, FP2_ZERO)
FP6_ONE = (FP2_ONE, FP2_ZERO, FP2_ZERO)


def fp6_add(a, b):
    return (fp2_add(a[0], b[0]), fp2_add(a[1], b[1]), fp2_add(a[2], b[2]))


def fp6_sub(a, b):
    return (fp2_sub(a[0], b[0]), fp2_sub(a[1], b[1]), fp2_sub(a[2], b[2]))


def fp6_neg(a):
    return (fp2_neg(a[0]), fp2_neg(a[1]), fp2_neg(a[2]))


def fp6_mul(a, b):
    # schoolbook, reduce with v^3 = xi
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0 = fp2_mul(a0, b0)
    t1 = fp2_add(fp2_mul(a0, b1), fp2_mul(a1, b0))
    t2 = fp2_add(fp2_add(fp2_mul(a0, b2), fp2_mul(a1, b1)), fp2_mul(a2, b0))
    t3 = fp2_add(fp2_mul(a1, b2), fp2_mul(a2, b1))
    t4 = fp2_mul(a2, b2)
    return (fp2_add(t0, fp2_mul_xi(t3)), fp2_add(t1, fp2_mul_xi(t4)), t2)


def fp6_mul_v(a):
    # (a0 + a1 v + a2 v^2) * v = xi*a2 + a0 v + a1 v^2
    return (fp2_mul_xi(a[2]), a[0], a[1])


def fp6_inv(a):
    a0, a1, a2 = a
    c0 = fp2_sub(fp2_sqr(a0), fp2_mul_xi(fp2_mul(a1, a2)))
    c1 = fp2_sub(fp2_mul_xi(fp2_sqr(a2)), fp2_mul(a0, a1))
    c2 = fp2_sub(fp2_sqr(a1), fp2_mul(a0, a2))
    t = fp2_add(fp2_mul(a0, c0), fp2_mul_xi(fp2_add(fp2_mul(a2, c1), fp2_mul(a1, c2))))
    ti = fp2_inv(t)
    return (fp2_mul(c0, ti), fp2_mul(c1, ti), fp2_mul(c2, ti))


# ----------------------------------------------------------------------------- Fp12 = Fp6[w]/(w^2 - v)
# elements are tuples (c0, c1) of Fp6.  As a vector over Fp2 with basis
# 1, v, v^2, w, vw, v^2 w  (w^2 = v, w^6 = xi).

FP12_ZERO = (FP6_ZERO, FP6_ZERO)
FP12_ONE = (FP6_ONE, FP6_ZERO)


def fp12_add(a, b):
    return (fp6_add(a[0], b[0]), fp6_add(a[1], b[1]))


def fp12_sub(a, b):
    return (fp6_sub(a[0], b[0]), fp6_sub(a[1], b[1]))


def fp12_neg(a):
    return (fp6_neg(a[0]), fp6_neg(a[1]))


def fp12_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    t0 = fp6_mul(a0, b0)
    t1 = fp6_mul(a1, b1)
    c0 = fp6_add(t0, fp6_mul_v(t1))
    c1 = fp6_add(fp6_mul(a0, b1), fp6_mul(a1, b0))
    return (c0, c1)


def fp12_sqr(a):
    return fp12_mul(a, a)


def fp12_conj(a):
    return (a[0], fp6_neg(a[1]))


def fp12_inv(a):
    a0, a1 = a
    t = fp6_sub(fp6_mul(a0, a0), fp6_mul_v(fp6_mul(a1, a1)))
    ti = fp6_inv(t)
    return (fp6_mul(a0, ti), fp6_neg(fp6_mul(a1, ti)))


def fp12_pow(a, e):
    if e < 0:
        return fp12_pow(fp12_inv(a), -e)
    r = FP12_ONE
    b = a
    while e:
        if e & 1:
            r = fp12_mul(r, b)
        b = fp12_sqr(b)
        e >>= 1
    return r


def fp12_from_fp(x):
    return (((x % P, 0), FP2_ZERO, FP2_ZERO), FP6_ZERO)


def fp12_coeffs(a):
    """The 12 Fp coefficients in tower order c0.a0.c0, c0.a0.c1, c0.a1.c0, ... c1.a2.c1."""
    out = []
    for c in a:
        for f2 in c:
            out.extend(f2)
    return out


def fp12_from_coeffs(cs):
    cs = [c % P for c in cs]
    return (((cs[0], cs[1]), (cs[2], cs[3]), (cs[4], cs[5])),
            ((cs[6], cs[7]), (cs[8], cs[9]), (cs[10], cs[11])))


# Frobenius: computed literally as x -> x^p coefficientwise using conjugation on Fp2 and
# gamma constants derived (not hard-coded) from xi.
_G1 = [fp2_pow(XI, i * (P - 1) // 6) for i in range(6)]        # xi^(i(p-1)/6)


def fp12_frobenius(a):
    """a^p.  basis element v^i w^j = w^(2i+j) maps to gamma_{2i+j} * w^(2i+j) after conjugating
    the Fp2 coefficient, since (w^k)^p = w^k * xi^(k(p-1)/6)."""
    (a0, a1, a2), (b0, b1, b2) = a
    return ((fp2_conj(a0), fp2_mul(fp2_conj(a1), _G1[2]), fp2_mul(fp2_conj(a2), _G1[4])),
            (fp2_mul(fp2_conj(b0), _G1[1]), fp2_mul(fp2_conj(b1), _G1[3]), fp2_mul(fp2_conj(b2), _G1[5])))


# ----------------------------------------------------------------------------- Fr

def fr_inv(a):
    a %= R
    if a == 0:
        raise ZeroDivisionError("Fr inverse of zero")
    return pow(a, R - 2, R)


def fr_from_be32_reduce(b):
    """`Fr::from_slice` of a 32-byte digest as used by src/utils/hash/mod.rs:16,27:
    big-endian integer reduced mod r (ASSUMPTION (i) of SURVEY.md 8c -- isolated here)."""
    if len(b) != 32:
        raise ValueError("InvalidSliceLength")
    return int.from_bytes(b, "big") % R


# ----------------------------------------------------------------------------- generic short-Weierstrass affine groups
# A point is None (infinity) or (x, y).  `F` bundles the field ops so the same code serves
# G1 (over Fp), G2 (over Fp2) and the untwisted image in E(Fp12).

class _Field:
    def __init__(self, add, sub, mul, inv, neg, zero, one, b):
        self.add, self.sub, self.mul, self.inv, self.neg = add, sub, mul, inv, neg
        self.zero, self.one, self.b = zero, one, b


FP = _Field(lambda a, b: (a + b) % P, lambda a, b: (a - b) % P, lambda a, b: (a * b) % P,
            lambda a: pow(a, P - 2, P), lambda a: (-a) % P, 0, 1, B1)
FP2 = _Field(fp2_add, fp2_sub, fp2_mul, fp2_inv, fp2_neg, FP2_ZERO, FP2_ONE, B2)
FP12 = _Field(fp12_add, fp12_sub, fp12_mul, fp12_inv, fp12_neg, FP12_ZERO, FP12_ONE, fp12_from_fp(3))


def ec_is_on_curve(F, pt):
    if pt is None:
        return True
    x, y = pt
    return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), F.b)


def ec_neg(F, pt):
    if pt is None:
        return None
    return (pt[0], F.neg(pt[1]))


def ec_double(F, pt):
    if pt is None:
        return None
    x, y = pt
    if y == F.zero:
        return None
    xx = F.mul(x, x)
    lam = F.mul(F.add(F.add(xx, xx), xx), F.inv(F.add(y, y)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x), x)
    y3 = F.sub(F.mul(lam, F.sub(x, x3)), y)
    return (x3, y3)


def ec_add(F, p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if y1 == y2:
            return ec_double(F, p1)
        return None
    lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_sub(F, p1, p2):
    return ec_add(F, p1, ec_neg(F, p2))


def ec_mul(F, pt, k):
    """Left-to-right binary double-and-add (the operation order of a plain `G * Fr`)."""
    k %= R
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = ec_double(F, acc)
        if bit == "1":
            acc = ec_add(F, acc, pt)
    return acc


G1_GEN = (1, 2)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def g1_add(a, b): return ec_add(FP, a, b)
def g1_sub(a, b): return ec_sub(FP, a, b)
def g1_neg(a): return ec_neg(FP, a)
def g1_mul(a, k): return ec_mul(FP, a, k)
def g2_add(a, b): return ec_add(FP2, a, b)
def g2_sub(a, b): return ec_sub(FP2, a, b)
def g2_neg(a): return ec_neg(FP2, a)
def g2_mul(a, k): return ec_mul(FP2, a, k)


# ----------------------------------------------------------------------------- pairing (literal definition)

def _w_pow(k):
    """w^k as an Fp12 element (k = 2, 3 only needed)."""
    # basis order 1, v, v^2 | w, vw, v^2 w ; w^2 = v ; w^3 = v w
    if k == 2:
        return ((FP2_ZERO, FP2_ONE, FP2_ZERO), FP6_ZERO)
    if k == 3:
        return (FP6_ZERO, (FP2_ZERO, FP2_ONE, FP2_ZERO))
    raise ValueError


def _fp2_in_fp12(a):
    return ((a, FP2_ZERO, FP2_ZERO), FP6_ZERO)


_W2 = _w_pow(2)
_W3 = _w_pow(3)


def untwist(q):
    """psi: E'(Fp2) -> E(Fp12), (x', y') -> (x' w^2, y' w^3)."""
    if q is None:
        return None
    return (fp12_mul(_fp2_in_fp12(q[0]), _W2), fp12_mul(_fp2_in_fp12(q[1]), _W3))


def _line(t, q, pt):
    """Value at pt of the line through t and q (tangent if t == q), all in E(Fp12) affine.
    Vertical lines return x_P - x_T (they lie in a proper subfield and die in the final
    exponentiation, kept only so the function is total)."""
    xt, yt = t
    xq, yq = q
    xp, yp = pt
    if xt != xq:
        lam = fp12_mul(fp12_sub(yq, yt), fp12_inv(fp12_sub(xq, xt)))
    elif yt == yq:
        xx = fp12_mul(xt, xt)
        lam = fp12_mul(fp12_add(fp12_add(xx, xx), xx), fp12_inv(fp12_add(yt, yt)))
    else:
        return fp12_sub(xp, xt)
    return fp12_sub(fp12_sub(yp, yt), fp12_mul(lam, fp12_sub(xp, xt)))


def _frob_point12(q):
    return (fp12_frobenius(q[0]), fp12_frobenius(q[1]))


# Final exponent.  The zcash `bn` lineage (which `rabe-bn` forks, /root/reference/README.md:8)
# takes its `final_exponentiation_last_chunk` from libff's alt_bn128: after the easy part
# (p^6-1)(p^2+1) it raises to
#   lambda = p^3(12z^3+6z^2+4z-1) + p^2(12z^3+6z^2+6z) + p(12z^3+6z^2+4z) + (12z^3+12z^2+6z+1)
# (Fuentes-Castaneda et al.), which is 2z(6z^2+3z+1) TIMES the exact hard exponent
# (p^4-p^2+1)/r -- asserted below.  Both give a bilinear non-degenerate pairing; they differ by
# the fixed power FE_MULTIPLE.  `pairing` follows the lineage (ASSUMPTION (iii) of SURVEY.md 8c,
# corrected: the survey calls the exponent "exact"); `pairing_exact` is the textbook
# (p^12-1)/r variant.  This is the single place the choice is made in the oracle.
_Z = U
FE_EASY = (P**6 - 1) * (P**2 + 1)
FE_HARD_EXACT = (P**4 - P**2 + 1) // R
FE_HARD_LIBFF = (P**3 * (12 * _Z**3 + 6 * _Z**2 + 4 * _Z - 1) + P**2 * (12 * _Z**3 + 6 * _Z**2 + 6 * _Z)
                 + P * (12 * _Z**3 + 6 * _Z**2 + 4 * _Z) + (12 * _Z**3 + 12 * _Z**2 + 6 * _Z + 1))
FE_MULTIPLE = 2 * _Z * (6 * _Z**2 + 3 * _Z + 1)
assert (P**4 - P**2 + 1) % R == 0 and FE_EASY * FE_HARD_EXACT == (P**12 - 1) // R
assert FE_HARD_LIBFF == FE_MULTIPLE * FE_HARD_EXACT
FINAL_EXP_EXACT = FE_EASY * FE_HARD_EXACT
FINAL_EXP = FE_EASY * FE_HARD_LIBFF


def miller_loop(p1, q2):
    """f_{6u+2,Q}(P) * l_{[6u+2]Q, pi(Q)}(P) * l_{[6u+2]Q+pi(Q), -pi^2(Q)}(P), evaluated literally in Fp12."""
    if p1 is None or q2 is None:
        return FP12_ONE
    Q = untwist(q2)
    Pp = (fp12_from_fp(p1[0]), fp12_from_fp(p1[1]))
    T = Q
    f = FP12_ONE
    for bit in bin(ATE_LOOP)[3:]:
        f = fp12_mul(fp12_sqr(f), _line(T, T, Pp))
        T = ec_double(FP12, T)
        if bit == "1":
            f = fp12_mul(f, _line(T, Q, Pp))
            T = ec_add(FP12, T, Q)
    Q1 = _frob_point12(Q)
    Q2 = ec_neg(FP12, _frob_point12(Q1))
    f = fp12_mul(f, _line(T, Q1, Pp))
    T = ec_add(FP12, T, Q1)
    f = fp12_mul(f, _line(T, Q2, Pp))
    return f


def final_exponentiation(f):
    return fp12_pow(f, FINAL_EXP)


def pairing(p1, q2):
    """`rabe_bn::pairing(G1, G2) -> Gt` (call sites: ac17/mod.rs:148,415-416; bsw/mod.rs:108,292-293,308)."""
    return final_exponentiation(miller_loop(p1, q2))


def pairing_exact(p1, q2):
    """Optimal ate pairing with the exact exponent (p^12-1)/r; pairing == pairing_exact ** FE_MULTIPLE."""
    return fp12_pow(miller_loop(p1, q2), FINAL_EXP_EXACT)


def gt_mul(a, b): return fp12_mul(a, b)
def gt_inv(a): return fp12_inv(a)
def gt_pow(a, k): return fp12_pow(a, k % R)


GT_ONE = FP12_ONE

# ----------------------------------------------------------------------------- canonical encodings (spec freeze)

def fp_to_le(x):
    return int(x % P).to_bytes(32, "little")


def fr_to_le(x):
    return int(x % R).to_bytes(32, "little")


def g1_to_le(pt):
    """64 B: x || y little-endian canonical integers; infinity = all-zero."""
    if pt is None:
        return bytes(64)
    return fp_to_le(pt[0]) + fp_to_le(pt[1])


def g1_from_le(b):
    x = int.from_bytes(b[:32], "little")
    y = int.from_bytes(b[32:64], "little")
    if x == 0 and y == 0:
        return None
    return (x, y)


def g2_to_le(pt):
    """128 B: x.c0 || x.c1 || y.c0 || y.c1; infinity = all-zero."""
    if pt is None:
        return bytes(128)
    return fp_to_le(pt[0][0]) + fp_to_le(pt[0][1]) + fp_to_le(pt[1][0]) + fp_to_le(pt[1][1])


def g2_from_le(b):
    v = [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(4)]
    if not any(v):
        return None
    return ((v[0], v[1]), (v[2], v[3]))


def gt_to_le(a):
    """384 B: 12 Fp coefficients, tower order, each 32 B little-endian."""
    return b"".join(fp_to_le(c) for c in fp12_coeffs(a))


def gt_from_le(b):
    return fp12_from_coeffs([int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(12)])
