"""The membership test the engine reads off a finished Miller loop (engine_jobs.hip: k_walk_verdicts; include/rabe_hip.h:
rhip_ctx_collect_walk_verdicts): a point Q of the twist E'(Fp2) lies in G2 exactly when
    [6u+2]Q + psi(Q) - psi^2(Q) + psi^3(Q) = O,
the relation the optimal ate pairing rests on.  Checked here with exact integers (the argument) and with the oracle's curve arithmetic
(members satisfy it, twist points outside G2 do not)."""
import random
from math import gcd

from oracle import bn254 as bn

U = bn.U
P, R = bn.P, bn.R
T = 6 * U * U + 1                  # trace of Frobenius
H2 = P - 1 + T                     # cofactor of G2 in E'(Fp2): #E'(Fp2) = r * h2


def _resultant_with_chi(coeffs):
    """Res(f, chi) for f = sum coeffs[k] X^k and chi = X^2 - T X + P: reduce f mod chi to a X + b, then
    prod over the roots (a x + b) = a^2 P + a b T + b^2."""
    # powers of X mod chi as (a, b) = a X + b
    pw = [(0, 1), (1, 0)]
    while len(pw) < len(coeffs):
        a, b = pw[-1]                      # X * (a X + b) = a X^2 + b X = a (T X - P) + b X
        pw.append((a * T + b, -a * P))
    a = sum(c * pw[k][0] for k, c in enumerate(coeffs))
    b = sum(c * pw[k][1] for k, c in enumerate(coeffs))
    return a * a * P + a * b * T + b * b


def test_parameters():
    assert P == 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
    assert R == 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
    assert H2 == 2 * P - R
    assert (6 * U + 2 + P - P * P + P**3) % R == 0          # the eigenvalue of psi on G2 is p: f(p) = 0 mod r
    assert bn.ATE_LOOP == 6 * U + 2


def test_walk_relation_is_a_membership_test():
    """Q on the twist with f(psi) Q = O, f = X^3 - X^2 + X + (6u+2).  psi also satisfies chi(psi) = 0 there, so every integer combination
    of f and chi kills Q -- in particular their resultant.  Res = r * m with m coprime to r * h2, the group order: the order of Q divides
    r, and the points of order dividing r in E'(Fp2) are G2 (r^2 does not divide the order)."""
    res = _resultant_with_chi([6 * U + 2, 1, -1, 1])
    assert res % R == 0
    m = res // R
    assert gcd(m, R) == 1 and gcd(m, H2) == 1
    assert (R * H2) % (R * R) != 0
    # the same computation confirms the published test the stand-alone kernel uses: (u+1) + u X + u X^2 - 2u X^3
    res2 = _resultant_with_chi([U + 1, U, U, -2 * U])
    assert res2 % R == 0 and gcd(res2 // R, R * H2) == 1


def _psi12(q):
    return (bn.fp12_frobenius(q[0]), bn.fp12_frobenius(q[1]))


def _walk_sum(q2):
    """[6u+2]Q + pi(Q) - pi^2(Q) + pi^3(Q) in E(Fp12), where the twist's psi is the p-power Frobenius pi"""
    q = bn.untwist(q2)
    q1 = _psi12(q)
    q2_ = _psi12(q1)
    q3 = _psi12(q2_)
    acc = bn.ec_mul(bn.FP12, q, 6 * U + 2)
    acc = bn.ec_add(bn.FP12, acc, q1)
    acc = bn.ec_sub(bn.FP12, acc, q2_)
    return bn.ec_add(bn.FP12, acc, q3)


def _fp2_sqrt(a):
    # p = 3 mod 4: a^((p^2+7)/16) style shortcuts do not apply to every BN prime; use the norm method
    a0, a1 = a
    if a1 == 0:
        s = pow(a0, (P + 1) // 4, P)
        if s * s % P == a0:
            return (s, 0)
        s = pow((-a0) % P, (P + 1) // 4, P)          # sqrt(-a0) * u, u^2 = -1
        return (0, s) if s * s % P == (-a0) % P else None
    n = (a0 * a0 + a1 * a1) % P
    s = pow(n, (P + 1) // 4, P)
    if s * s % P != n:
        return None
    for sg in (s, (-s) % P):
        h = (a0 + sg) * pow(2, P - 2, P) % P
        x0 = pow(h, (P + 1) // 4, P)
        if x0 * x0 % P == h and x0:
            x1 = a1 * pow(2 * x0, P - 2, P) % P
            if bn.fp2_mul((x0, x1), (x0, x1)) == (a0 % P, a1 % P):
                return (x0, x1)
    return None


def test_members_satisfy_it_and_other_twist_points_do_not():
    rnd = random.Random(4)
    for _ in range(2):
        q = bn.g2_mul(bn.G2_GEN, rnd.randrange(1, R))
        assert _walk_sum(q) is None
    b2 = bn.fp2_mul((3, 0), bn.fp2_inv(bn.XI))
    found = 0
    while found < 2:
        x = (rnd.randrange(P), rnd.randrange(P))
        y = _fp2_sqrt(bn.fp2_add(bn.fp2_mul(bn.fp2_mul(x, x), x), b2))
        if y is None:
            continue
        t = (x, y)
        assert bn.ec_is_on_curve(bn.FP2, t)
        assert _walk_sum(t) is not None                      # a random twist point has a component in the cofactor
        cof = bn.g2_add(bn.ec_mul(bn.FP2, t, R - 1), t)      # r * T: its G2 component is gone
        shifted = bn.g2_add(bn.g2_mul(bn.G2_GEN, rnd.randrange(1, R)), cof)
        assert _walk_sum(shifted) is not None                # a member plus a cofactor-torsion point
        found += 1
