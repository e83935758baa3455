"""Host-side preparation of AC17 engine inputs for the tests and the benchmark (test/bench plumbing).

Everything here is string / Fr work the reference also does on the host before its group loops:
policy -> MSP (ac17/mod.rs:282-287), label hashing (utils/hash/mod.rs:23-31), pruning
(ac17/mod.rs:391-396).  It produces the numeric records of include/rabe_hip.h.  The Fr
pre-combination A[row][l][t] is the restructuring of SURVEY.md Appendix B.3.
"""
import hashlib

from oracle import bn254 as bn
from oracle import policy as pol


def h_fr(label):
    return int.from_bytes(hashlib.sha3_256(label.encode("utf-8")).digest(), "big") % bn.R


def le(x):
    return int(x % bn.R).to_bytes(32, "little")


def policy_table(policy, language):
    """Returns (pi, A_bytes) with A[row][l][t] = h(pi_row||l||t) + sum_j M[row][j] h("0"||(j+1)||l||t)."""
    tree = pol.parse(policy, language)
    m, pi, c = pol.calculate_msp(tree)
    cols = [[[h_fr("0" + str(j + 1) + str(l) + str(t)) for t in range(2)] for l in range(3)] for j in range(len(m[0]))]
    out = []
    for i, row in enumerate(m):
        for l in range(3):
            for t in range(2):
                v = h_fr(pi[i] + str(l) + str(t))
                for j, mij in enumerate(row):
                    if mij == 1:
                        v += cols[j][l][t]
                    elif mij == -1:
                        v -= cols[j][l][t]
                out.append(le(v))
    return pi, b"".join(out)


def keygen_tables(attributes):
    """H[y][l][t] = h(y||l||t) and H01[l][t] = h("01"||l||t) as canonical Fr bytes."""
    H = b"".join(le(h_fr(a + str(l) + str(t))) for a in attributes for l in range(3) for t in range(2))
    H01 = b"".join(le(h_fr("01" + str(l) + str(t))) for l in range(3) for t in range(2))
    return H, H01


def decrypt_selection(sk_attr_names, ct_row_names, policy, language):
    """The index lists the reference's name-matching loops (ac17/mod.rs:403-414) walk:
    for each pruned leaf, every ciphertext row and every key row with that attribute name.
    Returns (ok, ct_sel, sk_sel)."""
    tree = pol.parse(policy, language)
    if not pol.traverse_policy(sk_attr_names, tree):
        return False, [], []
    ok, lst = pol.calc_pruned(sk_attr_names, tree)
    if not ok:
        return False, [], []
    ct_sel, sk_sel = [], []
    for name, _name_col in lst:
        ct_sel += [i for i, n in enumerate(ct_row_names) if n == name]
        sk_sel += [i for i, n in enumerate(sk_attr_names) if n == name]
    return True, ct_sel, sk_sel
