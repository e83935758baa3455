"""The reference's own fixed-answer (structural) tests, restated against oracle/policy.py.

Sources (inputs and expected outputs are data taken from the reference's tests):
  msp.rs:157-199, secretsharing/mod.rs:228-324, pest/mod.rs:118-149, tools/mod.rs:76-129.
"""
import pytest

from oracle import bn254 as bn
from oracle import policy as pol
from oracle.tape import SeededRng


def test_msp_kat():
    # msp.rs:158-179 (unquoted keys are legal in the JSON grammar)
    p = r'''{name:"and", children:[{name:"A"}, {name:"or", "children":[{name:"D"}, {name:"and", "children":[{name:"B"},{name:"C"}]}]} ]}'''
    m, pi, c = pol.calculate_msp(pol.parse(p, pol.JSON))
    assert pi == ["A", "B", "C", "D"]
    assert m == [[1, 1, 0], [0, -1, 1], [0, 0, -1], [0, -1, 0]]
    assert c == 3


def test_pruning_kat():
    # secretsharing/mod.rs:286-324
    attrs = ["A", "B", "C"]
    pol1 = r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}, {"name": "and", "children": [{"name": "C"}, {"name": "D"}]}]}'''
    pol2 = r'''{"name": "or", "children": [{"name": "C"}, {"name": "and", "children": [{"name": "A"}, {"name": "E"}]}]}'''
    pol3 = r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "A"}, {"name": "C"}]}, {"name": "and", "children": [{"name": "C"}, {"name": "A"}]}]}'''
    assert pol.calc_pruned(attrs, pol.parse(pol1)) == (True, [("A", "A_68"), ("B", "B_83")])
    assert pol.calc_pruned(attrs, pol.parse(pol2)) == (True, [("C", "C_39")])
    assert pol.calc_pruned(attrs, pol.parse(pol3)) == (True, [("A", "A_68"), ("C", "C_83")])


@pytest.mark.parametrize("js,human", [
    (r'''{"name": "A"}''', "A"),
    (r'''{"name": "and", "children": [{"name": "B"}, {"name": "C"}]}''', "(B and C)"),
    (r'''{"name": "or", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}''', "(A or (B and C))"),
])
def test_parse_serialize_kat(js, human):
    # pest/mod.rs:118-149
    tree = pol.parse(js, pol.JSON)
    assert pol.serialize_policy(tree, pol.JSON) == js
    assert pol.serialize_policy(tree, pol.HUMAN) == human


def test_traverse_truth_table():
    # tools/mod.rs:76-129
    with pytest.raises(pol.PolicyError):
        pol.parse("what-the-heck?", pol.JSON)
    p1 = pol.parse(r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}''')
    p2 = pol.parse(r'''{"name": "or", "children": [{"name": "A"}, {"name": "B"}]}''')
    p3 = pol.parse(r'''{"name": "and", "children": [{"name":"or", "children": [{"name": "C"}, {"name": "D"}]}, {"name": "B"}]}''')
    s0, s1, s2, s3 = ["X", "Y"], ["A", "B"], ["C", "D"], ["A", "B", "C", "D"]
    assert [pol.traverse_policy(s, p1) for s in (s0, s1, s2, s3)] == [False, True, False, True]
    assert [pol.traverse_policy(s, p2) for s in (s1, s2, s3)] == [True, False, True]
    assert [pol.traverse_policy(s, p3) for s in (s1, s2, s3)] == [False, False, True]
    assert pol.traverse_policy([], p2) is False


def test_secret_sharing_properties():
    # secretsharing/mod.rs:228-283 (recover_secret == secret for OR 1-of-2 and AND 2-of-2, k-ary AND)
    rng = SeededRng(7)
    for p in [r'''{"name":"or", "children": [{"name": "A"}, {"name": "B"}]}''',
              r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}''',
              r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}, {"name": "C"}, {"name": "D"}]}''',
              r'''{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "B"}, {"name": "and", "children": [{"name": "C"}, {"name": "D"}, {"name": "E"}]}]}]}''']:
        tree = pol.parse(p)
        secret = rng.fr()
        shares = pol.gen_shares_policy(secret, tree, rng)
        coeffs = pol.calc_coefficients(tree, 1)
        assert [s[0] for s in shares] == [c[0] for c in coeffs]
        ok, pruned = pol.calc_pruned(["A", "B", "C", "D", "E"], tree)
        assert ok
        keys = {k for _, k in pruned}
        rec = sum(dict(shares)[k] * dict(coeffs)[k] for k in keys) % bn.R
        assert rec == secret


def test_human_grammar():
    t = pol.parse('"A" and "B"', pol.HUMAN)
    assert t == ("and", [("leaf", "A", 2), ("leaf", "B", 10)])
    t = pol.parse('("A" and "B") or "C"', pol.HUMAN)
    assert t[0] == "or" and t[1][0][0] == "and" and t[1][1] == ("leaf", "C", 19)
    # k-ary chain is ONE node (human.policy.pest:13-18)
    t = pol.parse('"A" and "B" and "C"', pol.HUMAN)
    assert t[0] == "and" and len(t[1]) == 3
    # mixed operators at one level do not parse
    with pytest.raises(pol.PolicyError):
        pol.parse('"A" and "B" or "C"', pol.HUMAN)
    # unquoted operands do not parse
    with pytest.raises(pol.PolicyError):
        pol.parse('A and B', pol.HUMAN)


def test_lw_panics():
    with pytest.raises(pol.PolicyPanic):
        pol.calculate_msp(pol.parse('"A" and "B" and "C"', pol.HUMAN))
    with pytest.raises(pol.PolicyPanic):
        pol.calculate_msp(pol.parse(r'''{"name": "and", "children": [{"name": "A"}]}'''))


def test_msp_rows_sum_to_unit_over_pruned_set():
    # the property AC17 decryption relies on (all reconstruction coefficients are 1)
    p = r'''{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "D"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}]}'''
    tree = pol.parse(p)
    m, pi, c = pol.calculate_msp(tree)
    for attrs in (["A", "D"], ["A", "B", "C"], ["A", "B", "C", "D"]):
        ok, pruned = pol.calc_pruned(attrs, tree)
        assert ok
        names = [n for n, _ in pruned]
        tot = [0] * c
        for n in names:
            row = m[pi.index(n)]
            tot = [a + b for a, b in zip(tot, row)]
        assert tot == [1] + [0] * (c - 1)


# dnf.rs:245-307 (test_dnf_from): which policies are in DNF, and how many conjunctions `json_to_dnf` builds (3 / 1 / 5 -- the
# "2 x child index" placement of dnf.rs:162-164 included)
DNF_IN = [r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}, {"name": "and", "children":  [{"name": "A"}, {"name": "C"}]}]}''',
          r'''{"name": "and", "children": [{"name": "C"}, {"name": "D"}]}''',
          r'''{"name": "or", "children": [{"name": "C"}, {"name": "and",  "children": [{"name": "A"}, {"name": "C"}]}, {"name" :"and",  "children": [{"name": "A"}, {"name": "D"}]}]}''']
DNF_OUT = [r'''{"name": "or", "children":  [{"name": "and",  "children": [{"name": "or",  "children": [{"name": "C"}, {"name": "D"}]}, {"name": "B"}]}, {"name": "and",  "children": [{"name": "C"}, {"name": "D"}]}]}''',
           r'''{"name": "and", "children": [{"name": "or",  "children": [{"name": "A"}, {"name": "B"}]}, {"name": "and",  "children": [{"name": "C"}, {"name": "D"}]}]}''']
DNF_TERMS = [[["A"], ["C"], ["A", "B"]], [["C", "D"]], [["C"], ["A"], ["C"], ["A"], ["D"]]]


def test_dnf_kat():
    ops = (lambda a, b: a + b, lambda a, b: a + b, lambda a, b: a + b)       # the term structure does not depend on the groups
    pks = [(n, 1, 1, 1, 1) for n in "ABCD"]
    for policy, want in zip(DNF_IN, DNF_TERMS):
        tree = pol.parse(policy, pol.JSON)
        assert pol.policy_in_dnf(tree)
        terms = pol.json_to_dnf(tree, pks, ops)
        assert [t[0] for t in terms] == want
        assert all(t[3] == len(t[0]) for t in terms)                         # every attribute added its key once
    for policy in DNF_OUT:
        assert not pol.policy_in_dnf(pol.parse(policy, pol.JSON))
    # an AND below an AND passes policy_in_dnf and fails in json_to_dnf (encrypt's `.unwrap()` panics there)
    nested = r'''{"name": "and", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}'''
    tree = pol.parse(nested, pol.JSON)
    assert pol.policy_in_dnf(tree)
    with pytest.raises(pol.PolicyPanic):
        pol.json_to_dnf(tree, pks, ops)
    # a name that matches no key is dropped, one that matches two keys takes both
    tree = pol.parse(r'''{"name": "and", "children": [{"name": "A"}, {"name": "Z"}]}''', pol.JSON)
    assert [t[0] for t in pol.json_to_dnf(tree, pks + [("A", 1, 1, 1, 1)], ops)] == [["A", "A"]]
