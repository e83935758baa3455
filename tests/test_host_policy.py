"""CPU tests of the C++ host layer's string / Fr half (no GPU): the reference's structural KATs
(msp.rs:157-199, secretsharing/mod.rs:286-324, pest/mod.rs:118-149, tools/mod.rs:76-129) and a fuzz
comparison against the oracle's independent restatement (oracle/policy.py)."""
import random

import pytest

from oracle import bn254 as bn
from oracle import policy as opol
from oracle.tape import ListRng
from rabe_amd import build, hostlib as hl


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build()


def test_msp_kat():
    p = r'''{name:"and", children:[{name:"A"}, {name:"or", "children":[{name:"D"}, {name:"and", "children":[{name:"B"},{name:"C"}]}]} ]}'''
    d = hl.policy_msp(p)
    assert d["pi"] == ["A", "B", "C", "D"] and d["c"] == 3
    assert d["m"] == [[1, 1, 0], [0, -1, 1], [0, 0, -1], [0, -1, 0]]


def test_pruning_kat():
    attrs = ["A", "B", "C"]
    pol1 = r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}, {"name": "and", "children": [{"name": "C"}, {"name": "D"}]}]}'''
    pol2 = r'''{"name": "or", "children": [{"name": "C"}, {"name": "and", "children": [{"name": "A"}, {"name": "E"}]}]}'''
    pol3 = r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "A"}, {"name": "C"}]}, {"name": "and", "children": [{"name": "C"}, {"name": "A"}]}]}'''
    assert hl.policy_pruned(pol1, attrs) == (True, [("A", "A_68"), ("B", "B_83")])
    assert hl.policy_pruned(pol2, attrs) == (True, [("C", "C_39")])
    assert hl.policy_pruned(pol3, attrs) == (True, [("A", "A_68"), ("C", "C_83")])


@pytest.mark.parametrize("js,human", [
    (r'''{"name": "A"}''', "A"),
    (r'''{"name": "and", "children": [{"name": "B"}, {"name": "C"}]}''', "(B and C)"),
    (r'''{"name": "or", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}''', "(A or (B and C))"),
])
def test_parse_serialize_kat(js, human):
    assert hl.policy_parse(js, hl.JSON_POLICY, hl.JSON_POLICY) == js
    assert hl.policy_parse(js, hl.JSON_POLICY, hl.HUMAN_POLICY) == human


def test_traverse_truth_table():
    with pytest.raises(hl.RabeError):
        hl.policy_parse("what-the-heck?")
    p1 = r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}'''
    p2 = r'''{"name": "or", "children": [{"name": "A"}, {"name": "B"}]}'''
    p3 = r'''{"name": "and", "children": [{"name":"or", "children": [{"name": "C"}, {"name": "D"}]}, {"name": "B"}]}'''
    s0, s1, s2, s3 = ["X", "Y"], ["A", "B"], ["C", "D"], ["A", "B", "C", "D"]
    assert [hl.policy_traverse(p1, s) for s in (s0, s1, s2, s3)] == [False, True, False, True]
    assert [hl.policy_traverse(p2, s) for s in (s1, s2, s3)] == [True, False, True]
    assert [hl.policy_traverse(p3, s) for s in (s1, s2, s3)] == [False, False, True]
    assert hl.policy_traverse(p2, []) is False


def test_panics_and_errors():
    with pytest.raises(hl.RabePanic):
        hl.policy_msp('"A" and "B" and "C"', hl.HUMAN_POLICY)          # lw: AND must be binary
    with pytest.raises(hl.RabePanic):
        hl.policy_msp(r'''{"name": "and", "children": [{"name": "A"}]}''')
    with pytest.raises(hl.RabeError):
        hl.policy_parse('"A" and "B" or "C"', hl.HUMAN_POLICY)
    with pytest.raises(hl.RabeError):
        hl.policy_parse('A and B', hl.HUMAN_POLICY)


def _rand_tree(rnd, names, depth=0, binary=False):
    if depth > 3 or rnd.random() < 0.3:
        return ("leaf", rnd.choice(names))
    k = 2 if binary else rnd.randint(2, 4)
    return (rnd.choice(["and", "or"]), [_rand_tree(rnd, names, depth + 1, binary) for _ in range(k)])


def _json(t, rnd):
    sp = lambda: " " * rnd.randint(0, 2)
    if t[0] == "leaf":
        return '{%s"name"%s:%s"%s"%s}' % (sp(), sp(), sp(), t[1], sp())
    return '{"name":%s"%s",%s"children":%s[%s]}' % (sp(), t[0], sp(), sp(), ("," + sp()).join(_json(c, rnd) for c in t[1]))


def _human(t, rnd):
    if t[0] == "leaf":
        return '"%s"' % t[1]
    return "(" + (" %s " % t[0]).join(_human(c, rnd) for c in t[1]) + ")"


def test_fuzz_against_oracle_policy():
    rnd = random.Random(5)
    names = ["A", "B", "C", "D", "E", "attr-x", "Z9"]
    for it in range(60):
        binary = it % 2 == 0
        t = _rand_tree(rnd, names, binary=binary)
        if t[0] == "leaf":
            continue
        for lang, text in ((hl.JSON_POLICY, _json(t, rnd)), (hl.HUMAN_POLICY, _human(t, rnd))):
            olang = opol.JSON if lang == hl.JSON_POLICY else opol.HUMAN
            tree = opol.parse(text, olang)
            assert hl.policy_parse(text, lang, hl.JSON_POLICY) == opol.serialize_policy(tree, opol.JSON)
            attrs = rnd.sample(names, rnd.randint(1, len(names)))
            assert hl.policy_traverse(text, attrs, lang) == opol.traverse_policy(attrs, tree)
            assert hl.policy_pruned(text, attrs, lang) == opol.calc_pruned(attrs, tree)
            assert hl.policy_coeffs(text, lang) == opol.calc_coefficients(tree, 1)
            tape = [rnd.randrange(bn.R) for _ in range(64)]
            secret = rnd.randrange(bn.R)
            assert hl.policy_shares(text, secret, tape, lang) == opol.gen_shares_policy(secret, tree, ListRng(tape))
            if binary:
                m, pi, c = opol.calculate_msp(tree)
                d = hl.policy_msp(text, lang)
                assert (d["m"], d["pi"], d["c"]) == (m, pi, c)


def test_symmetric_roundtrip_and_tamper():
    gt = bytes(range(256)) + bytes(128)
    nonce = bytes(range(12))
    ct = hl.encrypt_symmetric(gt, b"dance like no one's watching, encrypt like everyone is!", nonce)
    assert ct[:12] == nonce and len(ct) == 12 + 55 + 16
    assert hl.decrypt_symmetric(gt, ct) == b"dance like no one's watching, encrypt like everyone is!"
    bad = bytearray(ct)
    bad[20] ^= 1
    with pytest.raises(hl.RabeError):
        hl.decrypt_symmetric(gt, bytes(bad))
    with pytest.raises(hl.RabeError):
        hl.decrypt_symmetric(bytes(384), ct)          # a different Gt derives a different key


def test_hash_to_fr_and_wide_reduction():
    """sha3_hash_fr of the host layer against the oracle, and the fast 512-bit reduction against long division."""
    import ctypes
    from oracle import schemes as osch
    lib = hl._lib()
    rnd = random.Random(11)
    for label in ["A", "a10", "0111", "attr-x21", "", "ü-umlaut", "x" * 300]:
        o = ctypes.create_string_buffer(32)
        assert lib.rabe_hash_fr(label.encode("utf-8"), o) == 0
        assert int.from_bytes(o.raw, "little") == osch.sha3_hash_fr(label)
    edge = [0, 1, bn.R - 1, bn.R, bn.R + 1, (1 << 256) - 1, 1 << 256, (1 << 512) - 1, bn.R << 256, (bn.R << 256) + bn.R - 1]
    for x in edge + [rnd.getrandbits(512) for _ in range(200)]:
        a, b = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        lib.rabe_fr_reduce512(x.to_bytes(64, "little"), a, b)
        assert int.from_bytes(a.raw, "little") == int.from_bytes(b.raw, "little") == x % bn.R


def test_dnf_kat_and_fuzz():
    """dnf.rs:245-307 through the C ABI, then random AND/OR trees against the oracle's restatement"""
    from tests.test_oracle_policy import DNF_IN, DNF_OUT, DNF_TERMS
    for policy, want in zip(DNF_IN, DNF_TERMS):
        assert hl.policy_in_dnf(policy)
        assert hl.policy_dnf_terms(policy, list("ABCD")) == want
    for policy in DNF_OUT:
        assert not hl.policy_in_dnf(policy)
    with pytest.raises(hl.RabeError):
        hl.policy_dnf_terms(r'''{"name": "and", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}''', list("ABC"))
    rnd = random.Random(99)
    ops = (lambda a, b: a + b,) * 3

    def tree(depth):
        if depth == 0 or rnd.random() < 0.3:
            return '{"name": "%s"}' % rnd.choice("ABCDE")
        kids = ", ".join(tree(depth - 1) for _ in range(rnd.randint(2, 4)))
        return '{"name": "%s", "children": [%s]}' % (rnd.choice(["and", "or"]), kids)

    for _ in range(300):
        policy = tree(3)
        keys = [rnd.choice("ABCDEF") for _ in range(rnd.randint(1, 6))]
        t = opol.parse(policy, opol.JSON)
        assert hl.policy_in_dnf(policy) == opol.policy_in_dnf(t)
        try:
            want = [x[0] for x in opol.json_to_dnf(t, [(k, 1, 1, 1, 1) for k in keys], ops)]
        except opol.PolicyPanic:
            with pytest.raises(hl.RabeError):
                hl.policy_dnf_terms(policy, keys)
            continue
        assert hl.policy_dnf_terms(policy, keys) == want
