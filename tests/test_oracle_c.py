"""The C restatement (oracle/c/rabe_ref.c: reference operation order, CPU baseline) against the Python
big-int oracle (ground truth) and against the golden AC17 vectors."""
import hashlib
import json
import os
import random

import pytest

from oracle import bn254 as bn
from oracle import cport
from oracle import policy as pol
from oracle import schemes as sch

pytestmark = pytest.mark.skipif(not cport.available(), reason="gcc build of oracle/c failed")
RND = random.Random(99)
HERE = os.path.dirname(os.path.abspath(__file__))


def test_primitives_match_python_oracle():
    k1, k2 = RND.randrange(1, bn.R), RND.randrange(1, bn.R)
    p, q = bn.g1_mul(bn.G1_GEN, k1), bn.g2_mul(bn.G2_GEN, k1)
    assert cport.g1_mul(p, k2) == bn.g1_mul(p, k2)
    assert cport.g2_mul(q, k2) == bn.g2_mul(q, k2)
    assert cport.g1_mul(p, 0) is None
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    assert cport.pairing(bn.G1_GEN, bn.G2_GEN) == e
    assert cport.pairing(p, q) == bn.gt_pow(e, k1 * k1 % bn.R)
    assert cport.gt_pow(e, k2) == bn.gt_pow(e, k2)
    for label in ["A00", "", "x" * 200]:
        assert cport.hash_fr(label) == bn.fr_from_be32_reduce(hashlib.sha3_256(label.encode()).digest())


def test_ac17_golden_vectors():
    with open(os.path.join(HERE, "golden", "ac17.json")) as f:
        doc = json.load(f)
    hb = bytes.fromhex
    pk = (hb(doc["pk"]["g"]), b"".join(hb(x) for x in doc["pk"]["h_a"]), b"".join(hb(x) for x in doc["pk"]["e_gh_ka"]))
    for c in doc["cases"]:
        s0, s1 = [int.from_bytes(hb(x), "little") for x in c["encrypt_tape"]]
        pi, c0, cc, cp = cport.ac17_cp_encrypt_raw(pk, c["policy"], c["language"], s0, s1, hb(c["msg"]))
        assert pi == [n for n, _ in c["ct"]["c"]]
        assert c0 == b"".join(hb(x) for x in c["ct"]["c_0"])
        assert cc == b"".join(hb(p) for _, v in c["ct"]["c"] for p in v)
        assert cp == hb(c["ct"]["c_p"])
        tree = pol.parse(c["policy"], c["language"])
        ok, lst = pol.calc_pruned(c["attrs"], tree)
        ct_sel, sk_sel = [], []
        for name, _ in lst:
            ct_sel += [i for i, n in enumerate(pi) if n == name]
            sk_sel += [i for i, n in enumerate(c["attrs"]) if n == name]
        out = cport.ac17_cp_decrypt_raw(c0, cc, cp, b"".join(hb(x) for x in c["sk"]["k_0"]),
                                        b"".join(hb(p) for _, v in c["sk"]["k"] for p in v),
                                        b"".join(hb(x) for x in c["sk"]["k_p"]), ct_sel, sk_sel)
        assert out == hb(c["decrypted"])
