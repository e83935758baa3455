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


def test_c_port_scheme_loops_match_python_oracle():
    """The reference-order bsw / lsw / aw11 loops over the C primitives (the CPU baselines of bench.py --config 3/4/5)
    produce the Python oracle's bytes on the same tape and decrypt to the message."""
    from oracle import schemes as sch
    from oracle.tape import ListRng, SeededRng
    rng = SeededRng(77)
    pk, msk = sch.bsw_setup(rng)
    policy = '{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "B"}, {"name": "C"}]}, {"name": "D"}]}'
    sk = sch.bsw_keygen(pk, msk, ["A", "C", "D"], rng)
    msg = bn.gt_pow(pk["e_gg_alpha"], 4242)
    tape = [rng.fr() for _ in range(8)]
    ct = sch.bsw_encrypt(pk, policy, pol.JSON, ListRng(tape), msg)
    pkb = {"g1": bn.g1_to_le(pk["g1"]), "g2": bn.g2_to_le(pk["g2"]), "h": bn.g1_to_le(pk["h"]), "e_gg_alpha": bn.gt_to_le(pk["e_gg_alpha"])}
    ctb = cport.bsw_encrypt_raw(pkb, policy, ListRng(tape), bn.gt_to_le(msg))
    assert ctb["c"] == bn.g1_to_le(ct["c"]) and ctb["c_p"] == bn.gt_to_le(ct["c_p"])
    assert [(n, a, b) for n, a, b in ctb["c_y"]] == [(y["string"], bn.g1_to_le(y["g1"]), bn.g2_to_le(y["g2"])) for y in ct["c_y"]]
    skb = {"d": bn.g2_to_le(sk["d"]), "d_j": [(d["string"], bn.g1_to_le(d["g1"]), bn.g2_to_le(d["g2"])) for d in sk["d_j"]]}
    assert cport.bsw_decrypt_raw(skb, ctb) == bn.gt_to_le(msg) == bn.gt_to_le(sch.bsw_decrypt(sk, ct))
    # lsw / aw11: the workload drivers assert decrypt(...) == msg on every item
    assert cport.lsw_keygen_dec(4, 1, tree="flat", seed=3) > 0
    assert cport.aw11_encdec(10, 1, tree="mixed", seed=3) > 0
