#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the oracle (oracle/*.py).

The reference cannot run here (Rust, and its arithmetic crate rabe-bn is not vendored), and it holds no
known-answer vectors for group values, so these vectors pin the ORACLE: inputs (keys,
policy strings, explicit-randomness tapes) and every output element in the canonical wire format of
include/rabe_hip.h, hex-encoded.  The policy strings and attribute sets are the ones the reference's own
tests use (ac17/mod.rs:756-809, bsw/mod.rs:344-602, lsw/mod.rs:300-374, aw11/mod.rs:400-561, bdabe/mod.rs:477-663,
mke08/mod.rs:472-619).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bn254 as bn  # noqa: E402
from oracle import policy as pol  # noqa: E402
from oracle import schemes as sch  # noqa: E402
from oracle.tape import ListRng, SeededRng  # noqa: E402


def hx(b):
    return b.hex()


def g1(p): return hx(bn.g1_to_le(p))
def g2(p): return hx(bn.g2_to_le(p))
def gt(x): return hx(bn.gt_to_le(x))
def fr(x): return hx(bn.fr_to_le(x))


class RecRng(SeededRng):
    """SeededRng whose tape is recorded so the vectors carry the explicit randomness."""


def primitives():
    rng = SeededRng(100)
    out = {"fp_mul": [], "fr_inv": [], "g1_mul": [], "g2_mul": [], "g1_add": [], "pairing": [], "gt_pow": [], "fr_from_digest": []}
    import random
    rnd = random.Random(1)
    for _ in range(4):
        a, b = rnd.randrange(bn.P), rnd.randrange(bn.P)
        out["fp_mul"].append({"a": hx(bn.fp_to_le(a)), "b": hx(bn.fp_to_le(b)), "out": hx(bn.fp_to_le(a * b % bn.P))})
    for _ in range(3):
        a = rng.fr_nonzero()
        out["fr_inv"].append({"a": fr(a), "out": fr(bn.fr_inv(a))})
    for _ in range(3):
        k, s = rng.fr_nonzero(), rng.fr()
        p = bn.g1_mul(bn.G1_GEN, k)
        q = bn.g2_mul(bn.G2_GEN, k)
        out["g1_mul"].append({"p": g1(p), "k": fr(s), "out": g1(bn.g1_mul(p, s))})
        out["g2_mul"].append({"p": g2(q), "k": fr(s), "out": g2(bn.g2_mul(q, s))})
        p2 = bn.g1_mul(bn.G1_GEN, s)
        out["g1_add"].append({"a": g1(p), "b": g1(p2), "out": g1(bn.g1_add(p, p2))})
    # EIP-196 public vector: 2*(1,2)
    out["g1_mul"].append({"p": g1(bn.G1_GEN), "k": fr(2), "out": g1(bn.g1_mul(bn.G1_GEN, 2)), "note": "EIP-196 doubling vector"})
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    out["pairing"].append({"p": g1(bn.G1_GEN), "q": g2(bn.G2_GEN), "out": gt(e), "note": "generators; libff/zcash-bn final exponent"})
    k1, k2 = rng.fr_nonzero(), rng.fr_nonzero()
    out["pairing"].append({"p": g1(bn.g1_mul(bn.G1_GEN, k1)), "q": g2(bn.g2_mul(bn.G2_GEN, k2)), "out": gt(bn.gt_pow(e, k1 * k2 % bn.R))})
    k = rng.fr()
    out["gt_pow"].append({"a": gt(e), "k": fr(k), "out": gt(bn.gt_pow(e, k))})
    import hashlib
    for label in ["A00", "01 20", "attribute", ""]:
        d = hashlib.sha3_256(label.encode()).digest()
        out["fr_from_digest"].append({"label": label, "digest_be": d.hex(), "out": fr(bn.fr_from_be32_reduce(d))})
    return out


E_GEN = bn.pairing(bn.G1_GEN, bn.G2_GEN)


def ac17_cases():
    rng = SeededRng(17)
    pk, msk = sch.ac17_setup(rng)
    doc = {"pk": {"g": g1(pk["g"]), "h_a": [g2(x) for x in pk["h_a"]], "e_gh_ka": [gt(x) for x in pk["e_gh_ka"]]},
           "msk": {"g": g1(msk["g"]), "h": g2(msk["h"]), "g_k": [g1(x) for x in msk["g_k"]], "a": [fr(x) for x in msk["a"]],
                   "b": [fr(x) for x in msk["b"]]}, "cases": []}
    cases = [('"A" and "B"', pol.HUMAN, ["A", "B"]),
             (r'''{"name": "or", "children": [{"name": "X"}, {"name": "and", "children": [{"name": "A"}, {"name": "B"}]}]}''', pol.JSON, ["A", "B"]),
             (r'''{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "D"}, {"name": "and", "children": [{"name": "B"},{"name": "C"}]}]}]}''', pol.JSON, ["A", "B", "C"])]
    for policy, lang, attrs in cases:
        kt = [rng.fr() for _ in range(2 + len(attrs) + 1)]
        sk = sch.ac17_cp_keygen(msk, attrs, ListRng(kt))
        et = [rng.fr(), rng.fr()]
        rho = rng.fr_nonzero()
        msg = bn.gt_pow(E_GEN, rho)       # `rng.gen::<Gt>()` modelled as e(G1::one(), G2::one())^rho
        ct = sch.ac17_cp_encrypt(pk, policy, lang, ListRng(et), msg)
        dec = sch.ac17_cp_decrypt(sk, ct)
        assert dec == msg
        doc["cases"].append({
            "policy": policy, "language": lang, "attrs": attrs,
            "keygen_tape": [fr(x) for x in kt], "encrypt_tape": [fr(x) for x in et], "msg_rho": fr(rho), "msg": gt(msg),
            "sk": {"k_0": [g2(x) for x in sk["sk"]["k_0"]], "k": [[n, [g1(p) for p in v]] for n, v in sk["sk"]["k"]],
                   "k_p": [g1(x) for x in sk["sk"]["k_p"]]},
            "ct": {"c_0": [g2(x) for x in ct["ct"]["c_0"]], "c": [[n, [g1(p) for p in v]] for n, v in ct["ct"]["c"]],
                   "c_p": gt(ct["ct"]["c_p"])},
            "decrypted": gt(dec)})
    return doc


def bsw_cases():
    rng = SeededRng(18)
    pk, msk = sch.bsw_setup(rng)
    doc = {"pk": {"g1": g1(pk["g1"]), "g2": g2(pk["g2"]), "h": g1(pk["h"]), "f": g2(pk["f"]), "e_gg_alpha": gt(pk["e_gg_alpha"])},
           "msk": {"beta": fr(msk["beta"]), "g2_alpha": g2(msk["g2_alpha"])}, "cases": []}
    cases = [(r'''{"name": "and", "children":  [{"name": "A"}, {"name": "B"}, {"name": "C"}]}''', pol.JSON, ["A", "B", "C"]),
             (r'''{"name": "or", "children": [{"name": "and", "children":  [{"name": "A"}, {"name": "B"}]}, {"name": "and", "children":  [{"name": "C"}, {"name": "D"}]}]}''', pol.JSON, ["C", "D"]),
             ('"A" or ("B" and "C")', pol.HUMAN, ["B", "C"])]
    for policy, lang, attrs in cases:
        kt = [rng.fr() for _ in range(1 + len(attrs))]
        sk = sch.bsw_keygen(pk, msk, attrs, ListRng(kt))
        rec = RecRng(rng.fr() % (1 << 62))
        rho = rng.fr_nonzero()
        msg = bn.gt_pow(E_GEN, rho)
        ct = sch.bsw_encrypt(pk, policy, lang, rec, msg)
        dec = sch.bsw_decrypt(sk, ct)
        assert dec == msg
        doc["cases"].append({
            "policy": policy, "language": lang, "attrs": attrs, "keygen_tape": [fr(x) for x in kt],
            "encrypt_tape": [fr(x) for x in rec.log], "msg_rho": fr(rho), "msg": gt(msg),
            "sk": {"d": g2(sk["d"]), "d_j": [[x["string"], g1(x["g1"]), g2(x["g2"])] for x in sk["d_j"]]},
            "ct": {"c": g1(ct["c"]), "c_p": gt(ct["c_p"]), "c_y": [[x["string"], g1(x["g1"]), g2(x["g2"])] for x in ct["c_y"]]},
            "decrypted": gt(dec)})
    return doc


def lsw_cases():
    rng = SeededRng(19)
    pk, msk = sch.lsw_setup(rng)
    doc = {"pk": {k: (gt(v) if k == "e_gg_alpha" else g2(v) if k == "g2" else g1(v)) for k, v in pk.items()},
           "msk": {"alpha1": fr(msk["alpha1"]), "alpha2": fr(msk["alpha2"]), "b": fr(msk["b"]), "h_g1": g1(msk["h_g1"]), "h_g2": g2(msk["h_g2"])},
           "cases": []}
    cases = [(r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}''', pol.JSON, ["A", "B", "C"]),
             (r'''{"name": "or", "children": [{"name": "X"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}, {"name": "A"}]}]}''', pol.JSON, ["A", "B", "C"])]
    for policy, lang, attrs in cases:
        rk = RecRng(rng.fr() % (1 << 62))
        sk = sch.lsw_keygen(pk, msk, policy, lang, rk)
        re_ = RecRng(rng.fr() % (1 << 62))
        rho = rng.fr_nonzero()
        msg = bn.gt_pow(E_GEN, rho)
        ct = sch.lsw_encrypt(pk, attrs, re_, msg)
        dec = sch.lsw_decrypt(sk, ct)
        assert dec == msg

        def pt(x, f):
            return f(x)
        doc["cases"].append({
            "policy": policy, "language": lang, "attrs": attrs, "keygen_tape": [fr(x) for x in rk.log],
            "encrypt_tape": [fr(x) for x in re_.log], "msg_rho": fr(rho), "msg": gt(msg),
            "sk": {"dj": [[d[0], g1(d[1]), g2(d[2]), g1(d[3]), g1(d[4]), g1(d[5])] for d in sk["dj"]]},
            "ct": {"e1": gt(ct["e1"]), "e2": g2(ct["e2"]), "ej": [[e[0], g1(e[1]), g1(e[2]), g1(e[3])] for e in ct["ej"]]},
            "decrypted": gt(dec)})
    return doc


def aw11_cases():
    rng = SeededRng(20)
    gk = sch.aw11_setup(rng)
    ta1 = [rng.fr() for _ in range(4)]
    ta2 = [rng.fr() for _ in range(2)]
    pk1, msk1 = sch.aw11_authgen(gk, ["a", "b"], ListRng(ta1))
    pk2, msk2 = sch.aw11_authgen(gk, ["C"], ListRng(ta2))
    doc = {"gk": {"g1": g1(gk["g1"]), "g2": g2(gk["g2"])},
           "authorities": [{"attrs": ["a", "b"], "tape": [fr(x) for x in ta1], "pk": [[n, gt(e), g2(y)] for n, e, y in pk1["attr"]],
                            "msk": [[n, fr(a), fr(y)] for n, a, y in msk1["attr"]]},
                           {"attrs": ["C"], "tape": [fr(x) for x in ta2], "pk": [[n, gt(e), g2(y)] for n, e, y in pk2["attr"]],
                            "msk": [[n, fr(a), fr(y)] for n, a, y in msk2["attr"]]}],
           "cases": []}
    sk = sch.aw11_keygen(gk, msk1, "alice", ["A", "B"])
    policy = r'''{"name": "or", "children": [{"name": "C"}, {"name": "and", "children": [{"name": "A"}, {"name": "B"}]}]}'''
    rec = RecRng(rng.fr() % (1 << 62))
    rho = rng.fr_nonzero()
    msg = bn.gt_pow(E_GEN, rho)
    ct = sch.aw11_encrypt(gk, [pk1, pk2], policy, pol.JSON, rec, msg)
    dec = sch.aw11_decrypt(gk, sk, ct)
    assert dec == msg
    doc["cases"].append({
        "policy": policy, "language": pol.JSON, "gid": "alice", "key_authority": 0, "key_attrs": ["A", "B"],
        "encrypt_tape": [fr(x) for x in rec.log], "msg_rho": fr(rho), "msg": gt(msg),
        "sk": [[n, g1(p)] for n, p in sk["attr"]],
        "ct": {"c_0": gt(ct["c_0"]), "c": [[n, gt(c1), g2(c2), g2(c3)] for n, c1, c2, c3 in ct["c"]]},
        "decrypted": gt(dec)})
    return doc


def ac17_kp_cases():
    """KP-ABE variant (ac17/mod.rs:439-675); the second policy has 3 MSP columns, which pins the reference's
    un-reset `_temp` accumulation across columns (:496-516)."""
    rng = SeededRng(21)
    pk, msk = sch.ac17_setup(rng)
    doc = {"pk": {"g": g1(pk["g"]), "h_a": [g2(x) for x in pk["h_a"]], "e_gh_ka": [gt(x) for x in pk["e_gh_ka"]]},
           "msk": {"g": g1(msk["g"]), "h": g2(msk["h"]), "g_k": [g1(x) for x in msk["g_k"]], "a": [fr(x) for x in msk["a"]],
                   "b": [fr(x) for x in msk["b"]]}, "cases": []}
    cases = [(r'''{"name": "or", "children": [{"name": "X"}, {"name": "and", "children": [{"name": "A"}, {"name": "B"}]}]}''', pol.JSON, ["A", "B"]),
             ('"A" and ("B" and ("C" or "D"))', pol.HUMAN, ["A", "B", "D"])]
    for policy, lang, attrs in cases:
        rk = RecRng(rng.fr() % (1 << 62))
        sk = sch.ac17_kp_keygen(msk, policy, lang, rk)
        et = [rng.fr(), rng.fr()]
        rho = rng.fr_nonzero()
        msg = bn.gt_pow(E_GEN, rho)
        ct = sch.ac17_kp_encrypt(pk, attrs, ListRng(et), msg)
        dec = sch.ac17_kp_decrypt(sk, ct)
        doc["cases"].append({
            "policy": policy, "language": lang, "attrs": attrs, "keygen_tape": [fr(x) for x in rk.log],
            "encrypt_tape": [fr(x) for x in et], "msg_rho": fr(rho), "msg": gt(msg),
            "sk": {"k_0": [g2(x) for x in sk["sk"]["k_0"]], "k": [[n, [g1(p) for p in v]] for n, v in sk["sk"]["k"]]},
            "ct": {"c_0": [g2(x) for x in ct["ct"]["c_0"]], "c": [[n, [g1(p) for p in v]] for n, v in ct["ct"]["c"]], "c_p": gt(ct["ct"]["c_p"])},
            "decrypted": gt(dec), "decrypts_to_msg": dec == msg})
    return doc


def bsw_delegate_cases():
    """bsw::delegate (bsw/mod.rs:162-206; reference test delegate_ab :569-602)."""
    rng = SeededRng(22)
    pk, msk = sch.bsw_setup(rng)
    doc = {"pk": {"g1": g1(pk["g1"]), "g2": g2(pk["g2"]), "h": g1(pk["h"]), "f": g2(pk["f"]), "e_gg_alpha": gt(pk["e_gg_alpha"])},
           "msk": {"beta": fr(msk["beta"]), "g2_alpha": g2(msk["g2_alpha"])}, "cases": []}
    kt = [rng.fr() for _ in range(4)]
    sk = sch.bsw_keygen(pk, msk, ["A", "B", "C"], ListRng(kt))
    dt = [rng.fr() for _ in range(3)]
    dsk = sch.bsw_delegate(pk, sk, ["A", "B"], ListRng(dt))
    policy = r'''{"name": "and", "children":  [{"name": "A"}, {"name": "B"}]}'''
    rec = RecRng(rng.fr() % (1 << 62))
    rho = rng.fr_nonzero()
    msg = bn.gt_pow(E_GEN, rho)
    ct = sch.bsw_encrypt(pk, policy, pol.JSON, rec, msg)
    dec = sch.bsw_decrypt(dsk, ct)
    assert dec == msg
    doc["cases"].append({
        "attrs": ["A", "B", "C"], "keygen_tape": [fr(x) for x in kt], "subset": ["A", "B"], "delegate_tape": [fr(x) for x in dt],
        "policy": policy, "language": pol.JSON, "encrypt_tape": [fr(x) for x in rec.log], "msg_rho": fr(rho), "msg": gt(msg),
        "sk": {"d": g2(sk["d"]), "d_j": [[x["string"], g1(x["g1"]), g2(x["g2"])] for x in sk["d_j"]]},
        "delegated": {"d": g2(dsk["d"]), "d_j": [[x["string"], g1(x["g1"]), g2(x["g2"])] for x in dsk["d_j"]]},
        "decrypted": gt(dec)})
    return doc


def ghw11_cases():
    """ghw11 (ghw11/mod.rs:92-305; reference tests or / and / or_and :308-449): keygen -> tkgen -> encrypt -> transform -> decrypt_out."""
    rng = SeededRng(23)
    pk, msk = sch.ghw11_setup(rng)

    def pkd(p):
        return {"g1": g1(p["g1"]), "g2": g2(p["g2"]), "g1_a": g1(p["g1_a"]), "g2_a": g2(p["g2_a"]), "e_gg_alpha": gt(p["e_gg_alpha"])}
    doc = {"pk": pkd(pk), "msk": {"g2_alpha": g2(msk["g2_alpha"])}, "cases": []}
    cases = [(r'''{"name": "or", "children": [{"name": "A"}, {"name": "B"}]}''', pol.JSON, ["D", "B"]),
             (r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}, {"name": "C"}]}''', pol.JSON, ["A", "B", "C"]),
             ('"X" or ("B" and "C")', pol.HUMAN, ["B", "C", "D"])]
    for policy, lang, attrs in cases:
        kt = [rng.fr()]
        sk = sch.ghw11_keygen(pk, msk, attrs, ListRng(kt))
        zt = [rng.fr_nonzero()]
        tk, rk = sch.ghw11_tkgen(sk, ListRng(zt))
        rec = RecRng(rng.fr() % (1 << 62))
        rho = rng.fr_nonzero()
        msg = bn.gt_pow(E_GEN, rho)
        ct = sch.ghw11_encrypt(pk, policy, lang, rec, msg)
        pct = sch.ghw11_transform(ct, tk)
        dec = sch.ghw11_decrypt_out(pct, rk)
        assert dec == msg
        doc["cases"].append({
            "policy": policy, "language": lang, "attrs": attrs, "keygen_tape": [fr(x) for x in kt], "tkgen_tape": [fr(x) for x in zt],
            "encrypt_tape": [fr(x) for x in rec.log], "msg_rho": fr(rho), "msg": gt(msg),
            "sk": {"k": g2(sk["k"]), "l": g2(sk["l"]), "attr_key": [[a["string"], g2(a["k_x"])] for a in sk["attr_key"]]},
            "tk": {"k_z": g2(tk["k_z"]), "l_z": g2(tk["l_z"]), "attr_key_z": [[a["string"], g2(a["k_x"])] for a in tk["attr_key_z"]]},
            "ct": {"c": gt(ct["c"]), "c1": g1(ct["c1"]), "ci_di": [[n, g1(c), g1(d)] for n, c, d in ct["ci_di"]]},
            "t": gt(pct["t"]), "decrypted": gt(dec)})
    return doc


def _dnf_world(setup, seed):
    rng = SeededRng(seed)
    return rng, setup(rng)


def bdabe_cases():
    """bdabe/mod.rs:477-663: `and`, `or`, `or_and` -- including or_and's shadowed `_att2_pk` (the ciphertext only carries the
    conjunction whose public attribute key was passed, :600-626) -- with every element of every key and ciphertext."""
    rng = SeededRng(30)
    ts = [rng.fr_nonzero() for _ in range(5)]
    pk, msk = sch.bdabe_setup(ListRng(ts))
    doc = {"setup_tape": [fr(x) for x in ts],
           "pk": {"g1": g1(pk["g1"]), "g2": g2(pk["g2"]), "p1": g1(pk["p1"]), "p2": g2(pk["p2"]), "e_gg_y": gt(pk["e_gg_y"])},
           "msk": {"y": fr(msk["y"])}, "authorities": [], "cases": []}
    auths = {}
    for name in ("aa1", "aa2", "aa3"):
        ta = [rng.fr(), rng.fr()]
        a = sch.bdabe_authgen(pk, msk, name, ListRng(ta))
        auths[name] = a
        doc["authorities"].append({"name": name, "tape": [fr(x) for x in ta], "a1": g1(a["a1"]), "a2": g2(a["a2"]), "a3": fr(a["a3"])})
    cases = [("and", "aa1", ["aa1::A", "aa2::B"], ["aa1::A", "aa2::B"],
              r'''{"name": "and", "children": [{"name": "aa1::A"}, {"name": "aa2::B"}]}'''),
             ("or", "aa2", ["aa1::C", "aa2::B"], ["aa1::C", "aa2::B"],
              r'''{"name": "or", "children": [{"name": "aa1::A"}, {"name": "aa2::B"}]}'''),
             ("or_and", "aa2", ["aa1::A", "aa2::B", "aa3::C"], ["aa1::A", "aa3::C"],
              r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "aa3::C"}, {"name": "aa2::B"}]}, {"name": "aa1::X"}]}''')]
    for label, key_auth, sk_attrs, pk_attrs, policy in cases:
        tk = [rng.fr()]
        sk = sch.bdabe_keygen(pk, auths[key_auth], "u1", ListRng(tk))
        for a in sk_attrs:
            sk["sk_a"].append(sch.bdabe_request_attribute_sk(sk["pk"], auths[a.split("::")[0]], a))
        pkas = [sch.bdabe_request_attribute_pk(pk, auths[a.split("::")[0]], a) for a in pk_attrs]
        rec = RecRng(rng.fr() % (1 << 62))
        ct, msg = sch.bdabe_encrypt(pk, pkas, policy, pol.JSON, rec)
        dec = sch.bdabe_decrypt(sk, ct)
        assert dec == msg
        doc["cases"].append({
            "label": label, "policy": policy, "language": pol.JSON, "key_authority": key_auth, "keygen_tape": [fr(x) for x in tk],
            "sk_attrs": sk_attrs, "pk_attrs": pk_attrs, "encrypt_tape": [fr(x) for x in rec.log], "msg": gt(msg),
            "uk": {"sk": {"u1": g1(sk["sk"]["u1"]), "u2": g2(sk["sk"]["u2"])}, "pk": {"u1": g1(sk["pk"]["u1"]), "u2": g2(sk["pk"]["u2"])},
                   "sk_a": [[k["attr"], g1(k["au1"]), g2(k["au2"])] for k in sk["sk_a"]]},
            "pkas": [[k["attr"], g1(k["a1"]), g2(k["a2"]), gt(k["a3"])] for k in pkas],
            "ct": [[t["attr"], gt(t["e1"]), g1(t["e2"]), g2(t["e3"]), g1(t["e4"]), g2(t["e5"])] for t in ct["j"]],
            "decrypted": gt(dec)})
    return doc


def mke08_cases():
    """mke08/mod.rs:472-619: `and`, `or`, `or_and`."""
    rng = SeededRng(31)
    ts = [rng.fr_nonzero() for _ in range(6)]
    pk, msk = sch.mke08_setup(ListRng(ts))
    doc = {"setup_tape": [fr(x) for x in ts],
           "pk": {"g1": g1(pk["g1"]), "g2": g2(pk["g2"]), "p1": g1(pk["p1"]), "p2": g2(pk["p2"]), "e_gg_y1": gt(pk["e_gg_y1"]), "e_gg_y2": gt(pk["e_gg_y2"])},
           "msk": {"g1": g1(msk["g1"]), "g2": g2(msk["g2"])}, "authorities": [], "cases": []}
    auths = {}
    for name in ("auth1", "auth2", "auth3"):
        ta = [rng.fr()]
        auths[name] = sch.mke08_authgen(name, ListRng(ta))
        doc["authorities"].append({"name": name, "tape": [fr(x) for x in ta]})
    cases = [("and", ["auth1::A", "auth2::B"], ["auth1::A", "auth2::B"],
              r'''{"name": "and", "children": [{"name": "auth1::A"}, {"name": "auth2::B"}]}'''),
             ("or", ["auth1::C", "auth2::B"], ["auth1::C", "auth2::B"],
              r'''{"name": "or", "children": [{"name": "auth1::A"}, {"name": "auth2::B"}]}'''),
             ("or_and", ["auth1::A", "auth2::B", "auth2::X"], ["auth1::A", "auth2::B", "auth2::X"],
              r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "auth1::A"}, {"name": "auth2::B"}]}, {"name": "auth2::X"}]}'''),
             # three conjunctions, the middle child of the OR lands on term index 2 (dnf.rs:162-164): terms [[C], [A], [A, B], [C]] after the sort
             ("three_terms", ["auth1::A", "auth3::C"], ["auth1::A", "auth2::B", "auth3::C"],
              r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "auth1::A"}, {"name": "auth2::B"}]}, {"name": "auth3::C"}, {"name": "and", "children": [{"name": "auth1::A"}, {"name": "auth3::C"}]}]}''')]
    for label, sk_attrs, pk_attrs, policy in cases:
        tk = [rng.fr()]
        sk = sch.mke08_keygen(pk, msk, "user1", ListRng(tk))
        for a in sk_attrs:
            sk["sk_a"].append(sch.mke08_request_authority_sk(sk["pk"], a, auths[a.split("::")[0]]))
        pkas = [sch.mke08_request_authority_pk(pk, a, auths[a.split("::")[0]]) for a in pk_attrs]
        rec = RecRng(rng.fr() % (1 << 62))
        ct, msg = sch.mke08_encrypt(pk, pkas, policy, pol.JSON, rec)
        dec = sch.mke08_decrypt(sk, ct)
        assert dec == msg
        doc["cases"].append({
            "label": label, "policy": policy, "language": pol.JSON, "keygen_tape": [fr(x) for x in tk],
            "sk_attrs": sk_attrs, "pk_attrs": pk_attrs, "encrypt_tape": [fr(x) for x in rec.log], "msg": gt(msg),
            "uk": {"sk": {"g1": g1(sk["sk"]["g1"]), "g2": g2(sk["sk"]["g2"])}, "pk": {"g1": g1(sk["pk"]["g1"]), "g2": g2(sk["pk"]["g2"])},
                   "sk_a": [[k["attr"], g1(k["g1"]), g2(k["g2"])] for k in sk["sk_a"]]},
            "pkas": [[k["attr"], g1(k["g1"]), g2(k["g2"]), gt(k["gt1"]), gt(k["gt2"])] for k in pkas],
            "ct": [[t["str"], gt(t["j1"]), gt(t["j2"]), g1(t["j3"]), g2(t["j4"]), g1(t["j5"]), g2(t["j6"])] for t in ct["e"]],
            "decrypted": gt(dec)})
    return doc


def main():
    import sys as _sys
    only = set(_sys.argv[1:])
    for name, fn in (("bn254_primitives", primitives), ("ac17", ac17_cases), ("bsw", bsw_cases), ("lsw", lsw_cases), ("aw11", aw11_cases),
                     ("ac17_kp", ac17_kp_cases), ("bsw_delegate", bsw_delegate_cases), ("ghw11", ghw11_cases), ("bdabe", bdabe_cases),
                     ("mke08", mke08_cases)):
        if only and name not in only:
            continue
        doc = fn()
        path = os.path.join(HERE, name + ".json")
        with open(path, "w") as f:
            json.dump(doc, f, indent=1)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
