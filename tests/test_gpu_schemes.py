"""GPU tests of the host layer (rabe::schemes::* mirror): (1) the reference's own round-trip tests, restated;
(2) bit-exact parity of every key / ciphertext element with the golden vectors (oracle, reference operation
order) under explicit randomness."""
import json
import os

import pytest

from rabe_amd import hostlib as hl
from rabe_amd.schemes import ac17, aw11, bdabe, bsw, lsw, mke08

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PLAINTEXT = b"dance like no one's watching, encrypt like everyone is!"     # ac17/mod.rs:764, bsw/mod.rs:331 ...


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


def load(name):
    with open(os.path.join(HERE, "golden", name + ".json")) as f:
        return json.load(f)


def hb(x):
    return bytes.fromhex(x)


def fri(x):
    return int.from_bytes(hb(x), "little")


LANG = {"json": hl.JSON_POLICY, "human": hl.HUMAN_POLICY}

# ------------------------------------------------------------------------------------------------ reference-style round trips


def test_ac17_cp_and_or_or_and_and(host):
    # ac17/mod.rs:756-809: test_cp_and, test_cp_or, test_cp_or_and_and
    pk, msk = ac17.setup(host)
    cases = [('"A" and "B"', ["A", "B"], True), ('"A" and "B"', ["A", "C"], False), ('"A" or "B"', ["B"], True),
             ('"X" or ("B" and ("A" and "C"))', ["A", "B", "C"], True), ('"X" or ("B" and ("A" and "C"))', ["A", "B"], False)]
    for policy, attrs, ok in cases:
        ct = ac17.cp_encrypt(host, pk, policy, PLAINTEXT, hl.HUMAN_POLICY)
        sk = ac17.cp_keygen(host, msk, attrs)
        if ok:
            assert ac17.cp_decrypt(host, sk, ct) == PLAINTEXT
        else:
            with pytest.raises(hl.RabeError):
                ac17.cp_decrypt(host, sk, ct)
    with pytest.raises(hl.RabeError):
        ac17.cp_keygen(host, msk, [])                      # "empty attributes!" :197-199


def test_ac17_batch_api(host):
    pk, msk = ac17.setup(host)
    policies = ['"A" and "B"', '"A" or "C"', '"A" and "B"', '"D" and ("A" or "B")']
    pts = [PLAINTEXT + bytes([i]) for i in range(4)]
    cts = ac17.cp_encrypt_batch(host, pk, policies, pts, hl.HUMAN_POLICY)
    sk_ab = ac17.cp_keygen(host, msk, ["A", "B"])
    got = ac17.cp_decrypt_batch(host, [sk_ab] * 4, cts)
    assert got[0] == pts[0] and got[1] == pts[1] and got[2] == pts[2] and got[3] is None


def test_bsw_reference_cases(host):
    # bsw/mod.rs:320-602: or, and (10 attrs), nested, or3, and, dual attributes, and3, or_and; keygen(None)
    pk, msk = bsw.setup(host)
    assert bsw.keygen(host, pk, msk, []) is None
    and10 = '{"name": "and", "children": [%s]}' % ", ".join('{"name": "attr%d"}' % i for i in range(1, 11))
    nested = '{"name":"and", "children": [{"name": "a2"}, {"name": "a1"}]}'
    for i in range(3, 9):
        nested = '{"name":"and", "children":[{"name": "a%d"}, %s]}' % (i, nested)
    cases = [(r'''{"name": "or", "children": [{"name": "A"}, {"name": "B"}]}''', ["C", "B"], True),
             (and10, ["attr%d" % i for i in range(1, 11)], True),
             (and10, ["attr%d" % i for i in range(1, 10)], False),
             (nested, ["a%d" % i for i in range(1, 9)], True),
             (r'''{"name": "or", "children": [{"name": "X"}, {"name": "Y"}, {"name": "A"}]}''', ["A", "B", "C"], True),
             (r'''{"name": "and", "children":  [{"name": "A"}, {"name": "B"}]}''', ["A", "B"], True),
             (r'''{"name": "or", "children": [{"name": "and", "children":  [{"name": "A"}, {"name": "B"}]}, {"name": "and", "children":  [{"name": "B"}, {"name": "C"}]}]}''', ["B", "C"], True),
             (r'''{"name": "and", "children":  [{"name": "A"}, {"name": "B"}, {"name": "C"}]}''', ["A", "B", "C"], True),
             (r'''{"name": "and", "children":  [{"name": "A"}, {"name": "B"}, {"name": "C"}]}''', ["A", "B"], False)]
    for policy, attrs, ok in cases:
        ct = bsw.encrypt(host, pk, policy, hl.JSON_POLICY, PLAINTEXT)
        sk = bsw.keygen(host, pk, msk, attrs)
        if ok:
            assert bsw.decrypt(host, sk, ct) == PLAINTEXT
        else:
            with pytest.raises(hl.RabeError):
                bsw.decrypt(host, sk, ct)


def test_lsw_reference_cases(host):
    # lsw/mod.rs:292-374: and, or, or_and; non-matching
    pk, msk = lsw.setup(host)
    cases = [(r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}''', ["A", "B"], True),
             (r'''{"name": "or", "children": [{"name": "A"}, {"name": "B"}]}''', ["B", "C"], True),
             (r'''{"name": "or", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}''', ["B", "C"], True),
             (r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}''', ["A", "C"], False)]
    for policy, attrs, ok in cases:
        sk = lsw.keygen(host, pk, msk, policy, hl.JSON_POLICY)
        ct = lsw.encrypt(host, pk, attrs, PLAINTEXT)
        if ok:
            assert lsw.decrypt(host, sk, ct) == PLAINTEXT
        else:
            with pytest.raises(hl.RabeError):
                lsw.decrypt(host, sk, ct)
    with pytest.raises(hl.RabeError):
        lsw.encrypt(host, pk, [], PLAINTEXT)


def test_aw11_reference_cases(host):
    # aw11/mod.rs:392-561: two authorities, and / or policies, add_to_attribute, non-matching key
    gk = aw11.setup(host)
    assert aw11.authgen(host, gk, []) is None
    pk1, msk1 = aw11.authgen(host, gk, ["A", "B"])
    pk2, msk2 = aw11.authgen(host, gk, ["C", "D"])
    sk = aw11.keygen(host, gk, msk1, "bob", ["A"])
    aw11.add_to_attribute(host, gk, msk2, "C", sk)
    pol_and = r'''{"name": "and", "children": [{"name": "A"}, {"name": "C"}]}'''
    ct = aw11.encrypt(host, gk, [pk1, pk2], pol_and, hl.JSON_POLICY, PLAINTEXT)
    assert aw11.decrypt(host, gk, sk, ct) == PLAINTEXT
    pol_or = r'''{"name": "or", "children": [{"name": "B"}, {"name": "and", "children": [{"name": "A"}, {"name": "C"}]}]}'''
    ct = aw11.encrypt(host, gk, [pk1, pk2], pol_or, hl.JSON_POLICY, PLAINTEXT)
    assert aw11.decrypt(host, gk, sk, ct) == PLAINTEXT
    sk_bad = aw11.keygen(host, gk, msk1, "eve", ["A"])
    with pytest.raises(hl.RabeError):
        aw11.decrypt(host, gk, sk_bad, ct)
    with pytest.raises(hl.RabeError):
        aw11.keygen(host, gk, msk1, "", ["A"])


# ------------------------------------------------------------------------------------------------ parity with the golden vectors

def test_ac17_matches_golden(host):
    doc = load("ac17")
    pk_bytes = hb(doc["pk"]["g"]) + (3).to_bytes(4, "little") + b"".join(hb(x) for x in doc["pk"]["h_a"]) + (2).to_bytes(4, "little") + b"".join(hb(x) for x in doc["pk"]["e_gh_ka"])
    pk = hl.Obj.deserialize("ac17_pk", pk_bytes)
    m = doc["msk"]
    msk_bytes = hb(m["g"]) + hb(m["h"]) + (3).to_bytes(4, "little") + b"".join(hb(x) for x in m["g_k"]) + (2).to_bytes(4, "little") + b"".join(hb(x) for x in m["a"]) + (2).to_bytes(4, "little") + b"".join(hb(x) for x in m["b"])
    msk = hl.Obj.deserialize("ac17_msk", msk_bytes)
    assert pk.serialize() == pk_bytes and msk.serialize() == msk_bytes
    for c in doc["cases"]:
        host.set_tape([fri(x) for x in c["keygen_tape"]])
        sk = ac17.cp_keygen(host, msk, c["attrs"])
        got = hl.parse_obj("ac17_cp_sk", sk.serialize())
        assert got["attr"] == c["attrs"]
        assert got["k_0"] == [hb(x) for x in c["sk"]["k_0"]]
        assert got["k"] == [(n, [hb(p) for p in v]) for n, v in c["sk"]["k"]]
        assert got["k_p"] == [hb(x) for x in c["sk"]["k_p"]]
        host.set_tape([fri(x) for x in c["encrypt_tape"]] + [fri(c["msg_rho"]), 7])
        ct = ac17.cp_encrypt(host, pk, c["policy"], PLAINTEXT, LANG[c["language"]])
        g = hl.parse_obj("ac17_cp_ct", ct.serialize())
        assert g["policy"] == (c["policy"], LANG[c["language"]])
        assert g["c_0"] == [hb(x) for x in c["ct"]["c_0"]]
        assert g["c"] == [(n, [hb(p) for p in v]) for n, v in c["ct"]["c"]]
        assert g["c_p"] == hb(c["ct"]["c_p"])
        host.clear_tape()
        assert ac17.cp_decrypt_gt(host, sk, ct) == hb(c["decrypted"])
        assert ac17.cp_decrypt(host, sk, ct) == PLAINTEXT
        # the AES layer: nonce came from the tape (low 12 bytes of 7), key = SHA3-256(bytes(msg))
        assert g["ct"] == hl.encrypt_symmetric(hb(c["msg"]), PLAINTEXT, (7).to_bytes(32, "little")[:12])


def test_bsw_matches_golden(host):
    doc = load("bsw")
    p = doc["pk"]
    pk = hl.Obj.deserialize("bsw_pk", hb(p["g1"]) + hb(p["g2"]) + hb(p["h"]) + hb(p["f"]) + hb(p["e_gg_alpha"]))
    msk = hl.Obj.deserialize("bsw_msk", hb(doc["msk"]["beta"]) + hb(doc["msk"]["g2_alpha"]))
    for c in doc["cases"]:
        host.set_tape([fri(x) for x in c["keygen_tape"]])
        sk = bsw.keygen(host, pk, msk, c["attrs"])
        g = hl.parse_obj("bsw_sk", sk.serialize())
        assert g["d"] == hb(c["sk"]["d"])
        assert g["d_j"] == [(n, hb(a), hb(b)) for n, a, b in c["sk"]["d_j"]]
        et = [fri(x) for x in c["encrypt_tape"]]
        host.set_tape([et[0], fri(c["msg_rho"])] + et[1:] + [9])          # secret, msg, gate coefficients, nonce
        ct = bsw.encrypt(host, pk, c["policy"], LANG[c["language"]], PLAINTEXT)
        g = hl.parse_obj("bsw_ct", ct.serialize())
        assert g["c"] == hb(c["ct"]["c"]) and g["c_p"] == hb(c["ct"]["c_p"])
        assert g["c_y"] == [(n, hb(a), hb(b)) for n, a, b in c["ct"]["c_y"]]
        host.clear_tape()
        assert bsw.decrypt_gt(host, sk, ct) == hb(c["decrypted"])
        assert bsw.decrypt(host, sk, ct) == PLAINTEXT


def test_lsw_matches_golden(host):
    doc = load("lsw")
    p, m = doc["pk"], doc["msk"]
    pk = hl.Obj.deserialize("lsw_pk", hb(p["g1"]) + hb(p["g2"]) + hb(p["g1_b"]) + hb(p["g1_b2"]) + hb(p["h_b"]) + hb(p["e_gg_alpha"]))
    msk = hl.Obj.deserialize("lsw_msk", hb(m["alpha1"]) + hb(m["alpha2"]) + hb(m["b"]) + hb(m["h_g1"]) + hb(m["h_g2"]))
    for c in doc["cases"]:
        host.set_tape([fri(x) for x in c["keygen_tape"]])
        sk = lsw.keygen(host, pk, msk, c["policy"], LANG[c["language"]])
        g = hl.parse_obj("lsw_sk", sk.serialize())
        assert g["dj"] == [(d[0], hb(d[1]), hb(d[2]), hb(d[3]), hb(d[4]), hb(d[5])) for d in c["sk"]["dj"]]
        host.set_tape([fri(x) for x in c["encrypt_tape"]] + [fri(c["msg_rho"]), 11])   # secret, sx.., msg, nonce
        ct = lsw.encrypt(host, pk, c["attrs"], PLAINTEXT)
        g = hl.parse_obj("lsw_ct", ct.serialize())
        assert g["e1"] == hb(c["ct"]["e1"]) and g["e2"] == hb(c["ct"]["e2"])
        assert g["ej"] == [(e[0], hb(e[1]), hb(e[2]), hb(e[3])) for e in c["ct"]["ej"]]
        host.clear_tape()
        assert lsw.decrypt_gt(host, sk, ct) == hb(c["decrypted"])
        assert lsw.decrypt(host, sk, ct) == PLAINTEXT


def test_aw11_matches_golden(host):
    doc = load("aw11")
    gk = hl.Obj.deserialize("aw11_gk", hb(doc["gk"]["g1"]) + hb(doc["gk"]["g2"]))
    auths = []
    for a in doc["authorities"]:
        host.set_tape([fri(x) for x in a["tape"]])
        pk, msk = aw11.authgen(host, gk, a["attrs"])
        assert hl.parse_obj("aw11_pk", pk.serialize())["attr"] == [(n, hb(e), hb(y)) for n, e, y in a["pk"]]
        assert hl.parse_obj("aw11_msk", msk.serialize())["attr"] == [(n, hb(x), hb(y)) for n, x, y in a["msk"]]
        auths.append((pk, msk))
    host.clear_tape()
    for c in doc["cases"]:
        sk = aw11.keygen(host, gk, auths[c["key_authority"]][1], c["gid"], c["key_attrs"])
        assert hl.parse_obj("aw11_sk", sk.serialize())["attr"] == [(n, hb(p)) for n, p in c["sk"]]
        et = [fri(x) for x in c["encrypt_tape"]]
        n_rows = len(c["ct"]["c"])
        head = len(et) - n_rows                                            # s + gate coefficients of both sharings
        host.set_tape(et[:head] + [fri(c["msg_rho"])] + et[head:] + [13])  # ..., msg, r_x per row, nonce
        ct = aw11.encrypt(host, gk, [a[0] for a in auths], c["policy"], LANG[c["language"]], PLAINTEXT)
        g = hl.parse_obj("aw11_ct", ct.serialize())
        assert g["c_0"] == hb(c["ct"]["c_0"])
        assert g["c"] == [(n, hb(a), hb(b), hb(d)) for n, a, b, d in c["ct"]["c"]]
        host.clear_tape()
        assert aw11.decrypt_gt(host, gk, sk, ct) == hb(c["decrypted"])
        assert aw11.decrypt(host, gk, sk, ct) == PLAINTEXT


def test_ac17_kp_reference_cases_and_golden(host):
    # ac17/mod.rs:677-754 (kp_and, kp_or_and, kp_or, non-matching) + golden parity (3-column MSP pins the `_temp` quirk)
    pk, msk = ac17.setup(host)
    pol_ = r'''{"name": "or", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}'''
    sk = ac17.kp_keygen(host, msk, pol_, hl.JSON_POLICY)
    assert ac17.kp_decrypt(host, sk, ac17.kp_encrypt(host, pk, ["B", "C"], PLAINTEXT)) == PLAINTEXT
    assert ac17.kp_decrypt(host, sk, ac17.kp_encrypt(host, pk, ["A"], PLAINTEXT)) == PLAINTEXT
    with pytest.raises(hl.RabeError):
        ac17.kp_decrypt(host, sk, ac17.kp_encrypt(host, pk, ["B", "D"], PLAINTEXT))
    doc = load("ac17_kp")
    pkb = hb(doc["pk"]["g"]) + (3).to_bytes(4, "little") + b"".join(hb(x) for x in doc["pk"]["h_a"]) + (2).to_bytes(4, "little") + b"".join(hb(x) for x in doc["pk"]["e_gh_ka"])
    m = doc["msk"]
    mskb = hb(m["g"]) + hb(m["h"]) + (3).to_bytes(4, "little") + b"".join(hb(x) for x in m["g_k"]) + (2).to_bytes(4, "little") + b"".join(hb(x) for x in m["a"]) + (2).to_bytes(4, "little") + b"".join(hb(x) for x in m["b"])
    gpk, gmsk = hl.Obj.deserialize("ac17_pk", pkb), hl.Obj.deserialize("ac17_msk", mskb)
    for c in doc["cases"]:
        host.set_tape([fri(x) for x in c["keygen_tape"]])
        sk = ac17.kp_keygen(host, gmsk, c["policy"], LANG[c["language"]])
        g = hl.parse_obj("ac17_kp_sk", sk.serialize())
        assert g["k_0"] == [hb(x) for x in c["sk"]["k_0"]]
        assert g["k"] == [(n, [hb(p) for p in v]) for n, v in c["sk"]["k"]]
        assert g["k_p"] == []
        host.set_tape([fri(x) for x in c["encrypt_tape"]] + [fri(c["msg_rho"]), 5])
        ct = ac17.kp_encrypt(host, gpk, c["attrs"], PLAINTEXT)
        g = hl.parse_obj("ac17_kp_ct", ct.serialize())
        assert g["attr"] == c["attrs"]
        assert g["c_0"] == [hb(x) for x in c["ct"]["c_0"]]
        assert g["c"] == [(n, [hb(p) for p in v]) for n, v in c["ct"]["c"]]
        assert g["c_p"] == hb(c["ct"]["c_p"])
        host.clear_tape()
        assert ac17.kp_decrypt_gt(host, sk, ct) == hb(c["decrypted"])
        assert ac17.kp_decrypt(host, sk, ct) == PLAINTEXT


def test_bsw_delegate_matches_golden(host):
    doc = load("bsw_delegate")
    p = doc["pk"]
    pk = hl.Obj.deserialize("bsw_pk", hb(p["g1"]) + hb(p["g2"]) + hb(p["h"]) + hb(p["f"]) + hb(p["e_gg_alpha"]))
    msk = hl.Obj.deserialize("bsw_msk", hb(doc["msk"]["beta"]) + hb(doc["msk"]["g2_alpha"]))
    c = doc["cases"][0]
    host.set_tape([fri(x) for x in c["keygen_tape"]])
    sk = bsw.keygen(host, pk, msk, c["attrs"])
    host.set_tape([fri(x) for x in c["delegate_tape"]])
    dsk = bsw.delegate(host, pk, sk, c["subset"])
    g = hl.parse_obj("bsw_sk", dsk.serialize())
    assert g["d"] == hb(c["delegated"]["d"])
    assert g["d_j"] == [(n, hb(a), hb(b)) for n, a, b in c["delegated"]["d_j"]]
    et = [fri(x) for x in c["encrypt_tape"]]
    host.set_tape([et[0], fri(c["msg_rho"])] + et[1:] + [3])
    ct = bsw.encrypt(host, pk, c["policy"], LANG[c["language"]], PLAINTEXT)
    host.clear_tape()
    assert bsw.decrypt_gt(host, dsk, ct) == hb(c["decrypted"])
    assert bsw.decrypt(host, dsk, ct) == PLAINTEXT
    assert bsw.delegate(host, pk, sk, ["A", "Z"]) is None          # not a subset -> None (:173-176)
    assert bsw.delegate(host, pk, sk, []) is None


def test_ac17_reference_quirks_against_the_oracle(host):
    """AC17 labels are plain concatenations (attribute "01" collides with the k_p label "01"+l+t,
    row label "A1"+"0"+t with column-ish text, rows sort lexicographically "a10" < "a2"), and a policy may name an
    attribute twice.  The engine must reproduce whatever the reference's order of operations gives: compare the host
    layer with the oracle's statement-by-statement restatement on one tape."""
    import random as _random
    from oracle import bn254 as bn
    from oracle import policy as opol
    from oracle import schemes as sch
    from oracle.tape import ListRng
    rnd = _random.Random(77)
    R = bn.R
    pk, msk = ac17.setup(host)
    opk = hl.parse_obj("ac17_pk", pk.serialize())
    omsk = hl.parse_obj("ac17_msk", msk.serialize())
    o_pk = {"g": bn.g1_from_le(opk["g"]), "h_a": [bn.g2_from_le(x) for x in opk["h_a"]], "e_gh_ka": [bn.gt_from_le(x) for x in opk["e_gh_ka"]]}
    o_msk = {"g": bn.g1_from_le(omsk["g"]), "h": bn.g2_from_le(omsk["h"]), "g_k": [bn.g1_from_le(x) for x in omsk["g_k"]],
             "a": [int.from_bytes(x, "little") for x in omsk["a"]], "b": [int.from_bytes(x, "little") for x in omsk["b"]]}
    e_gen = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    # (policy, key attributes, does the reference's own algorithm recover the message?)
    cases = [('"01" and "A1"', ["01", "A1"], True),                              # label collisions
             ('("a10" and "a2") and ("a1" or "A")', ["a1", "a10", "a2"], True),    # lexicographic row order, OR branch
             # the same attribute on two leaves: the name-matching loops (:403-414) add both "A" rows once per pruned entry,
             # so the reference does NOT recover msg -- and neither may the engine
             ('"A" and ("B" or "A")', ["A"], False)]
    for policy, attrs, recovers in cases:
        kt = [rnd.randrange(1, R) for _ in range(3 + len(attrs))]
        host.set_tape(kt)
        sk = ac17.cp_keygen(host, msk, attrs)
        want_sk = sch.ac17_cp_keygen(o_msk, attrs, ListRng(kt))
        got = hl.parse_obj("ac17_cp_sk", sk.serialize())
        assert got["k_0"] == [bn.g2_to_le(x) for x in want_sk["sk"]["k_0"]]
        assert got["k"] == [(n, [bn.g1_to_le(p) for p in v]) for n, v in want_sk["sk"]["k"]]
        assert got["k_p"] == [bn.g1_to_le(x) for x in want_sk["sk"]["k_p"]]
        s0, s1, rho = rnd.randrange(1, R), rnd.randrange(1, R), rnd.randrange(1, R)
        host.set_tape([s0, s1, rho, 5])
        ct = ac17.cp_encrypt(host, pk, policy, PLAINTEXT, hl.HUMAN_POLICY)
        host.clear_tape()
        msg = bn.gt_pow(e_gen, rho)
        want_ct = sch.ac17_cp_encrypt(o_pk, policy, opol.HUMAN, ListRng([s0, s1]), msg)
        g = hl.parse_obj("ac17_cp_ct", ct.serialize())
        assert g["c_0"] == [bn.g2_to_le(x) for x in want_ct["ct"]["c_0"]]
        assert g["c"] == [(n, [bn.g1_to_le(p) for p in v]) for n, v in want_ct["ct"]["c"]]
        assert g["c_p"] == bn.gt_to_le(want_ct["ct"]["c_p"])
        got_gt = ac17.cp_decrypt_gt(host, sk, ct)
        assert got_gt == bn.gt_to_le(sch.ac17_cp_decrypt(want_sk, want_ct))
        assert (got_gt == bn.gt_to_le(msg)) == recovers
        if recovers:
            assert ac17.cp_decrypt(host, sk, ct) == PLAINTEXT
        else:
            with pytest.raises(hl.RabeError):
                ac17.cp_decrypt(host, sk, ct)          # "decryption error: aead::Error", as in the reference


# ------------------------------------------------------------------------------------------------ the DNF schemes
def test_bdabe_reference_cases(host):
    # bdabe/mod.rs:477-663: and, or, or_and, not
    pk, msk = bdabe.setup(host)
    a1, a2, a3 = (bdabe.authgen(host, pk, msk, n) for n in ("aa1", "aa2", "aa3"))
    sk = bdabe.keygen(host, pk, a1, "u1")
    p1 = bdabe.request_attribute_pk(host, pk, a1, "aa1::A")
    p2 = bdabe.request_attribute_pk(host, pk, a2, "aa2::B")
    p3 = bdabe.request_attribute_pk(host, pk, a3, "aa3::C")
    bdabe.request_attribute_sk(host, sk, a1, "aa1::A")
    bdabe.request_attribute_sk(host, sk, a2, "aa2::B")
    ct_and = bdabe.encrypt(host, pk, [p1, p2], r'''{"name": "and", "children": [{"name": "aa1::A"}, {"name": "aa2::B"}]}''', hl.JSON_POLICY, PLAINTEXT)
    assert bdabe.decrypt(host, sk, ct_and) == PLAINTEXT
    ct_or = bdabe.encrypt(host, pk, [p1, p2], r'''{"name": "or", "children": [{"name": "aa1::C"}, {"name": "aa2::B"}]}''', hl.JSON_POLICY, PLAINTEXT)
    assert bdabe.decrypt(host, sk, ct_or) == PLAINTEXT
    ct_human = bdabe.encrypt(host, pk, [p1], '"aa1::A" or "aa1::B"', hl.HUMAN_POLICY, PLAINTEXT)       # the doc example, :13-27
    assert bdabe.decrypt(host, sk, ct_human) == PLAINTEXT
    # or_and: a key with C and B, the policy's X unknown; the shadowed pk list leaves only [aa3::C] in the ciphertext
    sk3 = bdabe.keygen(host, pk, a2, "u1")
    for a, n in ((a1, "aa1::A"), (a2, "aa2::B"), (a3, "aa3::C")):
        bdabe.request_attribute_sk(host, sk3, a, n)
    pol3 = r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "aa3::C"}, {"name": "aa2::B"}]}, {"name": "aa1::X"}]}'''
    ct3 = bdabe.encrypt(host, pk, [p1, p3], pol3, hl.JSON_POLICY, PLAINTEXT)
    assert [t[0] for t in hl.parse_obj("bdabe_ct", ct3.serialize())["j"]] == [["aa3::C"]]
    assert bdabe.decrypt(host, sk3, ct3) == PLAINTEXT
    # not: no attribute of the key in the policy
    ct_not = bdabe.encrypt(host, pk, [p1, p2], r'''{"name": "or", "children": [{"name": "aa1::B"}, {"name": "aa2::A"}]}''', hl.JSON_POLICY, PLAINTEXT)
    with pytest.raises(hl.RabeError):
        bdabe.decrypt(host, sk, ct_not)
    with pytest.raises(hl.RabeError):
        bdabe.request_attribute_pk(host, pk, a1, "aa2::B")                    # not from that authority
    with pytest.raises(hl.RabeError):
        bdabe.encrypt(host, pk, [p1, p2], r'''{"name": "and", "children": [{"name": "or", "children": [{"name": "aa1::A"}, {"name": "aa2::B"}]}, {"name": "aa1::A"}]}''',
                      hl.JSON_POLICY, PLAINTEXT)                              # not in DNF
    # the batch form: per-item failures, the rest decrypt
    got = bdabe.decrypt_batch(host, [sk, sk, sk3, sk], [ct_and, ct_not, ct3, ct_or])
    assert got == [PLAINTEXT, None, PLAINTEXT, PLAINTEXT]
    # the packed form (one blob of records under one key): the same plaintexts, a record the key does not satisfy, a truncated one and one
    # with an element off its group fail alone
    import numpy as np
    recs = [c.serialize() for c in (ct_and, ct_not, ct_or, ct_human)] * 9
    pts_want = [PLAINTEXT, None, PLAINTEXT, PLAINTEXT] * 9
    bad = bytearray(recs[4])
    bad[len(bad) // 2] ^= 0x40                       # somewhere inside an element or the sealed part
    recs[4] = bytes(bad)
    pts_want[4] = None
    recs[8] = recs[8][:50]
    pts_want[8] = None
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    for trusted in (False, True):
        out, oo, st = bdabe.decrypt_packed(host, sk, b"".join(recs), off, trusted=trusted)
        got = [bytes(out[int(oo[i]):int(oo[i + 1])]) if st[i] == 0 else None for i in range(len(recs))]
        assert got == pts_want
    # a key that passes traverse_policy (it holds one OR branch) but satisfies no conjunction carried by the ciphertext: AES fails
    sk_x = bdabe.keygen(host, pk, a1, "u2")
    bdabe.request_attribute_sk(host, sk_x, a1, "aa1::X")
    with pytest.raises(hl.RabeError):
        bdabe.decrypt(host, sk_x, ct3)
    # serialize / deserialize round trip of every new object kind
    for o in (pk, msk, a1, sk3, p1, ct3):
        assert hl.Obj.deserialize(o.kind, o.serialize(), host).serialize() == o.serialize()


def test_mke08_reference_cases(host):
    # mke08/mod.rs:472-619: and, or, or_and
    pk, msk = mke08.setup(host)
    sk = mke08.keygen(host, pk, msk, "user1")
    a1, a2 = mke08.authgen(host, "auth1"), mke08.authgen(host, "auth2")
    names = ["auth1::A", "auth2::B", "auth2::X"]
    auth = {"auth1": a1, "auth2": a2}
    pks = [mke08.request_authority_pk(host, pk, n, auth[n.split("::")[0]]) for n in names]
    for n in names:
        mke08.request_authority_sk(host, sk, n, auth[n.split("::")[0]])
    pol_and = r'''{"name": "and", "children": [{"name": "auth1::A"}, {"name": "auth2::B"}]}'''
    pol_or = r'''{"name": "or", "children": [{"name": "auth1::A"}, {"name": "auth2::B"}]}'''
    pol_or_and = r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "auth1::A"}, {"name": "auth2::B"}]}, {"name": "auth2::X"}]}'''
    cts = [mke08.encrypt(host, pk, pks[:2], pol_and, hl.JSON_POLICY, PLAINTEXT), mke08.encrypt(host, pk, pks[:2], pol_or, hl.JSON_POLICY, PLAINTEXT),
           mke08.encrypt(host, pk, pks, pol_or_and, hl.JSON_POLICY, PLAINTEXT)]
    for ct in cts:
        assert mke08.decrypt(host, sk, ct) == PLAINTEXT
    sk_other = mke08.keygen(host, pk, msk, "user2")
    mke08.request_authority_sk(host, sk_other, "auth1::A", a1)
    assert mke08.decrypt_batch(host, [sk, sk_other, sk_other, sk], [cts[0], cts[0], cts[1], cts[2]]) == [PLAINTEXT, None, PLAINTEXT, PLAINTEXT]
    import numpy as np
    recs = [c.serialize() for c in cts] * 11
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    out, oo, st = mke08.decrypt_packed(host, sk, b"".join(recs), off)
    assert not st.any() and all(bytes(out[int(oo[i]):int(oo[i + 1])]) == PLAINTEXT for i in range(len(recs)))
    out, oo, st = mke08.decrypt_packed(host, sk_other, b"".join(recs), off, trusted=True)
    assert [int(x) for x in st] == [-1, 0, -1] * 11            # sk_other holds auth1::A alone: only the plain OR opens
    with pytest.raises(hl.RabeError):
        mke08.request_authority_pk(host, pk, "auth2::B", a1)
    for o in (pk, msk, a1, sk, pks[0], cts[2]):
        assert hl.Obj.deserialize(o.kind, o.serialize(), host).serialize() == o.serialize()


def test_bdabe_matches_golden(host):
    doc = load("bdabe")
    host.set_tape([fri(x) for x in doc["setup_tape"]])
    pk, msk = bdabe.setup(host)
    g = hl.parse_obj("bdabe_pk", pk.serialize())
    assert g == {k: hb(v) for k, v in doc["pk"].items()} and hl.parse_obj("bdabe_msk", msk.serialize())["y"] == hb(doc["msk"]["y"])
    auths = {}
    for a in doc["authorities"]:
        host.set_tape([fri(x) for x in a["tape"]])
        ska = bdabe.authgen(host, pk, msk, a["name"])
        assert hl.parse_obj("bdabe_ska", ska.serialize()) == {"name": a["name"], "a1": hb(a["a1"]), "a2": hb(a["a2"]), "a3": hb(a["a3"])}
        auths[a["name"]] = ska
    for c in doc["cases"]:
        host.set_tape([fri(x) for x in c["keygen_tape"]])
        uk = bdabe.keygen(host, pk, auths[c["key_authority"]], "u1")
        host.clear_tape()
        for a in c["sk_attrs"]:
            bdabe.request_attribute_sk(host, uk, auths[a.split("::")[0]], a)
        g = hl.parse_obj("bdabe_uk", uk.serialize())
        assert g["sk"] == {k: hb(v) for k, v in c["uk"]["sk"].items()}
        assert (g["pk"]["u1"], g["pk"]["u2"]) == (hb(c["uk"]["pk"]["u1"]), hb(c["uk"]["pk"]["u2"]))
        assert g["sk_a"] == [(n, hb(x), hb(y)) for n, x, y in c["uk"]["sk_a"]]
        pkas = [bdabe.request_attribute_pk(host, pk, auths[a.split("::")[0]], a) for a in c["pk_attrs"]]
        for o, w in zip(pkas, c["pkas"]):
            assert hl.parse_obj("bdabe_pka", o.serialize()) == {"attr": w[0], "a1": hb(w[1]), "a2": hb(w[2]), "a3": hb(w[3])}
        host.set_tape([fri(x) for x in c["encrypt_tape"]] + [17])                     # msg's G1, G2; r_j per term; nonce
        ct = bdabe.encrypt(host, pk, pkas, c["policy"], LANG[c["language"]], PLAINTEXT)
        host.clear_tape()
        assert hl.parse_obj("bdabe_ct", ct.serialize())["j"] == [(t[0], hb(t[1]), hb(t[2]), hb(t[3]), hb(t[4]), hb(t[5])) for t in c["ct"]]
        assert bdabe.decrypt_gt(host, uk, ct) == hb(c["decrypted"]) == hb(c["msg"])
        assert bdabe.decrypt(host, uk, ct) == PLAINTEXT


def test_mke08_matches_golden(host):
    doc = load("mke08")
    host.set_tape([fri(x) for x in doc["setup_tape"]])
    pk, msk = mke08.setup(host)
    assert hl.parse_obj("mke08_pk", pk.serialize()) == {k: hb(v) for k, v in doc["pk"].items()}
    assert hl.parse_obj("mke08_msk", msk.serialize()) == {k: hb(v) for k, v in doc["msk"].items()}
    auths = {}
    for a in doc["authorities"]:
        host.set_tape([fri(x) for x in a["tape"]])
        auths[a["name"]] = mke08.authgen(host, a["name"])
    for c in doc["cases"]:
        host.set_tape([fri(x) for x in c["keygen_tape"]])
        uk = mke08.keygen(host, pk, msk, "user1")
        host.clear_tape()
        for a in c["sk_attrs"]:
            mke08.request_authority_sk(host, uk, a, auths[a.split("::")[0]])
        g = hl.parse_obj("mke08_uk", uk.serialize())
        assert g["sk"] == {k: hb(v) for k, v in c["uk"]["sk"].items()}
        assert g["sk_a"] == [(n, hb(x), hb(y)) for n, x, y in c["uk"]["sk_a"]]
        pkas = [mke08.request_authority_pk(host, pk, a, auths[a.split("::")[0]]) for a in c["pk_attrs"]]
        for o, w in zip(pkas, c["pkas"]):
            assert hl.parse_obj("mke08_pka", o.serialize()) == {"attr": w[0], "g1": hb(w[1]), "g2": hb(w[2]), "gt1": hb(w[3]), "gt2": hb(w[4])}
        host.set_tape([fri(x) for x in c["encrypt_tape"]] + [19])                     # msg1's G1, G2; msg2's exponent; r_j per term; nonce
        ct = mke08.encrypt(host, pk, pkas, c["policy"], LANG[c["language"]], PLAINTEXT)
        host.clear_tape()
        assert hl.parse_obj("mke08_ct", ct.serialize())["e"] == [(t[0], hb(t[1]), hb(t[2]), hb(t[3]), hb(t[4]), hb(t[5]), hb(t[6])) for t in c["ct"]]
        assert mke08.decrypt_gt(host, uk, ct) == hb(c["decrypted"]) == hb(c["msg"])
        assert mke08.decrypt(host, uk, ct) == PLAINTEXT
