"""Round-trip properties of the oracle's scheme restatements, mirroring the reference's own
scheme tests (ac17/mod.rs:677-809, bsw/mod.rs:320-602, lsw/mod.rs:292-374, aw11/mod.rs:392-561):
decrypt(keygen, encrypt(msg)) == msg for matching keys, error for non-matching keys.
Kept small: each oracle pairing costs ~0.3 s of pure-Python big-int work."""
import pytest

from oracle import bn254 as bn
from oracle import policy as pol
from oracle import schemes as sch
from oracle.tape import SeededRng

E_GEN = None


def gt_sample(rng):
    global E_GEN
    if E_GEN is None:
        E_GEN = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    return bn.gt_pow(E_GEN, rng.fr_nonzero())


def test_ac17_cp_and_or():
    rng = SeededRng(11)
    pk, msk = sch.ac17_setup(rng)
    # ac17/mod.rs:756-774 (and), :777-792 (or): matching and non-matching keys
    policy = '"A" and "B"'
    msg = gt_sample(rng)
    ct = sch.ac17_cp_encrypt(pk, policy, pol.HUMAN, rng, msg)
    sk = sch.ac17_cp_keygen(msk, ["A", "B"], rng)
    assert sch.ac17_cp_decrypt(sk, ct) == msg
    sk_bad = sch.ac17_cp_keygen(msk, ["A", "C"], rng)
    with pytest.raises(ValueError):
        sch.ac17_cp_decrypt(sk_bad, ct)
    policy = r'''{"name": "or", "children": [{"name": "X"}, {"name": "and", "children": [{"name": "A"}, {"name": "B"}]}]}'''
    ct = sch.ac17_cp_encrypt(pk, policy, pol.JSON, rng, msg)
    assert sch.ac17_cp_decrypt(sk, ct) == msg


def test_bsw_or_and():
    rng = SeededRng(12)
    pk, msk = sch.bsw_setup(rng)
    msg = gt_sample(rng)
    sk = sch.bsw_keygen(pk, msk, ["A", "B"], rng)
    ct = sch.bsw_encrypt(pk, r'''{"name": "and", "children":  [{"name": "A"}, {"name": "B"}]}''', pol.JSON, rng, msg)
    assert sch.bsw_decrypt(sk, ct) == msg
    ct = sch.bsw_encrypt(pk, r'''{"name": "or", "children": [{"name": "X"}, {"name": "Y"}, {"name": "A"}]}''', pol.JSON, rng, msg)
    assert sch.bsw_decrypt(sk, ct) == msg
    sk_bad = sch.bsw_keygen(pk, msk, ["C"], rng)
    with pytest.raises(ValueError):
        sch.bsw_decrypt(sk_bad, ct)


def test_lsw_and():
    rng = SeededRng(13)
    pk, msk = sch.lsw_setup(rng)
    msg = gt_sample(rng)
    sk = sch.lsw_keygen(pk, msk, r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}''', pol.JSON, rng)
    ct = sch.lsw_encrypt(pk, ["A", "B", "C"], rng, msg)
    assert sch.lsw_decrypt(sk, ct) == msg
    ct = sch.lsw_encrypt(pk, ["A", "C"], rng, msg)
    with pytest.raises(ValueError):
        sch.lsw_decrypt(sk, ct)


def test_aw11_two_authorities():
    rng = SeededRng(14)
    gk = sch.aw11_setup(rng)
    pk1, msk1 = sch.aw11_authgen(gk, ["a", "b"], rng)
    pk2, msk2 = sch.aw11_authgen(gk, ["C"], rng)
    sk = sch.aw11_keygen(gk, msk1, "alice", ["A", "B"])
    msg = gt_sample(rng)
    policy = r'''{"name": "or", "children": [{"name": "C"}, {"name": "and", "children": [{"name": "A"}, {"name": "B"}]}]}'''
    ct = sch.aw11_encrypt(gk, [pk1, pk2], policy, pol.JSON, rng, msg)
    assert sch.aw11_decrypt(gk, sk, ct) == msg


def test_ac17_kp_and_delegate():
    # ac17/mod.rs:677-754 (kp_and, kp_or ...) and bsw/mod.rs:569-602 (delegate_ab)
    rng = SeededRng(15)
    pk, msk = sch.ac17_setup(rng)
    msg = gt_sample(rng)
    policy = r'''{"name": "or", "children": [{"name": "X"}, {"name": "and", "children": [{"name": "A"}, {"name": "B"}]}]}'''
    sk = sch.ac17_kp_keygen(msk, policy, pol.JSON, rng)
    ct = sch.ac17_kp_encrypt(pk, ["A", "B"], rng, msg)
    assert sch.ac17_kp_decrypt(sk, ct) == msg
    ct_bad = sch.ac17_kp_encrypt(pk, ["A", "C"], rng, msg)
    with pytest.raises(ValueError):
        sch.ac17_kp_decrypt(sk, ct_bad)
    bpk, bmsk = sch.bsw_setup(rng)
    bsk = sch.bsw_keygen(bpk, bmsk, ["A", "B", "C"], rng)
    dsk = sch.bsw_delegate(bpk, bsk, ["A", "B"], rng)
    bct = sch.bsw_encrypt(bpk, r'''{"name": "and", "children":  [{"name": "A"}, {"name": "B"}]}''', pol.JSON, rng, msg)
    assert sch.bsw_decrypt(dsk, bct) == msg
    assert sch.bsw_delegate(bpk, bsk, ["A", "Z"], rng) is None


def test_bdabe_reference_cases():
    # bdabe/mod.rs:477-663: and / or / not (or_and with its shadowed key is in the golden vectors)
    rng = SeededRng(16)
    pk, msk = sch.bdabe_setup(rng)
    a1, a2 = sch.bdabe_authgen(pk, msk, "aa1", rng), sch.bdabe_authgen(pk, msk, "aa2", rng)
    sk = sch.bdabe_keygen(pk, a1, "u1", rng)
    p1, p2 = sch.bdabe_request_attribute_pk(pk, a1, "aa1::A"), sch.bdabe_request_attribute_pk(pk, a2, "aa2::B")
    sk["sk_a"].append(sch.bdabe_request_attribute_sk(sk["pk"], a1, "aa1::A"))
    sk["sk_a"].append(sch.bdabe_request_attribute_sk(sk["pk"], a2, "aa2::B"))
    ct, msg = sch.bdabe_encrypt(pk, [p1, p2], r'''{"name": "and", "children": [{"name": "aa1::A"}, {"name": "aa2::B"}]}''', pol.JSON, rng)
    assert [t["attr"] for t in ct["j"]] == [["aa1::A", "aa2::B"]] and sch.bdabe_decrypt(sk, ct) == msg
    ct, msg = sch.bdabe_encrypt(pk, [p1, p2], r'''{"name": "or", "children": [{"name": "aa1::B"}, {"name": "aa2::A"}]}''', pol.JSON, rng)
    with pytest.raises(ValueError):                       # `not`: no attribute of the key appears in the policy
        sch.bdabe_decrypt(sk, ct)
    with pytest.raises(ValueError):                       # an attribute of another authority
        sch.bdabe_request_attribute_pk(pk, a1, "aa2::B")
    with pytest.raises(ValueError):
        sch.bdabe_encrypt(pk, [p1, p2], r'''{"name": "and", "children": [{"name": "or", "children": [{"name": "aa1::A"}, {"name": "aa2::B"}]}, {"name": "aa1::A"}]}''', pol.JSON, rng)


def test_mke08_reference_cases():
    # mke08/mod.rs:472-619: and / or_and
    rng = SeededRng(17)
    pk, msk = sch.mke08_setup(rng)
    sk = sch.mke08_keygen(pk, msk, "user1", rng)
    a1, a2 = sch.mke08_authgen("auth1", rng), sch.mke08_authgen("auth2", rng)
    names = ["auth1::A", "auth2::B", "auth2::X"]
    pks = [sch.mke08_request_authority_pk(pk, n, a1 if n.startswith("auth1") else a2) for n in names]
    for n in names[:2]:
        sk["sk_a"].append(sch.mke08_request_authority_sk(sk["pk"], n, a1 if n.startswith("auth1") else a2))
    policy = r'''{"name": "or", "children": [{"name": "and", "children": [{"name": "auth1::A"}, {"name": "auth2::B"}]}, {"name": "auth2::X"}]}'''
    ct, msg = sch.mke08_encrypt(pk, pks, policy, pol.JSON, rng)
    assert [t["str"] for t in ct["e"]] == [["auth2::X"], ["auth1::A", "auth2::B"]]
    assert sch.mke08_decrypt(sk, ct) == msg
