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
