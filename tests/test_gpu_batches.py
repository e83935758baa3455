"""Batch entry points of bsw / lsw / aw11 (BASELINE configs 3-5 ask for batches): on the same randomness tape,
item i of a *_batch call is byte-identical to the i-th of n single calls (which tests/test_gpu_schemes.py and
the golden vectors pin to the oracle), and a key that does not satisfy its item's policy fails that item only."""
import random

import pytest

from rabe_amd import hostlib as hl
from rabe_amd.schemes import aw11, bsw, lsw

pytestmark = pytest.mark.gpu
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
PTS = [b"item %d: dance like no one's watching" % i for i in range(5)]


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


def tape(seed, n=4000):
    rnd = random.Random(seed)
    return [rnd.randrange(1, R) for _ in range(n)]


def leaf(a):
    return '{"name": "%s"}' % a


def gate(op, *ch):
    return '{"name": "%s", "children": [%s]}' % (op, ", ".join(ch))


def test_bsw_encrypt_decrypt_batch(host):
    pk, msk = bsw.setup(host)
    attrs = ["A", "B", "C", "D"]
    sk = bsw.keygen(host, pk, msk, attrs)
    sk_ab = bsw.keygen(host, pk, msk, ["A", "B"])
    policies = [gate("and", leaf("A"), leaf("B")), gate("or", leaf("C"), gate("and", leaf("A"), leaf("D"))),
                gate("and", leaf("A"), leaf("B"), leaf("C"), leaf("D")), leaf("B"), gate("and", leaf("C"), leaf("D"))]
    t = tape(3)
    host.set_tape(t)
    batch = bsw.encrypt_batch(host, pk, policies, hl.JSON_POLICY, PTS)
    host.set_tape(t)
    singles = [bsw.encrypt(host, pk, p, hl.JSON_POLICY, pt) for p, pt in zip(policies, PTS)]
    host.clear_tape()
    assert [c.serialize() for c in batch] == [c.serialize() for c in singles]
    assert bsw.decrypt_batch(host, [sk] * 5, batch) == PTS
    # sk_ab satisfies items 0 and 3 only
    assert bsw.decrypt_batch(host, [sk_ab] * 5, batch) == [PTS[0], None, None, PTS[3], None]
    assert bsw.decrypt_batch(host, [], []) == []


def test_lsw_keygen_decrypt_batch(host):
    pk, msk = lsw.setup(host)
    policies = [gate("and", leaf("A"), leaf("B")), gate("or", leaf("C"), gate("and", leaf("A"), leaf("D"))), leaf("B"),
                gate("and", leaf("A"), gate("or", leaf("B"), leaf("E")), leaf("C"))]
    t = tape(4)
    host.set_tape(t)
    batch = lsw.keygen_batch(host, pk, msk, policies, hl.JSON_POLICY)
    host.set_tape(t)
    singles = [lsw.keygen(host, pk, msk, p, hl.JSON_POLICY) for p in policies]
    host.clear_tape()
    assert [k.serialize() for k in batch] == [k.serialize() for k in singles]
    ct_abc = lsw.encrypt(host, pk, ["A", "B", "C"], PTS[0])
    ct_b = lsw.encrypt(host, pk, ["B"], PTS[1])
    assert lsw.decrypt_batch(host, batch, [ct_abc] * 4) == [PTS[0]] * 4
    assert lsw.decrypt_batch(host, batch, [ct_b] * 4) == [None, None, PTS[1], None]
    assert lsw.decrypt_batch(host, batch[:2] + batch[2:], [ct_abc, ct_b, ct_b, ct_abc]) == [PTS[0], None, PTS[1], PTS[0]]


def test_aw11_encrypt_decrypt_batch(host):
    gk = aw11.setup(host)
    pk1, msk1 = aw11.authgen(host, gk, ["A", "B"])
    pk2, msk2 = aw11.authgen(host, gk, ["C", "D"])
    alice = aw11.keygen(host, gk, msk1, "alice", ["A", "B"])
    for a in ("C", "D"):
        aw11.add_to_attribute(host, gk, msk2, a, alice)
    bob = aw11.keygen(host, gk, msk1, "bob", ["A"])
    policies = [gate("and", leaf("A"), leaf("C")), gate("or", leaf("D"), gate("and", leaf("A"), leaf("B"))), leaf("A"),
                gate("and", gate("and", leaf("A"), leaf("B")), gate("and", leaf("C"), leaf("D")))]
    t = tape(5)
    host.set_tape(t)
    batch = aw11.encrypt_batch(host, gk, [pk1, pk2], policies, hl.JSON_POLICY, PTS[:4])
    host.set_tape(t)
    singles = [aw11.encrypt(host, gk, [pk1, pk2], p, hl.JSON_POLICY, pt) for p, pt in zip(policies, PTS)]
    host.clear_tape()
    assert [c.serialize() for c in batch] == [c.serialize() for c in singles]
    assert aw11.decrypt_batch(host, gk, [alice] * 4, batch) == PTS[:4]
    assert aw11.decrypt_batch(host, gk, [bob, alice, bob, bob], batch) == [None, PTS[1], PTS[2], None]


def test_fixed_base_tables_do_not_change_results(host):
    """The host layer serves repeated-base G*Fr / Gt^Fr from cached window tables once a call is large enough;
    forcing the table path and forcing the generic path must give identical bytes (bsw, lsw, aw11 on one tape)."""
    pk, msk = bsw.setup(host)
    lpk, lmsk = lsw.setup(host)
    gk = aw11.setup(host)
    apk, amsk = aw11.authgen(host, gk, ["A", "B", "C"])
    policy = gate("and", leaf("A"), gate("or", leaf("B"), leaf("C")))
    outs = []
    for n_min in (1, 1 << 40):
        host.set_fixed_base_min(n_min)
        host.set_tape(tape(9))
        ct = bsw.encrypt(host, pk, policy, hl.JSON_POLICY, PTS[0])
        sk = bsw.keygen(host, pk, msk, ["A", "B"])
        lsk = lsw.keygen(host, lpk, lmsk, policy, hl.JSON_POLICY)
        lct = lsw.encrypt(host, lpk, ["A", "C"], PTS[1])
        act = aw11.encrypt(host, gk, [apk], policy, hl.JSON_POLICY, PTS[2])
        host.clear_tape()
        outs.append([x.serialize() for x in (ct, sk, lsk, lct, act)])
        assert bsw.decrypt(host, sk, ct) == PTS[0]
        assert lsw.decrypt(host, lsk, lct) == PTS[1]
    host.set_fixed_base_min(4096)
    assert outs[0] == outs[1]


def test_ac17_kp_encrypt_decrypt_batch(host):
    from rabe_amd.schemes import ac17
    pk, msk = ac17.setup(host)
    policies = [gate("and", leaf("A"), leaf("B")), gate("or", leaf("C"), gate("and", leaf("A"), leaf("D")))]
    sks = [ac17.kp_keygen(host, msk, p, hl.JSON_POLICY) for p in policies]
    sets = [["A", "B"], ["C"], ["A", "B", "C", "D"], ["D"], ["A", "D"]]
    t = tape(6)
    host.set_tape(t)
    batch = ac17.kp_encrypt_batch(host, pk, sets, PTS)
    host.set_tape(t)
    singles = [ac17.kp_encrypt(host, pk, s, pt) for s, pt in zip(sets, PTS)]
    host.clear_tape()
    assert [c.serialize() for c in batch] == [c.serialize() for c in singles]
    # key 0 ("A" and "B") opens items 0 and 2; key 1 (C or (A and D)) opens items 1, 2 and 4
    assert ac17.kp_decrypt_batch(host, [sks[0]] * 5, batch) == [PTS[0], None, PTS[2], None, None]
    assert ac17.kp_decrypt_batch(host, [sks[1]] * 5, batch) == [None, PTS[1], PTS[2], None, PTS[4]]


def test_release_before_final_exp_pipelines_two_contexts_with_identical_results():
    """rhip_ctx_release_before_final_exp: context B's decrypt releases context A's stream after its Miller kernel; A's work then runs
    beside B's final exponentiation.  Results must be what the serial order gives, and nothing may hang when the request is unused or
    withdrawn."""
    import random
    from rabe_amd import Engine
    from oracle import bn254 as bn
    rnd = random.Random(5)
    a, b = Engine(0), Engine(0)
    a.set_pairing_mode(1)                                         # the one-lane kernels (k_miller + k_final_exp), whatever the launch size
    b.set_pairing_mode(1)
    n = 700                                                       # a few blocks of the four-wave final exponentiation
    ks = [rnd.randrange(1, bn.R) for _ in range(8)]
    P = [bn.g1_to_le(bn.g1_mul(bn.G1_GEN, k)) for k in ks]
    Q = [bn.g2_to_le(bn.g2_mul(bn.G2_GEN, k + 1)) for k in ks]
    ps, qs = [P[i % 8] for i in range(n)], [Q[(3 * i) % 8] for i in range(n)]
    off = list(range(0, n + 1, 7))                                # items of seven pairs: the shared-accumulator kernel + final exponentiation
    want = a.pairing_product(off, ps, qs)
    b.release_before_final_exp(a)                                 # A is held until B's Miller loops are done and its blocks resident
    got_b = b.pairing_product(off, ps, qs)
    got_a = a.pairing_product(off, ps, qs)
    assert got_a == want and got_b == want
    b.release_before_final_exp(a)
    b._check(b.lib.rhip_ctx_release_before_final_exp(b.ctx, None))          # withdrawn: nothing is held
    assert a.pairing_product(off, ps, qs) == want
    acc = bn.GT_ONE
    for i in range(7):
        acc = bn.gt_mul(acc, bn.pairing(bn.g1_from_le(ps[i]), bn.g2_from_le(qs[i])))
    assert want[0] == bn.gt_to_le(acc)
    a.close()
    b.close()


def test_release_when_miller_resident_gives_identical_results():
    """rhip_ctx_release_when_miller_resident: context A's stream continues as soon as the blocks of context B's next Miller launch are
    resident (k_miller_multi_rr counts them), i.e. A's kernels run beside B's Miller loops.  The bytes must be those of the serial order,
    for uniform pair lists (the early release), for ragged ones and other pairing modes (the request falls back to the release before
    the final exponentiation) and for items without pairs; an unused or withdrawn request holds nothing."""
    import random
    from rabe_amd import Engine
    from oracle import bn254 as bn
    rnd = random.Random(11)
    a, b = Engine(0), Engine(0)
    a.set_pairing_mode(1)
    n_items = 333                                                 # two blocks, the second partly filled
    ks = [rnd.randrange(1, bn.R) for _ in range(8)]
    P = [bn.g1_to_le(bn.g1_mul(bn.G1_GEN, k)) for k in ks]
    Q = [bn.g2_to_le(bn.g2_mul(bn.G2_GEN, k + 1)) for k in ks]
    for per_item in (6, 1):
        n = n_items * per_item
        ps, qs = [P[(i * 5 + 1) % 8] for i in range(n)], [Q[(3 * i) % 8] for i in range(n)]
        off = list(range(0, n + 1, per_item))
        want = a.pairing_product(off, ps, qs)
        for mode in (0, 29):
            b.set_pairing_mode(mode)
            b.release_when_miller_resident(a)
            got_b = b.pairing_jobs(off, ps, qs)
            got_a = a.pairing_product(off, ps, qs)                # behind the hold
            assert got_b == want and got_a == want, (per_item, mode)
        assert b.pairing_jobs(off, ps, qs) == want                # the request was one-shot
    # ragged lists (an item without pairs among them) and explicit other modes: the fallback path, same values
    n = 500
    ps, qs = [P[i % 8] for i in range(n)], [Q[(3 * i + 1) % 8] for i in range(n)]
    off = [0, 0]
    while off[-1] < n:
        off.append(min(n, off[-1] + rnd.randrange(0, 9)))
    want = a.pairing_product(off, ps, qs)
    for mode in (0, 6, 1):
        b.set_pairing_mode(mode)
        b.release_when_miller_resident(a)
        assert b.pairing_jobs(off, ps, qs) == want, mode
        assert a.pairing_product(off, ps, qs) == want
    b.release_when_miller_resident(a)
    b.release_when_miller_resident(None)                         # withdrawn
    assert a.pairing_product(off, ps, qs) == want
    acc = bn.GT_ONE
    for i in range(off[2]):
        acc = bn.gt_mul(acc, bn.pairing(bn.g1_from_le(ps[i]), bn.g2_from_le(qs[i])))
    assert want[0] == bn.gt_to_le(bn.GT_ONE) and want[1] == bn.gt_to_le(acc)
    a.close()
    b.close()
