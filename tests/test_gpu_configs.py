"""BASELINE.json configs 3-5 at their full attribute counts (reduced batch): size-independent round-trip
property encrypt -> decrypt == plaintext through the host layer, plus a non-matching key per config.
(Config 2 -- AC17, 50 attributes, batch 4096 -- is bench.py's workload and checks the same property on
every item of the full batch.)"""
import random

import pytest

from rabe_amd import hostlib as hl
from rabe_amd import hostprep as hp
from rabe_amd.schemes import ac17, aw11, bsw, lsw

pytestmark = pytest.mark.gpu
PT = b"dance like no one's watching, encrypt like everyone is!"


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


def test_config1_ac17_five_attribute_and(host):
    # config 1: AC17 CP-ABE, 5-attribute right-nested binary AND
    policy = '{"name":"and","children":[{"name":"A"},{"name":"and","children":[{"name":"B"},{"name":"and","children":[{"name":"C"},{"name":"and","children":[{"name":"D"},{"name":"E"}]}]}]}]}'
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, list("ABCDE"))
    assert ac17.cp_decrypt(host, sk, ac17.cp_encrypt(host, pk, policy, PT, hl.JSON_POLICY)) == PT
    with pytest.raises(hl.RabeError):
        ac17.cp_decrypt(host, ac17.cp_keygen(host, msk, list("ABCD")), ac17.cp_encrypt(host, pk, policy, PT, hl.JSON_POLICY))


def test_config2_ac17_fifty_attributes_batch(host):
    rnd = random.Random(2)
    attrs = ["a%d" % (i + 1) for i in range(50)]
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, attrs)
    policies = [hp.to_json(hp.random_binary_tree(attrs, rnd)) for _ in range(4)]
    items = [policies[i % 4] for i in range(32)]
    pts = [PT + bytes([i]) for i in range(32)]
    cts = ac17.cp_encrypt_batch(host, pk, items, pts, hl.JSON_POLICY)
    assert ac17.cp_decrypt_batch(host, [sk] * 32, cts) == pts          # < 64 items: six independent Miller loops per item
    # >= 64 items: the host layer prepares each distinct key once and runs the paired Miller kernel; two keys
    # alternate, a third one (one attribute short of most policies) fails its items only
    sk2 = ac17.cp_keygen(host, msk, attrs)
    sk_few = ac17.cp_keygen(host, msk, attrs[:1])
    n = 96
    items = [policies[i % 4] for i in range(n)]
    pts = [PT + bytes([i]) for i in range(n)]
    cts = ac17.cp_encrypt_batch(host, pk, items, pts, hl.JSON_POLICY)
    keys = [sk_few if i % 7 == 3 else (sk if i % 2 else sk2) for i in range(n)]
    singles = []
    for k, c in zip(keys, cts):
        try:
            singles.append(ac17.cp_decrypt(host, k, c))
        except hl.RabeError:
            singles.append(None)
    got = ac17.cp_decrypt_batch(host, keys, cts)
    assert got == singles
    assert [g for i, g in enumerate(got) if i % 7 != 3] == [p for i, p in enumerate(pts) if i % 7 != 3]


def test_config3_bsw_hundred_leaf_tree(host):
    # 100 leaves: flat 100-way AND (201 pairings, one final exponentiation) and an AND of ORs
    attrs = ["b%d" % i for i in range(100)]
    flat = '{"name": "and", "children": [%s]}' % ", ".join('{"name": "%s"}' % a for a in attrs)
    mixed = '{"name": "and", "children": [%s]}' % ", ".join(
        '{"name": "or", "children": [{"name": "%s"}, {"name": "%s"}]}' % (attrs[2 * i], attrs[2 * i + 1]) for i in range(50))
    pk, msk = bsw.setup(host)
    sk = bsw.keygen(host, pk, msk, attrs)
    for policy in (flat, mixed):
        assert bsw.decrypt(host, sk, bsw.encrypt(host, pk, policy, hl.JSON_POLICY, PT)) == PT
    sk99 = bsw.keygen(host, pk, msk, attrs[:99])
    with pytest.raises(hl.RabeError):
        bsw.decrypt(host, sk99, bsw.encrypt(host, pk, flat, hl.JSON_POLICY, PT))


def test_config4_lsw_two_hundred_attributes(host):
    attrs = ["c%d" % i for i in range(200)]
    policy = '{"name": "and", "children": [%s]}' % ", ".join('{"name": "%s"}' % a for a in attrs)
    pk, msk = lsw.setup(host)
    sk = lsw.keygen(host, pk, msk, policy, hl.JSON_POLICY)
    assert lsw.decrypt(host, sk, lsw.encrypt(host, pk, attrs, PT)) == PT
    with pytest.raises(hl.RabeError):
        lsw.decrypt(host, sk, lsw.encrypt(host, pk, attrs[:199], PT))


def test_config5_aw11_ten_authorities_twenty_attributes(host):
    gk = aw11.setup(host)
    auth = []
    names = []
    for a in range(10):
        n = ["AUTH%dX%d" % (a, k) for k in range(20)]
        names += n
        auth.append(aw11.authgen(host, gk, n))

    def nest(ns):          # binary ANDs (aw11 builds an MSP, msp.rs:132-134)
        if len(ns) == 1:
            return '{"name": "%s"}' % ns[0]
        h = len(ns) // 2
        return '{"name": "and", "children": [%s, %s]}' % (nest(ns[:h]), nest(ns[h:]))
    policy = nest(names)
    sk = aw11.keygen(host, gk, auth[0][1], "alice", names[:20])
    for a in range(1, 10):
        for n in names[20 * a:20 * a + 20]:
            aw11.add_to_attribute(host, gk, auth[a][1], n, sk)
    ct = aw11.encrypt(host, gk, [p for p, _ in auth], policy, hl.JSON_POLICY, PT)
    assert aw11.decrypt(host, gk, sk, ct) == PT
    sk_few = aw11.keygen(host, gk, auth[0][1], "bob", names[:20])
    with pytest.raises(hl.RabeError):
        aw11.decrypt(host, gk, sk_few, ct)


# ---- BASELINE configs 3-5 at their full per-GPU batch sizes: the size-independent property
# decrypt(encrypt(x)) == x for every item of the batch, through the batch entry points.
def test_config3_bsw_full_batch_round_trip(host):
    n = 4096
    attrs = ["b%d" % i for i in range(100)]
    flat = '{"name": "and", "children": [%s]}' % ", ".join('{"name": "%s"}' % a for a in attrs)
    pk, msk = bsw.setup(host)
    sk = bsw.keygen(host, pk, msk, attrs)
    pts = [PT + i.to_bytes(2, "little") for i in range(n)]
    cts = bsw.encrypt_batch(host, pk, [flat] * n, hl.JSON_POLICY, pts)
    assert bsw.decrypt_batch(host, [sk] * n, cts) == pts


def test_config4_lsw_full_per_gpu_batch_round_trip(host):
    n = 16384 // 8          # config 4 shards 16384 items over 8 GPUs
    attrs = ["c%d" % i for i in range(200)]
    policy = '{"name": "and", "children": [%s]}' % ", ".join('{"name": "%s"}' % a for a in attrs)
    pk, msk = lsw.setup(host)
    ct = lsw.encrypt(host, pk, attrs, PT)
    sks = lsw.keygen_batch(host, pk, msk, [policy] * n, hl.JSON_POLICY)      # the config's timed op: keygen ...
    assert lsw.decrypt_batch(host, sks, [ct] * n) == [PT] * n                # ... + decrypt, every fresh key must open it


def test_config5_aw11_full_per_gpu_batch_round_trip(host):
    n = 8192 // 8
    gk = aw11.setup(host)
    auth, names = [], []
    for a in range(10):
        nm = ["AUTH%dX%d" % (a, k) for k in range(20)]
        names += nm
        auth.append(aw11.authgen(host, gk, nm))

    def nest(ns):
        if len(ns) == 1:
            return '{"name": "%s"}' % ns[0]
        h = len(ns) // 2
        return '{"name": "and", "children": [%s, %s]}' % (nest(ns[:h]), nest(ns[h:]))
    policy = nest(names)
    sk = aw11.keygen(host, gk, auth[0][1], "alice", names[:20])
    for a in range(1, 10):
        for nm in names[20 * a:20 * a + 20]:
            aw11.add_to_attribute(host, gk, auth[a][1], nm, sk)
    pts = [PT + i.to_bytes(2, "little") for i in range(n)]
    cts = aw11.encrypt_batch(host, gk, [p for p, _ in auth], [policy] * n, hl.JSON_POLICY, pts)
    assert aw11.decrypt_batch(host, gk, [sk] * n, cts) == pts


def test_config2_ac17_full_batch_round_trip_through_the_host_layer(host):
    """bench.py drives config 2 at the device level; this is the same batch through rabe::schemes::ac17's batch API."""
    n = 4096
    rnd = random.Random(22)
    attrs = ["a%d" % (i + 1) for i in range(50)]
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, attrs)
    policies = [hp.to_json(hp.random_binary_tree(attrs, rnd)) for _ in range(16)]
    pts = [PT + i.to_bytes(2, "little") for i in range(n)]
    cts = ac17.cp_encrypt_batch(host, pk, [policies[i % 16] for i in range(n)], pts, hl.JSON_POLICY)
    assert ac17.cp_decrypt_batch(host, [sk] * n, cts) == pts
