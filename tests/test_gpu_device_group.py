"""A device GROUP (include/rabe_host.h: rabe_host_open_group): the packed entry points of ac17 / bsw / lsw / aw11 shard their items into one
contiguous block per engine.  On a one-GPU box the group lists device 0 several times (separate engines = streams, workspaces, key-table
replicas on the same GPU): the blobs, offsets, status entries and plaintexts must be byte-identical to the single-engine call on the same
tape -- for 2 and 8 engines, for item counts that do not divide evenly and for fewer items than engines."""
import numpy as np
import pytest

from rabe_amd import hostlib as hl

pytestmark = pytest.mark.gpu

AC_POLS = ['"A" and "B"', '"A" or "C"', '("D" and "B") or "C"']
BSW_POLS = ['"A" and "B" and "C"', '"A" or ("B" and "D")', '("C" or "D") and ("A" or "E") and "B"']
LSW_POLS = ['{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}', '{"name": "or", "children": [{"name": "A"}, {"name": "D"}]}',
            '{"name": "and", "children": [{"name": "C"}, {"name": "or", "children": [{"name": "B"}, {"name": "E"}]}]}']
AW_POLS = ['{"name": "and", "children": [{"name": "A"}, {"name": "D"}]}', '{"name": "or", "children": [{"name": "B"}, {"name": "E"}]}',
           '{"name": "and", "children": [{"name": "C"}, {"name": "or", "children": [{"name": "A"}, {"name": "E"}]}]}']


def offsets(items):
    return np.concatenate([[0], np.cumsum([len(p) for p in items])]).astype(np.uint64)


@pytest.fixture(scope="module")
def hosts():
    hs = {1: hl.Host(0), 2: hl.Host(devices=[0, 0]), 8: hl.Host(devices=[0] * 8)}
    assert [h.group_size() for h in hs.values()] == [1, 2, 8]
    yield hs
    for h in hs.values():
        h.close()


def same(a, b):
    return all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))


@pytest.mark.parametrize("n", [21, 5])
def test_ac17_group_equals_single_engine(hosts, n):
    from rabe_amd.schemes import ac17
    pk, msk = ac17.setup(hosts[1])
    item_pol = [i % 3 for i in range(n)]
    pts = [b"ac17-%d " % i * (i % 4 + 1) for i in range(n)]
    tape = [1000003 * (i + 7) + 11 for i in range(4 * n)]
    sk_abc = ac17.cp_keygen(hosts[1], msk, ["A", "B", "C"])
    sk_a = ac17.cp_keygen(hosts[1], msk, ["A"])                     # satisfies policy 1 only: the other items fail alone
    ref = None
    for g, h in hosts.items():
        h.set_tape(tape)
        blob, ct_off = ac17.cp_encrypt_packed(h, pk, AC_POLS, item_pol, b"".join(pts), offsets(pts), hl.HUMAN_POLICY)
        h.clear_tape()
        dec = ac17.cp_decrypt_packed(h, sk_abc, blob, ct_off)
        dec_some = ac17.cp_decrypt_packed(h, sk_a, blob, ct_off, trusted=True)
        got = (blob, ct_off) + tuple(dec) + tuple(dec_some)
        if ref is None:
            ref = got
            assert not dec[2].any() and dec[0].tobytes() == b"".join(pts)
            assert [int(s) for s in dec_some[2]] == [0 if p == 1 else -1 for p in item_pol]
        else:
            assert same(ref, got), "group of %d engines differs from the single engine" % g


def test_bsw_group_equals_single_engine(hosts):
    from rabe_amd.schemes import bsw
    pk, msk = bsw.setup(hosts[1])
    n = 13
    item_pol = [i % 3 for i in range(n)]
    pts = [b"bsw plaintext %d " % i * (i % 3 + 1) for i in range(n)]
    tape = [1000003 * (i + 5) + 17 for i in range(40 * n)]
    sk = bsw.keygen(hosts[1], pk, msk, ["A", "B", "C", "D"])
    sk_ab = bsw.keygen(hosts[1], pk, msk, ["A", "B"])
    ref = None
    for g, h in hosts.items():
        h.set_tape(tape)
        blob, ct_off = bsw.encrypt_packed(h, pk, BSW_POLS, item_pol, b"".join(pts), offsets(pts), hl.HUMAN_POLICY)
        h.clear_tape()
        got = (blob, ct_off) + tuple(bsw.decrypt_packed(h, sk, blob, ct_off)) + tuple(bsw.decrypt_packed(h, sk_ab, blob, ct_off, trusted=True))
        if ref is None:
            ref = got
            assert not got[4].any() and got[2].tobytes() == b"".join(pts)
            assert got[7].any() and not got[7].all()
        else:
            assert same(ref, got), "group of %d engines differs from the single engine" % g


def test_lsw_group_equals_single_engine(hosts):
    from rabe_amd.schemes import lsw
    pk, msk = lsw.setup(hosts[1])
    n = 11
    item_pol = [i % 3 for i in range(n)]
    pt = b"lsw: one ciphertext, a fresh key per item"
    ct = lsw.encrypt(hosts[1], pk, ["A", "B", "C"], pt)
    ct2 = lsw.encrypt(hosts[1], pk, ["A", "E"], pt)                  # policy 1 only
    tape = [1000003 * (i + 9) + 29 for i in range(20 * n)]
    ref = None
    for g, h in hosts.items():
        h.set_tape(tape)
        blob, sk_off = lsw.keygen_packed(h, pk, msk, LSW_POLS, item_pol, hl.JSON_POLICY)
        h.clear_tape()
        got = (blob, sk_off) + tuple(lsw.decrypt_packed(h, ct, blob, sk_off)) + tuple(lsw.decrypt_packed(h, ct2, blob, sk_off, trusted=True))
        if ref is None:
            ref = got
            assert not got[4].any() and got[2].tobytes() == pt * n
            assert got[7].any() and not got[7].all()
        else:
            assert same(ref, got), "group of %d engines differs from the single engine" % g


def test_aw11_group_equals_single_engine(hosts):
    from rabe_amd.schemes import aw11
    h1 = hosts[1]
    gk = aw11.setup(h1)
    pk1, msk1 = aw11.authgen(h1, gk, ["A", "B", "C"])
    pk2, msk2 = aw11.authgen(h1, gk, ["D", "E"])
    n = 10
    item_pol = [i % 3 for i in range(n)]
    pts = [b"aw11 plaintext %d " % i * (i % 3 + 1) for i in range(n)]
    tape = [1000003 * (i + 3) + 41 for i in range(40 * n)]
    sk = aw11.keygen(h1, gk, msk1, "alice", ["A", "B", "C"])
    aw11.add_to_attribute(h1, gk, msk2, "D", sk)
    aw11.add_to_attribute(h1, gk, msk2, "E", sk)
    ref = None
    for g, h in hosts.items():
        h.set_tape(tape)
        blob, ct_off = aw11.encrypt_packed(h, gk, [pk1, pk2], AW_POLS, item_pol, b"".join(pts), offsets(pts), hl.JSON_POLICY)
        h.clear_tape()
        got = (blob, ct_off) + tuple(aw11.decrypt_packed(h, gk, sk, blob, ct_off))
        if ref is None:
            ref = got
            assert not got[4].any() and got[2].tobytes() == b"".join(pts)
        else:
            assert same(ref, got), "group of %d engines differs from the single engine" % g


@pytest.mark.parametrize("bad", [0, 1])
def test_a_worker_that_fails_before_its_block_is_an_error_not_a_hang(hosts, bad, monkeypatch):
    """pipeline.cpp: fan_out -- in a group, chunk w belongs to worker w alone; if that worker fails before run(w) (injected here), the chunk
    never takes its turn at the draw gate.  The call has to come back with the error, and the group has to work afterwards."""
    import threading
    from rabe_amd.schemes import ac17
    h = hosts[2]
    pk, _msk = ac17.setup(hosts[1])
    n = 6
    pts = [b"fault-%d" % i for i in range(n)]
    tape = [7919 * (i + 3) for i in range(4 * n)]
    box = {}

    def call():
        try:
            h.set_tape(tape)
            box["out"] = ac17.cp_encrypt_packed(h, pk, AC_POLS, [i % 3 for i in range(n)], b"".join(pts), offsets(pts), hl.HUMAN_POLICY)
        except Exception as ex:          # the C ABI's error, as hostlib raises it
            box["err"] = ex
        finally:
            h.clear_tape()

    monkeypatch.setenv("RABE_FAULT_GROUP_WORKER", str(bad))
    t = threading.Thread(target=call, daemon=True)
    t.start()
    t.join(60)
    assert not t.is_alive(), "the packed call hangs when worker %d fails before its block" % bad
    assert "err" in box and "injected fault" in str(box["err"]), box
    monkeypatch.delenv("RABE_FAULT_GROUP_WORKER")
    h.set_tape(tape)
    blob, off = ac17.cp_encrypt_packed(h, pk, AC_POLS, [i % 3 for i in range(n)], b"".join(pts), offsets(pts), hl.HUMAN_POLICY)
    h.clear_tape()
    hosts[1].set_tape(tape)
    blob1, off1 = ac17.cp_encrypt_packed(hosts[1], pk, AC_POLS, [i % 3 for i in range(n)], b"".join(pts), offsets(pts), hl.HUMAN_POLICY)
    hosts[1].clear_tape()
    assert np.array_equal(blob, blob1) and np.array_equal(off, off1)
