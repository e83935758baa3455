"""The submission queue of the host layer (include/rabe_host.h: rabe_host_set_coalescing, rabe_*_submit, rabe_ticket_wait): the reference's
one-call-per-ciphertext API (ac17/mod.rs:274-279, :385-388; bsw/mod.rs:217, :260; lsw/mod.rs:180, :228; aw11/mod.rs:241, :298) issued by many
threads at once, collected into packed batches.  Results must be what the unqueued calls give: byte-identical on a fixed tape for a single
caller, correct and isolated (a failing request fails alone) under concurrency."""
import ctypes
import threading

import pytest

from rabe_amd import hostlib as hl

pytestmark = pytest.mark.gpu
PT = b"dance like no one's watching, encrypt like everyone is!"


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.set_coalescing(False)
    h.close()


def tape(seed, n=400):
    import random
    rnd = random.Random(seed)
    return [rnd.randrange(1, 1 << 253) for _ in range(n)]


def test_a_single_caller_on_a_tape_gets_the_bytes_of_the_unqueued_path(host):
    from rabe_amd.schemes import ac17, aw11, bsw, lsw
    host.set_coalescing(False)
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, ["A", "B", "C"])
    bpk, bmsk = bsw.setup(host)
    bsk = bsw.keygen(host, bpk, bmsk, ["A", "B", "C"])
    lpk, lmsk = lsw.setup(host)
    lsk = lsw.keygen(host, lpk, lmsk, '"A" and "B"', hl.HUMAN_POLICY)
    gk = aw11.setup(host)
    apk, amsk = aw11.authgen(host, gk, ["A", "B"])
    ask = aw11.keygen(host, gk, amsk, "alice", ["A", "B"])
    jand = '{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}'

    def all_four():
        out = []
        host.set_tape(tape(1))
        out.append(ac17.cp_encrypt(host, pk, '"A" and ("B" or "C")', PT, hl.HUMAN_POLICY))
        host.set_tape(tape(2))
        out.append(bsw.encrypt(host, bpk, '"A" and "B" and "C"', hl.HUMAN_POLICY, PT))
        host.set_tape(tape(3))
        out.append(lsw.encrypt(host, lpk, ["A", "B"], PT))
        host.set_tape(tape(4))
        out.append(aw11.encrypt(host, gk, [apk], jand, hl.JSON_POLICY, PT))
        host.clear_tape()
        return out
    plain = all_four()
    host.set_coalescing(True)
    queued = all_four()
    for a, b in zip(plain, queued):
        assert a.serialize() == b.serialize()
    # and they decrypt through the queue
    assert ac17.cp_decrypt(host, sk, queued[0]) == PT
    assert bsw.decrypt(host, bsk, queued[1]) == PT
    assert lsw.decrypt(host, lsk, queued[2]) == PT
    assert aw11.decrypt(host, gk, ask, queued[3]) == PT
    host.set_coalescing(False)


def test_many_threads_one_call_at_a_time(host):
    """32 threads, each a loop of blocking encrypt -> decrypt calls on ONE host: every plaintext comes back, a key that does not satisfy its
    policy and a policy that does not parse fail alone (their neighbours in the same batch succeed)"""
    from rabe_amd.schemes import ac17, bsw
    host.set_coalescing(False)
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, ["A", "B", "C"])
    sk_poor = ac17.cp_keygen(host, msk, ["A"])
    bpk, bmsk = bsw.setup(host)
    bsk = bsw.keygen(host, bpk, bmsk, ["A", "B", "C"])
    host.set_coalescing(True)
    pols = ['"A" and "B"', '"A" or ("B" and "C")', '"C" and ("A" or "B")']
    errors, done = [], []

    def worker(tid):
        try:
            for k in range(6):
                pt = PT + bytes([tid, k])
                if tid % 2 == 0:
                    ct = ac17.cp_encrypt(host, pk, pols[(tid + k) % 3], pt, hl.HUMAN_POLICY)
                    assert ac17.cp_decrypt(host, sk, ct) == pt
                    if k == 2:                                  # this key holds "A" only
                        ct2 = ac17.cp_encrypt(host, pk, '"A" and "B"', pt, hl.HUMAN_POLICY)
                        with pytest.raises(hl.RabeError):
                            ac17.cp_decrypt(host, sk_poor, ct2)
                    if k == 3:
                        with pytest.raises((hl.RabeError, hl.RabePanic)):
                            ac17.cp_encrypt(host, pk, '"A" and', pt, hl.HUMAN_POLICY)
                else:
                    ct = bsw.encrypt(host, bpk, pols[(tid + k) % 3], hl.HUMAN_POLICY, pt)
                    assert bsw.decrypt(host, bsk, ct) == pt
            done.append(tid)
        except BaseException as ex:          # noqa: BLE001 -- reported below, in the main thread
            errors.append((tid, repr(ex)))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(32)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    host.set_coalescing(False)
    assert not errors, errors[:3]
    assert sorted(done) == list(range(32))


def test_submit_many_then_wait(host):
    """the asynchronous form: one thread keeps hundreds of calls in flight; they run as a few packed batches"""
    from rabe_amd.schemes import ac17, aw11, lsw
    host.set_coalescing(False)
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, ["A", "B", "C"])
    lpk, lmsk = lsw.setup(host)
    lsk = lsw.keygen(host, lpk, lmsk, '"A" and "B"', hl.HUMAN_POLICY)
    gk = aw11.setup(host)
    apk, amsk = aw11.authgen(host, gk, ["A", "B"])
    ask = aw11.keygen(host, gk, amsk, "alice", ["A", "B"])
    jand = '{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}'
    host.set_coalescing(True)
    n = 150
    pts = [PT + i.to_bytes(2, "little") for i in range(n)]
    pols = ['"A" and "B"', '"A" or ("B" and "C")']
    tick = [host.submit("rabe_ac17_cp_encrypt_submit", pk.ptr, pols[i % 2].encode(), hl.HUMAN_POLICY, pts[i], ctypes.c_size_t(len(pts[i]))) for i in range(n)]
    arr, cnt = hl._strs(["A", "B"])
    ltick = [host.submit("rabe_lsw_encrypt_submit", lpk.ptr, arr, cnt, pts[i], ctypes.c_size_t(len(pts[i]))) for i in range(20)]
    pka = (ctypes.c_void_p * 1)(apk.ptr)
    atick = [host.submit("rabe_aw11_encrypt_submit", gk.ptr, pka, ctypes.c_size_t(1), jand.encode(), hl.JSON_POLICY, pts[i], ctypes.c_size_t(len(pts[i])))
             for i in range(20)]
    cts = [host.wait(t, "ac17_cp_ct") for t in tick]
    lcts = [host.wait(t, "lsw_ct") for t in ltick]
    acts = [host.wait(t, "aw11_ct") for t in atick]
    dt = [host.submit("rabe_ac17_cp_decrypt_submit", sk.ptr, c.ptr) for c in cts]
    ldt = [host.submit("rabe_lsw_decrypt_submit", lsk.ptr, c.ptr) for c in lcts]
    adt = [host.submit("rabe_aw11_decrypt_submit", gk.ptr, ask.ptr, c.ptr) for c in acts]
    assert [host.wait(t) for t in dt] == pts
    assert [host.wait(t) for t in ldt] == pts[:20]
    assert [host.wait(t) for t in adt] == pts[:20]
    host.set_coalescing(False)
