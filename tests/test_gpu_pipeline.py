"""Pipelined packed batches (rabe_amd/csrc/host/pipeline.cpp): a packed call cut into chunks that run a few at a time on their own engine
lanes returns exactly what the unchunked call returns -- records byte for byte on the same randomness tape (the chunks draw in item
order), plaintexts, offsets and per-item statuses with failures of every kind spread over the chunks -- for all four schemes; plus a
larger concurrent run on OS randomness."""
import os

import numpy as np
import pytest

from rabe_amd import hostlib as hl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


class cut:
    """RABE_PACKED_CHUNK / RABE_PACKED_LANES for the calls inside (read per call)"""

    def __init__(self, chunk, lanes):
        self.env = {"RABE_PACKED_CHUNK": str(chunk), "RABE_PACKED_LANES": str(lanes)}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def offsets(items):
    return np.concatenate([[0], np.cumsum([len(p) for p in items])]).astype(np.uint64)


def both(fn):
    """fn() unchunked and in chunks of >= 2 items on 3 lanes"""
    with cut(1 << 30, 1):
        a = fn()
    with cut(2, 3):
        b = fn()
    return a, b


def same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))


N = 11
PTS = [b"pipelined item %d " % i * (i % 4 + 1) for i in range(N)]
TAPE = [1000003 * (i + 5) + 17 for i in range(60 * N)]


def damage(blob, off):
    """(blob, offsets) with a tampered sealed part (item 1) and non-monotone offsets (item 9; item 10 then starts inside item 9's
    record): failures in different chunks"""
    raw = bytearray(np.asarray(blob).tobytes())
    raw[int(off[2]) - 3] ^= 0x40
    o = np.array(off, dtype=np.uint64).copy()
    o[10] = np.uint64(int(o[9]) - 8)
    return bytes(raw), o, len(raw)


def test_ac17_chunks_equal_the_unchunked_call(host):
    from rabe_amd.schemes import ac17
    pols = ['"A" and "B"', '"A" or ("B" and "C")', '"C" and ("A" or "D")']
    pk, msk = ac17.setup(host)
    item_pol = [i % 3 for i in range(N)]

    def enc():
        host.set_tape(TAPE)
        try:
            return ac17.cp_encrypt_packed(host, pk, pols, item_pol, b"".join(PTS), offsets(PTS), hl.HUMAN_POLICY)
        finally:
            host.clear_tape()
    a, b = both(enc)
    same(a, b)
    blob, off = a
    sk = ac17.cp_keygen(host, msk, ["A", "B"])                     # does not satisfy policy 2
    raw, o, _ = damage(blob, off)
    for trusted in (False, True):
        a, b = both(lambda: ac17.cp_decrypt_packed(host, sk, raw, o, trusted=trusted))
        same(a, b)
    out, out_off, status = b
    assert list(status[:10]) == [0, -1, -1, 0, 0, -1, 0, 0, -1, -1]
    assert out[int(out_off[7]):int(out_off[8])].tobytes() == PTS[7] and out[int(out_off[3]):int(out_off[4])].tobytes() == PTS[3]


def test_bsw_chunks_equal_the_unchunked_call(host):
    from rabe_amd.schemes import bsw
    pols = ['"A" and "B" and "C"', '"A" or ("B" and "D")', '("C" or "D") and ("A" or "E") and "B"']
    pk, msk = bsw.setup(host)
    item_pol = [i % 3 for i in range(N)]

    def enc():
        host.set_tape(TAPE)
        try:
            return bsw.encrypt_packed(host, pk, pols, item_pol, b"".join(PTS), offsets(PTS), hl.HUMAN_POLICY)
        finally:
            host.clear_tape()
    a, b = both(enc)
    same(a, b)
    blob, off = a
    sk = bsw.keygen(host, pk, msk, ["A", "B"])                     # policy 1 only
    raw, o, _ = damage(blob, off)
    a, b = both(lambda: bsw.decrypt_packed(host, sk, raw, o))
    same(a, b)
    assert list(b[2][:10]) == [-1, -1, -1, -1, 0, -1, -1, 0, -1, -1]
    sk = bsw.keygen(host, pk, msk, ["A", "B", "C", "D", "E"])
    a, b = both(lambda: bsw.decrypt_packed(host, sk, blob, off, trusted=True))
    same(a, b)
    assert not b[2].any() and b[0].tobytes() == b"".join(PTS)


def test_lsw_chunks_equal_the_unchunked_call(host):
    from rabe_amd.schemes import lsw
    pols = ['{"name": "and", "children": [{"name": "A"}, {"name": "!B"}, {"name": "or", "children": [{"name": "!C"}, {"name": "D"}]}]}',
            '{"name": "or", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "D"}]}]}',
            '{"name": "and", "children": [{"name": "or", "children": [{"name": "C"}, {"name": "D"}]}, {"name": "B"}]}']
    pk, msk = lsw.setup(host)
    item_pol = [i % 3 for i in range(N)]

    def gen():
        host.set_tape(TAPE)
        try:
            return lsw.keygen_packed(host, pk, msk, pols, item_pol, hl.JSON_POLICY)
        finally:
            host.clear_tape()
    a, b = both(gen)
    same(a, b)
    blob, off = a
    pt = b"one ciphertext, a key per item"
    ct = lsw.encrypt(host, pk, ["A", "D"], pt)                     # policies 0 (A, not B, D) and 1 (A); not 2 (needs B)
    raw = bytearray(blob.tobytes())
    o = off.copy()
    o[10] = np.uint64(int(o[9]) - 8)
    a, b = both(lambda: lsw.decrypt_packed(host, ct, bytes(raw), o))
    same(a, b)
    st = list(b[2])
    assert st[1] == 0 and st[4] == 0 and st[7] == 0 and st[2] == -1 and st[9] == -1, st        # policy 1 (A) decrypts, policy 2 (needs B) does not
    assert b[0].tobytes() == pt * st.count(0)


def test_aw11_chunks_equal_the_unchunked_call(host):
    from rabe_amd.schemes import aw11
    pols = ['{"name": "and", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "D"}, {"name": "or", "children": [{"name": "B"}, {"name": "C"}]}]}]}',
            '{"name": "or", "children": [{"name": "and", "children": [{"name": "E"}, {"name": "A"}]}, {"name": "and", "children": [{"name": "C"}, {"name": "D"}]}]}',
            '{"name": "and", "children": [{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}, {"name": "and", "children": [{"name": "C"}, {"name": "D"}]}]}']
    gk = aw11.setup(host)
    pk1, msk1 = aw11.authgen(host, gk, ["A", "B", "C"])
    pk2, msk2 = aw11.authgen(host, gk, ["D", "E"])
    item_pol = [i % 3 for i in range(N)]

    def enc():
        host.set_tape(TAPE)
        try:
            return aw11.encrypt_packed(host, gk, [pk1, pk2], pols, item_pol, b"".join(PTS), offsets(PTS), hl.JSON_POLICY)
        finally:
            host.clear_tape()
    a, b = both(enc)
    same(a, b)
    blob, off = a
    bob = aw11.keygen(host, gk, msk1, "bob", ["C"])                # policy 1's second branch only
    aw11.add_to_attribute(host, gk, msk2, "D", bob)
    raw, o, _ = damage(blob, off)
    a, b = both(lambda: aw11.decrypt_packed(host, gk, bob, raw, o))
    same(a, b)
    assert list(b[2][:10]) == [-1, -1, -1, -1, 0, -1, -1, 0, -1, -1]


def test_many_chunks_in_flight_round_trip_on_os_randomness(host):
    """3000 items in chunks of >= 250 on 4 lanes: every lane runs several chunks back to back (buffers and arenas are reused)"""
    from rabe_amd.schemes import ac17
    pols = ['"A" and "B"', '"A" or ("B" and "C")', '("C" and "A") or "D"']
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, ["A", "B", "C"])
    n = 3000
    pts = [b"item %05d" % i * (i % 3 + 1) for i in range(n)]
    item_pol = np.arange(n, dtype=np.uint32) % 3
    with cut(250, 4):
        for _ in range(2):
            blob, off = ac17.cp_encrypt_packed(host, pk, pols, item_pol, b"".join(pts), offsets(pts), hl.HUMAN_POLICY)
            out, out_off, status = ac17.cp_decrypt_packed(host, sk, blob, off)
            assert not status.any() and out.tobytes() == b"".join(pts) and (out_off == offsets(pts)).all()
    # records written by the chunks sit exactly where the offsets say: an unchunked decrypt reads them
    with cut(1 << 30, 1):
        out, out_off, status = ac17.cp_decrypt_packed(host, sk, blob, off, trusted=True)
    assert not status.any() and out.tobytes() == b"".join(pts)
