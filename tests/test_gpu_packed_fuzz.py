"""Untrusted bytes into the packed decrypt / transform entry points: random corruption of records (bit flips, truncation through the offsets,
length fields overwritten) must never take the process down or disturb the OTHER items of the batch -- an item either comes back with
status 0 and its exact plaintext, or with status -1.  (The entry points validate offsets against the blob, every length field against
its record, and -- in checked mode -- every decoded element's membership.)"""
import random

import numpy as np
import pytest

from rabe_amd import hostlib as hl

pytestmark = pytest.mark.gpu
PT = b"dance like no one's watching, encrypt like everyone is!"


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


def offsets(items):
    return np.concatenate([[0], np.cumsum([len(p) for p in items])]).astype(np.uint64)


def corrupt(rnd, recs, victims):
    """damage the victims' records in place (lengths unchanged) or cut them short; returns (blob, offsets)"""
    out = []
    for i, r in enumerate(recs):
        b = bytearray(r)
        if i in victims:
            mode = rnd.randrange(4)
            if mode == 0:                                  # a few bit flips anywhere
                for _ in range(rnd.randrange(1, 6)):
                    b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
            elif mode == 1:                                # a length field (first 64 bytes hold the policy length; later ones appear at random)
                pos = rnd.choice([0, rnd.randrange(0, max(1, len(b) - 4))])
                b[pos:pos + 4] = rnd.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", (len(b) * 2).to_bytes(4, "little")])
            elif mode == 2:                                # truncated record
                b = b[:rnd.randrange(0, len(b))]
            else:                                          # random bytes over a stretch (elements become non-members / non-canonical)
                p0 = rnd.randrange(len(b))
                for k in range(p0, min(len(b), p0 + rnd.randrange(1, 200))):
                    b[k] = rnd.randrange(256)
        out.append(bytes(b))
    return b"".join(out), offsets(out)


def check(status, plains, want, victims):
    for i in range(len(want)):
        if i in victims:
            assert status[i] in (0, -1)
            if status[i] == 0:
                assert plains[i] == want[i]               # damage that left the plaintext path intact (e.g. an unused row's name)
        else:
            assert status[i] == 0 and plains[i] == want[i], i


def split(out, off, n):
    return [out[int(off[i]):int(off[i + 1])].tobytes() for i in range(n)]


@pytest.mark.parametrize("trusted", [False, True], ids=["checked", "trusted"])
def test_corrupted_records_fail_alone(host, trusted):
    from rabe_amd.schemes import ac17, aw11, bsw, ghw11, lsw
    rnd = random.Random(20250929 + int(trusted))
    n = 12
    pts = [PT + bytes([i]) for i in range(n)]
    # ---- AC17
    pk, msk = ac17.setup(host)
    pols = ['"A" and "B"', '"A" or ("B" and "C")']
    blob, off = ac17.cp_encrypt_packed(host, pk, pols, [i % 2 for i in range(n)], b"".join(pts), offsets(pts), hl.HUMAN_POLICY)
    recs = split(blob, off, n)
    sk = ac17.cp_keygen(host, msk, ["A", "B", "C"])
    for _ in range(12):
        victims = set(rnd.sample(range(n), 4))
        b, o = corrupt(rnd, recs, victims)
        out, oo, st = ac17.cp_decrypt_packed(host, sk, b, o, trusted=trusted)
        check(st, split(out, oo, n), pts, victims)
    # ---- AC17 KP (the same core with the record head and the roles swapped)
    blob, off = ac17.kp_encrypt_packed(host, pk, [["A", "B"], ["A", "C", "D"]], [i % 2 for i in range(n)], b"".join(pts), offsets(pts))
    recs = split(blob, off, n)
    ksk = ac17.kp_keygen(host, msk, '"A" and ("B" or "C")', hl.HUMAN_POLICY)
    for _ in range(12):
        victims = set(rnd.sample(range(n), 4))
        b, o = corrupt(rnd, recs, victims)
        out, oo, st = ac17.kp_decrypt_packed(host, ksk, b, o, trusted=trusted)
        check(st, split(out, oo, n), pts, victims)
    # ---- BSW
    pk, msk = bsw.setup(host)
    blob, off = bsw.encrypt_packed(host, pk, ['"A" and "B" and "C"', '"A" or ("B" and "D")'], [i % 2 for i in range(n)], b"".join(pts), offsets(pts),
                                   hl.HUMAN_POLICY)
    recs = split(blob, off, n)
    sk = bsw.keygen(host, pk, msk, ["A", "B", "C", "D"])
    for _ in range(12):
        victims = set(rnd.sample(range(n), 4))
        b, o = corrupt(rnd, recs, victims)
        out, oo, st = bsw.decrypt_packed(host, sk, b, o, trusted=trusted)
        check(st, split(out, oo, n), pts, victims)
    # ---- LSW (n keys, one ciphertext)
    pk, msk = lsw.setup(host)
    lpols = ['{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}', '{"name": "or", "children": [{"name": "A"}, {"name": "C"}]}']
    blob, off = lsw.keygen_packed(host, pk, msk, lpols, [i % 2 for i in range(n)], hl.JSON_POLICY)
    recs = split(blob, off, n)
    ct = lsw.encrypt(host, pk, ["A", "B"], PT)
    for _ in range(12):
        victims = set(rnd.sample(range(n), 4))
        b, o = corrupt(rnd, recs, victims)
        out, oo, st = lsw.decrypt_packed(host, ct, b, o, trusted=trusted)
        check(st, split(out, oo, n), [PT] * n, victims)
    # ---- AW11
    gk = aw11.setup(host)
    pk1, msk1 = aw11.authgen(host, gk, ["A", "B"])
    pk2, msk2 = aw11.authgen(host, gk, ["C"])
    apols = ['{"name": "and", "children": [{"name": "A"}, {"name": "C"}]}', '{"name": "or", "children": [{"name": "B"}, {"name": "C"}]}']
    blob, off = aw11.encrypt_packed(host, gk, [pk1, pk2], apols, [i % 2 for i in range(n)], b"".join(pts), offsets(pts), hl.JSON_POLICY)
    recs = split(blob, off, n)
    sk = aw11.keygen(host, gk, msk1, "alice", ["A", "B"])
    aw11.add_to_attribute(host, gk, msk2, "C", sk)
    for _ in range(12):
        victims = set(rnd.sample(range(n), 4))
        b, o = corrupt(rnd, recs, victims)
        out, oo, st = aw11.decrypt_packed(host, gk, sk, b, o, trusted=trusted)
        check(st, split(out, oo, n), pts, victims)
    # ---- GHW11 transform
    pk, msk = ghw11.setup(host)
    tk, rk = ghw11.tkgen(host, ghw11.keygen(host, pk, msk, ["A", "B", "C"]))
    cts = [ghw11.encrypt(host, pk, lpols[i % 2], hl.JSON_POLICY, pts[i]) for i in range(n)]
    recs = [c.serialize() for c in cts]
    good = [ghw11.transform(host, c, tk).serialize() for c in cts]
    for _ in range(12):
        victims = set(rnd.sample(range(n), 4))
        b, o = corrupt(rnd, recs, victims)
        out, st = ghw11.transform_packed(host, tk, b, o, trusted=trusted)
        for i in range(n):
            if i not in victims:
                assert st[i] == 0 and out[i].tobytes() == good[i], i
            else:
                assert st[i] in (0, -1)


# ---------------------------------------------------------------- targeted damage (errors that are not RabeError)
def _head(rec):
    plen = int.from_bytes(rec[:4], "little")
    return plen, 4 + plen + 1                             # policy length, offset of what follows (policy text, language byte)


def rename_first_row(rec, fixed):
    """the name of the first row (it follows `fixed` bytes of elements and the row count) gets another first letter: the record stays
    well-formed, but the name the policy asks for is gone -- the reference's `.unwrap()` on a `None` (a panic there, -1 here)"""
    b = bytearray(rec)
    _plen, o = _head(rec)
    o += fixed + 4
    nlen = int.from_bytes(b[o:o + 4], "little")
    assert 0 < nlen < 32
    b[o + 4] = ord("Z")
    return bytes(b)


def with_policy(rec, text, lang_byte=None):
    """the same record under another policy text"""
    plen, o = _head(rec)
    raw = text.encode()
    lang = rec[4 + plen:4 + plen + 1] if lang_byte is None else bytes([lang_byte])
    return len(raw).to_bytes(4, "little") + raw + lang + rec[o:]


PANIC_JSON = ['{"name": "and", "children": [{"name": "A"}]}',          # a gate with a single child (secretsharing/mod.rs:167,187)
              '{"name": "or", "children": [{"name": 7}, {"name": "A"}]}',   # a numeric leaf (pest/json.rs:14-17: unwrap on an atomic rule)
              '{"name": "and"']                                          # not a policy at all


@pytest.mark.parametrize("trusted", [False, True], ids=["checked", "trusted"])
def test_a_renamed_row_or_a_panicking_policy_fails_alone(host, trusted):
    from rabe_amd.schemes import aw11, bsw, lsw
    n = 20                                                   # > 16: the host's parallel_for takes its threaded path
    pts = [PT + bytes([i]) for i in range(n)]
    victims = {3: "rename", 7: 0, 11: 1, 17: 2, 19: "rename"}

    def damage(recs, fixed):
        out = list(recs)
        for i, how in victims.items():
            out[i] = rename_first_row(recs[i], fixed) if how == "rename" else with_policy(recs[i], PANIC_JSON[how], 0)
        return b"".join(out), offsets(out)

    def expect(st, plains, want):
        for i in range(n):
            if i in victims:
                assert st[i] == -1, (i, victims[i], st[i])
            else:
                assert st[i] == 0 and plains[i] == want[i], i

    jpols = ['{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}', '{"name": "and", "children": [{"name": "A"}, {"name": "C"}]}']
    # ---- BSW ciphertext: policy, language, c (64), c_p (384), rows
    pk, msk = bsw.setup(host)
    blob, off = bsw.encrypt_packed(host, pk, jpols, [i % 2 for i in range(n)], b"".join(pts), offsets(pts), hl.JSON_POLICY)
    sk = bsw.keygen(host, pk, msk, ["A", "B", "C"])
    b, o = damage(split(blob, off, n), 64 + 384)
    out, oo, st = bsw.decrypt_packed(host, sk, b, o, trusted=trusted)
    expect(st, split(out, oo, n), pts)
    # ---- LSW key record: policy, language, rows
    pk, msk = lsw.setup(host)
    blob, off = lsw.keygen_packed(host, pk, msk, jpols, [i % 2 for i in range(n)], hl.JSON_POLICY)
    ct = lsw.encrypt(host, pk, ["A", "B", "C"], PT)
    b, o = damage(split(blob, off, n), 0)
    out, oo, st = lsw.decrypt_packed(host, ct, b, o, trusted=trusted)
    expect(st, split(out, oo, n), [PT] * n)
    # ---- AW11 ciphertext: policy, language, c_0 (384), rows
    gk = aw11.setup(host)
    pk1, msk1 = aw11.authgen(host, gk, ["A", "B"])
    pk2, msk2 = aw11.authgen(host, gk, ["C"])
    blob, off = aw11.encrypt_packed(host, gk, [pk1, pk2], jpols, [i % 2 for i in range(n)], b"".join(pts), offsets(pts), hl.JSON_POLICY)
    sk = aw11.keygen(host, gk, msk1, "alice", ["A", "B"])
    aw11.add_to_attribute(host, gk, msk2, "C", sk)
    b, o = damage(split(blob, off, n), 384)
    out, oo, st = aw11.decrypt_packed(host, gk, sk, b, o, trusted=trusted)
    expect(st, split(out, oo, n), pts)
