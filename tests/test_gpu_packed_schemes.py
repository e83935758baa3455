"""Packed batches of bsw / lsw / aw11 (rabe_{bsw,lsw,aw11}_*_packed): records byte-identical to serialising the objects of the
per-object API on the same tape, round trips, per-item failures, and interoperability with the object API (the packed entry points
run the device-resident Level B paths, the object API the general pairing-job path: two routes to the same bytes)."""
import numpy as np
import pytest

from rabe_amd import hostlib as hl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


def offsets(items):
    return np.concatenate([[0], np.cumsum([len(p) for p in items])]).astype(np.uint64)


BSW_POLS = ['"A" and "B" and "C"', '"A" or ("B" and "D")', '("C" or "D") and ("A" or "E") and "B"']


def test_bsw_packed_equals_object_api_and_round_trips(host):
    from rabe_amd.schemes import bsw
    pk, msk = bsw.setup(host)
    n = 9
    item_pol = [i % 3 for i in range(n)]
    pts = [b"bsw plaintext %d " % i * (i + 1) for i in range(n)]
    # same tape through both APIs: secret, msg exponent, gate coefficients, nonce per item
    tape = [1000003 * (i + 5) + 17 for i in range(40 * n)]
    host.set_tape(tape)
    objs = bsw.encrypt_batch(host, pk, [BSW_POLS[p] for p in item_pol], hl.HUMAN_POLICY, pts)
    host.set_tape(tape)
    blob, ct_off = bsw.encrypt_packed(host, pk, BSW_POLS, item_pol, b"".join(pts), offsets(pts), hl.HUMAN_POLICY)
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(ct_off[i]):int(ct_off[i + 1])].tobytes(), i
    sk = bsw.keygen(host, pk, msk, ["A", "B", "C", "D"])
    out, out_off, status = bsw.decrypt_packed(host, sk, blob, ct_off)
    assert not status.any() and out.tobytes() == b"".join(pts) and (out_off == offsets(pts)).all()
    # the packed records are ordinary ciphertexts: the object API (general pairing-job path) decrypts them too
    for i in (0, 4, 8):
        assert bsw.decrypt(host, sk, hl.Obj.deserialize("bsw_ct", blob[int(ct_off[i]):int(ct_off[i + 1])].tobytes())) == pts[i]
    # and the other way round: object-API ciphertexts in a packed decrypt
    blob2 = b"".join(o.serialize() for o in objs)
    out, _, status = bsw.decrypt_packed(host, sk, blob2, offsets([o.serialize() for o in objs]), trusted=True)
    assert not status.any() and out.tobytes() == b"".join(pts)


def test_bsw_packed_items_fail_individually(host):
    from rabe_amd.schemes import bsw
    pk, msk = bsw.setup(host)
    n = 6
    item_pol = [i % 3 for i in range(n)]
    pts = [b"item %d" % i for i in range(n)]
    blob, ct_off = bsw.encrypt_packed(host, pk, BSW_POLS, item_pol, b"".join(pts), offsets(pts), hl.HUMAN_POLICY)
    sk_ab = bsw.keygen(host, pk, msk, ["A", "B"])                      # satisfies policy 1 only ("A" or ...)
    out, out_off, status = bsw.decrypt_packed(host, sk_ab, blob, ct_off)
    assert list(status) == [-1, 0, -1, -1, 0, -1]
    assert [out[int(out_off[i]):int(out_off[i + 1])].tobytes() for i in range(n)] == [b"", pts[1], b"", b"", pts[4], b""]
    sk = bsw.keygen(host, pk, msk, ["A", "B", "C", "D", "E"])
    raw = blob.tobytes()
    bad = bytearray(raw)
    bad[int(ct_off[2]) - 3] ^= 0x40                                    # item 1's sealed bytes
    o = ct_off.copy()
    o[6] = np.uint64(len(raw) + 1000)                                  # item 5 claims bytes past the blob
    out, out_off, status = bsw.decrypt_packed(host, sk, bytes(bad), o)
    assert list(status) == [0, -1, 0, 0, 0, -1]
    # a leaf point off its curve fails the membership pass of that item only
    rec = int(ct_off[3])
    pl = int.from_bytes(raw[rec:rec + 4], "little")
    first_leaf = rec + 4 + pl + 1 + 64 + 384 + 4
    nl = int.from_bytes(raw[first_leaf:first_leaf + 4], "little")
    bad = bytearray(raw)
    bad[first_leaf + 4 + nl] ^= 1
    out, out_off, status = bsw.decrypt_packed(host, sk, bytes(bad), ct_off)
    assert list(status) == [0, 0, 0, -1, 0, 0]


LSW_POLS = ['{"name": "and", "children": [{"name": "A"}, {"name": "B"}, {"name": "C"}]}',
            '{"name": "or", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "D"}]}]}',
            '{"name": "and", "children": [{"name": "or", "children": [{"name": "C"}, {"name": "D"}]}, {"name": "B"}]}']


def test_lsw_packed_keygen_equals_object_api_and_decrypts(host):
    from rabe_amd.schemes import lsw
    pk, msk = lsw.setup(host)
    n = 9
    item_pol = [i % 3 for i in range(n)]
    pt = b"lsw: one ciphertext, a fresh key per item"
    ct = lsw.encrypt(host, pk, ["A", "B", "C"], pt)                 # satisfies policies 0 and 1, and 2 (C, B)
    tape = [1000003 * (i + 9) + 29 for i in range(20 * n)]
    host.set_tape(tape)
    objs = lsw.keygen_batch(host, pk, msk, [LSW_POLS[p] for p in item_pol], hl.JSON_POLICY)
    host.set_tape(tape)
    blob, sk_off = lsw.keygen_packed(host, pk, msk, LSW_POLS, item_pol, hl.JSON_POLICY)
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(sk_off[i]):int(sk_off[i + 1])].tobytes(), i
    out, out_off, status = lsw.decrypt_packed(host, ct, blob, sk_off)
    assert not status.any() and out.tobytes() == pt * n
    # packed keys are ordinary keys: the object API decrypts with them
    for i in (0, 4, 8):
        assert lsw.decrypt(host, hl.Obj.deserialize("lsw_sk", blob[int(sk_off[i]):int(sk_off[i + 1])].tobytes()), ct) == pt
    # a ciphertext that satisfies only some of the key policies: those items fail alone
    ct2 = lsw.encrypt(host, pk, ["A", "E"], pt)                     # policy 1 only ("A" or ...)
    out, out_off, status = lsw.decrypt_packed(host, ct2, blob, sk_off, trusted=True)
    assert list(status) == [-1, 0, -1] * 3
    # a key component off its curve: the membership pass fails that key only
    raw = bytearray(blob.tobytes())
    rec = int(sk_off[4])
    pl = int.from_bytes(raw[rec:rec + 4], "little")
    first = rec + 4 + pl + 1 + 4
    nl = int.from_bytes(raw[first:first + 4], "little")
    raw[first + 4 + nl] ^= 1
    out, out_off, status = lsw.decrypt_packed(host, ct, bytes(raw), sk_off)
    assert list(status) == [0, 0, 0, 0, -1, 0, 0, 0, 0]


def test_lsw_packed_keygen_with_negative_attributes_equals_object_api(host):
    from rabe_amd.schemes import lsw
    pk, msk = lsw.setup(host)
    pols = ['{"name": "and", "children": [{"name": "A"}, {"name": "!B"}, {"name": "or", "children": [{"name": "!C"}, {"name": "D"}]}]}', LSW_POLS[0]]
    item_pol = [0, 1, 0, 0, 1]
    tape = [1000003 * (i + 2) + 53 for i in range(60)]
    host.set_tape(tape)
    objs = lsw.keygen_batch(host, pk, msk, [pols[p] for p in item_pol], hl.JSON_POLICY)
    host.set_tape(tape)
    blob, sk_off = lsw.keygen_packed(host, pk, msk, pols, item_pol, hl.JSON_POLICY)
    host.clear_tape()
    for i in range(len(item_pol)):
        assert objs[i].serialize() == blob[int(sk_off[i]):int(sk_off[i + 1])].tobytes(), i


AW_POLS = ['{"name": "and", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "D"}, {"name": "or", "children": [{"name": "B"}, {"name": "C"}]}]}]}',
           '{"name": "or", "children": [{"name": "and", "children": [{"name": "E"}, {"name": "A"}]}, {"name": "and", "children": [{"name": "C"}, {"name": "D"}]}]}',
           '{"name": "and", "children": [{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}, {"name": "and", "children": [{"name": "C"}, {"name": "D"}]}]}']


def test_aw11_packed_equals_object_api_and_round_trips(host):
    from rabe_amd.schemes import aw11
    gk = aw11.setup(host)
    pk1, msk1 = aw11.authgen(host, gk, ["A", "B", "C"])
    pk2, msk2 = aw11.authgen(host, gk, ["D", "E"])
    n = 9
    item_pol = [i % 3 for i in range(n)]
    pts = [b"aw11 plaintext %d " % i * (i + 1) for i in range(n)]
    tape = [1000003 * (i + 3) + 41 for i in range(40 * n)]
    host.set_tape(tape)
    objs = aw11.encrypt_batch(host, gk, [pk1, pk2], [AW_POLS[p] for p in item_pol], hl.JSON_POLICY, pts)
    host.set_tape(tape)
    blob, ct_off = aw11.encrypt_packed(host, gk, [pk1, pk2], AW_POLS, item_pol, b"".join(pts), offsets(pts), hl.JSON_POLICY)
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(ct_off[i]):int(ct_off[i + 1])].tobytes(), i
    sk = aw11.keygen(host, gk, msk1, "alice", ["A", "B", "C"])
    aw11.add_to_attribute(host, gk, msk2, "D", sk)
    aw11.add_to_attribute(host, gk, msk2, "E", sk)
    out, out_off, status = aw11.decrypt_packed(host, gk, sk, blob, ct_off)
    assert not status.any() and out.tobytes() == b"".join(pts) and (out_off == offsets(pts)).all()
    for i in (1, 5):
        assert aw11.decrypt(host, gk, sk, hl.Obj.deserialize("aw11_ct", blob[int(ct_off[i]):int(ct_off[i + 1])].tobytes())) == pts[i]
    # a key that satisfies only policy 1's second branch (C and D)
    bob = aw11.keygen(host, gk, msk1, "bob", ["C"])
    aw11.add_to_attribute(host, gk, msk2, "D", bob)
    out, out_off, status = aw11.decrypt_packed(host, gk, bob, blob, ct_off, trusted=True)
    assert list(status) == [-1, 0, -1] * 3
    assert [out[int(out_off[i]):int(out_off[i + 1])].tobytes() for i in (1, 4, 7)] == [pts[1], pts[4], pts[7]]
    # tampered c_0 (an arbitrary Fq12 value is outside the order-r subgroup): that item only
    raw = bytearray(blob.tobytes())
    rec = int(ct_off[2])
    pl = int.from_bytes(raw[rec:rec + 4], "little")
    c0 = rec + 4 + pl + 1
    raw[c0:c0 + 384] = b"".join((11 + i).to_bytes(32, "little") for i in range(12))
    out, out_off, status = aw11.decrypt_packed(host, gk, sk, bytes(raw), ct_off)
    assert list(status) == [0, 0, -1, 0, 0, 0, 0, 0, 0]


def test_bsw_packed_keygen_equals_object_keygen_on_the_same_tape(host):
    """rabe_bsw_keygen_packed: n keys in one call (three window-table launches: d = g2_alpha/beta + g2*(r/beta), g1*r_j, g2*(r + h(j) r_j)) =
    n calls of bsw::keygen on the same randomness, byte for byte; the keys decrypt; an empty attribute list fails the call."""
    from rabe_amd.schemes import bsw
    pk, msk = bsw.setup(host)
    sets = [["A", "B"], ["A", "B", "C", "D"], ["C"]]
    item_set = [0, 1, 2, 1, 0, 1, 2, 0, 1]
    n = len(item_set)
    tape = [1000003 * (i + 13) + 23 for i in range(6 * n)]                  # r, then r_j per attribute, per item
    host.set_tape(tape)
    objs = [bsw.keygen(host, pk, msk, sets[s]) for s in item_set]
    host.set_tape(tape)
    blob, off = bsw.keygen_packed(host, pk, msk, sets, item_set)
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(off[i]):int(off[i + 1])].tobytes(), i
    ct = bsw.encrypt(host, pk, '"A" and ("C" or "D")', hl.HUMAN_POLICY, b"bulk bsw keys")
    for i in (1, 3):
        assert bsw.decrypt(host, hl.Obj.deserialize("bsw_sk", blob[int(off[i]):int(off[i + 1])].tobytes()), ct) == b"bulk bsw keys"
    with pytest.raises(hl.RabeError):
        bsw.keygen_packed(host, pk, msk, [["A"], []], [0, 1])
    attrs = ["a%d" % i for i in range(100)]
    blob, off = bsw.keygen_packed(host, pk, msk, [attrs, attrs[:40]], np.arange(1500, dtype=np.uint32) % 2)
    ct = bsw.encrypt(host, pk, " and ".join('"a%d"' % i for i in (1, 7, 13, 19, 39)), hl.HUMAN_POLICY, b"x" * 40)
    for i in (0, 1, 1498, 1499):
        assert bsw.decrypt(host, hl.Obj.deserialize("bsw_sk", blob[int(off[i]):int(off[i + 1])].tobytes()), ct) == b"x" * 40


def test_aw11_packed_keygen_equals_object_keygen(host):
    """rabe_aw11_keygen_packed: n users' keys from one authority in one call (K_x = g1 * (alpha_x + h(gid) y_x): one window-table launch) =
    n calls of aw11::keygen, byte for byte (no randomness involved); the keys decrypt; an attribute of another authority fails the call."""
    from rabe_amd.schemes import aw11
    gk = aw11.setup(host)
    pk1, msk1 = aw11.authgen(host, gk, ["A", "B", "C", "D"])
    sets = [["A", "B"], ["A", "B", "C", "D"], ["C"]]
    item_set = [0, 1, 2, 1, 0, 1]
    gids = ["user-%d" % i for i in range(len(item_set))]
    objs = [aw11.keygen(host, gk, msk1, gids[i], sets[s]) for i, s in enumerate(item_set)]
    blob, off = aw11.keygen_packed(host, gk, msk1, gids, sets, item_set)
    for i in range(len(item_set)):
        assert objs[i].serialize() == blob[int(off[i]):int(off[i + 1])].tobytes(), i
    ct = aw11.encrypt(host, gk, [pk1], '{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "C"}, {"name": "D"}]}]}',
                      hl.JSON_POLICY, b"bulk aw11 keys")
    for i in (1, 3):
        assert aw11.decrypt(host, gk, hl.Obj.deserialize("aw11_sk", blob[int(off[i]):int(off[i + 1])].tobytes()), ct) == b"bulk aw11 keys"
    with pytest.raises(hl.RabePanic):
        aw11.keygen_packed(host, gk, msk1, ["x"], [["E"]], [0])            # not this authority's attribute: the reference's unwrap panic (:213)
    with pytest.raises(hl.RabeError):
        aw11.keygen_packed(host, gk, msk1, [""], [["A"]], [0])
    n = 3000
    blob, off = aw11.keygen_packed(host, gk, msk1, ["u%05d" % i for i in range(n)], sets, np.arange(n, dtype=np.uint32) % 3)
    k = aw11.keygen(host, gk, msk1, "u02998", sets[2998 % 3])
    assert k.serialize() == blob[int(off[2998]):int(off[2999])].tobytes()


def test_lsw_packed_encrypt_equals_object_encrypt_on_the_same_tape(host):
    """rabe_lsw_encrypt_packed: n ciphertexts in one call (window-table launches over the public key's elements, the reference's sx[0] quirk
    included) = n calls of lsw::encrypt on the same randomness, byte for byte; keys decrypt them, packed and object-wise."""
    from rabe_amd.schemes import lsw
    pk, msk = lsw.setup(host)
    sets = [["A", "B"], ["A", "B", "C", "D"], ["C"]]
    item_set = [0, 1, 2, 1, 0, 1, 2]
    n = len(item_set)
    pts = [b"lsw packed plaintext %d " % i * (i % 3 + 1) for i in range(n)]
    tape = [1000003 * (i + 17) + 31 for i in range(8 * n)]                  # secret, sx per attribute, msg exponent, nonce per item
    host.set_tape(tape)
    objs = [lsw.encrypt(host, pk, sets[s], pts[i]) for i, s in enumerate(item_set)]
    host.set_tape(tape)
    blob, off = lsw.encrypt_packed(host, pk, sets, item_set, b"".join(pts), offsets(pts))
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(off[i]):int(off[i + 1])].tobytes(), i
    sk = lsw.keygen(host, pk, msk, '{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}', hl.JSON_POLICY)
    for i in (0, 1, 3):
        assert lsw.decrypt(host, sk, hl.Obj.deserialize("lsw_ct", blob[int(off[i]):int(off[i + 1])].tobytes())) == pts[i]
    with pytest.raises(hl.RabeError):
        lsw.encrypt_packed(host, pk, [["A"], []], [0, 1], b"xy", [0, 1, 2])             # `attributes or data empty` (:185)
    with pytest.raises(hl.RabeError):
        lsw.encrypt_packed(host, pk, [["A"]], [0, 0], b"x", [0, 1, 1])                  # an empty plaintext
    # a bulk run on OS randomness: every ciphertext opens with a key whose policy its attributes satisfy
    attrs = ["a%d" % i for i in range(60)]
    m = 1200
    big = [b"p%04d" % i for i in range(m)]
    blob, off = lsw.encrypt_packed(host, pk, [attrs, attrs[:30]], np.arange(m, dtype=np.uint32) % 2, b"".join(big), offsets(big))
    sk = lsw.keygen(host, pk, msk, '{"name": "and", "children": [{"name": "a3"}, {"name": "or", "children": [{"name": "a29"}, {"name": "zz"}]}]}', hl.JSON_POLICY)
    for i in (0, 1, m - 2, m - 1):
        assert lsw.decrypt(host, sk, hl.Obj.deserialize("lsw_ct", blob[int(off[i]):int(off[i + 1])].tobytes())) == big[i]


def test_ac17_kp_packed_keygen_equals_object_keygen_on_the_same_tape(host):
    """rabe_ac17_kp_keygen_packed: n calls of ac17::kp_keygen (ac17/mod.rs:439-547; the `_temp` accumulation over the columns :496 included)
    as Fr work on the host cores + one fixed-base launch set, records written on the device = the objects' bytes on the same tape; the keys
    decrypt KP ciphertexts; policies with and without a +/-1 in the first MSP column (the +/- g_k additions), one-column policies."""
    from rabe_amd.schemes import ac17
    pk, msk = ac17.setup(host)
    pols = ['"A" and ("B" or "C")', '"A" or "B"', '"A"', '("A" and "B") and ("C" and ("D" or "E"))']
    item_pol = [0, 1, 2, 3, 3, 0, 2, 1, 0]
    n = len(item_pol)
    tape = [1000003 * (i + 5) + 11 for i in range(16 * n)]
    host.set_tape(tape)
    objs = [ac17.kp_keygen(host, msk, pols[p], hl.HUMAN_POLICY) for p in item_pol]
    host.set_tape(tape)
    blob, off = ac17.kp_keygen_packed(host, msk, pols, item_pol, hl.HUMAN_POLICY)
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(off[i]):int(off[i + 1])].tobytes(), i
    ct = ac17.kp_encrypt(host, pk, ["A", "B", "C", "D"], b"bulk kp keys")
    for i in range(n):
        sk = hl.Obj.deserialize("ac17_kp_sk", blob[int(off[i]):int(off[i + 1])].tobytes())
        assert ac17.kp_decrypt(host, sk, ct) == b"bulk kp keys", i
    # a larger batch on OS randomness: every key decrypts
    attrs = ["a%d" % i for i in range(40)]
    big = [" and ".join('"%s"' % a for a in attrs[:k]) if k < 3 else '("%s" and "%s") and ("%s" or "%s")' % tuple(attrs[k:k + 4]) for k in (1, 2, 5, 9)]
    blob, off = ac17.kp_keygen_packed(host, msk, big, np.arange(1200, dtype=np.uint32) % 4, hl.HUMAN_POLICY)
    ct = ac17.kp_encrypt(host, pk, attrs, b"y" * 33)
    for i in (0, 1, 2, 3, 1198, 1199):
        assert ac17.kp_decrypt(host, hl.Obj.deserialize("ac17_kp_sk", blob[int(off[i]):int(off[i + 1])].tobytes()), ct) == b"y" * 33
    with pytest.raises((hl.RabeError, hl.RabePanic)):
        ac17.kp_keygen_packed(host, msk, ['"A" and'], [0], hl.HUMAN_POLICY)


def test_bsw_packed_delegate_equals_object_delegate_on_the_same_tape(host):
    """rabe_bsw_delegate_packed: n calls of bsw::delegate (bsw/mod.rs:162-206) on one key = the objects' bytes on the same tape; the delegated
    keys decrypt what their subset satisfies and nothing else; a subset outside the key fails the call (delegate returns None)."""
    from rabe_amd.schemes import bsw
    pk, msk = bsw.setup(host)
    sk = bsw.keygen(host, pk, msk, ["A", "B", "C", "D", "E"])
    subsets = [["A", "B"], ["C"], ["E", "A", "D"], ["A", "B", "C", "D", "E"]]
    item = [0, 1, 2, 3, 2, 0, 1]
    n = len(item)
    tape = [1000003 * (i + 29) + 7 for i in range(7 * n)]
    host.set_tape(tape)
    objs = [bsw.delegate(host, pk, sk, subsets[s]) for s in item]
    host.set_tape(tape)
    blob, off = bsw.delegate_packed(host, pk, sk, subsets, item)
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(off[i]):int(off[i + 1])].tobytes(), i
    ct_ab = bsw.encrypt(host, pk, '"A" and "B"', hl.HUMAN_POLICY, b"delegated")
    ct_c = bsw.encrypt(host, pk, '"C"', hl.HUMAN_POLICY, b"delegated c")
    keys = [hl.Obj.deserialize("bsw_sk", blob[int(off[i]):int(off[i + 1])].tobytes()) for i in range(n)]
    assert bsw.decrypt(host, keys[0], ct_ab) == b"delegated" and bsw.decrypt(host, keys[3], ct_ab) == b"delegated"
    assert bsw.decrypt(host, keys[1], ct_c) == b"delegated c"
    with pytest.raises(hl.RabeError):
        bsw.decrypt(host, keys[1], ct_ab)
    with pytest.raises(hl.RabeError):
        bsw.delegate_packed(host, pk, sk, [["A", "Z"]], [0])
    with pytest.raises(hl.RabeError):
        bsw.delegate_packed(host, pk, sk, [[]], [0])
    blob, off = bsw.delegate_packed(host, pk, sk, subsets, np.arange(2000, dtype=np.uint32) % 4)        # OS randomness, more than one launch tile
    for i in (0, 3, 1996, 1999):
        k = hl.Obj.deserialize("bsw_sk", blob[int(off[i]):int(off[i + 1])].tobytes())
        assert bsw.decrypt(host, k, ct_ab) == b"delegated"
