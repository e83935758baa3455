"""Packed AC17 batches (rabe_ac17_cp_{encrypt,decrypt}_packed): the records are byte-identical to serialising the objects the
per-object batch API returns on the same tape; decrypting them returns the plaintexts; a key that does not satisfy one policy,
a truncated record and a tampered record fail their own item only."""
import numpy as np
import pytest

from rabe_amd import hostlib as hl
from rabe_amd.schemes import ac17

pytestmark = pytest.mark.gpu
POLS = ['"A" and "B"', '"A" or ("B" and "C")', '"C" and ("A" or "D")']


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


def test_packed_records_equal_object_serialisation_on_the_same_tape(host):
    pk, msk = ac17.setup(host)
    n = 9
    item_pol = [i % 3 for i in range(n)]
    pts = [b"plaintext-%d" % i * (i + 1) for i in range(n)]
    tape = [1000003 * (i + 7) + 11 for i in range(4 * n)]                    # s0, s1, msg exponent, nonce per item
    host.set_tape(tape)
    objs = ac17.cp_encrypt_batch(host, pk, [POLS[p] for p in item_pol], pts, hl.HUMAN_POLICY)
    host.set_tape(tape)
    off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
    blob, ct_off = ac17.cp_encrypt_packed(host, pk, POLS, item_pol, b"".join(pts), off, hl.HUMAN_POLICY)
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(ct_off[i]):int(ct_off[i + 1])].tobytes(), i
    sk = ac17.cp_keygen(host, msk, ["A", "B", "C"])
    out, out_off, status = ac17.cp_decrypt_packed(host, sk, blob, ct_off)
    assert not status.any() and out.tobytes() == b"".join(pts) and (out_off == off).all()
    # the packed records are ordinary ciphertexts: the object API decrypts them too
    for i in (0, 4):
        assert ac17.cp_decrypt(host, sk, hl.Obj.deserialize("ac17_cp_ct", blob[int(ct_off[i]):int(ct_off[i + 1])].tobytes())) == pts[i]


def test_packed_decrypt_fails_items_individually(host):
    pk, msk = ac17.setup(host)
    n = 6
    item_pol = [i % 3 for i in range(n)]
    pts = [b"item %d" % i for i in range(n)]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
    blob, ct_off = ac17.cp_encrypt_packed(host, pk, POLS, item_pol, b"".join(pts), off, hl.HUMAN_POLICY)
    sk_ab = ac17.cp_keygen(host, msk, ["A", "B"])                          # does not satisfy policy 2 ("C" and ...)
    out, out_off, status = ac17.cp_decrypt_packed(host, sk_ab, blob, ct_off)
    assert list(status) == [0, 0, -1, 0, 0, -1]
    assert [out[int(out_off[i]):int(out_off[i + 1])].tobytes() for i in range(n)] == [pts[0], pts[1], b"", pts[3], pts[4], b""]
    # tamper with item 1's sealed bytes, truncate item 3's record (by lying about its extent)
    blob = blob.tobytes()
    bad = bytearray(blob)
    bad[int(ct_off[2]) - 3] ^= 0x40
    off2 = ct_off.copy()
    out, out_off, status = ac17.cp_decrypt_packed(host, sk_ab, bytes(bad), off2)
    assert list(status) == [0, -1, -1, 0, 0, -1]
    cut = blob[:int(ct_off[3])] + blob[int(ct_off[3]):int(ct_off[4]) - 40] + blob[int(ct_off[4]):]
    off3 = ct_off.copy()
    off3[4:] -= 40
    out, out_off, status = ac17.cp_decrypt_packed(host, sk_ab, cut, off3)
    assert list(status) == [0, 0, -1, -1, 0, -1]


def test_packed_decrypt_validates_offsets_and_membership(host):
    """the packed decrypt is the entry point for external data -- offsets are checked against the blob length before
    anything is read, and decoded elements go through the batched membership pass (coordinate < p, curve, subgroup)."""
    pk, msk = ac17.setup(host)
    n = 5
    pts = [b"item %d" % i for i in range(n)]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
    blob, ct_off = ac17.cp_encrypt_packed(host, pk, POLS, [0] * n, b"".join(pts), off, hl.HUMAN_POLICY)
    sk = ac17.cp_keygen(host, msk, ["A", "B", "C"])
    blob = blob.tobytes()
    # non-monotone and out-of-range offsets: those items fail, nothing outside the blob is read, the others decrypt
    o = ct_off.copy()
    o[2] = o[3] + 5                                                       # item 1 ends after item 2 starts, item 2 has negative extent
    out, out_off, status = ac17.cp_decrypt_packed(host, sk, blob, o)
    assert status[2] == -1 and status[0] == 0 and status[3] == 0 and status[4] == 0
    o = ct_off.copy()
    o[5] = np.uint64(len(blob) + (1 << 40))                               # last record claims to run far past the blob
    out, out_off, status = ac17.cp_decrypt_packed(host, sk, blob, o)
    assert list(status) == [0, 0, 0, 0, -1]
    # a row coordinate >= p (x + p: the same residue, a second encoding) and a point off the curve
    P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    rec0 = int(ct_off[0])
    pol_len = int.from_bytes(blob[rec0:rec0 + 4], "little")
    rows_at = rec0 + 4 + pol_len + 1 + 4 + 384 + 4                        # first row record: name length, name, count, 3 x G1
    nl = int.from_bytes(blob[rows_at:rows_at + 4], "little")
    x_at = rows_at + 4 + nl + 4
    x = int.from_bytes(blob[x_at:x_at + 32], "little")
    bad = bytearray(blob)
    if x + P < 1 << 256:
        bad[x_at:x_at + 32] = (x + P).to_bytes(32, "little")
        out, out_off, status = ac17.cp_decrypt_packed(host, sk, bytes(bad), ct_off)
        assert list(status) == [-1, 0, 0, 0, 0], "a coordinate >= p must be rejected, not reduced"
        # the trusted form skips the pass: the reduced coordinate is the same point, the item decrypts
        out, out_off, status = ac17.cp_decrypt_packed(host, sk, bytes(bad), ct_off, trusted=True)
        assert not status.any()
    bad = bytearray(blob)
    bad[x_at] ^= 1                                                        # off the curve
    out, out_off, status = ac17.cp_decrypt_packed(host, sk, bytes(bad), ct_off)
    assert list(status) == [-1, 0, 0, 0, 0]
    # c_p outside the order-r subgroup (an arbitrary Fq12 element): item 3
    rec3 = int(ct_off[3])
    end3 = int(ct_off[4])
    sealed_len = len(pts[3]) + 28
    cp_at = end3 - sealed_len - 4 - 384
    bad = bytearray(blob)
    bad[cp_at:cp_at + 384] = b"".join((7 + i).to_bytes(32, "little") for i in range(12))
    out, out_off, status = ac17.cp_decrypt_packed(host, sk, bytes(bad), ct_off)
    assert list(status) == [0, 0, 0, -1, 0]
    assert rec3 < cp_at


def test_packed_keygen_equals_object_keygen_on_the_same_tape(host):
    """rabe_ac17_cp_keygen_packed: n keys in one call (items that share an attribute list are one launch of the Level B keygen kernels) =
    n calls of cp_keygen on the same randomness, byte for byte; the keys decrypt."""
    pk, msk = ac17.setup(host)
    sets = [["A", "B"], ["A", "B", "C", "D"], ["C"]]
    item_set = [0, 1, 2, 1, 0, 1, 2, 0, 1]
    n = len(item_set)
    tape = [1000003 * (i + 11) + 7 for i in range(8 * n)]                   # r0, r1, sigma per attribute, sigma' per item
    host.set_tape(tape)
    objs = [ac17.cp_keygen(host, msk, sets[s]) for s in item_set]
    host.set_tape(tape)
    blob, off = ac17.cp_keygen_packed(host, msk, sets, item_set)
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(off[i]):int(off[i + 1])].tobytes(), i
    ct = ac17.cp_encrypt(host, pk, '"A" and ("C" or "D")', b"bulk keys", hl.HUMAN_POLICY)
    for i in (1, 3):
        assert ac17.cp_decrypt(host, hl.Obj.deserialize("ac17_cp_sk", blob[int(off[i]):int(off[i + 1])].tobytes()), ct) == b"bulk keys"
    with pytest.raises(hl.RabeError):
        ac17.cp_keygen_packed(host, msk, [["A"], []], [0, 1])               # `empty attributes!` (:199) fails the call
    # a bulk run on OS randomness: 2000 keys over 50 attributes, every key opens a ciphertext under an AND of five of its attributes
    attrs = ["a%d" % i for i in range(50)]
    blob, off = ac17.cp_keygen_packed(host, msk, [attrs, attrs[:25]], np.arange(2000, dtype=np.uint32) % 2)
    ct = ac17.cp_encrypt(host, pk, '"a1" and ("a7" and ("a13" and ("a19" and "a24")))', b"x" * 40, hl.HUMAN_POLICY)     # AC17: binary ANDs
    for i in (0, 1, 1998, 1999):
        assert ac17.cp_decrypt(host, hl.Obj.deserialize("ac17_cp_sk", blob[int(off[i]):int(off[i + 1])].tobytes()), ct) == b"x" * 40


def test_kp_packed_equals_object_api_and_fails_items_alone(host):
    """rabe_ac17_kp_{encrypt,decrypt}_packed: Ac17KpCiphertext records byte-identical to kp_encrypt_batch on the same tape; one key's policy
    against every ciphertext's attribute list -- a list that does not satisfy it, a tampered record and bad offsets fail their own item."""
    pk, msk = ac17.setup(host)
    sets = [["B", "C"], ["A"], ["B", "D"], ["A", "C", "D"]]
    n = 11
    item_set = [i % 4 for i in range(n)]
    pts = [b"kp plaintext %d " % i * (i % 3 + 1) for i in range(n)]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
    tape = [1000003 * (i + 3) + 19 for i in range(4 * n)]                    # s0, s1, msg exponent, nonce per item
    host.set_tape(tape)
    objs = ac17.kp_encrypt_batch(host, pk, [sets[s] for s in item_set], pts)
    host.set_tape(tape)
    blob, ct_off = ac17.kp_encrypt_packed(host, pk, sets, item_set, b"".join(pts), off)
    host.clear_tape()
    for i in range(n):
        assert objs[i].serialize() == blob[int(ct_off[i]):int(ct_off[i + 1])].tobytes(), i
    sk = ac17.kp_keygen(host, msk, '{"name": "or", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}', hl.JSON_POLICY)
    for trusted in (False, True):
        out, out_off, status = ac17.kp_decrypt_packed(host, sk, blob, ct_off, trusted=trusted)
        want = [0, 0, -1, 0] * 3                                               # ["B", "D"] does not satisfy A or (B and C)
        assert list(status) == want[:n]
        for i in range(n):
            got = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
            assert got == (pts[i] if want[i] == 0 else b"")
            if want[i] == 0:
                assert ac17.kp_decrypt(host, sk, hl.Obj.deserialize("ac17_kp_ct", blob[int(ct_off[i]):int(ct_off[i + 1])].tobytes())) == pts[i]
    raw = bytearray(blob.tobytes())
    raw[int(ct_off[2]) - 3] ^= 0x40                                            # item 1's sealed bytes
    o = ct_off.copy()
    o[10] = np.uint64(int(o[9]) - 8)                                           # item 9: non-monotone offsets
    out, out_off, status = ac17.kp_decrypt_packed(host, sk, bytes(raw), o)
    assert list(status[:10]) == [0, -1, -1, 0, 0, 0, -1, 0, 0, -1]


def test_a_batch_cut_into_parts_gives_the_bytes_of_its_halves():
    """A packed encrypt can go through the device in parts whose copies out run beside the next part's arithmetic (schemes.cpp:
    encrypt_packed_core, RABE_AC17_ENC_PARTS; off by default -- measured no faster).  The cut must not show in the bytes: 32 768 items in
    one call on a tape = the first 16 384 and the last 16 384 in two calls (one part each) continuing the same tape; the batch decrypts.
    Runs in a child process: the setting is read once per process."""
    import os
    import subprocess
    import sys
    if os.environ.get("RABE_AC17_ENC_PARTS") != "4":
        env = dict(os.environ, RABE_AC17_ENC_PARTS="4")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        pr = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "cut_into_parts"], env=env, cwd=root,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert pr.returncode == 0, pr.stdout.decode()[-3000:]
        return
    host = hl.Host(0)
    pk, msk = ac17.setup(host)
    n = 32768
    item_pol = np.arange(n, dtype=np.uint32) % 3
    pts = [b"part %d" % i for i in range(n)]
    pt_blob = b"".join(pts)
    off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
    tape = [(1000003 * (i + 3) + 17) % (1 << 200) for i in range(4 * n)]
    host.set_tape(tape)
    blob, ct_off = ac17.cp_encrypt_packed(host, pk, POLS, item_pol, pt_blob, off, hl.HUMAN_POLICY)
    host.set_tape(tape)
    h = n // 2
    b1, o1 = ac17.cp_encrypt_packed(host, pk, POLS, item_pol[:h], pt_blob[:int(off[h])], off[:h + 1], hl.HUMAN_POLICY)
    b2, o2 = ac17.cp_encrypt_packed(host, pk, POLS, item_pol[h:], pt_blob[int(off[h]):], off[h:] - off[h], hl.HUMAN_POLICY)
    host.clear_tape()
    assert blob[:int(ct_off[h])].tobytes() == b1.tobytes()
    assert blob[int(ct_off[h]):].tobytes() == b2.tobytes()
    sk = ac17.cp_keygen(host, msk, ["A", "B", "C", "D"])
    out, out_off, status = ac17.cp_decrypt_packed(host, sk, blob, ct_off, trusted=True)
    assert not status.any() and out.tobytes() == pt_blob
    host.close()
