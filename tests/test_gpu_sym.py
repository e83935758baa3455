"""Level S on the device (rabe_amd/csrc/engine_sym.hip): SHA3-256, label hashes to Fr, the Gt KDF and AES-256-GCM -- the KEM -> DEM step
of src/utils/aes/mod.rs:10-55 and the hashing of src/utils/hash/mod.rs:10-31 -- against the published vectors (FIPS 202, FIPS 197 C.3, GCM
specification test cases 13-15), hashlib, the independent pure-Python AES-GCM of tests/test_host_kats.py and the host layer's C++ code
(itself pinned by the same vectors).  Byte-exact."""
import hashlib
import random

import pytest

from rabe_amd import Engine
from rabe_amd import hostlib as hl
from rabe_amd import symlib as sym
from tests.test_host_kats import GCM_SPEC_256, aes256_block, aes256_gcm, c_gcm_encrypt

pytestmark = pytest.mark.gpu
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


def test_sha3_256_fips202_and_ragged_lengths(eng):
    rnd = random.Random(3)
    msgs = [b"", b"abc"] + [rnd.randbytes(n) for n in [1, 7, 8, 9, 55, 135, 136, 137, 271, 272, 273, 1000]] + [b"a10" + bytes([48 + l, 48 + t]) for l in range(3) for t in range(2)]
    msgs += [rnd.randbytes(rnd.randrange(0, 300)) for _ in range(300)]          # more than one wave, a ragged last one
    got = sym.sha3_256(eng, msgs)
    assert got[0].hex() == "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a"
    assert got[1].hex() == "3a985da74fe225b2045c172d6bd390bd855f086e3e9d525b46bfe24511431532"
    for m, d in zip(msgs, got):
        assert d == hashlib.sha3_256(m).digest(), len(m)


def test_label_hashes_reduce_like_fr_from_slice(eng):
    """sha3_hash_fr (hash/mod.rs:23-31): the digest as a big-endian integer mod r"""
    labels = [b"A00", b"a5010", b"0111", b"01" + b"21", b""] + [("attr%d%d%d" % (i, i % 3, i % 2)).encode() for i in range(200)]
    got = sym.sha3_fr(eng, labels)
    for lab, fr in zip(labels, got):
        want = int.from_bytes(hashlib.sha3_256(lab).digest(), "big") % R
        assert int.from_bytes(fr, "little") == want, lab
    # all five subtractions of the reduction (5 r < 2^256 - 1 < 6 r) are exercised by digests >= 5 r: none of these labels needs to hit
    # that range for the loop to be the same code; Level E's rhip_fr_from_be32_reduce is the other implementation of the same map
    digs = [hashlib.sha3_256(lab).digest() for lab in labels]
    assert eng.fr_from_be32_reduce(digs) == got


def test_gt_kdf_is_sha3_of_the_big_endian_coefficients(eng):
    rnd = random.Random(4)
    gts = [b"".join(rnd.randrange(1 << 254).to_bytes(32, "little") for _ in range(12)) for _ in range(70)]
    want = [hashlib.sha3_256(b"".join(g[32 * i:32 * i + 32][::-1] for i in range(12))).digest() for g in gts]
    assert sym.gt_kdf(eng, gts) == want
    idx = [rnd.randrange(70) for _ in range(130)]
    assert sym.gt_kdf(eng, gts, idx) == [want[i] for i in idx]


def test_aes256_fips197_block_vector_and_random_blocks(eng):
    key = bytes(range(32))
    rnd = random.Random(5)
    keys = [key] + [rnd.randbytes(32) for _ in range(199)]
    blocks = [bytes.fromhex("00112233445566778899aabbccddeeff")] + [rnd.randbytes(16) for _ in range(199)]
    got = sym.aes256_blocks(eng, keys, blocks)
    assert got[0].hex() == "8ea2b7ca516745bfeafc49904b496089"          # FIPS 197 appendix C.3
    for k, b, g in zip(keys, blocks, got):
        assert g == aes256_block(k, b)


def test_aes256_gcm_specification_vectors(eng):
    keys = [bytes.fromhex(v[0]) for v in GCM_SPEC_256]
    ivs = [bytes.fromhex(v[1]) for v in GCM_SPEC_256]
    pts = [bytes.fromhex(v[2]) for v in GCM_SPEC_256]
    want = [iv + bytes.fromhex(v[3]) + bytes.fromhex(v[4]) for iv, v in zip(ivs, GCM_SPEC_256)]
    assert sym.gcm_seal(eng, keys, ivs, pts) == want
    back, ok = sym.gcm_open(eng, keys, want)
    assert back == pts and ok == [1, 1, 1]
    bad = [bytearray(w) for w in want]
    bad[0][-1] ^= 1                      # tag
    bad[1][12] ^= 0x80                   # ciphertext
    bad[2][3] ^= 2                       # nonce
    back, ok = sym.gcm_open(eng, keys, [bytes(b) for b in bad])
    assert ok == [0, 0, 0]
    assert back == [bytes(len(p)) for p in pts]          # a failed tag releases no plaintext


def test_aes256_gcm_lengths_around_blocks_and_segments(eng):
    """one lane per block, GHASH per 64-block segment folded with powers of H: lengths around 16 and around 1024 bytes, several segments"""
    rnd = random.Random(9)
    lens = [0, 1, 15, 16, 17, 31, 32, 33, 55, 59, 64, 100, 1008, 1023, 1024, 1025, 1040, 2047, 2048, 2049, 5000] + [rnd.randrange(0, 200) for _ in range(150)]
    keys = [rnd.randbytes(32) for _ in lens]
    ivs = [rnd.randbytes(12) for _ in lens]
    pts = [rnd.randbytes(n) for n in lens]
    got = sym.gcm_seal(eng, keys, ivs, pts, len_prefix=True)
    for k, iv, p, g in zip(keys, ivs, pts, got):
        assert g[:4] == (len(p) + 28).to_bytes(4, "little")
        if len(p) <= 2049:
            assert g[4:] == iv + b"".join(aes256_gcm(k, iv, p)), len(p)                    # the independent Python implementation
        assert g[4:] == iv + c_gcm_encrypt(k, iv, p), len(p)                               # the host layer's C++
    back, ok = sym.gcm_open(eng, keys, [g[4:] for g in got])
    assert ok == [1] * len(lens) and back == pts


def test_seal_and_open_are_the_references_symmetric_pair(eng):
    """encrypt_symmetric / decrypt_symmetric (aes/mod.rs:10-44) over a batch: the same bytes as the host layer's one-at-a-time functions"""
    rnd = random.Random(10)
    n = 100
    gts = [b"".join(rnd.randrange(1 << 250).to_bytes(32, "little") for _ in range(12)) for _ in range(n)]
    nonces = [rnd.randbytes(12) for _ in range(n)]
    pts = [b"dance like no one's watching, encrypt like everyone is!" + bytes(rnd.randrange(256) for _ in range(i % 7)) for i in range(n)]
    got = sym.seal(eng, gts, nonces, pts)
    for g, nonce, pt, s in zip(gts, nonces, pts, got):
        assert s == hl.encrypt_symmetric(g, pt, nonce)
    back, ok = sym.open_(eng, gts, got)
    assert back == pts and ok == [1] * n
    # through an index (one Gt per slot, items name their slot) and with the wrong Gt for some
    idx = list(range(n))
    idx[5], idx[6] = 6, 5
    back, ok = sym.open_(eng, gts, got, idx)
    assert ok == [0 if i in (5, 6) else 1 for i in range(n)]
    assert all(back[i] == pts[i] for i in range(n) if i not in (5, 6))


def test_long_plaintexts_fold_hundreds_of_ghash_segments(eng):
    """a 300 KB and a 1 MB plaintext beside short ones in one batch: CTR is one lane per block, GHASH one lane per 64-block segment folded
    with H^64 per item -- the same bytes as the host layer's one-at-a-time function (AES-NI / PCLMULQDQ or the portable code)"""
    rnd = random.Random(11)
    lens = [300 * 1024 + 5, 7, 1 << 20, 0, 1024 * 64, 1024 * 64 + 1]
    gts = [b"".join(rnd.randrange(1 << 250).to_bytes(32, "little") for _ in range(12)) for _ in lens]
    nonces = [rnd.randbytes(12) for _ in lens]
    pts = [rnd.randbytes(n) for n in lens]
    got = sym.seal(eng, gts, nonces, pts)
    for g, nonce, pt, s in zip(gts, nonces, pts, got):
        assert s == hl.encrypt_symmetric(g, pt, nonce), len(pt)
    back, ok = sym.open_(eng, gts, got)
    assert ok == [1] * len(lens) and back == pts
    bad = bytearray(got[2])
    bad[12 + 700000] ^= 4                      # one bit deep inside the 1 MB ciphertext
    back, ok = sym.open_(eng, gts, got[:2] + [bytes(bad)] + got[3:])
    assert ok == [1, 1, 0, 1, 1, 1] and back[2] == bytes(len(pts[2]))


def test_empty_batches_are_no_ops(eng):
    assert sym.sha3_256(eng, []) == [] and sym.gt_kdf(eng, []) == [] and sym.seal(eng, [], [], []) == []
