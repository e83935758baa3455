"""Public known-answer vectors for the host layer's hand-written SHA3-256 and AES-256-GCM (rabe_amd/csrc/host/sha3.h,
aes_gcm.h -- the reference uses the sha3 and aes-gcm crates, Cargo.toml:28,36; src/utils/aes/mod.rs:10-55).

Vectors: FIPS 202 SHA3-256 of "" and "abc" (and hashlib's independent implementation on ragged lengths around the
136-byte rate); FIPS 197 appendix C.3 (AES-256 block); test cases 13-16 of the GCM specification (McGrew & Viega), the
ones with a 256-bit key.  A small pure-Python AES-GCM written from the two standards, itself pinned by the FIPS 197 block
vector, cross-checks both the remembered vectors and the C++ code on random inputs."""
import ctypes
import hashlib
import random

import pytest

from rabe_amd import hostlib as hl


def lib():
    return hl._lib()


# ---------------------------------------------------------------------------------------------------- independent AES-GCM (test helper)
def _sbox():
    # multiplicative inverse in GF(2^8) followed by the affine map (FIPS 197 section 5.1.1)
    def mul(a, b):
        r = 0
        while b:
            if b & 1:
                r ^= a
            a = ((a << 1) ^ 0x11b) if a & 0x80 else (a << 1)
            b >>= 1
        return r & 0xff
    inv = [0] * 256
    for a in range(1, 256):
        for b in range(1, 256):
            if mul(a, b) == 1:
                inv[a] = b
                break
    box = []
    for a in range(256):
        x = inv[a]
        y = x
        for _ in range(4):
            x = ((x << 1) | (x >> 7)) & 0xff
            y ^= x
        box.append(y ^ 0x63)
    return box, mul


SBOX, GMUL = _sbox()


def aes256_block(key, block):
    w = [list(key[4 * i:4 * i + 4]) for i in range(8)]
    rcon = 1
    for i in range(8, 60):
        t = list(w[i - 1])
        if i % 8 == 0:
            t = [SBOX[t[1]] ^ rcon, SBOX[t[2]], SBOX[t[3]], SBOX[t[0]]]
            rcon = GMUL(rcon, 2)
        elif i % 8 == 4:
            t = [SBOX[x] for x in t]
        w.append([a ^ b for a, b in zip(w[i - 8], t)])
    s = [block[i] ^ w[i // 4][i % 4] for i in range(16)]
    for rnd in range(1, 15):
        s = [SBOX[x] for x in s]
        s = [s[(i + 4 * (i % 4)) % 16] for i in range(16)]                       # ShiftRows (column-major state)
        if rnd != 14:
            o = []
            for c in range(4):
                a = s[4 * c:4 * c + 4]
                o += [GMUL(a[0], 2) ^ GMUL(a[1], 3) ^ a[2] ^ a[3], a[0] ^ GMUL(a[1], 2) ^ GMUL(a[2], 3) ^ a[3],
                      a[0] ^ a[1] ^ GMUL(a[2], 2) ^ GMUL(a[3], 3), GMUL(a[0], 3) ^ a[1] ^ a[2] ^ GMUL(a[3], 2)]
            s = o
        s = [s[i] ^ w[4 * rnd + i // 4][i % 4] for i in range(16)]
    return bytes(s)


def _gf128_mul(x, y):
    z, v = 0, x
    for i in range(128):
        if (y >> (127 - i)) & 1:
            z ^= v
        v = (v >> 1) ^ (0xe1 << 120) if v & 1 else v >> 1
    return z


def aes256_gcm(key, nonce, pt):
    h = int.from_bytes(aes256_block(key, bytes(16)), "big")
    j0 = nonce + b"\x00\x00\x00\x01"
    ct = b""
    for i in range(0, len(pt), 16):
        ctr = nonce + ((i // 16) + 2).to_bytes(4, "big")
        ks = aes256_block(key, ctr)
        ct += bytes(a ^ b for a, b in zip(pt[i:i + 16], ks))
    y = 0
    padded = ct + bytes(-len(ct) % 16)
    for i in range(0, len(padded), 16):
        y = _gf128_mul(y ^ int.from_bytes(padded[i:i + 16], "big"), h)
    y = _gf128_mul(y ^ (len(ct) * 8), h)
    tag = bytes(a ^ b for a, b in zip(y.to_bytes(16, "big"), aes256_block(key, j0)))
    return ct, tag


# ---------------------------------------------------------------------------------------------------- the host layer through its C ABI
def c_sha3(data):
    o = ctypes.create_string_buffer(32)
    assert lib().rabe_sha3_256(data, ctypes.c_size_t(len(data)), o) == 0
    return o.raw


def c_gcm_encrypt(key, nonce, pt):
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    assert lib().rabe_aes256_gcm_encrypt(key, nonce, pt, ctypes.c_size_t(len(pt)), ctypes.byref(p), ctypes.byref(n)) == 0
    return hl._take_bytes(p, n)


def c_gcm_decrypt(key, nonce, ct_tag):
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    rc = lib().rabe_aes256_gcm_decrypt(key, nonce, ct_tag, ctypes.c_size_t(len(ct_tag)), ctypes.byref(p), ctypes.byref(n))
    return hl._take_bytes(p, n) if rc == 0 else None


def test_sha3_256_fips202_vectors():
    assert c_sha3(b"").hex() == "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a"
    assert c_sha3(b"abc").hex() == "3a985da74fe225b2045c172d6bd390bd855f086e3e9d525b46bfe24511431532"
    rnd = random.Random(3)
    for n in [1, 55, 135, 136, 137, 271, 272, 273, 1000]:              # around the 136-byte rate
        d = rnd.randbytes(n)
        assert c_sha3(d) == hashlib.sha3_256(d).digest(), n


def test_python_helper_matches_fips197_block_vector():
    key = bytes(range(32))
    assert aes256_block(key, bytes.fromhex("00112233445566778899aabbccddeeff")).hex() == "8ea2b7ca516745bfeafc49904b496089"


GCM_SPEC_256 = [
    # (key, iv, plaintext, ciphertext, tag): test cases 13, 14, 15 of the GCM specification (no AAD, 96-bit IV)
    ("00" * 32, "00" * 12, "", "", "530f8afbc74536b9a963b4f1c4cb738b"),
    ("00" * 32, "00" * 12, "00" * 16, "cea7403d4d606b6e074ec5d3baf39d18", "d0d1c8a799996bf0265b98b5d48ab919"),
    ("feffe9928665731c6d6a8f9467308308feffe9928665731c6d6a8f9467308308", "cafebabefacedbaddecaf888",
     "d9313225f88406e5a55909c5aff5269a86a7a9531534f7da2e4c303d8a318a721c3c0c95956809532fcf0e2449a6b525b16aedf5aa0de657ba637b391aafd255",
     "522dc1f099567d07f47f37a32a84427d643a8cdcbfe5c0c97598a2bd2555d1aa8cb08e48590dbb3da7b08b1056828838c5f61e6393ba7a0abcc9f662898015ad",
     "b094dac5d93471bdec1a502270e3cc6c"),
]


@pytest.mark.parametrize("key,iv,pt,ct,tag", GCM_SPEC_256)
def test_aes256_gcm_specification_vectors(key, iv, pt, ct, tag):
    key, iv, pt = bytes.fromhex(key), bytes.fromhex(iv), bytes.fromhex(pt)
    want = bytes.fromhex(ct) + bytes.fromhex(tag)
    assert b"".join(aes256_gcm(key, iv, pt)) == want            # the published vector, by an independent implementation
    assert c_gcm_encrypt(key, iv, pt) == want                   # the host layer
    assert c_gcm_decrypt(key, iv, want) == pt
    bad = bytearray(want)
    bad[-1] ^= 1
    assert c_gcm_decrypt(key, iv, bytes(bad)) is None


def test_aes256_gcm_random_inputs_against_independent_implementation():
    rnd = random.Random(9)
    for n in [0, 1, 15, 16, 17, 55, 64, 100]:
        key, iv, pt = rnd.randbytes(32), rnd.randbytes(12), rnd.randbytes(n)
        assert c_gcm_encrypt(key, iv, pt) == b"".join(aes256_gcm(key, iv, pt)), n


def test_encrypt_symmetric_is_kdf_then_gcm():
    """encrypt_symmetric (src/utils/aes/mod.rs:10-26): key = SHA3-256(bytes(Gt)), output = nonce || ct || tag"""
    rnd = random.Random(10)
    gt = b"".join(rnd.randrange(1 << 250).to_bytes(32, "little") for _ in range(12))
    nonce, pt = rnd.randbytes(12), b"dance like no one's watching, encrypt like everyone is!"
    key = hashlib.sha3_256(b"".join(gt[32 * i:32 * i + 32][::-1] for i in range(12))).digest()      # 12 x 32-byte big-endian
    assert hl.encrypt_symmetric(gt, pt, nonce) == nonce + b"".join(aes256_gcm(key, nonce, pt))


def test_both_host_implementations_pass_the_vectors():
    """aes_gcm.h has two forms: AES-NI + PCLMULQDQ (no table indexed by secret bytes) where the CPU has them, and the portable table form.
    The process-wide choice is made once (RABE_AES_PORTABLE=1 forces the portable one): the vectors above run again in a child under it."""
    import os
    import subprocess
    import sys
    if os.environ.get("RABE_AES_PORTABLE"):
        pytest.skip("already the forced-portable child")
    env = dict(os.environ, RABE_AES_PORTABLE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-k", "gcm or symmetric or fips197"], env=env, cwd=root,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert pr.returncode == 0, pr.stdout.decode()[-2000:]
