"""Mixed-shape batches of >= 64 items through the gather kernels of the decrypts (k_bsw_dec_pairs, k_lsw_dec_pairs, k_aw11_dec_pairs,
k_ghw11_pairs): a ragged batch is cut into tiles of 64 consecutive items and a wave holds the j-th pair of a tile's items
(engine_jobs.hip: pair_lane / k_tile_offsets), items in any order, tiles whose items differ in size, a last tile that is not full.
Every plaintext must come back (the AES tag only verifies under the exact Gt), and the transform's records must be the ones the
one-call API produces."""
import random

import numpy as np
import pytest

from rabe_amd import hostlib as hl

pytestmark = pytest.mark.gpu
PT = b"dance like no one's watching, encrypt like everyone is!"


def _offsets(items):
    return np.concatenate([[0], np.cumsum([len(p) for p in items])]).astype(np.uint64)


def _split(out, off, n):
    return [bytes(out[int(off[i]):int(off[i + 1])]) for i in range(n)]


def _and(names):
    """left-deep binary conjunction (the MSP schemes take two children per gate: src/utils/policy/msp.rs panics otherwise)"""
    t = '{"name": "%s"}' % names[0]
    for a in names[1:]:
        t = '{"name": "and", "children": [%s, {"name": "%s"}]}' % (t, a)
    return t


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


def test_shuffled_mixed_shapes_round_trip(host):
    from rabe_amd.schemes import aw11, bsw, ghw11, lsw
    rnd = random.Random(5)
    attrs = ["A%d" % i for i in range(12)]
    pols = [_and(attrs[:1] + attrs[1:2]), _and(attrs[:5]), _and(attrs[:12]), '{"name": "or", "children": [{"name": "A0"}, %s]}' % _and(attrs[3:9])]
    n = 203                                          # three full tiles and a short one
    item_pol = [rnd.randrange(len(pols)) for _ in range(n)]
    pts = [PT + i.to_bytes(2, "little") for i in range(n)]
    # ---- BSW
    pk, msk = bsw.setup(host)
    sk = bsw.keygen(host, pk, msk, attrs)
    blob, off = bsw.encrypt_packed(host, pk, pols, item_pol, b"".join(pts), _offsets(pts), hl.JSON_POLICY)
    out, oo, st = bsw.decrypt_packed(host, sk, blob, off)
    assert not st.any() and _split(out, oo, n) == pts
    # ---- LSW: n keys of mixed policies, one ciphertext
    lpk, lmsk = lsw.setup(host)
    kblob, koff = lsw.keygen_packed(host, lpk, lmsk, pols, item_pol, hl.JSON_POLICY)
    ct = lsw.encrypt(host, lpk, attrs, PT)
    out, oo, st = lsw.decrypt_packed(host, ct, kblob, koff)
    assert not st.any() and _split(out, oo, n) == [PT] * n
    # ---- AW11
    gk = aw11.setup(host)
    apk, amsk = aw11.authgen(host, gk, attrs)
    ask = aw11.keygen(host, gk, amsk, "alice", attrs)
    blob, off = aw11.encrypt_packed(host, gk, [apk], pols, item_pol, b"".join(pts), _offsets(pts), hl.JSON_POLICY)
    out, oo, st = aw11.decrypt_packed(host, gk, ask, blob, off)
    assert not st.any() and _split(out, oo, n) == pts
    # ---- GHW11 transform: the packed form against the one-call form, record by record
    gpk, gmsk = ghw11.setup(host)
    tk, rk = ghw11.tkgen(host, ghw11.keygen(host, gpk, gmsk, attrs))
    m = 130
    cts = [ghw11.encrypt(host, gpk, pols[item_pol[i]], hl.JSON_POLICY, pts[i]) for i in range(m)]
    recs = [c.serialize() for c in cts]
    got, st = ghw11.transform_packed(host, tk, b"".join(recs), _offsets(recs))
    assert not np.asarray(st).any()
    for i in (0, 1, 63, 64, 65, 127, 128, 129):
        want = ghw11.transform(host, cts[i], tk).serialize()
        assert got[i].tobytes() == want, i
