"""GPU parity for GHW11 (CP-ABE with outsourced decryption, src/schemes/ghw11/mod.rs:92-305 -- SURVEY.md 8f item 1):
the host layer against the oracle's golden vectors on the same explicit randomness, the reference's own test cases
(or / and / or_and, :308-449), and the batched `transform` a decryption service would run."""
import json
import os
import random

import pytest

from rabe_amd import hostlib as hl
from rabe_amd.schemes import ghw11

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PLAINTEXT = b"dance like no one's watching, encrypt like everyone is!"
LANG = {"json": hl.JSON_POLICY, "human": hl.HUMAN_POLICY}
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def hb(s):
    return bytes.fromhex(s)


def fri(x):
    return int.from_bytes(hb(x), "little")


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


def test_ghw11_matches_golden(host):
    with open(os.path.join(HERE, "golden", "ghw11.json")) as f:
        doc = json.load(f)
    p = doc["pk"]
    pkb = hb(p["g1"]) + hb(p["g2"]) + hb(p["g1_a"]) + hb(p["g2_a"]) + hb(p["e_gg_alpha"])
    pk = hl.Obj.deserialize("ghw11_pk", pkb)
    msk = hl.Obj.deserialize("ghw11_msk", hb(doc["msk"]["g2_alpha"]) + pkb)
    for c in doc["cases"]:
        host.set_tape([fri(x) for x in c["keygen_tape"]])
        sk = ghw11.keygen(host, pk, msk, c["attrs"])
        g = hl.parse_obj("ghw11_sk", sk.serialize())
        assert (g["k"], g["l"]) == (hb(c["sk"]["k"]), hb(c["sk"]["l"]))
        assert g["attr_key"] == [(n, hb(k)) for n, k in c["sk"]["attr_key"]]
        host.set_tape([fri(x) for x in c["tkgen_tape"]])
        tk, rk = ghw11.tkgen(host, sk)
        g = hl.parse_obj("ghw11_tk", tk.serialize())
        assert (g["k_z"], g["l_z"]) == (hb(c["tk"]["k_z"]), hb(c["tk"]["l_z"]))
        assert g["attr_key_z"] == [(n, hb(k)) for n, k in c["tk"]["attr_key_z"]]
        assert hl.parse_obj("ghw11_rk", rk.serialize())["z"] == hb(c["tkgen_tape"][0])
        et = [fri(x) for x in c["encrypt_tape"]]
        host.set_tape([et[0], fri(c["msg_rho"])] + et[1:] + [13])      # secret, msg, gate coefficients, t_i.., nonce
        ct = ghw11.encrypt(host, pk, c["policy"], LANG[c["language"]], PLAINTEXT)
        g = hl.parse_obj("ghw11_ct", ct.serialize())
        assert (g["c"], g["c1"]) == (hb(c["ct"]["c"]), hb(c["ct"]["c1"]))
        assert g["ci_di"] == [(n, hb(a), hb(b)) for n, a, b in c["ct"]["ci_di"]]
        host.clear_tape()
        tct = ghw11.transform(host, ct, tk)
        g = hl.parse_obj("ghw11_tct", tct.serialize())
        assert g["t"] == hb(c["t"]) and g["c"] == hb(c["ct"]["c"])
        rec = ct.serialize()                                         # the device-resident packed path against the same golden value
        out, st = ghw11.transform_packed(host, tk, rec, [0, len(rec)])
        assert st[0] == 0 and out[0].tobytes() == hb(c["ct"]["c"]) + hb(c["t"])
        assert ghw11.decrypt_out_gt(host, tct, rk) == hb(c["decrypted"]) == hb(c["msg"])
        assert ghw11.decrypt_out(host, tct, rk, ct) == PLAINTEXT


def test_ghw11_reference_cases(host):
    # ghw11/mod.rs:308-449: or (matching / not matching), and, or_and; keygen(None) for no attributes
    pk, msk = ghw11.setup(host)
    assert ghw11.keygen(host, pk, msk, []) is None
    cases = [(r'''{"name": "or", "children": [{"name": "A"}, {"name": "B"}]}''', ["D", "B"], True),
             (r'''{"name": "or", "children": [{"name": "A"}, {"name": "B"}]}''', ["C", "D"], False),
             (r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}''', ["A", "B"], True),
             (r'''{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}''', ["A", "C"], False),
             (r'''{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "D"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}]}''', ["A", "B", "C"], True)]
    for policy, attrs, ok in cases:
        ct = ghw11.encrypt(host, pk, policy, hl.JSON_POLICY, PLAINTEXT)
        tk, rk = ghw11.tkgen(host, ghw11.keygen(host, pk, msk, attrs))
        if ok:
            assert ghw11.decrypt_out(host, ghw11.transform(host, ct, tk), rk, ct) == PLAINTEXT
        else:
            with pytest.raises(hl.RabeError):
                ghw11.transform(host, ct, tk)
    # a wrong retrieve key does not open the ciphertext
    tk2, rk2 = ghw11.tkgen(host, ghw11.keygen(host, pk, msk, ["A", "B"]))
    ct = ghw11.encrypt(host, pk, cases[2][0], hl.JSON_POLICY, PLAINTEXT)
    tct = ghw11.transform(host, ct, tk2)
    _tk3, rk3 = ghw11.tkgen(host, ghw11.keygen(host, pk, msk, ["A", "B"]))
    with pytest.raises(hl.RabeError):
        ghw11.decrypt_out(host, tct, rk3, ct)
    assert ghw11.decrypt_out(host, tct, rk2, ct) == PLAINTEXT


def test_ghw11_transform_batch(host):
    """The service case: many (ciphertext, transform key) pairs in one launch set; element i equals transform(ct_i, tk_i),
    a key that does not satisfy its ciphertext fails that item only."""
    rnd = random.Random(31)
    pk, msk = ghw11.setup(host)
    attrs = ["a%d" % i for i in range(12)]

    def tree(ns):
        if len(ns) == 1:
            return '{"name": "%s"}' % ns[0]
        h = rnd.randrange(1, len(ns))
        return '{"name": "%s", "children": [%s, %s]}' % (rnd.choice(["and", "or"]), tree(ns[:h]), tree(ns[h:]))
    users = [ghw11.tkgen(host, ghw11.keygen(host, pk, msk, attrs)), ghw11.tkgen(host, ghw11.keygen(host, pk, msk, attrs[:1]))]
    n = 40
    pts = [PLAINTEXT + bytes([i]) for i in range(n)]
    cts = [ghw11.encrypt(host, pk, tree(rnd.sample(attrs, 8)), hl.JSON_POLICY, pts[i]) for i in range(n)]
    who = [1 if i % 5 == 2 else 0 for i in range(n)]
    batch = ghw11.transform_batch(host, cts, [users[w][0] for w in who])
    for i in range(n):
        tk, rk = users[who[i]]
        try:
            single = ghw11.transform(host, cts[i], tk)
        except hl.RabeError:
            single = None
        assert (batch[i] is None) == (single is None)
        if single is not None:
            assert batch[i].serialize() == single.serialize()
            assert ghw11.decrypt_out(host, batch[i], rk, cts[i]) == pts[i]
    assert all(batch[i] is not None for i in range(n) if who[i] == 0)


def test_ghw11_transform_packed_equals_object_api(host):
    """rabe_ghw11_transform_packed (device-resident: every Miller loop replays the transform key's prepared lines, the rows that share l_z
    are one pairing of a multi-scalar sum) against the object API (general pairing jobs, every G2 argument walked): the same 768 bytes per
    item; a policy the key does not satisfy, a tampered row, a truncated record and bad offsets fail their own item only."""
    import numpy as np
    rnd = random.Random(77)
    pk, msk = ghw11.setup(host)
    attrs = ["a%d" % i for i in range(14)]

    def tree(ns):
        if len(ns) == 1:
            return '{"name": "%s"}' % ns[0]
        h = rnd.randrange(1, len(ns))
        return '{"name": "%s", "children": [%s, %s]}' % (rnd.choice(["and", "or"]), tree(ns[:h]), tree(ns[h:]))
    tk, rk = ghw11.tkgen(host, ghw11.keygen(host, pk, msk, attrs[:11]))                    # a3 .. never needs a11+; some policies do
    pols = [tree(rnd.sample(attrs[:11], 7)) for _ in range(4)] + ['{"name": "and", "children": [{"name": "a1"}, {"name": "a13"}]}',
                                                                  '{"name": "a2"}']
    n = 26
    pts = [PLAINTEXT + bytes([i]) for i in range(n)]
    cts = [ghw11.encrypt(host, pk, pols[i % len(pols)], hl.JSON_POLICY, pts[i]) for i in range(n)]
    recs = [c.serialize() for c in cts]
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    blob = b"".join(recs)
    for trusted in (False, True):
        out, status = ghw11.transform_packed(host, tk, blob, off, trusted=trusted)
        for i in range(n):
            if i % len(pols) == 4:                                  # needs a13: the key does not satisfy the policy
                assert status[i] == -1 and not out[i].any()
                continue
            assert status[i] == 0
            single = ghw11.transform(host, cts[i], tk)
            assert out[i].tobytes() == single.serialize(), i
            assert ghw11.decrypt_out(host, hl.Obj.deserialize("ghw11_tct", out[i].tobytes()), rk, cts[i]) == pts[i]
    # damage: item 1's first row point off its curve (checked mode), item 7 truncated by its offsets, item 12 with non-monotone offsets
    raw = bytearray(blob)
    rec = int(off[1])
    pl = int.from_bytes(raw[rec:rec + 4], "little")
    first = rec + 4 + pl + 1 + 384 + 64 + 4
    nl = int.from_bytes(raw[first:first + 4], "little")
    raw[first + 4 + nl] ^= 1
    o = off.copy()
    cut = bytes(raw[:int(off[8]) - 30]) + bytes(raw[int(off[8]):])
    o[8:] -= 30
    o2 = o.copy()
    o2[13] = np.uint64(int(o2[12]) - 4)
    out, status = ghw11.transform_packed(host, tk, cut, o2)
    want = [0 if i % len(pols) != 4 else -1 for i in range(n)]
    for i in (1, 7, 12, 13):
        want[i] = -1
    assert list(status) == want
    good = ghw11.transform(host, cts[2], tk).serialize()
    assert out[2].tobytes() == good
