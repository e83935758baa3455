"""GPU parity, Level B / AC17: the engine's restructured batch kernels (fixed-base tables, Fr
pre-combination, multi-Miller + one final exponentiation) against the oracle's statement-by-statement
restatement of ac17::{cp_keygen, cp_encrypt, cp_decrypt} on the same explicit randomness -- bit-exact."""
import pytest

from oracle import bn254 as bn
from oracle import policy as pol
from oracle import schemes as sch
from oracle.tape import ListRng, SeededRng
from tests import ac17_host as host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from rabe_amd import Engine
    from rabe_amd import engine as E
    eng = Engine(0)
    rng = SeededRng(2024)
    pk, msk = sch.ac17_setup(rng)
    dpk = E.Ac17Pk(eng, bn.g1_to_le(pk["g"]), [bn.g2_to_le(x) for x in pk["h_a"]], [bn.gt_to_le(x) for x in pk["e_gh_ka"]])
    yield eng, E, pk, msk, dpk, rng
    dpk.destroy()
    eng.close()


POLICIES = [
    ('"A" and "B"', pol.HUMAN, ["A", "B"]),
    (r'''{"name": "or", "children": [{"name": "X"}, {"name": "and", "children": [{"name": "A"}, {"name": "B"}]}]}''', pol.JSON, ["A", "B"]),
    (r'''{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "D"}, {"name": "and", "children": [{"name": "B"},{"name": "C"}]}]}]}''', pol.JSON, ["A", "B", "C"]),
]


def gpu_encrypt(eng, E, dpk, policy, lang, s_list, msgs):
    pi, A = host.policy_table(policy, lang)
    n_items, n_rows = len(s_list), len(pi)
    dA = eng.upload(A)
    ds = eng.upload(b"".join(host.le(s0) + host.le(s1) for s0, s1 in s_list))
    dmsg = eng.upload(b"".join(bn.gt_to_le(m) for m in msgs))
    dc0, dc, dcp = eng.alloc(n_items * 3 * 128), eng.alloc(n_items * n_rows * 3 * 64), eng.alloc(n_items * 384)
    E.ac17_encrypt_dev(eng, dpk, n_items, dA, eng.upload_u32([0] * n_items), eng.upload_u32([i * n_rows for i in range(n_items + 1)]),
                       n_items * n_rows, ds, dmsg, dc0, dc, dcp)
    return pi, eng.download(dc0), eng.download(dc), eng.download(dcp), (dc0, dc, dcp)


@pytest.mark.parametrize("policy,lang,_attrs", POLICIES)
def test_cp_encrypt_matches_reference_order(env, policy, lang, _attrs):
    eng, E, pk, msk, dpk, rng = env
    e_gen = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    items = [(rng.fr(), rng.fr()) for _ in range(3)]
    msgs = [bn.gt_pow(e_gen, rng.fr_nonzero()) for _ in items]
    pi, c0, c, cp, _ = gpu_encrypt(eng, E, dpk, policy, lang, items, msgs)
    n_rows = len(pi)
    # the oracle is slow (variable-base everything): check item 0 fully, the others on c_0 / c_p and one row
    for i, ((s0, s1), msg) in enumerate(zip(items, msgs)):
        if i == 0:
            ct = sch.ac17_cp_encrypt(pk, policy, lang, ListRng([s0, s1]), msg)
            assert [n for n, _ in ct["ct"]["c"]] == pi
            want_c = b"".join(bn.g1_to_le(p) for _, vec in ct["ct"]["c"] for p in vec)
            assert c[i * n_rows * 192:(i + 1) * n_rows * 192] == want_c
            assert c0[i * 384:(i + 1) * 384] == b"".join(bn.g2_to_le(x) for x in ct["ct"]["c_0"])
            assert cp[i * 384:(i + 1) * 384] == bn.gt_to_le(ct["ct"]["c_p"])
        else:
            want_c0 = [bn.g2_mul(pk["h_a"][0], s0), bn.g2_mul(pk["h_a"][1], s1), bn.g2_mul(pk["h_a"][2], (s0 + s1) % bn.R)]
            assert c0[i * 384:(i + 1) * 384] == b"".join(bn.g2_to_le(x) for x in want_c0)
            want_cp = bn.gt_mul(bn.gt_mul(bn.gt_pow(pk["e_gh_ka"][0], s0), bn.gt_pow(pk["e_gh_ka"][1], s1)), msg)
            assert cp[i * 384:(i + 1) * 384] == bn.gt_to_le(want_cp)


def test_cp_keygen_matches_reference_order(env):
    eng, E, pk, msk, dpk, rng = env
    attrs = ["A", "B", "C"]
    r0, r1 = rng.fr(), rng.fr()
    sig = [rng.fr() for _ in attrs]
    sigp = rng.fr()
    tape = [r0, r1] + sig + [sigp]
    want = sch.ac17_cp_keygen(msk, attrs, ListRng(tape))
    g_tab = eng.g1_table(bn.g1_to_le(msk["g"]))
    h_tab = eng.g2_table(bn.g2_to_le(msk["h"]))
    H, H01 = host.keygen_tables(attrs)
    dgk = eng.upload(b"".join(bn.g1_to_le(x) for x in msk["g_k"]))
    dainv = eng.upload(b"".join(host.le(bn.fr_inv(a)) for a in msk["a"]))
    db = eng.upload(b"".join(host.le(b) for b in msk["b"]))
    n_items = 2     # second item: same randomness -> same key (exercises the batch indexing)
    dr = eng.upload((host.le(r0) + host.le(r1)) * n_items)
    dsig = eng.upload(b"".join(host.le(x) for x in sig) * n_items)
    dsigp = eng.upload(host.le(sigp) * n_items)
    dk0, dk, dkp = eng.alloc(n_items * 384), eng.alloc(n_items * len(attrs) * 192), eng.alloc(n_items * 192)
    E.ac17_keygen_dev(eng, g_tab, h_tab, dgk, dainv, db, n_items, len(attrs), eng.upload(H), eng.upload(H01), dr, dsig, dsigp, dk0, dk, dkp)
    first = (eng.download(dk0), eng.download(dk), eng.download(dkp))
    # the same launch over 16-bit windows (g, h) and then signed 18-bit windows for g: identical key bytes
    g_tab.add_w16(); h_tab.add_w16()
    E.ac17_keygen_dev(eng, g_tab, h_tab, dgk, dainv, db, n_items, len(attrs), eng.upload(H), eng.upload(H01), dr, dsig, dsigp, dk0, dk, dkp)
    assert (eng.download(dk0), eng.download(dk), eng.download(dkp)) == first
    g_tab.add_wide(18)
    E.ac17_keygen_dev(eng, g_tab, h_tab, dgk, dainv, db, n_items, len(attrs), eng.upload(H), eng.upload(H01), dr, dsig, dsigp, dk0, dk, dkp)
    assert (eng.download(dk0), eng.download(dk), eng.download(dkp)) == first
    k0, k, kp = eng.download(dk0), eng.download(dk), eng.download(dkp)
    w_k0 = b"".join(bn.g2_to_le(x) for x in want["sk"]["k_0"])
    w_k = b"".join(bn.g1_to_le(p) for _, vec in want["sk"]["k"] for p in vec)
    w_kp = b"".join(bn.g1_to_le(p) for p in want["sk"]["k_p"])
    assert k0 == w_k0 * n_items
    assert k == w_k * n_items
    assert kp == w_kp * n_items
    g_tab.destroy(); h_tab.destroy()


@pytest.mark.parametrize("policy,lang,attrs", POLICIES)
def test_cp_decrypt_roundtrip_and_reference(env, policy, lang, attrs):
    eng, E, pk, msk, dpk, rng = env
    e_gen = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    sk = sch.ac17_cp_keygen(msk, attrs, rng)          # oracle key (reference order)
    items = [(rng.fr(), rng.fr()) for _ in range(2)]
    msgs = [bn.gt_pow(e_gen, rng.fr_nonzero()) for _ in items]
    pi, c0, c, cp, (dc0, dc, dcp) = gpu_encrypt(eng, E, dpk, policy, lang, items, msgs)
    n_rows, n_items = len(pi), len(items)
    ok, ct_sel, sk_sel = host.decrypt_selection(sk["attr"], pi, policy, lang)
    assert ok
    dsk_k0 = eng.upload(b"".join(bn.g2_to_le(x) for x in sk["sk"]["k_0"]))
    dsk_k = eng.upload(b"".join(bn.g1_to_le(p) for _, vec in sk["sk"]["k"] for p in vec))
    dsk_kp = eng.upload(b"".join(bn.g1_to_le(p) for p in sk["sk"]["k_p"]))
    dout = eng.alloc(n_items * 384)
    E.ac17_decrypt_dev(eng, n_items, dc0, dc, eng.upload_u32([i * n_rows for i in range(n_items + 1)]), dcp,
                       dsk_k0, dsk_k, eng.upload_u32([0, len(attrs)]), dsk_kp, eng.upload_u32([0] * n_items),
                       eng.upload_u32(ct_sel * n_items), eng.upload_u32([i * len(ct_sel) for i in range(n_items + 1)]),
                       eng.upload_u32(sk_sel * n_items), eng.upload_u32([i * len(sk_sel) for i in range(n_items + 1)]), dout)
    out = eng.download(dout)
    # prepared-key path (k_0 lines computed once, paired Miller loops): identical bytes
    lines = E.Ac17SkLines(eng, 1, dsk_k0)
    dout2 = eng.alloc(n_items * 384)
    E.ac17_decrypt_prepared_dev(eng, n_items, dc0, dc, eng.upload_u32([i * n_rows for i in range(n_items + 1)]), dcp,
                                lines, dsk_k, eng.upload_u32([0, len(attrs)]), dsk_kp, eng.upload_u32([0] * n_items),
                                eng.upload_u32(ct_sel * n_items), eng.upload_u32([i * len(ct_sel) for i in range(n_items + 1)]),
                                eng.upload_u32(sk_sel * n_items), eng.upload_u32([i * len(sk_sel) for i in range(n_items + 1)]), dout2)
    assert eng.download(dout2) == out
    lines.destroy()
    # round trip: the decrypted Gt is the encrypted msg
    for i, msg in enumerate(msgs):
        assert out[i * 384:(i + 1) * 384] == bn.gt_to_le(msg)
    # and it equals what the reference-order decrypt computes from the same ciphertext bytes (item 0)
    ct = {"policy": (policy, lang),
          "ct": {"c_0": [bn.g2_from_le(c0[j * 128:(j + 1) * 128]) for j in range(3)],
                 "c": [(pi[r], [bn.g1_from_le(c[(r * 3 + l) * 64:(r * 3 + l + 1) * 64]) for l in range(3)]) for r in range(n_rows)],
                 "c_p": bn.gt_from_le(cp[:384])}}
    assert bn.gt_to_le(sch.ac17_cp_decrypt(sk, ct)) == out[:384]


@pytest.mark.parametrize("w_bits", [17, 19, 22, 26])          # 26 = bench.py's default (19 GB table)
def test_cp_encrypt_rows_with_wide_signed_windows(w_bits):
    """signed w-bit fixed-base windows for g (rhip_ac17_pk_set_g_window): the same ciphertext bytes as the
    16-bit tables, which the tests above pin to the oracle.  Scalars with extreme digits included."""
    from rabe_amd import Engine
    from rabe_amd import engine as E
    eng = Engine(0)
    rng = SeededRng(77 + w_bits)
    pk, _msk = sch.ac17_setup(rng)
    dpk = E.Ac17Pk(eng, bn.g1_to_le(pk["g"]), [bn.g2_to_le(x) for x in pk["h_a"]], [bn.gt_to_le(x) for x in pk["e_gh_ka"]])
    policy, lang, _ = POLICIES[2]
    e_gen = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    half = 1 << (w_bits - 1)
    edge = [sum(half << (w_bits * i) for i in range(254 // w_bits)) % bn.R,             # every digit exactly 2^(w-1)
            sum((half + 1) << (w_bits * i) for i in range(254 // w_bits)) % bn.R,       # every digit just above: all negative with carries
            bn.R - 1, 1, (1 << 253) + 12345, 0, 1 << (w_bits * 3), (1 << (w_bits * 9)) + (1 << w_bits)]    # zero digits too
    items = [(rng.fr(), rng.fr()) for _ in range(3)] + [(edge[i], edge[(i + 1) % len(edge)]) for i in range(len(edge))] + [(0, 0)]
    msgs = [bn.gt_pow(e_gen, rng.fr_nonzero()) for _ in items]
    pi, c0, c, cp, _ = gpu_encrypt(eng, E, dpk, policy, lang, items, msgs)
    dpk.set_g_window(w_bits)
    pi2, c0w, cw, cpw, _ = gpu_encrypt(eng, E, dpk, policy, lang, items, msgs)
    assert (pi2, c0w, cpw) == (pi, c0, cp)
    assert cw == c
    # and directly against the oracle for the first item
    ct = sch.ac17_cp_encrypt(pk, policy, lang, ListRng(list(items[0])), msgs[0])
    n_rows = len(pi)
    assert cw[:n_rows * 192] == b"".join(bn.g1_to_le(p) for _, vec in ct["ct"]["c"] for p in vec)
    dpk.destroy()
    eng.close()
