"""Decoding of untrusted keys / ciphertexts: structural checks in rabe_obj_deserialize (fixed-size AC17 vectors,
scalars < r, coordinates < p) and group membership in rabe_obj_deserialize_checked (G1 on the curve, G2 in the r-torsion of the
twist, Gt in the order-r subgroup); one malformed item fails alone inside an AC17 decrypt batch."""
import ctypes

import pytest

from oracle import bn254 as bn
from rabe_amd import hostlib as hl
from rabe_amd.schemes import ac17, bsw

PT = b"dance like no one's watching, encrypt like everyone is!"


def fp2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = bn.fp2_mul(r, a)
        a = bn.fp2_mul(a, a)
        e >>= 1
    return r


def fp2_sqrt(a):
    """square root in Fq2 = Fq[u]/(u^2+1), p = 3 mod 4 (test helper); None for a non-square"""
    a1 = fp2_pow(a, (bn.P - 3) // 4)
    alpha = bn.fp2_mul(bn.fp2_mul(a1, a1), a)
    x0 = bn.fp2_mul(a1, a)
    if alpha == (bn.P - 1, 0):
        x = bn.fp2_mul((0, 1), x0)
    else:
        x = bn.fp2_mul(fp2_pow(bn.fp2_add((1, 0), alpha), (bn.P - 1) // 2), x0)
    return x if bn.fp2_mul(x, x) == (a[0] % bn.P, a[1] % bn.P) else None


def test_structural_checks_need_no_gpu():
    # an Ac17PublicKey whose h_a has 2 elements instead of 3: g | u32 count | ...
    g = bn.g1_to_le(bn.G1_GEN)
    h = bn.g2_to_le(bn.G2_GEN)
    e = bytes(384)
    bad = g + (2).to_bytes(4, "little") + h * 2 + (2).to_bytes(4, "little") + e * 2
    with pytest.raises(hl.RabeError):
        hl.Obj.deserialize("ac17_pk", bad)
    good = g + (3).to_bytes(4, "little") + h * 3 + (2).to_bytes(4, "little") + e * 2
    hl.Obj.deserialize("ac17_pk", good)
    # a coordinate that is not below p (a second encoding of the same residue)
    x_plus_p = (1 + bn.P).to_bytes(32, "little") + (2).to_bytes(32, "little")
    with pytest.raises(hl.RabeError):
        hl.Obj.deserialize("ac17_pk", x_plus_p + good[64:])
    # a scalar that is not below r (bsw msk: beta | g2_alpha)
    with pytest.raises(hl.RabeError):
        hl.Obj.deserialize("bsw_msk", bn.R.to_bytes(32, "little") + h)
    hl.Obj.deserialize("bsw_msk", (bn.R - 1).to_bytes(32, "little") + h)
    # a hostile element count cannot make the reader allocate
    with pytest.raises(hl.RabeError):
        hl.Obj.deserialize("ac17_pk", g + (0xFFFFFFFF).to_bytes(4, "little"))


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


@pytest.mark.gpu
def test_membership_checks(host):
    pk, msk = bsw.setup(host)
    pkb = pk.serialize()
    hl.Obj.deserialize("bsw_pk", pkb, host=host)                       # an honest key passes
    # G1 off the curve
    bad = bytearray(pkb)
    bad[33] ^= 1
    hl.Obj.deserialize("bsw_pk", bytes(bad))                           # the unchecked form cannot know
    with pytest.raises(hl.RabeError):
        hl.Obj.deserialize("bsw_pk", bytes(bad), host=host)
    # a point ON the twist but outside its r-torsion: take x = 1, 2, ... until x^3 + b' is a square in Fq2
    def twist_point():
        bp = bn.fp2_mul((3, 0), bn.fp2_inv((9, 1)))
        x = (1, 0)
        while True:
            rhs = bn.fp2_add(bn.fp2_mul(bn.fp2_mul(x, x), x), bp)
            y = fp2_sqrt(rhs)
            if y is not None:
                return (x, y)
            x = (x[0] + 1, 0)
    q = twist_point()
    assert bn.g2_add(bn.g2_mul(q, bn.R - 1), q) is not None            # r * q != O: not in the r-torsion (the cofactor is huge)
    off = 64                                                           # g1 | g2 | h | f | e_gg_alpha
    bad = pkb[:off] + bn.g2_to_le(q) + pkb[off + 128:]
    with pytest.raises(hl.RabeError):
        hl.Obj.deserialize("bsw_pk", bad, host=host)
    # a Gt value outside the order-r subgroup (an arbitrary Fq12 element)
    gt_off = 64 + 128 + 64 + 128
    bad = pkb[:gt_off] + b"".join((i + 2).to_bytes(32, "little") for i in range(12))
    with pytest.raises(hl.RabeError):
        hl.Obj.deserialize("bsw_pk", bad, host=host)


@pytest.mark.gpu
def test_one_malformed_policy_fails_its_item_only(host):
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, ["A", "B"])
    cts = ac17.cp_encrypt_batch(host, pk, ['"A" and "B"'] * 3, [PT] * 3, hl.HUMAN_POLICY)
    blob = cts[1].serialize()
    # corrupt the policy text of item 1 (u32 length, then the string): it no longer parses
    n = int.from_bytes(blob[:4], "little")
    broken = hl.Obj.deserialize("ac17_cp_ct", blob[:4] + b"(" * n + blob[4 + n:])
    got = ac17.cp_decrypt_batch(host, [sk] * 3, [cts[0], broken, cts[2]])
    assert got[0] == PT and got[2] == PT and got[1] is None


@pytest.mark.gpu
def test_fast_g2_subgroup_test_agrees_with_the_order_test():
    """rhip_g2_in_subgroup uses the BN-specific criterion [u+1]Q + psi([u]Q) + psi^2([u]Q) = psi^3([2u]Q); it must give the verdicts
    of the definition r * Q = O on members, on points of the twist outside G2 (a random twist point almost surely has a component in
    the cofactor), on members shifted by a cofactor-torsion point, on infinity, on points off the twist and on non-canonical encodings."""
    import random
    import struct
    from rabe_amd import Engine
    from rabe_amd.engine import _sz
    rnd = random.Random(99)
    b2 = bn.fp2_mul((3, 0), bn.fp2_inv(bn.XI))
    twist = []
    while len(twist) < 12:
        x = (rnd.randrange(bn.P), rnd.randrange(bn.P))
        y = fp2_sqrt(bn.fp2_add(bn.fp2_mul(bn.fp2_mul(x, x), x), b2))
        if y is not None:
            twist.append((x, y))
    members = [bn.g2_mul(bn.G2_GEN, rnd.randrange(1, bn.R)) for _ in range(12)]
    # r * T lies in the cofactor torsion (T's G2 component is killed): a member plus such a point is on the twist, outside G2
    cof = [bn.g2_add(bn.ec_mul(bn.FP2, t, bn.R - 1), t) for t in twist[:4]]
    assert all(c is not None for c in cof)
    shifted = [bn.g2_add(m, c) for m, c in zip(members, cof)]
    off = [(m[0], bn.fp2_add(m[1], (1, 0))) for m in members[:3]]
    pts = members + twist + cof + shifted + off + [None]
    raw = [bn.g2_to_le(p) for p in pts]
    x0, x1 = members[0][0]
    raw.append((x0 + bn.P).to_bytes(32, "little") + bn.g2_to_le(members[0])[32:])          # x.c0 + p: the same point, second encoding
    want = [1] * 12 + [0] * 12 + [0] * 4 + [0] * 4 + [0] * 3 + [1] + [0]
    eng = Engine(0)
    d = eng.upload(b"".join(raw))
    n = len(raw)
    for fn in ("rhip_g2_in_subgroup", "rhip_g2_in_subgroup_by_order"):
        out = eng.alloc(4 * n)
        eng._check(getattr(eng.lib, fn)(eng.ctx, _sz(n), d.ptr, out.ptr))
        assert list(struct.unpack("<%dI" % n, eng.download(out))) == want, fn
    eng.close()


@pytest.mark.gpu
def test_fast_gt_membership_test_agrees_with_the_order_test():
    """rhip_gt_is_member: cyclotomic test by Frobenius maps, then f^p = f^(6u^2) instead of f^r = 1.  Same verdicts as the definition
    on members, on elements of the cyclotomic subgroup outside Gt (a random Fq12 value raised to the easy part of the final
    exponentiation), on arbitrary Fq12 values, on 1 and on non-canonical encodings."""
    import random
    import struct
    from rabe_amd import Engine
    from rabe_amd.engine import _sz
    rnd = random.Random(98)
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    members = [bn.gt_pow(e, rnd.randrange(1, bn.R)) for _ in range(6)] + [bn.GT_ONE]
    rand12 = [bn.fp12_from_coeffs([rnd.randrange(bn.P) for _ in range(12)]) for _ in range(6)]
    cyc = [bn.fp12_pow(f, bn.FE_EASY) for f in rand12[:4]]                      # order divides p^4 - p^2 + 1, almost surely not r
    mixed = [bn.fp12_mul(m, c) for m, c in zip(members, cyc)]
    vals = members + rand12 + cyc + mixed
    raw = [bn.gt_to_le(v) for v in vals]
    c0 = int.from_bytes(raw[0][:32], "little")
    if c0 + bn.P < 1 << 256:
        raw.append((c0 + bn.P).to_bytes(32, "little") + raw[0][32:])              # a member with a non-canonical first coefficient
    raw.append(bytes(384))                                                          # 0: every Frobenius identity holds for it, yet it is no unit
    want = [1] * 7 + [0] * 6 + [0] * 4 + [0] * 4 + [0] * (len(raw) - 21)
    eng = Engine(0)
    d = eng.upload(b"".join(raw))
    n = len(raw)
    for fn in ("rhip_gt_is_member", "rhip_gt_is_member_by_order"):
        out = eng.alloc(4 * n)
        eng._check(getattr(eng.lib, fn)(eng.ctx, _sz(n), d.ptr, out.ptr))
        assert list(struct.unpack("<%dI" % n, eng.download(out))) == want, fn
    eng.close()
