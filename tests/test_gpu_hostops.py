"""Host-value forms of the Level E operators (rhip_host_*: one element per call, host pointers) -- what the `rabe_bn`
replacement crate in integration/rabe-bn-shim/ binds -- against the big-integer oracle."""
import ctypes
import hashlib
import random

import pytest

from oracle import bn254 as bn

pytestmark = pytest.mark.gpu
RND = random.Random(4242)


@pytest.fixture(scope="module")
def eng():
    from rabe_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def call(eng, fn, *ins, out):
    o = ctypes.create_string_buffer(out)
    eng._check(getattr(eng.lib, fn)(eng.ctx, *[x if not isinstance(x, bytes) else ctypes.c_char_p(x) for x in ins], o))
    return o.raw


def test_host_value_operators(eng):
    a, b = RND.randrange(1, bn.R), RND.randrange(1, bn.R)
    le = bn.fr_to_le
    i32 = ctypes.c_int32
    assert call(eng, "rhip_host_fr_op", i32(0), le(a), le(b), out=32) == le((a + b) % bn.R)
    assert call(eng, "rhip_host_fr_op", i32(1), le(a), le(b), out=32) == le((a - b) % bn.R)
    assert call(eng, "rhip_host_fr_op", i32(2), le(a), le(b), out=32) == le(a * b % bn.R)
    assert call(eng, "rhip_host_fr_op", i32(3), le(a), None, out=32) == le((-a) % bn.R)
    assert call(eng, "rhip_host_fr_op", i32(4), le(a), None, out=32) == le(pow(a, bn.R - 2, bn.R))
    d = hashlib.sha3_256(b"attribute").digest()
    assert call(eng, "rhip_host_fr_from_be32_reduce", d, out=32) == le(bn.fr_from_be32_reduce(d))
    p1, p2 = bn.g1_mul(bn.G1_GEN, a), bn.g1_mul(bn.G1_GEN, b)
    q1, q2 = bn.g2_mul(bn.G2_GEN, a), bn.g2_mul(bn.G2_GEN, b)
    assert call(eng, "rhip_host_g1_add", bn.g1_to_le(p1), bn.g1_to_le(p2), out=64) == bn.g1_to_le(bn.g1_add(p1, p2))
    assert call(eng, "rhip_host_g1_neg", bn.g1_to_le(p1), out=64) == bn.g1_to_le(bn.g1_neg(p1))
    assert call(eng, "rhip_host_g1_mul", bn.g1_to_le(p1), le(b), out=64) == bn.g1_to_le(bn.g1_mul(p1, b))
    assert call(eng, "rhip_host_g2_add", bn.g2_to_le(q1), bn.g2_to_le(q2), out=128) == bn.g2_to_le(bn.g2_add(q1, q2))
    assert call(eng, "rhip_host_g2_neg", bn.g2_to_le(q1), out=128) == bn.g2_to_le(bn.g2_neg(q1))
    assert call(eng, "rhip_host_g2_mul", bn.g2_to_le(q1), le(b), out=128) == bn.g2_to_le(bn.g2_mul(q1, b))
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    e1 = call(eng, "rhip_host_pairing", bn.g1_to_le(p1), bn.g2_to_le(q2), out=384)
    assert e1 == bn.gt_to_le(bn.gt_pow(e, a * b % bn.R))
    e2 = bn.gt_to_le(bn.gt_pow(e, b))
    assert call(eng, "rhip_host_gt_mul", e1, e2, out=384) == bn.gt_to_le(bn.gt_pow(e, (a * b + b) % bn.R))
    assert call(eng, "rhip_host_gt_inv", e2, out=384) == bn.gt_to_le(bn.gt_pow(e, (-b) % bn.R))
    assert call(eng, "rhip_host_gt_pow", e2, le(a), out=384) == e1


def test_fr_pow_and_membership(eng):
    """`Fr::pow(Fr)` (src/utils/secretsharing/mod.rs:218) and the curve-membership test behind `FieldError::NotMember`"""
    le = bn.fr_to_le
    for a, e in [(3, 0), (3, 1), (5, 77), (RND.randrange(bn.R), RND.randrange(bn.R)), (0, 5), (bn.R - 1, bn.R - 1), (RND.randrange(bn.R), 99)]:
        assert call(eng, "rhip_host_fr_pow", le(a), le(e), out=32) == le(pow(a, e, bn.R)), (a, e)
    ok = ctypes.c_int32(7)
    p = bn.g1_to_le(bn.g1_mul(bn.G1_GEN, 12345))
    eng._check(eng.lib.rhip_host_g1_on_curve(eng.ctx, ctypes.c_char_p(p), ctypes.byref(ok)))
    assert ok.value == 1
    bad = p[:32] + (int.from_bytes(p[32:], "little") ^ 1).to_bytes(32, "little")
    eng._check(eng.lib.rhip_host_g1_on_curve(eng.ctx, ctypes.c_char_p(bad), ctypes.byref(ok)))
    assert ok.value == 0
    q = bn.g2_to_le(bn.g2_mul(bn.G2_GEN, 54321))
    eng._check(eng.lib.rhip_host_g2_on_curve(eng.ctx, ctypes.c_char_p(q), ctypes.byref(ok)))
    assert ok.value == 1
    badq = q[:96] + (int.from_bytes(q[96:], "little") ^ 1).to_bytes(32, "little")
    eng._check(eng.lib.rhip_host_g2_on_curve(eng.ctx, ctypes.c_char_p(badq), ctypes.byref(ok)))
    assert ok.value == 0
