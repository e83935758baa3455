"""Exact-integer check of the carry-capture plan of the device field multiplication (rabe_amd/csrc/bn254/fp.h).

The gfx950 multiplication accumulates each column of limb products in a 64-bit register pair and banks a carry-out only for
the products the plan marks as able to overflow.  This test takes the plan from the very header the kernels are compiled from
(through tests/hostsim) and recomputes, with Python integers and worst-case limbs, the largest value the accumulator can hold
before every product that issues WITHOUT a carry capture: it must stay below 2^64.  No GPU needed; the adversarial-limb
comparison on the device is tests/test_gpu_field_edge.py."""
import ctypes

import pytest

from oracle import bn254 as bn
from tests.hostsim import build as hs_build

try:
    HS = hs_build.load()
except Exception:  # pragma: no cover
    HS = None

pytestmark = pytest.mark.skipif(HS is None, reason="hostsim library could not be built")

W = (1 << 32) - 1          # largest limb
FULL = (1 << 64) - 1


def limbs(x):
    return [(x >> (32 * i)) & W for i in range(8)]


def plan(field, kind):
    out = (ctypes.c_uint16 * 17)()
    HS.hs_column_plan(field, kind, out)
    return list(out[:16]), out[16]


def top_products(k):
    """indices i of the a_i * b_(k-i) products that involve a top limb (a_7 or b_7)"""
    lo, hi = (0, k) if k < 8 else (k - 7, 7)
    return [i for i in range(lo, hi + 1) if k >= 7 and (i == 7 or k - i == 7)]


@pytest.mark.parametrize("field,mod", [(0, bn.P), (1, bn.R)])
def test_reduction_plan_never_overflows(field, mod):
    """redc2: column k = incoming + W[k] + sum m_i * p_(k-i) (+ m_k * p_0 for k < 8)."""
    p = limbs(mod)
    safe, last = plan(field, 0)
    incoming = 0
    n_uncaptured = 0
    for k in range(16):
        lo, hi = (0, k - 1) if k < 8 else (k - 7, 7)
        acc = incoming + W                                   # + W[k], issued without capture
        assert acc <= FULL
        banked = 0
        for i in range(lo, hi + 1):
            if (safe[k] >> i) & 1:
                acc += W * p[k - i]
                assert acc <= FULL, (k, i)
                n_uncaptured += 1
        for i in range(lo, hi + 1):
            if not (safe[k] >> i) & 1:
                acc += W * p[k - i]; banked += 1
        if k < 8:
            if (last >> k) & 1:
                assert banked == 0, "the closing product may skip the capture only in a column without banked products"
                acc += W * p[0]
                assert acc <= FULL, k
                n_uncaptured += 1
            else:
                acc += W * p[0]
        incoming = acc >> 32                                  # high word + banked carries of the true (unbounded) sum
        assert incoming < 1 << 37
    assert n_uncaptured >= 16                                 # the plan actually saves something


@pytest.mark.parametrize("field,mod,scale", [(0, bn.P, 1), (1, bn.R, 1), (0, bn.P, 2)])
def test_product_plan_never_overflows(field, mod, scale):
    """mont_mul{,2,3}_raw: operands < scale * mod (scale 2: the a0 + a1 sums of the lazy Fq2 product use the same top-limb rule
    in wide_mul3, which has no reduction products: checked with an empty reduction plan)."""
    p = limbs(mod)
    top = ((scale * mod - 1) >> 224)                          # largest top limb of an operand
    safe, last = plan(field, 1) if scale == 1 else ([0] * 16, 0)
    incoming = 0
    for k in range(16 if scale == 1 else 15):
        lo, hi = (0, k) if k < 8 else (k - 7, 7)
        tops = top_products(k)
        acc = incoming
        for i in tops:                                        # top-limb products first, uncaptured
            acc += (top * top) if (i == 7 and k - i == 7) else top * W
            assert acc <= FULL, (k, i)
        banked = 0
        if scale == 1:
            for i in range(lo, hi + 1):
                if not (k < 8 and i == k) and (safe[k] >> i) & 1:
                    acc += W * p[k - i]
                    assert acc <= FULL, (k, i)
        for i in range(lo, hi + 1):
            if i not in tops:
                acc += W * W; banked += 1
        if scale == 1:
            for i in range(lo, hi + 1):
                if not (k < 8 and i == k) and not (safe[k] >> i) & 1:
                    acc += W * p[k - i]; banked += 1
            if k < 8:
                if (last >> k) & 1:
                    assert banked == 0
                    acc += W * p[0]
                    assert acc <= FULL, k
                else:
                    acc += W * p[0]
        incoming = acc >> 32
        assert incoming < 1 << 37


def test_unbounded_second_operand_rule():
    """to_mont / to_mont_reduce256 (B_REDUCED = false): only a_7 * b_(k-7) skips the capture; b is any 256-bit value."""
    for mod in (bn.P, bn.R):
        top = (mod - 1) >> 224
        incoming = (1 << 37) - 1
        assert incoming + top * W <= FULL


def test_generated_header_is_current(tmp_path, monkeypatch):
    """rabe_amd/csrc/bn254/fp_gfx950_gen.h is what tools/gen_fp_asm.py writes today (its static_asserts tie it to ColumnPlan)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_fp_asm", os.path.join(root, "tools", "gen_fp_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(root, "rabe_amd", "csrc", "bn254", "fp_gfx950_gen.h")).read()
    real_open = open
    written = {}

    class Capture:
        def __init__(self, path):
            self.path = path
        def __enter__(self):
            return self
        def __exit__(self, *a):
            return False
        def write(self, text):
            written[self.path] = written.get(self.path, "") + text

    def fake_open(path, mode="r", *a, **k):
        if "w" in mode:
            return Capture(path)
        return real_open(path, mode, *a, **k)

    monkeypatch.setattr(gen, "open", fake_open, raising=False)
    gen.main()
    assert list(written.values()) == [committed]
    # and the generator's plan is the header's plan (the compile-time static_asserts say the same)
    mod = gen.fp_mod()
    safe, last = plan(0, 0)
    assert gen.column_plan(mod, False, 0, 0) == (safe, last)
    safe, last = plan(0, 1)
    assert gen.column_plan(mod, True, mod[7] + 1, mod[7] + 1) == (safe, last)


def test_wide_mac3_uncaptured_products_and_sum_bounds():
    """wide_mac3 (tools/gen_fp_asm.py; the multi-product sums of bn254/coop6.h): a column of the accumulate form is
    incoming + T[k] (a multiply-add by 1, no capture) + the column's TOP-limb products (no capture) + the captured rest.  With exact
    integers and worst-case limbs: nothing that issues without a capture can leave the 64-bit accumulator -- for operands below p and for
    the Karatsuba sums below 2p -- and the sums themselves stay below 2^512 for the six products an operation accumulates at most."""
    p = bn.P
    for bound_a, bound_b in ((p - 1, p - 1), (2 * p - 2, 2 * p - 2)):
        top_a, top_b = bound_a >> 224, bound_b >> 224                       # largest top limbs
        incoming = 0
        for k in range(16):
            lo, hi = (0, k) if k < 8 else (k - 7, 7)
            tops = top_products(k) if k < 15 else []
            acc = incoming + W                                               # + T[k]
            assert acc <= FULL
            for i in tops:
                acc += (top_a if i == 7 else W) * (top_b if k - i == 7 else W)
                assert acc <= FULL, (k, i)
            banked = 0
            for i in range(lo, hi + 1):
                if k < 15 and i not in tops:
                    acc += W * W
                    banked += acc >> 64
                    acc &= FULL
            incoming = (acc >> 32) + (banked << 32)                          # the next column starts from the high word + the banked carries
            assert incoming < 1 << 37
    # three sums of at most six products: T0, T1 < 6 p^2, T2 < 6 (2p)^2 = 24 p^2 < 2^512; the reduction's inputs are below 12 p^2 and
    # leave less than 3.27 p (three conditional subtractions bring that below p)
    assert 24 * p * p < 1 << 512
    assert (12 * p * p) // (1 << 256) + p < 3.27 * p
    for n, extra0, extra1 in ((2, 1, 0), (3, 1, 1), (4, 1, 1), (6, 2, 2)):    # coop6.h: wide3_finish's conditional subtractions by n
        c0_max = ((n + 6) * p * p) // (1 << 256) + p                           # REDC output bound: (W + m p) / R < W / R + p
        c1_max = (2 * n * p * p) // (1 << 256) + p
        assert c0_max < (2 + extra0) * p and c1_max < (2 + extra1) * p
