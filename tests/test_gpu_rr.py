"""The reduced-radix Miller kernel (rabe_amd/csrc/engine_rr.hip: k_miller_multi_rr, 9 signed 29-bit limbs) against the 8 x 32-bit kernel
(k_miller_multi) on the same inputs -- the bytes must be identical -- and against the Python oracle."""
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

from oracle import bn254 as bn  # noqa: E402

RND = random.Random(2929)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    from rabe_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def _pairs(n):
    ks = [(RND.randrange(1, bn.R), RND.randrange(1, bn.R)) for _ in range(n)]
    p = [bn.g1_to_le(bn.g1_mul(bn.G1_GEN, a)) for a, _ in ks]
    q = [bn.g2_to_le(bn.g2_mul(bn.G2_GEN, b)) for _, b in ks]
    return ks, p, q


def _both_modes(eng, fn):
    eng.set_pairing_mode(1)
    a = fn()
    eng.set_pairing_mode(29)
    b = fn()
    eng.set_pairing_mode(0)
    return a, b


@pytest.mark.parametrize("shape", [[1], [6], [7], [2, 0, 5], [13, 1, 6, 12], [40]])
def test_pairing_jobs_reduced_radix_equal_8x32_and_oracle(eng, shape):
    """item i: product of shape[i] pairings (walking pairs), an empty item gives 1; a lead factor multiplies in"""
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    off = [0]
    for s in shape:
        off.append(off[-1] + s)
    ks, p, q = _pairs(off[-1])
    lead_k = [RND.randrange(1, bn.R) for _ in shape]
    lead = [bn.gt_to_le(bn.gt_pow(e, k)) for k in lead_k]
    one, rr = _both_modes(eng, lambda: eng.pairing_jobs(off, p, q, lead=lead))
    assert one == rr
    for i, s in enumerate(shape):
        exp = (sum(a * b for a, b in ks[off[i]:off[i + 1]]) + lead_k[i]) % bn.R
        assert rr[i] == bn.gt_to_le(bn.gt_pow(e, exp))


def test_many_items_with_arguments_at_infinity(eng):
    """700 items of 6 pairs (several four-wave blocks, the last one partly filled), arguments at infinity mixed in"""
    n, c = 700, 6
    ks, p, q = _pairs(8)
    idx = [(RND.randrange(8), RND.randrange(8)) for _ in range(n * c)]
    pp = [p[a] for a, _ in idx]
    qq = [q[b] for _, b in idx]
    for t in (3, 77, 500, 4100):
        pp[t] = bytes(64)
    qq[91] = bytes(128)
    off = [c * i for i in range(n + 1)]
    one, rr = _both_modes(eng, lambda: eng.pairing_jobs(off, pp, qq))
    assert one == rr
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    for i in (0, 12, 15, 83, 683, 699):
        exp = sum(ks[a][0] * ks[b][1] for t, (a, b) in enumerate(idx[c * i:c * i + c], start=c * i) if t not in (3, 77, 500, 4100, 91)) % bn.R
        assert rr[i] == bn.gt_to_le(bn.gt_pow(e, exp))


# The scheme suites under THIS family of kernels: since round 6 the whole GPU suite runs in the engine's cross-check mode (tests/conftest.py:
# RABE_PAIRING_MODE=99) -- every pairing launch of test_gpu_ac17 / _bsw_dev / _lsw_aw11_dev / _ghw11 / _ragged_plan / _fullsize_parity /
# _configs runs with the automatic selection AND with this family forced, compared byte for byte on the device -- so the modules are no
# longer re-run in subprocesses (that recomputed the Python oracle per family: 150 s per family).  What the cross-check does not cover is
# the walk verdicts, which are read off the points THIS family's Miller kernel ends on: that module is still re-run with the family forced.
@pytest.mark.parametrize("module", ["tests/test_gpu_walk_verdicts.py"])
def test_scheme_suites_pass_with_the_reduced_radix_kernel_forced(module):
    """the scheme-level GPU tests (every byte against the oracle / the golden fixtures / the reference-order port) with RABE_PAIRING_MODE=29:
    every multi-pairing launch of every decrypt goes through k_miller_multi_rr -- prepared lines, walking pairs, ragged plans and the
    walk verdicts read off its final points included -- at every size, the five BASELINE configurations at full size among them"""
    env = dict(os.environ, RABE_PAIRING_MODE="29")
    r = subprocess.run([sys.executable, "-m", "pytest", module, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
