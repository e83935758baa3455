"""The six-lane cooperative pairing kernels (rabe_amd/csrc/engine_coop.hip: k_miller_c6, k_final_exp_c6) against the one-lane kernels
(k_miller_multi, k_final_exp) on the same inputs -- the bytes must be identical -- and against the Python oracle."""
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

from oracle import bn254 as bn  # noqa: E402

RND = random.Random(605)
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
    eng.set_pairing_mode(6)
    b = fn()
    eng.set_pairing_mode(0)
    return a, b


@pytest.mark.parametrize("shape", [[1], [6], [7], [2, 0, 5], [13, 1, 6, 12], [40]])
def test_pairing_jobs_six_lanes_equal_one_lane_and_oracle(eng, shape):
    """item i: product of shape[i] pairings (walking pairs), an empty item gives 1; a lead factor multiplies in"""
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    off = [0]
    for s in shape:
        off.append(off[-1] + s)
    ks, p, q = _pairs(off[-1])
    lead_k = [RND.randrange(1, bn.R) for _ in shape]
    lead = [bn.gt_to_le(bn.gt_pow(e, k)) for k in lead_k]
    one, six = _both_modes(eng, lambda: eng.pairing_jobs(off, p, q, lead=lead))
    assert one == six
    for i, s in enumerate(shape):
        exp = (sum(a * b for a, b in ks[off[i]:off[i + 1]]) + lead_k[i]) % bn.R
        assert six[i] == bn.gt_to_le(bn.gt_pow(e, exp))


def test_many_items_fill_several_waves(eng):
    """130 items of 6 pairs (13 waves of ten groups, the last one partly filled), arguments at infinity mixed in"""
    n, c = 130, 6
    ks, p, q = _pairs(8)
    idx = [(RND.randrange(8), RND.randrange(8)) for _ in range(n * c)]
    pp = [p[a] for a, _ in idx]
    qq = [q[b] for _, b in idx]
    for t in (3, 77, 500):
        pp[t] = bytes(64)
    qq[91] = bytes(128)
    off = [c * i for i in range(n + 1)]
    one, six = _both_modes(eng, lambda: eng.pairing_jobs(off, pp, qq))
    assert one == six
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    for i in (0, 12, 15, 83, 129):
        exp = sum(ks[a][0] * ks[b][1] for t, (a, b) in enumerate(idx[c * i:c * i + c], start=c * i) if t not in (3, 77, 500, 91)) % bn.R
        assert six[i] == bn.gt_to_le(bn.gt_pow(e, exp))


def test_fixed_base_gt_powers_six_lanes_equal_one_lane_and_oracle(eng):
    """rhip_gt_table_pow through k_gt_table_pow_c6 (one running product per group of six lanes) and through k_table_pow_gt"""
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    base = bn.gt_pow(e, 0xABCDEF12345)
    tbl = eng.gt_table(bn.gt_to_le(base))
    ks = [0, 1, 2, 255, 256, bn.R - 1, (1 << 253) + 5] + [RND.randrange(bn.R) for _ in range(30)]
    sc = [int(k).to_bytes(32, "little") for k in ks]
    one, six = _both_modes(eng, lambda: tbl.mul(sc))
    tbl.destroy()
    assert one == six
    for k, got in list(zip(ks, six))[:12]:
        assert got == bn.gt_to_le(bn.gt_pow(base, k))


# The scheme suites under THIS family of kernels: since round 6 the whole GPU suite runs in the engine's cross-check mode (tests/conftest.py:
# RABE_PAIRING_MODE=99) -- every pairing launch of test_gpu_ac17 / _bsw_dev / _lsw_aw11_dev / _ghw11 / _ragged_plan / _fullsize_parity /
# _configs runs with the automatic selection AND with this family forced, compared byte for byte on the device -- so the modules are no
# longer re-run in subprocesses (that recomputed the Python oracle per family: 150 s per family).  What the cross-check does not cover is
# the walk verdicts, which are read off the points THIS family's Miller kernel ends on: that module is still re-run with the family forced.
@pytest.mark.parametrize("module", ["tests/test_gpu_walk_verdicts.py"])
def test_scheme_suites_pass_with_six_lane_kernels_forced(module):
    """the scheme-level GPU tests (every byte against the oracle / the golden fixtures) with RABE_PAIRING_MODE=6: every pairing
    product of every decrypt goes through k_miller_c6 + k_final_exp_c6, prepared lines, walking pairs, ragged plans and walk verdicts
    included -- and the five BASELINE configurations at their full attribute counts and per-GPU batches (test_gpu_fullsize_parity: every
    byte against the reference-order port; test_gpu_configs: full batches through the host layer)"""
    env = dict(os.environ, RABE_PAIRING_MODE="6")
    r = subprocess.run([sys.executable, "-m", "pytest", module, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
