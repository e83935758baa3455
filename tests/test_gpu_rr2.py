"""The two-lane reduced-radix Miller kernel (rabe_amd/csrc/engine_rr2.hip: k_miller_pair_rr -- one (item, chunk) unit on two adjacent lanes, two waves
per SIMD; pairing mode 58) against the one-lane kernels on the same inputs -- the bytes must be identical -- and against the Python oracle.
The whole GPU suite cross-checks this family on every pairing launch as well (tests/conftest.py: mode 99)."""
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

from oracle import bn254 as bn  # noqa: E402

RND = random.Random(5858)
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


def _modes(eng, fn, modes=(1, 29, 58)):
    out = []
    for m in modes:
        eng.set_pairing_mode(m)
        out.append(fn())
    eng.set_pairing_mode(0)
    return out


@pytest.mark.parametrize("shape", [[1], [2], [3], [6], [7], [2, 0, 5], [13, 1, 6, 12], [40], [64, 63, 1]])
def test_pairing_jobs_two_lanes_equal_one_lane_and_oracle(eng, shape):
    """item i: product of shape[i] pairings (walking pairs: odd and even counts, so that lane 1 of a pair is idle in a chunk's last round), an
    empty item gives 1; a lead factor multiplies in"""
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    off = [0]
    for s in shape:
        off.append(off[-1] + s)
    ks, p, q = _pairs(off[-1])
    lead_k = [RND.randrange(1, bn.R) for _ in shape]
    lead = [bn.gt_to_le(bn.gt_pow(e, k)) for k in lead_k]
    one, rr, two = _modes(eng, lambda: eng.pairing_jobs(off, p, q, lead=lead))
    assert one == rr == two
    for i, s in enumerate(shape):
        exp = (sum(a * b for a, b in ks[off[i]:off[i + 1]]) + lead_k[i]) % bn.R
        assert two[i] == bn.gt_to_le(bn.gt_pow(e, exp))


def test_many_items_with_arguments_at_infinity(eng):
    """700 items of 6 pairs (several blocks, the last one partly filled), arguments at infinity mixed in (skipped pairs shift the lanes'
    shares of the walking pairs)"""
    n, c = 700, 6
    ks, p, q = _pairs(8)
    idx = [(RND.randrange(8), RND.randrange(8)) for _ in range(n * c)]
    pp = [p[a] for a, _ in idx]
    qq = [q[b] for _, b in idx]
    for t in (3, 77, 500, 4100):
        pp[t] = bytes(64)
    qq[91] = bytes(128)
    off = [c * i for i in range(n + 1)]
    one, two = _modes(eng, lambda: eng.pairing_jobs(off, pp, qq), modes=(1, 58))
    assert one == two
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    for i in (0, 12, 15, 83, 683, 699):
        exp = sum(ks[a][0] * ks[b][1] for t, (a, b) in enumerate(idx[c * i:c * i + c], start=c * i) if t not in (3, 77, 500, 4100, 91)) % bn.R
        assert two[i] == bn.gt_to_le(bn.gt_pow(e, exp))


@pytest.mark.parametrize("module", ["tests/test_gpu_walk_verdicts.py"])
def test_scheme_suites_pass_with_the_two_lane_kernel_forced(module):
    """walk verdicts are read off the points THIS kernel's lanes end on (the cross-check mode compares pairing values only): that module with
    RABE_PAIRING_MODE=58.  (Prepared + walking pairs side by side in one chunk -- AC17 -- run through this kernel in the cross-check mode.)"""
    env = dict(os.environ, RABE_PAIRING_MODE="58")
    r = subprocess.run([sys.executable, "-m", "pytest", module, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
