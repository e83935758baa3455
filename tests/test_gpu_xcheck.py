"""The engine's cross-check pairing mode (include/rabe_hip.h: rhip_ctx_set_pairing_mode 99; engine_jobs.hip: run_pair_lists): a launch runs as
the automatic selection runs it, then with each family of pairing kernels forced, and the results are compared on the device.  The GPU suite
runs in this mode (tests/conftest.py) -- this module checks the mode itself: it returns the automatic selection's bytes, it agrees with the
oracle, and it NOTICES a wrong result in any family (a bit flipped in that family's output before the comparison)."""
import random

import pytest

pytestmark = pytest.mark.gpu

from oracle import bn254 as bn  # noqa: E402

RND = random.Random(9901)


@pytest.fixture(scope="module")
def eng():
    from rabe_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def _job(n_items, pairs_each):
    ks = [(RND.randrange(1, bn.R), RND.randrange(1, bn.R)) for _ in range(5)]
    p = [bn.g1_to_le(bn.g1_mul(bn.G1_GEN, a)) for a, _ in ks]
    q = [bn.g2_to_le(bn.g2_mul(bn.G2_GEN, b)) for _, b in ks]
    idx = [(RND.randrange(5), RND.randrange(5)) for _ in range(n_items * pairs_each)]
    off = [pairs_each * i for i in range(n_items + 1)]
    return ks, idx, off, [p[a] for a, _ in idx], [q[b] for _, b in idx]


@pytest.mark.parametrize("n_items,pairs_each", [(3, 4), (200, 6)])
def test_cross_check_returns_the_automatic_result_and_matches_the_oracle(eng, n_items, pairs_each):
    ks, idx, off, pp, qq = _job(n_items, pairs_each)
    eng.set_pairing_mode(0)
    auto = eng.pairing_jobs(off, pp, qq)
    eng.set_pairing_mode(99)
    checked = eng.pairing_jobs(off, pp, qq)
    assert checked == auto
    e = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    for i in (0, n_items - 1):
        exp = sum(ks[a][0] * ks[b][1] for a, b in idx[off[i]:off[i + 1]]) % bn.R
        assert checked[i] == bn.gt_to_le(bn.gt_pow(e, exp))


@pytest.mark.parametrize("family,name", [(1, "one-lane 8x32"), (6, "six-lane"), (29, "reduced-radix")])
def test_cross_check_notices_a_wrong_result_in_each_family(eng, family, name, monkeypatch):
    from rabe_amd import EngineError
    _ks, _idx, off, pp, qq = _job(70, 3)
    eng.set_pairing_mode(99)
    monkeypatch.setenv("RABE_XCHECK_FAULT", str(family))
    with pytest.raises(EngineError) as err:
        eng.pairing_jobs(off, pp, qq)
    assert "cross-check" in str(err.value) and name in str(err.value)
    assert all(other not in str(err.value) for other in ("one-lane 8x32", "six-lane", "reduced-radix") if other != name)
    monkeypatch.delenv("RABE_XCHECK_FAULT")
    assert eng.pairing_jobs(off, pp, qq)          # and the context goes on working, still in mode 99
    eng.set_pairing_mode(0)


def test_the_suite_runs_in_cross_check_mode():
    """tests/conftest.py presets RABE_PAIRING_MODE=99 for every context this process (and its subprocesses) creates"""
    import os
    assert os.environ.get("RABE_PAIRING_MODE") in ("99", "1", "3", "6", "29"), "conftest.py no longer selects a pairing mode for the GPU suite"
