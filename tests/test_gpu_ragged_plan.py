"""Ragged pair lists (items of different pair counts in one batch) run k_miller_multi from a work list planned on the device
(engine_jobs.hip: k_plan_hist / k_plan_choose / k_plan_scan / k_plan_fill): chunk size chosen from the batch's histogram, the chunks that
exist sorted by size.  The values must be those of the one-lane-per-pair kernels (k_miller + k_final_exp, pinned to the oracle in
test_gpu_batches.py / test_gpu_elements.py) bit for bit, for every shape a list can take: empty items, one pair, more pairs than a chunk
holds, items in any order -- and the same with the plan switched off (RABE_NO_MILLER_PLAN: the static chunk rows)."""
import os
import random

import pytest

pytestmark = pytest.mark.gpu


def _elements(n_distinct, seed):
    from oracle import bn254 as bn
    rnd = random.Random(seed)
    ks = [rnd.randrange(1, bn.R) for _ in range(n_distinct)]
    P = [bn.g1_to_le(bn.g1_mul(bn.G1_GEN, k)) for k in ks]
    Q = [bn.g2_to_le(bn.g2_mul(bn.G2_GEN, k + 1)) for k in ks]
    return rnd, P, Q


@pytest.mark.parametrize("counts_kind", ["tiny", "mixed", "long_tail"])
def test_ragged_pair_lists_match_the_one_lane_kernels(counts_kind):
    from rabe_amd import Engine
    rnd, P, Q = _elements(11, 7)
    if counts_kind == "tiny":            # fewer lanes than one wave; an empty item first and last
        counts = [0, 1, 3, 2, 0, 5, 1, 0]
    elif counts_kind == "mixed":         # a few hundred items of 0 .. 40 pairs in random order
        counts = [rnd.choice([0, 1, 2, 3, 6, 7, 13, 20, 33, 40]) for _ in range(300)]
    else:                                # many short items and a handful far beyond one chunk (> 64 pairs)
        counts = [rnd.choice([1, 2, 4]) for _ in range(500)] + [150, 97, 65, 64, 129]
        rnd.shuffle(counts)
    off = [0]
    for c in counts:
        off.append(off[-1] + c)
    n = off[-1]
    ps, qs = [P[rnd.randrange(11)] for _ in range(n)], [Q[rnd.randrange(11)] for _ in range(n)]
    eng = Engine(0)
    eng.set_pairing_mode(1)
    want = eng.pairing_product(off, ps, qs)
    os.environ.pop("RABE_NO_MILLER_PLAN", None)
    got = eng.pairing_jobs(off, ps, qs)
    os.environ["RABE_NO_MILLER_PLAN"] = "1"
    try:
        got_static = eng.pairing_jobs(off, ps, qs)
    finally:
        os.environ.pop("RABE_NO_MILLER_PLAN", None)
    assert got == want
    assert got_static == want
    # an empty item is the unit of Gt
    from oracle import bn254 as bn
    for i, c in enumerate(counts):
        if c == 0:
            assert got[i] == bn.gt_to_le(bn.GT_ONE)
            break
    eng.close()


def test_ragged_lists_with_scalars_and_leading_factors():
    """the general job form on a ragged batch: out[i] = lead[i] * FE(prod ML(scal * base, q)); bilinearity ties it to the plain product"""
    from rabe_amd import Engine
    from oracle import bn254 as bn
    rnd, P, Q = _elements(5, 11)
    counts = [rnd.choice([1, 2, 9, 30]) for _ in range(90)]
    off = [0]
    for c in counts:
        off.append(off[-1] + c)
    n = off[-1]
    pi, qi = [rnd.randrange(5) for _ in range(n)], [rnd.randrange(5) for _ in range(n)]
    ps, qs = [P[i] for i in pi], [Q[i] for i in qi]
    scal = [rnd.randrange(1, bn.R) for _ in range(n)]
    eng = Engine(0)
    eng.set_pairing_mode(1)
    lead = eng.pairing_product(list(range(len(counts) + 1)), [P[0]] * len(counts), [Q[1]] * len(counts))
    got = eng.pairing_jobs(off, ps, qs, scal=[s.to_bytes(32, "little") for s in scal], lead=lead)
    # the same through scaled points prepared by the element-level kernels
    scaled = eng.g1_mul(ps, [s.to_bytes(32, "little") for s in scal])
    plain = eng.pairing_jobs(off, scaled, qs, lead=lead)
    assert got == plain
    eng.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_shapes_match_the_one_lane_kernels(seed):
    """random batch sizes and pair-count distributions (all equal but one, geometric, bimodal, a few giants): the planned launch against
    k_miller + k_final_exp, item by item"""
    from rabe_amd import Engine
    rnd, P, Q = _elements(9, 100 + seed)
    n_items = rnd.choice([1, 2, 63, 64, 65, 200, 513, 700])
    kind = seed % 4
    if kind == 0:
        counts = [5] * n_items
        counts[rnd.randrange(n_items)] = 6                       # uniform but for one item
    elif kind == 1:
        counts = [min(200, int(rnd.expovariate(0.15))) for _ in range(n_items)]
    elif kind == 2:
        counts = [rnd.choice([2, 2, 2, 90]) for _ in range(n_items)]
    else:
        counts = [rnd.randrange(0, 4) for _ in range(n_items)]
        for _ in range(min(3, n_items)):
            counts[rnd.randrange(n_items)] = rnd.randrange(130, 260)
    if sum(counts) == 0:
        counts[0] = 1
    off = [0]
    for c in counts:
        off.append(off[-1] + c)
    n = off[-1]
    ps, qs = [P[rnd.randrange(9)] for _ in range(n)], [Q[rnd.randrange(9)] for _ in range(n)]
    eng = Engine(0)
    eng.set_pairing_mode(1)
    assert eng.pairing_jobs(off, ps, qs) == eng.pairing_product(off, ps, qs)
    eng.close()
