"""The three-lane cooperative pairing kernels (bn254/coop3.h) give bit-identical results to the one-lane kernels
and to the oracle: pairing, pairing products (ragged), AC17 decrypt."""
import random

import pytest

from oracle import bn254 as bn

pytestmark = pytest.mark.gpu
RND = random.Random(31337)


@pytest.fixture(scope="module")
def eng():
    from rabe_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def test_pairing_modes_agree_with_oracle(eng):
    e_gen = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    ks = [(RND.randrange(1, bn.R), RND.randrange(1, bn.R)) for _ in range(50)]      # 50 pairs: more than two waves of triples
    P = [bn.g1_to_le(bn.g1_mul(bn.G1_GEN, a)) for a, _ in ks[:4]] * 13
    Q = [bn.g2_to_le(bn.g2_mul(bn.G2_GEN, b)) for _, b in ks[:4]] * 13
    P[5], Q[7] = bytes(64), bytes(128)                                              # infinity inputs inside a wave
    want = [bn.gt_to_le(bn.gt_pow(e_gen, a * b % bn.R)) for a, b in ks[:4]] * 13
    want[5] = want[7] = bn.gt_to_le(bn.GT_ONE)
    out = {}
    for mode in (1, 3):
        eng.set_pairing_mode(mode)
        out[mode] = eng.pairing(P, Q)
        assert out[mode] == want
        offs = [0, 3, 3, 10, 52]
        out[(mode, "prod")] = eng.pairing_product(offs, P, Q)
    eng.set_pairing_mode(0)
    assert out[(1, "prod")] == out[(3, "prod")]
    assert out[(3, "prod")][1] == bn.gt_to_le(bn.GT_ONE)                            # empty product
