"""Tape-driven `setup` parity for all five schemes (SURVEY.md row A4 and its siblings): the host layer's setup functions on
the oracle's recorded tape produce the oracle's public / master keys element for element (ac17/mod.rs:141-182,
bsw/mod.rs:92-114, lsw/mod.rs:86-110, aw11/mod.rs:100-151, ghw11/mod.rs:92-111)."""
import pytest

from oracle import bn254 as bn
from oracle import schemes as sch
from oracle.tape import SeededRng
from rabe_amd import hostlib as hl
from rabe_amd.schemes import ac17, aw11, bsw, ghw11, lsw

pytestmark = pytest.mark.gpu
g1, g2, gt, fr = bn.g1_to_le, bn.g2_to_le, bn.gt_to_le, bn.fr_to_le


@pytest.fixture(scope="module")
def host():
    h = hl.Host(0)
    yield h
    h.close()


def replay(host, seed, oracle_fn, *args):
    rng = SeededRng(seed)
    want = oracle_fn(*args, rng)
    host.set_tape(rng.log)
    return want


def test_ac17_setup(host):
    pk, msk = replay(host, 101, sch.ac17_setup)
    hpk, hmsk = ac17.setup(host)
    host.clear_tape()
    assert hl.parse_obj("ac17_pk", hpk.serialize()) == {"g": g1(pk["g"]), "h_a": [g2(x) for x in pk["h_a"]], "e_gh_ka": [gt(x) for x in pk["e_gh_ka"]]}
    assert hl.parse_obj("ac17_msk", hmsk.serialize()) == {"g": g1(msk["g"]), "h": g2(msk["h"]), "g_k": [g1(x) for x in msk["g_k"]],
                                                          "a": [fr(x) for x in msk["a"]], "b": [fr(x) for x in msk["b"]]}


def test_bsw_setup(host):
    pk, msk = replay(host, 102, sch.bsw_setup)
    hpk, hmsk = bsw.setup(host)
    host.clear_tape()
    assert hl.parse_obj("bsw_pk", hpk.serialize()) == {"g1": g1(pk["g1"]), "g2": g2(pk["g2"]), "h": g1(pk["h"]), "f": g2(pk["f"]),
                                                       "e_gg_alpha": gt(pk["e_gg_alpha"])}
    assert hl.parse_obj("bsw_msk", hmsk.serialize()) == {"beta": fr(msk["beta"]), "g2_alpha": g2(msk["g2_alpha"])}


def test_lsw_setup(host):
    pk, msk = replay(host, 103, sch.lsw_setup)
    hpk, hmsk = lsw.setup(host)
    host.clear_tape()
    assert hl.parse_obj("lsw_pk", hpk.serialize()) == {"g1": g1(pk["g1"]), "g2": g2(pk["g2"]), "g1_b": g1(pk["g1_b"]), "g1_b2": g1(pk["g1_b2"]),
                                                       "h_b": g1(pk["h_b"]), "e_gg_alpha": gt(pk["e_gg_alpha"])}
    assert hl.parse_obj("lsw_msk", hmsk.serialize()) == {"alpha1": fr(msk["alpha1"]), "alpha2": fr(msk["alpha2"]), "b": fr(msk["b"]),
                                                         "h_g1": g1(msk["h_g1"]), "h_g2": g2(msk["h_g2"])}


def test_aw11_setup_and_authgen(host):
    rng = SeededRng(104)
    gk = sch.aw11_setup(rng)
    pk, msk = sch.aw11_authgen(gk, ["a", "B", "c9"], rng)
    host.set_tape(rng.log)
    hgk = aw11.setup(host)
    hpk, hmsk = aw11.authgen(host, hgk, ["a", "B", "c9"])
    host.clear_tape()
    assert hl.parse_obj("aw11_gk", hgk.serialize()) == {"g1": g1(gk["g1"]), "g2": g2(gk["g2"])}
    assert hl.parse_obj("aw11_pk", hpk.serialize()) == {"attr": [(n, gt(e), g2(y)) for n, e, y in pk["attr"]]}
    assert hl.parse_obj("aw11_msk", hmsk.serialize()) == {"attr": [(n, fr(a), fr(y)) for n, a, y in msk["attr"]]}


def test_ghw11_setup(host):
    pk, msk = replay(host, 105, sch.ghw11_setup)
    hpk, hmsk = ghw11.setup(host)
    host.clear_tape()
    want_pk = {"g1": g1(pk["g1"]), "g2": g2(pk["g2"]), "g1_a": g1(pk["g1_a"]), "g2_a": g2(pk["g2_a"]), "e_gg_alpha": gt(pk["e_gg_alpha"])}
    assert hl.parse_obj("ghw11_pk", hpk.serialize()) == want_pk
    assert hl.parse_obj("ghw11_msk", hmsk.serialize()) == {"g2_alpha": g2(msk["g2_alpha"]), "pk": want_pk}
