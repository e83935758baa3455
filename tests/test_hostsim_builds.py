"""The host build of the engine's headers (tests/hostsim) is what the test_hostsim_* modules run on; they SKIP when it cannot be loaded, so a
header edit that breaks the host build would silently take a third of the CPU suite with it.  This test does not skip."""
from tests.hostsim import build as hs_build


def test_the_host_build_of_the_headers_compiles_and_loads():
    lib = hs_build.load()
    for sym in ("hs_rr_miller_multi", "hs_rr_miller_pair", "hs_miller_multi", "hs_rr_final_exp", "hs_selftest_digest"):
        assert hasattr(lib, sym), sym
