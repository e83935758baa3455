"""The drop-in boundary from C: integration/c-example/roundtrip.c includes include/rabe_hip.h and include/rabe_host.h as strict C99
(-Wall -Wextra -Werror -pedantic), links against the in-tree library and runs the reference's `and` test case (ac17/mod.rs:688-705) through the
object API, the packed API and one element-level call.  Without a GPU the program must refuse to run (exit code 2): the product path has no
CPU fallback."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "integration", "c-example", "roundtrip.c")


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    from rabe_amd import build
    build.build()
    out = str(tmp_path_factory.mktemp("c_client") / "roundtrip")
    lib_dir = os.path.join(ROOT, "rabe_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), "-o", out, SRC, "-L" + lib_dir,
                    "-lrabe_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib"], check=True, timeout=300)
    return out


def test_headers_are_c99_and_the_client_refuses_without_a_device(binary):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run is test_c_client_round_trips")
    r = subprocess.run([binary], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "no usable HIP device" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_client_round_trips(binary):
    r = subprocess.run([binary], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr)
