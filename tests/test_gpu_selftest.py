"""The known-answer self-test every device passes before its first context exists (rabe_amd/csrc/bn254/selftest.h, engine_coop.hip:
rhip_device_selftest): it runs on every SIMD, and a device whose answers differ is refused."""
import ctypes
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_selftest_covers_every_simd():
    from rabe_amd import Engine
    eng = Engine(0)
    n_cu, _ = eng.device_info()
    n = ctypes.c_uint32(0)
    eng._check(eng.lib.rhip_ctx_selftest_info(eng.ctx, ctypes.byref(n)))
    assert n.value == 4 * n_cu, (n.value, n_cu)
    eng.close()


def test_a_wrong_answer_refuses_the_device():
    code = ("from rabe_amd import Engine\n"
            "try:\n    Engine(0)\n    print('CREATED')\n"
            "except Exception as e:\n    print('REFUSED', e)\n")
    env = dict(os.environ, RABE_SELFTEST_CORRUPT="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert "REFUSED" in r.stdout and "self-test failed" in r.stdout, r.stdout + r.stderr
    # the host layer refuses as well (its Engine constructor creates the first context)
    code2 = ("from rabe_amd import hostlib\n"
             "try:\n    hostlib.Host()\n    print('CREATED')\n"
             "except Exception as e:\n    print('REFUSED', e)\n")
    r = subprocess.run([sys.executable, "-c", code2], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert "REFUSED" in r.stdout, r.stdout + r.stderr
    # and skipping the check is an explicit decision
    env = dict(os.environ, RABE_SELFTEST_CORRUPT="1", RABE_NO_SELFTEST="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert "CREATED" in r.stdout, r.stdout + r.stderr
