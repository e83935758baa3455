"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU,
and exports every symbol include/rabe_hip.h declares.  No compute is launched."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from rabe_amd import build
    return ctypes.CDLL(build.build())


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rabe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rhip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ["rhip_ctx_create", "rhip_pairing_product", "rhip_ac17_cp_encrypt_batch", "rhip_ac17_cp_decrypt_batch",
                 "rhip_ac17_cp_keygen_batch", "rhip_g1_table_mul", "rhip_gt_pow"]:
        assert must in syms


def test_every_declared_symbol_is_exported(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert missing == []


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = ctypes.c_void_p()
    rc = lib.rhip_ctx_create(ctypes.c_int32(0), ctypes.byref(ctx))
    assert rc == -1 and not ctx.value          # RHIP_ERR_NO_DEVICE: the engine refuses to run
    from rabe_amd import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(0)
