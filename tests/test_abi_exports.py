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
    out = set()
    for header, prefix in (("rabe_hip.h", "rhip_"), ("rabe_host.h", "rabe_")):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text))
    return sorted(out)


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ["rhip_ctx_create", "rhip_pairing_product", "rhip_ac17_cp_encrypt_batch", "rhip_ac17_cp_decrypt_batch",
                 "rhip_ac17_cp_keygen_batch", "rhip_g1_table_mul", "rhip_gt_pow", "rhip_ac17_sk_prepare",
                 "rhip_ac17_cp_decrypt_batch_prepared", "rhip_g1_table_add_wide", "rabe_ac17_cp_encrypt_batch", "rabe_bsw_encrypt_batch",
                 "rabe_bsw_decrypt_batch", "rabe_lsw_keygen_batch", "rabe_lsw_decrypt_batch", "rabe_aw11_encrypt_batch",
                 "rabe_aw11_decrypt_batch"]:
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
