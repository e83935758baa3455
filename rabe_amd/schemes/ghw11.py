"""rabe::schemes::ghw11 (src/schemes/ghw11/mod.rs:92-305) over the host layer: CP-ABE with outsourced decryption.
`transform` is the part a server runs (m + 2 pairings per ciphertext), `decrypt_out` the client's one Gt power."""
import ctypes

from ..hostlib import JSON_POLICY, Obj, _strs


def setup(host):
    pk, msk = ctypes.c_void_p(), ctypes.c_void_p()
    host.call("rabe_ghw11_setup", ctypes.byref(pk), ctypes.byref(msk))
    return Obj("ghw11_pk", pk), Obj("ghw11_msk", msk)


def keygen(host, pk, msk, attributes):
    """Option<Ghw11SecretKey>: None for an empty attribute list."""
    arr, n = _strs(attributes)
    sk = ctypes.c_void_p()
    if host.call("rabe_ghw11_keygen", pk.ptr, msk.ptr, arr, n, ctypes.byref(sk)) is None:
        return None
    return Obj("ghw11_sk", sk)


def tkgen(host, sk):
    tk, rk = ctypes.c_void_p(), ctypes.c_void_p()
    host.call("rabe_ghw11_tkgen", sk.ptr, ctypes.byref(tk), ctypes.byref(rk))
    return Obj("ghw11_tk", tk), Obj("ghw11_rk", rk)


def encrypt(host, pk, policy, language, plaintext):
    ct = ctypes.c_void_p()
    host.call("rabe_ghw11_encrypt", pk.ptr, policy.encode("utf-8"), language, bytes(plaintext), ctypes.c_size_t(len(plaintext)), ctypes.byref(ct))
    return Obj("ghw11_ct", ct)


def transform(host, ct, tk):
    out = ctypes.c_void_p()
    host.call("rabe_ghw11_transform", ct.ptr, tk.ptr, ctypes.byref(out))
    return Obj("ghw11_tct", out)


def transform_batch(host, cts, tks):
    """n independent transforms in one launch set; None where the transform key does not satisfy the ciphertext's policy."""
    n = len(cts)
    a = (ctypes.c_void_p * max(1, n))(*[c.ptr for c in cts])
    b = (ctypes.c_void_p * max(1, n))(*[t.ptr for t in tks])
    status = (ctypes.c_int32 * max(1, n))()
    out = (ctypes.c_void_p * max(1, n))()
    host.call("rabe_ghw11_transform_batch", ctypes.c_size_t(n), a, b, status, out)
    return [Obj("ghw11_tct", ctypes.c_void_p(out[i])) if status[i] == 0 else None for i in range(n)]


def decrypt_out(host, tct, rk, ct):
    """`ct` supplies the symmetric data field (the reference passes it as a separate Vec<u8>)."""
    return host.out_bytes("rabe_ghw11_decrypt_out", tct.ptr, rk.ptr, ct.ptr)


def decrypt_out_gt(host, tct, rk):
    return host.out_gt("rabe_ghw11_decrypt_out_gt", tct.ptr, rk.ptr)


def transform_packed(host, tk, ct_blob, ct_off, trusted=False):
    """n transforms under one transform key over a blob of serialized ciphertexts (rabe_ghw11_transform_packed).
    Returns (tct: numpy uint8 [n, 768] -- row i = the Ghw11TransformCiphertext record c | t --, status: numpy int32 [n])."""
    import numpy as np
    from ..hostlib import PACKED_TRUSTED, _as_u8, _np_ptr
    n = len(ct_off) - 1
    ct = _as_u8(ct_blob)
    co = np.ascontiguousarray(ct_off, dtype=np.uint64)
    out = np.zeros((max(n, 1), 768), dtype=np.uint8)
    status = np.zeros(max(n, 1), dtype=np.int32)
    host.call("rabe_ghw11_transform_packed", tk.ptr, ctypes.c_size_t(n), _np_ptr(ct), ctypes.c_size_t(ct.size), _np_ptr(co),
              ctypes.c_uint32(PACKED_TRUSTED if trusted else 0), _np_ptr(status), _np_ptr(out), ctypes.c_size_t(out.size))
    return out[:n], status[:n]
