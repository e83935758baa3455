"""rabe::schemes::lsw (src/schemes/lsw/mod.rs:86-290) over the host layer."""
import ctypes

from ..hostlib import JSON_POLICY, Obj, _strs


def setup(host):
    pk, msk = ctypes.c_void_p(), ctypes.c_void_p()
    host.call("rabe_lsw_setup", ctypes.byref(pk), ctypes.byref(msk))
    return Obj("lsw_pk", pk), Obj("lsw_msk", msk)


def keygen(host, pk, msk, policy, language=JSON_POLICY):
    sk = ctypes.c_void_p()
    host.call("rabe_lsw_keygen", pk.ptr, msk.ptr, policy.encode("utf-8"), language, ctypes.byref(sk))
    return Obj("lsw_sk", sk)


def encrypt(host, pk, attributes, plaintext):
    arr, n = _strs(attributes)
    ct = ctypes.c_void_p()
    host.call("rabe_lsw_encrypt", pk.ptr, arr, n, bytes(plaintext), ctypes.c_size_t(len(plaintext)), ctypes.byref(ct))
    return Obj("lsw_ct", ct)


def decrypt(host, sk, ct):
    return host.out_bytes("rabe_lsw_decrypt", sk.ptr, ct.ptr)


def decrypt_gt(host, sk, ct):
    return host.out_gt("rabe_lsw_decrypt_gt", sk.ptr, ct.ptr)
