"""rabe::schemes::lsw (src/schemes/lsw/mod.rs:86-290) over the host layer."""
import ctypes

from ..hostlib import JSON_POLICY, Obj, _strs, batch_decrypt


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


def keygen_batch(host, pk, msk, policies, language=JSON_POLICY):
    """n independent keygen calls, one launch per operation type (BASELINE config 4)"""
    n = len(policies)
    pol, _ = _strs(policies)
    out = (ctypes.c_void_p * max(1, n))()
    host.call("rabe_lsw_keygen_batch", pk.ptr, msk.ptr, ctypes.c_size_t(n), pol, language, out)
    return [Obj("lsw_sk", ctypes.c_void_p(out[i])) for i in range(n)]


def decrypt_batch(host, sks, cts):
    return batch_decrypt(host, "rabe_lsw_decrypt_batch", (), sks, cts)


# ---- packed batches (rabe_lsw_{keygen,decrypt}_packed): the device-resident path behind the scheme API
def keygen_packed(host, pk, msk, policies, item_policy, language=JSON_POLICY, out=None):
    """n keys as one blob of KpAbeSecretKey records: item i's policy = policies[item_policy[i]] -> (sk_blob, sk_off)"""
    from ..hostlib import packed_produce
    return packed_produce(host, "rabe_lsw_keygen_packed", (pk.ptr, msk.ptr), policies, item_policy, language, (), out)


def decrypt_packed(host, ct, sk_blob, sk_off, out=None, trusted=False):
    """n keys against ONE ciphertext object -> (pt_blob, pt_off, status)"""
    from ..hostlib import packed_decrypt
    return packed_decrypt(host, "rabe_lsw_decrypt_packed", (ct.ptr,), sk_blob, sk_off, out, trusted)


def encrypt_packed(host, pk, attr_sets, item_set, pt_blob, pt_off, out=None):
    """n encrypts (rabe_lsw_encrypt_packed): item i under the attribute list attr_sets[item_set[i]]; records = KpAbeCiphertext.
    Returns (ct_blob view, ct_off uint64 [n+1])."""
    import numpy as np
    from ..hostlib import _as_u8, _check, _np_ptr, _strs
    n = len(item_set)
    arr, _ = _strs([a for s_ in attr_sets for a in s_])
    counts = (ctypes.c_size_t * max(len(attr_sets), 1))(*[len(s_) for s_ in attr_sets])
    it = np.ascontiguousarray(item_set, dtype=np.uint32)
    po = np.ascontiguousarray(pt_off, dtype=np.uint64)
    pt = _as_u8(pt_blob)
    co = np.zeros(n + 1, dtype=np.uint64)
    buf = out if out is not None else np.empty(0, dtype=np.uint8)
    for _ in range(2):
        rc = host.lib.rabe_lsw_encrypt_packed(host.h, pk.ptr, arr, counts, ctypes.c_size_t(len(attr_sets)), ctypes.c_size_t(n), _np_ptr(it), _np_ptr(pt),
                                              _np_ptr(po), _np_ptr(buf), ctypes.c_size_t(buf.size), _np_ptr(co))
        if rc != 1:
            break
        buf = np.empty(int(co[n]), dtype=np.uint8)
    _check(rc, host.h)
    return buf[:int(co[n])], co
