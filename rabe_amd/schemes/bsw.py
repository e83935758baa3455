"""rabe::schemes::bsw (src/schemes/bsw/mod.rs:92-318) over the host layer."""
import ctypes

from ..hostlib import JSON_POLICY, Obj, _strs, batch_decrypt, batch_items


def setup(host):
    pk, msk = ctypes.c_void_p(), ctypes.c_void_p()
    host.call("rabe_bsw_setup", ctypes.byref(pk), ctypes.byref(msk))
    return Obj("bsw_pk", pk), Obj("bsw_msk", msk)


def keygen(host, pk, msk, attributes):
    """Option<CpAbeSecretKey>: None for an empty attribute list."""
    arr, n = _strs(attributes)
    sk = ctypes.c_void_p()
    if host.call("rabe_bsw_keygen", pk.ptr, msk.ptr, arr, n, ctypes.byref(sk)) is None:
        return None
    return Obj("bsw_sk", sk)


def delegate(host, pk, sk, subset):
    """Option<CpAbeSecretKey>: None when `subset` is empty or not a subset of the key's attributes (:162-206)."""
    arr, n = _strs(subset)
    out = ctypes.c_void_p()
    if host.call("rabe_bsw_delegate", pk.ptr, sk.ptr, arr, n, ctypes.byref(out)) is None:
        return None
    return Obj("bsw_sk", out)


def encrypt(host, pk, policy, language, plaintext):
    ct = ctypes.c_void_p()
    host.call("rabe_bsw_encrypt", pk.ptr, policy.encode("utf-8"), language, bytes(plaintext), ctypes.c_size_t(len(plaintext)), ctypes.byref(ct))
    return Obj("bsw_ct", ct)


def decrypt(host, sk, ct):
    return host.out_bytes("rabe_bsw_decrypt", sk.ptr, ct.ptr)


def decrypt_gt(host, sk, ct):
    return host.out_gt("rabe_bsw_decrypt_gt", sk.ptr, ct.ptr)


def encrypt_batch(host, pk, policies, language, plaintexts):
    """n independent encrypt calls, one launch per operation type (BASELINE config 3)"""
    n = len(policies)
    pol, _ = _strs(policies)
    pts, lens = batch_items(plaintexts)
    out = (ctypes.c_void_p * max(1, n))()
    host.call("rabe_bsw_encrypt_batch", pk.ptr, ctypes.c_size_t(n), pol, language, pts, lens, out)
    return [Obj("bsw_ct", ctypes.c_void_p(out[i])) for i in range(n)]


def decrypt_batch(host, sks, cts):
    return batch_decrypt(host, "rabe_bsw_decrypt_batch", (), sks, cts)


# ---- packed batches (rabe_bsw_{encrypt,decrypt}_packed): the device-resident path behind the scheme API
def encrypt_packed(host, pk, policies, item_policy, pt_blob, pt_off, language=JSON_POLICY, out=None):
    """policies: distinct texts, item_policy[i] indexes them, plaintext i = pt_blob[pt_off[i]:pt_off[i+1]] -> (ct_blob, ct_off)"""
    import numpy as np
    from ..hostlib import _as_u8, packed_produce
    return packed_produce(host, "rabe_bsw_encrypt_packed", (pk.ptr,), policies, item_policy, language,
                          (_as_u8(pt_blob), np.ascontiguousarray(pt_off, dtype=np.uint64)), out)


def decrypt_packed(host, sk, ct_blob, ct_off, out=None, trusted=False):
    from ..hostlib import packed_decrypt
    return packed_decrypt(host, "rabe_bsw_decrypt_packed", (sk.ptr,), ct_blob, ct_off, out, trusted)


def keygen_packed(host, pk, msk, attr_sets, item_set, out=None):
    """n keys under one master key (rabe_bsw_keygen_packed): attr_sets = distinct attribute lists, item_set[i] indexes them.
    Returns (sk_blob: numpy uint8 view of the CpAbeSecretKey records, sk_off: numpy uint64 [n+1])."""
    import numpy as np
    from ..hostlib import _check, _np_ptr
    n = len(item_set)
    arr, _ = _strs([a for s_ in attr_sets for a in s_])
    counts = (ctypes.c_size_t * max(len(attr_sets), 1))(*[len(s_) for s_ in attr_sets])
    it = np.ascontiguousarray(item_set, dtype=np.uint32)
    so = np.zeros(n + 1, dtype=np.uint64)
    buf = out if out is not None else np.empty(0, dtype=np.uint8)
    for _ in range(2):
        rc = host.lib.rabe_bsw_keygen_packed(host.h, pk.ptr, msk.ptr, arr, counts, ctypes.c_size_t(len(attr_sets)), ctypes.c_size_t(n), _np_ptr(it),
                                             _np_ptr(buf), ctypes.c_size_t(buf.size), _np_ptr(so))
        if rc != 1:
            break
        buf = np.empty(int(so[n]), dtype=np.uint8)
    _check(rc, host.h)
    return buf[:int(so[n])], so


def delegate_packed(host, pk, sk, subsets, item_subset, out=None):
    """n delegations of ONE key (rabe_bsw_delegate_packed): item i gets the attribute list subsets[item_subset[i]].
    Returns (sk_blob view of the CpAbeSecretKey records, sk_off uint64 [n+1])."""
    import numpy as np
    from ..hostlib import _check, _np_ptr
    n = len(item_subset)
    arr, _ = _strs([a for s_ in subsets for a in s_])
    counts = (ctypes.c_size_t * max(len(subsets), 1))(*[len(s_) for s_ in subsets])
    it = np.ascontiguousarray(item_subset, dtype=np.uint32)
    so = np.zeros(n + 1, dtype=np.uint64)
    buf = out if out is not None else np.empty(0, dtype=np.uint8)
    for _ in range(2):
        rc = host.lib.rabe_bsw_delegate_packed(host.h, pk.ptr, sk.ptr, arr, counts, ctypes.c_size_t(len(subsets)), ctypes.c_size_t(n), _np_ptr(it),
                                               _np_ptr(buf), ctypes.c_size_t(buf.size), _np_ptr(so))
        if rc != 1:
            break
        buf = np.empty(int(so[n]), dtype=np.uint8)
    _check(rc, host.h)
    return buf[:int(so[n])], so
