"""rabe::schemes::aw11 (src/schemes/aw11/mod.rs:100-390) over the host layer."""
import ctypes

from ..hostlib import JSON_POLICY, Obj, _strs, batch_decrypt, batch_items


def setup(host):
    gk = ctypes.c_void_p()
    host.call("rabe_aw11_setup", ctypes.byref(gk))
    return Obj("aw11_gk", gk)


def authgen(host, gk, attributes):
    """Option<(Aw11PublicKey, Aw11MasterKey)>: None for an empty attribute list."""
    arr, n = _strs(attributes)
    pk, msk = ctypes.c_void_p(), ctypes.c_void_p()
    if host.call("rabe_aw11_authgen", gk.ptr, arr, n, ctypes.byref(pk), ctypes.byref(msk)) is None:
        return None
    return Obj("aw11_pk", pk), Obj("aw11_msk", msk)


def keygen(host, gk, msk, name, attributes):
    arr, n = _strs(attributes)
    sk = ctypes.c_void_p()
    host.call("rabe_aw11_keygen", gk.ptr, msk.ptr, name.encode("utf-8"), arr, n, ctypes.byref(sk))
    return Obj("aw11_sk", sk)


def add_to_attribute(host, gk, msk, attribute, sk):
    host.call("rabe_aw11_add_to_attribute", gk.ptr, msk.ptr, attribute.encode("utf-8"), sk.ptr)


def encrypt(host, gk, pks, policy, language, data):
    arr = (ctypes.c_void_p * max(1, len(pks)))(*[p.ptr for p in pks])
    ct = ctypes.c_void_p()
    host.call("rabe_aw11_encrypt", gk.ptr, arr, ctypes.c_size_t(len(pks)), policy.encode("utf-8"), language, bytes(data),
              ctypes.c_size_t(len(data)), ctypes.byref(ct))
    return Obj("aw11_ct", ct)


def decrypt(host, gk, sk, ct):
    return host.out_bytes("rabe_aw11_decrypt", gk.ptr, sk.ptr, ct.ptr)


def decrypt_gt(host, gk, sk, ct):
    return host.out_gt("rabe_aw11_decrypt_gt", gk.ptr, sk.ptr, ct.ptr)


def encrypt_batch(host, gk, pks, policies, language, datas):
    """n independent encrypt calls under the same authority keys, one launch per operation type (BASELINE config 5)"""
    n = len(policies)
    arr = (ctypes.c_void_p * max(1, len(pks)))(*[p.ptr for p in pks])
    pol, _ = _strs(policies)
    pts, lens = batch_items(datas)
    out = (ctypes.c_void_p * max(1, n))()
    host.call("rabe_aw11_encrypt_batch", gk.ptr, arr, ctypes.c_size_t(len(pks)), ctypes.c_size_t(n), pol, language, pts, lens, out)
    return [Obj("aw11_ct", ctypes.c_void_p(out[i])) for i in range(n)]


def decrypt_batch(host, gk, sks, cts):
    return batch_decrypt(host, "rabe_aw11_decrypt_batch", (gk.ptr,), sks, cts)


# ---- packed batches (rabe_aw11_{encrypt,decrypt}_packed): the device-resident path behind the scheme API
def encrypt_packed(host, gk, pks, policies, item_policy, pt_blob, pt_off, language=JSON_POLICY, out=None):
    import numpy as np
    from ..hostlib import _as_u8, packed_produce
    arr = (ctypes.c_void_p * max(1, len(pks)))(*[p.ptr for p in pks])
    return packed_produce(host, "rabe_aw11_encrypt_packed", (gk.ptr, arr, ctypes.c_size_t(len(pks))), policies, item_policy, language,
                          (_as_u8(pt_blob), np.ascontiguousarray(pt_off, dtype=np.uint64)), out)


def decrypt_packed(host, gk, sk, ct_blob, ct_off, out=None, trusted=False):
    from ..hostlib import packed_decrypt
    return packed_decrypt(host, "rabe_aw11_decrypt_packed", (gk.ptr, sk.ptr), ct_blob, ct_off, out, trusted)


def keygen_packed(host, gk, msk, gids, attr_sets, item_set, out=None):
    """n keys issued by one authority (rabe_aw11_keygen_packed): user gids[i] gets attr_sets[item_set[i]].
    Returns (sk_blob: numpy uint8 view of the Aw11SecretKey records, sk_off: numpy uint64 [n+1])."""
    import numpy as np
    from ..hostlib import _check, _np_ptr, _strs
    n = len(item_set)
    garr, _ = _strs(list(gids))
    arr, _ = _strs([a for s_ in attr_sets for a in s_])
    counts = (ctypes.c_size_t * max(len(attr_sets), 1))(*[len(s_) for s_ in attr_sets])
    it = np.ascontiguousarray(item_set, dtype=np.uint32)
    so = np.zeros(n + 1, dtype=np.uint64)
    buf = out if out is not None else np.empty(0, dtype=np.uint8)
    for _ in range(2):
        rc = host.lib.rabe_aw11_keygen_packed(host.h, gk.ptr, msk.ptr, garr, arr, counts, ctypes.c_size_t(len(attr_sets)), ctypes.c_size_t(n), _np_ptr(it),
                                              _np_ptr(buf), ctypes.c_size_t(buf.size), _np_ptr(so))
        if rc != 1:
            break
        buf = np.empty(int(so[n]), dtype=np.uint8)
    _check(rc, host.h)
    return buf[:int(so[n])], so
