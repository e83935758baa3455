"""rabe::schemes::ac17 (src/schemes/ac17/mod.rs:141-430) over the host layer."""
import ctypes

from ..hostlib import JSON_POLICY, Obj, _check as _hl_check, _strs, batch_decrypt, batch_items


def setup(host):
    pk, msk = ctypes.c_void_p(), ctypes.c_void_p()
    host.call("rabe_ac17_setup", ctypes.byref(pk), ctypes.byref(msk))
    return Obj("ac17_pk", pk), Obj("ac17_msk", msk)


def cp_keygen(host, msk, attributes):
    arr, n = _strs(attributes)
    sk = ctypes.c_void_p()
    host.call("rabe_ac17_cp_keygen", msk.ptr, arr, n, ctypes.byref(sk))
    return Obj("ac17_cp_sk", sk)


def cp_encrypt(host, pk, policy, plaintext, language=JSON_POLICY):
    ct = ctypes.c_void_p()
    host.call("rabe_ac17_cp_encrypt", pk.ptr, policy.encode("utf-8"), language, bytes(plaintext), ctypes.c_size_t(len(plaintext)), ctypes.byref(ct))
    return Obj("ac17_cp_ct", ct)


def cp_decrypt(host, sk, ct):
    return host.out_bytes("rabe_ac17_cp_decrypt", sk.ptr, ct.ptr)


def cp_decrypt_gt(host, sk, ct):
    return host.out_gt("rabe_ac17_cp_decrypt_gt", sk.ptr, ct.ptr)


def cp_encrypt_batch(host, pk, policies, plaintexts, language=JSON_POLICY):
    n = len(policies)
    pol, _ = _strs(policies)
    pts = (ctypes.c_char_p * max(1, n))(*[bytes(p) for p in plaintexts])
    lens = (ctypes.c_size_t * max(1, n))(*[len(p) for p in plaintexts])
    out = (ctypes.c_void_p * max(1, n))()
    host.call("rabe_ac17_cp_encrypt_batch", pk.ptr, ctypes.c_size_t(n), pol, language, pts, lens, out)
    return [Obj("ac17_cp_ct", ctypes.c_void_p(out[i])) for i in range(n)]


def cp_decrypt_batch(host, sks, cts):
    """Returns a list with the plaintext bytes, or None where the key does not satisfy the policy."""
    n = len(cts)
    a = (ctypes.c_void_p * max(1, n))(*[s.ptr for s in sks])
    b = (ctypes.c_void_p * max(1, n))(*[c.ptr for c in cts])
    status = (ctypes.c_int32 * max(1, n))()
    pts = (ctypes.c_void_p * max(1, n))()
    lens = (ctypes.c_size_t * max(1, n))()
    host.call("rabe_ac17_cp_decrypt_batch", ctypes.c_size_t(n), a, b, status, pts, lens)
    out = []
    for i in range(n):
        if status[i] == 0:
            out.append(ctypes.string_at(pts[i], lens[i]))
            host.lib.rabe_bytes_free(ctypes.c_void_p(pts[i]))
        else:
            out.append(None)
    return out


# ---- KP-ABE variant (src/schemes/ac17/mod.rs:439-675)
def kp_keygen(host, msk, policy, language=JSON_POLICY):
    sk = ctypes.c_void_p()
    host.call("rabe_ac17_kp_keygen", msk.ptr, policy.encode("utf-8"), language, ctypes.byref(sk))
    return Obj("ac17_kp_sk", sk)


def kp_encrypt(host, pk, attributes, data):
    arr, n = _strs(attributes)
    ct = ctypes.c_void_p()
    host.call("rabe_ac17_kp_encrypt", pk.ptr, arr, n, bytes(data), ctypes.c_size_t(len(data)), ctypes.byref(ct))
    return Obj("ac17_kp_ct", ct)


def kp_decrypt(host, sk, ct):
    return host.out_bytes("rabe_ac17_kp_decrypt", sk.ptr, ct.ptr)


def kp_decrypt_gt(host, sk, ct):
    return host.out_gt("rabe_ac17_kp_decrypt_gt", sk.ptr, ct.ptr)


def kp_encrypt_batch(host, pk, attribute_sets, datas):
    """n independent kp_encrypt calls in one launch set"""
    n = len(attribute_sets)
    flat = [a for s in attribute_sets for a in s]
    arr, _ = _strs(flat)
    counts = (ctypes.c_size_t * max(1, n))(*[len(s) for s in attribute_sets])
    pts, lens = batch_items(datas)
    out = (ctypes.c_void_p * max(1, n))()
    host.call("rabe_ac17_kp_encrypt_batch", pk.ptr, ctypes.c_size_t(n), arr, counts, pts, lens, out)
    return [Obj("ac17_kp_ct", ctypes.c_void_p(out[i])) for i in range(n)]


def kp_decrypt_batch(host, sks, cts):
    return batch_decrypt(host, "rabe_ac17_kp_decrypt_batch", (), sks, cts)


def hostlib_check(rc, host):
    _hl_check(rc, host.h)


# ---- packed batches: one blob + offsets on both sides (rabe_ac17_cp_{encrypt,decrypt}_packed), caller-allocated numpy buffers
def _np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _as_u8(b):
    import numpy as np
    return b if isinstance(b, np.ndarray) else np.frombuffer(bytes(b), dtype=np.uint8)


def cp_encrypt_packed(host, pk, policies, item_policy, pt_blob, pt_off, language=JSON_POLICY, out=None):
    """policies: distinct policy texts; item_policy[i] indexes them; plaintext i = pt_blob[pt_off[i]:pt_off[i+1]].
    out: an optional numpy uint8 buffer to write into (re-used across calls); allocated when missing or too small.
    Returns (ct_blob: numpy uint8 view of exactly the records, ct_off: numpy uint64 [n+1])."""
    import numpy as np
    n = len(item_policy)
    arr, npol = _strs(policies)
    ip = np.ascontiguousarray(item_policy, dtype=np.uint32)
    po = np.ascontiguousarray(pt_off, dtype=np.uint64)
    pt = _as_u8(pt_blob)
    co = np.zeros(n + 1, dtype=np.uint64)
    buf = out if out is not None else np.empty(0, dtype=np.uint8)
    for _ in range(2):
        rc = host.lib.rabe_ac17_cp_encrypt_packed(host.h, pk.ptr, arr, npol, language, ctypes.c_size_t(n), _np_ptr(ip), _np_ptr(pt), _np_ptr(po),
                                                  _np_ptr(buf), ctypes.c_size_t(buf.size), _np_ptr(co))
        if rc != 1:
            break
        buf = np.empty(int(co[n]), dtype=np.uint8)
    hostlib_check(rc, host)
    return buf[:int(co[n])], co


def cp_keygen_packed(host, msk, attr_sets, item_set, out=None):
    """n keys under one master key (rabe_ac17_cp_keygen_packed): attr_sets = distinct attribute lists, item_set[i] indexes them.
    Returns (sk_blob: numpy uint8 view of the Ac17CpSecretKey records, sk_off: numpy uint64 [n+1])."""
    import numpy as np
    n = len(item_set)
    flat = [a for s_ in attr_sets for a in s_]
    arr, _ = _strs(flat)
    counts = (ctypes.c_size_t * max(len(attr_sets), 1))(*[len(s_) for s_ in attr_sets])
    it = np.ascontiguousarray(item_set, dtype=np.uint32)
    so = np.zeros(n + 1, dtype=np.uint64)
    buf = out if out is not None else np.empty(0, dtype=np.uint8)
    for _ in range(2):
        rc = host.lib.rabe_ac17_cp_keygen_packed(host.h, msk.ptr, arr, counts, ctypes.c_size_t(len(attr_sets)), ctypes.c_size_t(n), _np_ptr(it),
                                                 _np_ptr(buf), ctypes.c_size_t(buf.size), _np_ptr(so))
        if rc != 1:
            break
        buf = np.empty(int(so[n]), dtype=np.uint8)
    hostlib_check(rc, host)
    return buf[:int(so[n])], so


def kp_keygen_packed(host, msk, policies, item_policy, language=JSON_POLICY, out=None):
    """n KP-ABE keys under one master key (rabe_ac17_kp_keygen_packed): item i's policy = policies[item_policy[i]].
    Returns (sk_blob view of the Ac17KpSecretKey records, sk_off uint64 [n+1])."""
    from ..hostlib import packed_produce
    return packed_produce(host, "rabe_ac17_kp_keygen_packed", (msk.ptr,), policies, item_policy, language, (), out)


def kp_encrypt_packed(host, pk, attr_sets, item_set, pt_blob, pt_off, out=None):
    """n KP-ABE encrypts (rabe_ac17_kp_encrypt_packed): item i under the attribute list attr_sets[item_set[i]]; records = Ac17KpCiphertext.
    Returns (ct_blob view, ct_off uint64 [n+1])."""
    import numpy as np
    n = len(item_set)
    arr, _ = _strs([a for s_ in attr_sets for a in s_])
    counts = (ctypes.c_size_t * max(len(attr_sets), 1))(*[len(s_) for s_ in attr_sets])
    it = np.ascontiguousarray(item_set, dtype=np.uint32)
    po = np.ascontiguousarray(pt_off, dtype=np.uint64)
    pt = _as_u8(pt_blob)
    co = np.zeros(n + 1, dtype=np.uint64)
    buf = out if out is not None else np.empty(0, dtype=np.uint8)
    for _ in range(2):
        rc = host.lib.rabe_ac17_kp_encrypt_packed(host.h, pk.ptr, arr, counts, ctypes.c_size_t(len(attr_sets)), ctypes.c_size_t(n), _np_ptr(it), _np_ptr(pt),
                                                  _np_ptr(po), _np_ptr(buf), ctypes.c_size_t(buf.size), _np_ptr(co))
        if rc != 1:
            break
        buf = np.empty(int(co[n]), dtype=np.uint8)
    hostlib_check(rc, host)
    return buf[:int(co[n])], co


def kp_decrypt_packed(host, sk, ct_blob, ct_off, out=None, trusted=False):
    """n KP-ABE decrypts with one key (rabe_ac17_kp_decrypt_packed).  Returns (pt_blob view, pt_off uint64 [n+1], status int32 [n])."""
    from ..hostlib import packed_decrypt
    return packed_decrypt(host, "rabe_ac17_kp_decrypt_packed", [sk.ptr], ct_blob, ct_off, out=out, trusted=trusted)


PACKED_TRUSTED = 1


def cp_decrypt_packed(host, sk, ct_blob, ct_off, out=None, trusted=False):
    """Returns (pt_blob: numpy uint8 view, pt_off: numpy uint64 [n+1], status: numpy int32 [n]); status[i] != 0: item i did not decrypt.
    trusted=True skips the batched group-membership pass over the decoded elements (only for ciphertexts this process produced)."""
    import numpy as np
    n = len(ct_off) - 1
    ct = _as_u8(ct_blob)
    co = np.ascontiguousarray(ct_off, dtype=np.uint64)
    po = np.zeros(n + 1, dtype=np.uint64)
    status = np.zeros(max(n, 1), dtype=np.int32)
    # capacity: the sum of the well-formed records' sizes (what the library checks; overlapping records can exceed the blob's size)
    lo, hi = co[:-1], co[1:]
    okm = (lo <= hi) & (hi <= ct.size)
    need = int((hi[okm] - lo[okm]).sum()) if n else 0
    buf = out if out is not None and out.size >= need else np.empty(max(need, 1), dtype=np.uint8)
    rc = host.lib.rabe_ac17_cp_decrypt_packed(host.h, sk.ptr, ctypes.c_size_t(n), _np_ptr(ct), ctypes.c_size_t(ct.size), _np_ptr(co),
                                              ctypes.c_uint32(PACKED_TRUSTED if trusted else 0), _np_ptr(status), _np_ptr(buf),
                                              ctypes.c_size_t(buf.size), _np_ptr(po))
    hostlib_check(rc, host)
    return buf[:int(po[n])], po, status[:n]
