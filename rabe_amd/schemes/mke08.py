"""rabe::schemes::mke08 (src/schemes/mke08/mod.rs:130-380) over the host layer."""
import ctypes

from ..hostlib import Obj, batch_decrypt


def setup(host):
    pk, msk = ctypes.c_void_p(), ctypes.c_void_p()
    host.call("rabe_mke08_setup", ctypes.byref(pk), ctypes.byref(msk))
    return Obj("mke08_pk", pk), Obj("mke08_msk", msk)


def keygen(host, pk, msk, name):
    uk = ctypes.c_void_p()
    host.call("rabe_mke08_keygen", pk.ptr, msk.ptr, name.encode("utf-8"), ctypes.byref(uk))
    return Obj("mke08_uk", uk)


def authgen(host, name):
    ska = ctypes.c_void_p()
    host.call("rabe_mke08_authgen", name.encode("utf-8"), ctypes.byref(ska))
    return Obj("mke08_ska", ska)


def request_authority_pk(host, pk, attribute, ska):
    pka = ctypes.c_void_p()
    host.call("rabe_mke08_request_authority_pk", pk.ptr, attribute.encode("utf-8"), ska.ptr, ctypes.byref(pka))
    return Obj("mke08_pka", pka)


def request_authority_sk(host, uk, attribute, ska):
    """the reference returns the Mke08SecretAttributeKey and its callers push it onto `sk.sk_a`; here it is appended"""
    host.call("rabe_mke08_request_authority_sk", uk.ptr, attribute.encode("utf-8"), ska.ptr)


def encrypt(host, pk, attr_pks, policy, language, data):
    arr = (ctypes.c_void_p * max(1, len(attr_pks)))(*[p.ptr for p in attr_pks])
    ct = ctypes.c_void_p()
    host.call("rabe_mke08_encrypt", pk.ptr, arr, ctypes.c_size_t(len(attr_pks)), policy.encode("utf-8"), language, bytes(data),
              ctypes.c_size_t(len(data)), ctypes.byref(ct))
    return Obj("mke08_ct", ct)


def decrypt(host, uk, ct):
    return host.out_bytes("rabe_mke08_decrypt", uk.ptr, ct.ptr)


def decrypt_gt(host, uk, ct):
    return host.out_gt("rabe_mke08_decrypt_gt", uk.ptr, ct.ptr)


def decrypt_batch(host, uks, cts):
    return batch_decrypt(host, "rabe_mke08_decrypt_batch", (), uks, cts)


def decrypt_packed(host, uk, ct_blob, ct_off, out=None, trusted=False):
    """n serialized Mke08Ciphertext records under ONE user key (rabe_mke08_decrypt_packed).
    Returns (pt_blob view, pt_off uint64 [n+1], status int32 [n])."""
    from ..hostlib import packed_decrypt
    return packed_decrypt(host, "rabe_mke08_decrypt_packed", (uk.ptr,), ct_blob, ct_off, out, trusted)
