"""rabe::schemes::bdabe (src/schemes/bdabe/mod.rs:149-399) over the host layer."""
import ctypes

from ..hostlib import Obj, batch_decrypt


def setup(host):
    pk, msk = ctypes.c_void_p(), ctypes.c_void_p()
    host.call("rabe_bdabe_setup", ctypes.byref(pk), ctypes.byref(msk))
    return Obj("bdabe_pk", pk), Obj("bdabe_msk", msk)


def authgen(host, pk, msk, name):
    ska = ctypes.c_void_p()
    host.call("rabe_bdabe_authgen", pk.ptr, msk.ptr, name.encode("utf-8"), ctypes.byref(ska))
    return Obj("bdabe_ska", ska)


def keygen(host, pk, ska, name):
    uk = ctypes.c_void_p()
    host.call("rabe_bdabe_keygen", pk.ptr, ska.ptr, name.encode("utf-8"), ctypes.byref(uk))
    return Obj("bdabe_uk", uk)


def request_attribute_pk(host, pk, ska, attribute):
    pka = ctypes.c_void_p()
    host.call("rabe_bdabe_request_attribute_pk", pk.ptr, ska.ptr, attribute.encode("utf-8"), ctypes.byref(pka))
    return Obj("bdabe_pka", pka)


def request_attribute_sk(host, uk, ska, attribute):
    """the reference returns the BdabeSecretAttributeKey and its callers push it onto `sk.sk_a`; here it is appended"""
    host.call("rabe_bdabe_request_attribute_sk", uk.ptr, ska.ptr, attribute.encode("utf-8"))


def encrypt(host, pk, attr_pks, policy, language, data):
    arr = (ctypes.c_void_p * max(1, len(attr_pks)))(*[p.ptr for p in attr_pks])
    ct = ctypes.c_void_p()
    host.call("rabe_bdabe_encrypt", pk.ptr, arr, ctypes.c_size_t(len(attr_pks)), policy.encode("utf-8"), language, bytes(data),
              ctypes.c_size_t(len(data)), ctypes.byref(ct))
    return Obj("bdabe_ct", ct)


def decrypt(host, uk, ct):
    return host.out_bytes("rabe_bdabe_decrypt", uk.ptr, ct.ptr)


def decrypt_gt(host, uk, ct):
    return host.out_gt("rabe_bdabe_decrypt_gt", uk.ptr, ct.ptr)


def decrypt_batch(host, uks, cts):
    return batch_decrypt(host, "rabe_bdabe_decrypt_batch", (), uks, cts)


def decrypt_packed(host, uk, ct_blob, ct_off, out=None, trusted=False):
    """n serialized BdabeCiphertext records under ONE user key (rabe_bdabe_decrypt_packed).
    Returns (pt_blob view, pt_off uint64 [n+1], status int32 [n])."""
    from ..hostlib import packed_decrypt
    return packed_decrypt(host, "rabe_bdabe_decrypt_packed", (uk.ptr,), ct_blob, ct_off, out, trusted)
