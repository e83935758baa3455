"""ctypes binding of include/rabe_host.h: the C++ host layer (rabe::schemes::* mirror).  Plumbing only."""
import ctypes
import json

from .engine import EngineError, load_library

JSON_POLICY, HUMAN_POLICY = 0, 1

KINDS = {"ac17_pk": 1, "ac17_msk": 2, "ac17_cp_sk": 3, "ac17_cp_ct": 4, "ac17_kp_sk": 5, "ac17_kp_ct": 6,
         "bsw_pk": 10, "bsw_msk": 11, "bsw_sk": 12, "bsw_ct": 13,
         "lsw_pk": 20, "lsw_msk": 21, "lsw_sk": 22, "lsw_ct": 23,
         "aw11_gk": 30, "aw11_pk": 31, "aw11_msk": 32, "aw11_sk": 33, "aw11_ct": 34,
         "ghw11_pk": 40, "ghw11_msk": 41, "ghw11_sk": 42, "ghw11_tk": 43, "ghw11_rk": 44, "ghw11_ct": 45, "ghw11_tct": 46,
         "bdabe_pk": 50, "bdabe_msk": 51, "bdabe_ska": 52, "bdabe_uk": 53, "bdabe_pka": 54, "bdabe_ct": 55,
         "mke08_pk": 60, "mke08_msk": 61, "mke08_ska": 62, "mke08_uk": 63, "mke08_pka": 64, "mke08_ct": 65}


class RabeError(Exception):
    """`RabeError` of the reference (src/error.rs:19-28)."""


class RabePanic(Exception):
    """A condition on which the reference panics."""


def _lib():
    lib = load_library()
    lib.rabe_host_last_error.restype = ctypes.c_char_p
    return lib


def _check(rc, host=None):
    if rc == 0:
        return
    msg = _lib().rabe_host_last_error(None)          # the calling thread's last error (a host may be shared by threads: submission queue)
    msg = msg.decode() if msg else ""
    if rc == -2:
        raise RabePanic(msg)
    raise RabeError(msg)


def _strs(items):
    arr = (ctypes.c_char_p * max(1, len(items)))(*[s.encode("utf-8") for s in items])
    return arr, ctypes.c_size_t(len(items))


def _take_bytes(p, n):
    out = ctypes.string_at(p, n.value)
    _lib().rabe_bytes_free(p)
    return out


def _take_text(p):
    s = ctypes.cast(p, ctypes.c_char_p).value.decode("utf-8")
    _lib().rabe_bytes_free(p)
    return s


class Obj:
    """An opaque key / ciphertext object owned by the host layer."""

    def __init__(self, kind, ptr):
        self.kind, self.ptr = kind, ptr

    def serialize(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(_lib().rabe_obj_serialize(KINDS[self.kind], self.ptr, ctypes.byref(p), ctypes.byref(n)))
        return _take_bytes(p, n)

    @classmethod
    def deserialize(cls, kind, data, host=None):
        """host given: the checked form (group membership of every element established on the GPU)"""
        p = ctypes.c_void_p()
        if host is not None:
            _check(_lib().rabe_obj_deserialize_checked(host.h, KINDS[kind], data, ctypes.c_size_t(len(data)), ctypes.byref(p)), host.h)
        else:
            _check(_lib().rabe_obj_deserialize(KINDS[kind], data, ctypes.c_size_t(len(data)), ctypes.byref(p)))
        return cls(kind, p)

    def __del__(self):
        try:
            if self.ptr:
                _lib().rabe_obj_free(KINDS[self.kind], self.ptr)
                self.ptr = None
        except Exception:
            pass


class Host:
    """One GPU context + randomness source.  `Host()` fails loudly without a HIP device."""

    def __init__(self, device=0, devices=None):
        """devices: a list of HIP device indices opens a device GROUP (include/rabe_host.h: rabe_host_open_group) -- the packed entry
        points of ac17 / bsw / lsw / aw11 then shard their items over the listed devices (a device may be listed more than once)"""
        self.lib = _lib()
        h = ctypes.c_void_p()
        if devices is not None:
            arr = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices])
            rc = self.lib.rabe_host_open_group_checked(ctypes.c_int32(self.lib.rabe_host_abi_version()), ctypes.c_size_t(len(devices)), arr, ctypes.byref(h))
        else:
            rc = self.lib.rabe_host_create(ctypes.c_int32(device), ctypes.byref(h))
        if rc != 0:
            raise EngineError("rabe_host_create failed: %s" % (self.lib.rabe_host_last_error(None) or b"").decode())
        self.h = h

    def group_size(self):
        return int(self.lib.rabe_host_group_size(self.h))

    def close(self):
        if self.h:
            self.lib.rabe_host_destroy(self.h)
            self.h = None

    def set_tape(self, fr_values):
        """Explicit randomness: the following calls draw these Fr integers in the reference's draw order."""
        data = b"".join(int(v).to_bytes(32, "little") for v in fr_values)
        _check(self.lib.rabe_host_set_tape(self.h, data, ctypes.c_size_t(len(fr_values))), self.h)

    def set_fixed_base_min(self, n):
        """elements sharing one base before G*Fr / Gt^Fr calls switch to a cached fixed-base table (results are the same)"""
        _check(self.lib.rabe_host_set_fixed_base_min(self.h, ctypes.c_size_t(int(n))), self.h)

    # ---- submission queue (include/rabe_host.h): one-call encrypt / decrypt from many threads, collected into packed batches
    def set_coalescing(self, on=True, window_us=0):
        _check(self.lib.rabe_host_set_coalescing(self.h, ctypes.c_int32(1 if on else 0), ctypes.c_uint32(window_us)), self.h)

    def submit(self, fn, *args):
        """queue one call (fn = a rabe_*_submit entry point) -> ticket for `wait`"""
        t = ctypes.c_void_p()
        _check(getattr(self.lib, fn)(self.h, *args, ctypes.byref(t)), self.h)
        return t

    def wait(self, ticket, kind=None):
        """result of a queued call: an Obj of `kind` (encrypt) or the plaintext bytes (decrypt, kind None)"""
        obj, p, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
        _check(self.lib.rabe_ticket_wait(self.h, ticket, ctypes.byref(obj), ctypes.byref(p), ctypes.byref(n)), self.h)
        return Obj(kind, obj) if kind else _take_bytes(p, n)

    def clear_tape(self):
        _check(self.lib.rabe_host_set_tape(self.h, None, ctypes.c_size_t(0)), self.h)

    def call(self, fn, *args):
        rc = getattr(self.lib, fn)(self.h, *args)
        if rc == 1:
            return None
        _check(rc, self.h)
        return True

    def out_bytes(self, fn, *args):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self.call(fn, *args, ctypes.byref(p), ctypes.byref(n))
        return _take_bytes(p, n)

    def out_gt(self, fn, *args):
        buf = ctypes.create_string_buffer(384)
        self.call(fn, *args, buf)
        return buf.raw


# ------------------------------------------------------------------ host-only policy utilities
def batch_items(datas):
    """ctypes arrays (pointers, lengths) for a list of byte strings"""
    n = len(datas)
    ptrs = (ctypes.c_char_p * max(1, n))(*[bytes(p) for p in datas])
    lens = (ctypes.c_size_t * max(1, n))(*[len(p) for p in datas])
    return ptrs, lens


def batch_decrypt(host, fn, pre_args, sks, cts):
    """Shared tail of the *_decrypt_batch wrappers: list of plaintexts, None where the key does not satisfy the policy."""
    n = len(cts)
    a = (ctypes.c_void_p * max(1, n))(*[s.ptr for s in sks])
    b = (ctypes.c_void_p * max(1, n))(*[c.ptr for c in cts])
    status = (ctypes.c_int32 * max(1, n))()
    pts = (ctypes.c_void_p * max(1, n))()
    lens = (ctypes.c_size_t * max(1, n))()
    host.call(fn, *pre_args, ctypes.c_size_t(n), a, b, status, pts, lens)
    out = []
    for i in range(n):
        if status[i] == 0:
            out.append(ctypes.string_at(pts[i], lens[i]))
            host.lib.rabe_bytes_free(ctypes.c_void_p(pts[i]))
        else:
            out.append(None)
    return out


def policy_parse(policy, language=JSON_POLICY, out_language=JSON_POLICY):
    p = ctypes.c_void_p()
    _check(_lib().rabe_policy_parse(policy.encode(), language, out_language, ctypes.byref(p)))
    return _take_text(p)


def policy_msp(policy, language=JSON_POLICY):
    p = ctypes.c_void_p()
    _check(_lib().rabe_policy_msp(policy.encode(), language, ctypes.byref(p)))
    return json.loads(_take_text(p))


def policy_pruned(policy, attributes, language=JSON_POLICY):
    arr, n = _strs(attributes)
    p = ctypes.c_void_p()
    _check(_lib().rabe_policy_pruned(policy.encode(), language, arr, n, ctypes.byref(p)))
    d = json.loads(_take_text(p))
    return d["match"], [tuple(x) for x in d["list"]]


def policy_traverse(policy, attributes, language=JSON_POLICY):
    arr, n = _strs(attributes)
    r = ctypes.c_int32()
    _check(_lib().rabe_policy_traverse(policy.encode(), language, arr, n, ctypes.byref(r)))
    return bool(r.value)


def policy_in_dnf(policy, language=JSON_POLICY):
    r = ctypes.c_int32()
    _check(_lib().rabe_policy_in_dnf(policy.encode(), language, ctypes.byref(r)))
    return bool(r.value)


def policy_dnf_terms(policy, key_attrs, language=JSON_POLICY):
    """the conjunctions `json_to_dnf` builds for public attribute keys with these names (dnf.rs:186-201); RabeError = its Err"""
    arr, n = _strs(key_attrs)
    p = ctypes.c_void_p()
    _check(_lib().rabe_policy_dnf_terms(policy.encode(), language, arr, n, ctypes.byref(p)))
    return json.loads(_take_text(p))


def policy_shares(policy, secret, tape, language=JSON_POLICY):
    p = ctypes.c_void_p()
    data = b"".join(int(v).to_bytes(32, "little") for v in tape)
    _check(_lib().rabe_policy_shares(policy.encode(), language, int(secret).to_bytes(32, "little"), data, ctypes.c_size_t(len(tape)), ctypes.byref(p)))
    return [(k, int.from_bytes(bytes.fromhex(v), "little")) for k, v in json.loads(_take_text(p))]


def policy_coeffs(policy, language=JSON_POLICY):
    p = ctypes.c_void_p()
    _check(_lib().rabe_policy_coeffs(policy.encode(), language, ctypes.byref(p)))
    return [(k, int.from_bytes(bytes.fromhex(v), "little")) for k, v in json.loads(_take_text(p))]


def encrypt_symmetric(gt, data, nonce):
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    _check(_lib().rabe_encrypt_symmetric(gt, data, ctypes.c_size_t(len(data)), nonce, ctypes.byref(p), ctypes.byref(n)))
    return _take_bytes(p, n)


def decrypt_symmetric(gt, data):
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    _check(_lib().rabe_decrypt_symmetric(gt, data, ctypes.c_size_t(len(data)), ctypes.byref(p), ctypes.byref(n)))
    return _take_bytes(p, n)


# ------------------------------------------------------------------ parser of the canonical byte form (the tests compare it with the CPU checker)
class Reader:
    def __init__(self, b):
        self.b, self.o = b, 0

    def u8(self):
        v = self.b[self.o]; self.o += 1; return v

    def u32(self):
        v = int.from_bytes(self.b[self.o:self.o + 4], "little"); self.o += 4; return v

    def raw(self, n):
        v = self.b[self.o:self.o + n]; self.o += n; return v

    def s(self):
        return self.raw(self.u32()).decode("utf-8")

    def vec(self, n):
        return [self.raw(n) for _ in range(self.u32())]

    def pol(self):
        return (self.s(), self.u8())

    def done(self):
        assert self.o == len(self.b)


def parse_obj(kind, data):
    r = Reader(data)
    if kind == "ac17_pk":
        o = {"g": r.raw(64), "h_a": r.vec(128), "e_gh_ka": r.vec(384)}
    elif kind == "ac17_msk":
        o = {"g": r.raw(64), "h": r.raw(128), "g_k": r.vec(64), "a": r.vec(32), "b": r.vec(32)}
    elif kind == "ac17_cp_sk":
        attr = [r.s() for _ in range(r.u32())]
        o = {"attr": attr, "k_0": r.vec(128), "k": [(r.s(), r.vec(64)) for _ in range(r.u32())], "k_p": r.vec(64)}
    elif kind == "ac17_cp_ct":
        o = {"policy": r.pol(), "c_0": r.vec(128), "c": [(r.s(), r.vec(64)) for _ in range(r.u32())], "c_p": r.raw(384), "ct": r.raw(r.u32())}
    elif kind == "ac17_kp_sk":
        o = {"policy": r.pol(), "k_0": r.vec(128), "k": [(r.s(), r.vec(64)) for _ in range(r.u32())], "k_p": r.vec(64)}
    elif kind == "ac17_kp_ct":
        attr = [r.s() for _ in range(r.u32())]
        o = {"attr": attr, "c_0": r.vec(128), "c": [(r.s(), r.vec(64)) for _ in range(r.u32())], "c_p": r.raw(384), "ct": r.raw(r.u32())}
    elif kind == "bsw_pk":
        o = {"g1": r.raw(64), "g2": r.raw(128), "h": r.raw(64), "f": r.raw(128), "e_gg_alpha": r.raw(384)}
    elif kind == "bsw_msk":
        o = {"beta": r.raw(32), "g2_alpha": r.raw(128)}
    elif kind == "bsw_sk":
        o = {"d": r.raw(128), "d_j": [(r.s(), r.raw(64), r.raw(128)) for _ in range(r.u32())]}
    elif kind == "bsw_ct":
        o = {"policy": r.pol(), "c": r.raw(64), "c_p": r.raw(384), "c_y": [(r.s(), r.raw(64), r.raw(128)) for _ in range(r.u32())], "data": r.raw(r.u32())}
    elif kind == "lsw_pk":
        o = {"g1": r.raw(64), "g2": r.raw(128), "g1_b": r.raw(64), "g1_b2": r.raw(64), "h_b": r.raw(64), "e_gg_alpha": r.raw(384)}
    elif kind == "lsw_msk":
        o = {"alpha1": r.raw(32), "alpha2": r.raw(32), "b": r.raw(32), "h_g1": r.raw(64), "h_g2": r.raw(128)}
    elif kind == "lsw_sk":
        o = {"policy": r.pol(), "dj": [(r.s(), r.raw(64), r.raw(128), r.raw(64), r.raw(64), r.raw(64)) for _ in range(r.u32())]}
    elif kind == "lsw_ct":
        o = {"e1": r.raw(384), "e2": r.raw(128), "ej": [(r.s(), r.raw(64), r.raw(64), r.raw(64)) for _ in range(r.u32())], "ct": r.raw(r.u32())}
    elif kind == "aw11_gk":
        o = {"g1": r.raw(64), "g2": r.raw(128)}
    elif kind == "aw11_pk":
        o = {"attr": [(r.s(), r.raw(384), r.raw(128)) for _ in range(r.u32())]}
    elif kind == "aw11_msk":
        o = {"attr": [(r.s(), r.raw(32), r.raw(32)) for _ in range(r.u32())]}
    elif kind == "aw11_sk":
        o = {"gid": r.s(), "attr": [(r.s(), r.raw(64)) for _ in range(r.u32())]}
    elif kind == "aw11_ct":
        o = {"policy": r.pol(), "c_0": r.raw(384), "c": [(r.s(), r.raw(384), r.raw(128), r.raw(128)) for _ in range(r.u32())], "ct": r.raw(r.u32())}
    elif kind == "ghw11_pk":
        o = {"g1": r.raw(64), "g2": r.raw(128), "g1_a": r.raw(64), "g2_a": r.raw(128), "e_gg_alpha": r.raw(384)}
    elif kind == "ghw11_msk":
        o = {"g2_alpha": r.raw(128), "pk": {"g1": r.raw(64), "g2": r.raw(128), "g1_a": r.raw(64), "g2_a": r.raw(128), "e_gg_alpha": r.raw(384)}}
    elif kind == "ghw11_sk":
        o = {"k": r.raw(128), "l": r.raw(128), "attr_key": [(r.s(), r.raw(128)) for _ in range(r.u32())]}
    elif kind == "ghw11_tk":
        o = {"k_z": r.raw(128), "l_z": r.raw(128), "attr_key_z": [(r.s(), r.raw(128)) for _ in range(r.u32())]}
    elif kind == "ghw11_rk":
        o = {"z": r.raw(32)}
    elif kind == "ghw11_ct":
        o = {"policy": r.pol(), "c": r.raw(384), "c1": r.raw(64), "ci_di": [(r.s(), r.raw(64), r.raw(64)) for _ in range(r.u32())], "data": r.raw(r.u32())}
    elif kind == "ghw11_tct":
        o = {"c": r.raw(384), "t": r.raw(384)}
    elif kind == "bdabe_pk":
        o = {"g1": r.raw(64), "g2": r.raw(128), "p1": r.raw(64), "p2": r.raw(128), "e_gg_y": r.raw(384)}
    elif kind == "bdabe_msk":
        o = {"y": r.raw(32)}
    elif kind == "bdabe_ska":
        o = {"name": r.s(), "a1": r.raw(64), "a2": r.raw(128), "a3": r.raw(32)}
    elif kind == "bdabe_uk":
        o = {"sk": {"u1": r.raw(64), "u2": r.raw(128)}, "pk": {"u": r.s(), "u1": r.raw(64), "u2": r.raw(128)},
             "sk_a": [(r.s(), r.raw(64), r.raw(128)) for _ in range(r.u32())]}
    elif kind == "bdabe_pka":
        o = {"attr": r.s(), "a1": r.raw(64), "a2": r.raw(128), "a3": r.raw(384)}
    elif kind == "bdabe_ct":
        o = {"policy": r.pol(), "j": [([r.s() for _ in range(r.u32())], r.raw(384), r.raw(64), r.raw(128), r.raw(64), r.raw(128)) for _ in range(r.u32())],
             "ct": r.raw(r.u32())}
    elif kind == "mke08_pk":
        o = {"g1": r.raw(64), "g2": r.raw(128), "p1": r.raw(64), "p2": r.raw(128), "e_gg_y1": r.raw(384), "e_gg_y2": r.raw(384)}
    elif kind == "mke08_msk":
        o = {"g1": r.raw(64), "g2": r.raw(128)}
    elif kind == "mke08_ska":
        o = {"name": r.s(), "r": r.raw(32)}
    elif kind == "mke08_uk":
        o = {"sk": {"g1": r.raw(64), "g2": r.raw(128)}, "pk": {"name": r.s(), "g1": r.raw(64), "g2": r.raw(128)},
             "sk_a": [(r.s(), r.raw(64), r.raw(128)) for _ in range(r.u32())]}
    elif kind == "mke08_pka":
        o = {"attr": r.s(), "g1": r.raw(64), "g2": r.raw(128), "gt1": r.raw(384), "gt2": r.raw(384)}
    elif kind == "mke08_ct":
        o = {"policy": r.pol(), "e": [([r.s() for _ in range(r.u32())], r.raw(384), r.raw(384), r.raw(64), r.raw(128), r.raw(64), r.raw(128))
                                      for _ in range(r.u32())], "ct": r.raw(r.u32())}
    else:
        raise ValueError(kind)
    r.done()
    return o


# ---- packed batches (one blob of canonical records + offsets per side, caller-allocated numpy buffers): shared plumbing of
# rabe_{ac17_cp,bsw,aw11}_encrypt_packed / rabe_lsw_keygen_packed and rabe_*_decrypt_packed
PACKED_TRUSTED = 1


def _np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _as_u8(b):
    import numpy as np
    return b if isinstance(b, np.ndarray) else np.frombuffer(bytes(b), dtype=np.uint8)


def packed_produce(host, fn, head_args, policies, item_policy, language, tail_arrays=(), out=None):
    """calls `fn(host, *head_args, policies, n_policies, language, n, item_policy, *tail_arrays, out_buf, out_cap, out_off)`, growing
    the output buffer once when the library reports the size it needs.  Returns (blob view, offsets uint64 [n+1])."""
    import numpy as np
    n = len(item_policy)
    arr, npol = _strs(policies)
    ip = np.ascontiguousarray(item_policy, dtype=np.uint32)
    co = np.zeros(n + 1, dtype=np.uint64)
    buf = out if out is not None else np.empty(0, dtype=np.uint8)
    tails = [_np_ptr(t) if hasattr(t, "ctypes") else t for t in tail_arrays]
    for _ in range(2):
        rc = getattr(host.lib, fn)(host.h, *head_args, arr, npol, language, ctypes.c_size_t(n), _np_ptr(ip), *tails, _np_ptr(buf),
                                   ctypes.c_size_t(buf.size), _np_ptr(co))
        if rc != 1:
            break
        buf = np.empty(int(co[n]), dtype=np.uint8)
    _check(rc, host.h)
    return buf[:int(co[n])], co


def packed_decrypt(host, fn, head_args, blob, off, out=None, trusted=False):
    """calls `fn(host, *head_args, n, blob, len, off, flags, status, pt_buf, pt_cap, pt_off)`.
    Returns (pt_blob view, pt_off uint64 [n+1], status int32 [n])."""
    import numpy as np
    n = len(off) - 1
    ct = _as_u8(blob)
    co = np.ascontiguousarray(off, dtype=np.uint64)
    po = np.zeros(n + 1, dtype=np.uint64)
    status = np.zeros(max(n, 1), dtype=np.int32)
    lo, hi = co[:-1], co[1:]
    okm = (lo <= hi) & (hi <= ct.size)
    need = int((hi[okm] - lo[okm]).sum()) if n else 0
    buf = out if out is not None and out.size >= need else np.empty(max(need, 1), dtype=np.uint8)
    rc = getattr(host.lib, fn)(host.h, *head_args, ctypes.c_size_t(n), _np_ptr(ct), ctypes.c_size_t(ct.size), _np_ptr(co),
                               ctypes.c_uint32(PACKED_TRUSTED if trusted else 0), _np_ptr(status), _np_ptr(buf), ctypes.c_size_t(buf.size), _np_ptr(po))
    _check(rc, host.h)
    return buf[:int(po[n])], po, status[:n]
