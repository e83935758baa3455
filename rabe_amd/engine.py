"""ctypes binding of include/rabe_hip.h (plumbing only).

`Engine()` fails loudly -- EngineError -- when librabe_hip.so is missing or there is no HIP device;
there is no CPU fallback.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))

FR, G1, G2, GT = 32, 64, 128, 384
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class EngineError(RuntimeError):
    pass


def lib_path():
    # RABE_HIP_LIB selects an alternative build of the same library (kernel-tuning A/B runs)
    return os.environ.get("RABE_HIP_LIB") or os.path.join(HERE, "librabe_hip.so")


_LIB = None


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (same SONAME as /opt/rocm's); whichever is loaded first wins, and if ours came first torch would
    later find "No HIP GPUs".  So when torch is installed, map its runtime before librabe_hip.so is
    resolved -- import order then no longer matters (bench.py and smoke() need torch for streams and
    torch.distributed in the same process)."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """dlopen the engine (works without a GPU; only rhip_ctx_create needs one)."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise EngineError("HIP engine not built: %s missing (run `python -m rabe_amd.build`)" % path)
        _share_hip_runtime_with_torch()
        lib = ctypes.CDLL(path)
        lib.rhip_last_error.restype = ctypes.c_char_p
        _LIB = lib
    return _LIB


def fr_bytes(x):
    return int(x % R_ORDER).to_bytes(32, "little")


class DevBuf:
    """A device allocation owned by an Engine."""

    def __init__(self, eng, nbytes):
        self.eng = eng
        self.nbytes = nbytes
        p = ctypes.c_void_p()
        eng._check(eng.lib.rhip_malloc(eng.ctx, ctypes.c_size_t(nbytes), ctypes.byref(p)))
        self.ptr = p

    def free(self):
        if self.ptr:
            self.eng.lib.rhip_free(self.eng.ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, device=0):
        self.lib = load_library()
        ctx = ctypes.c_void_p()
        rc = self.lib.rhip_ctx_create(ctypes.c_int32(device), ctypes.byref(ctx))
        if rc != 0:
            why = self.lib.rhip_last_error(None)
            raise EngineError("rhip_ctx_create(device=%d) failed with %d [%s]: no usable HIP device "
                              "(the engine has no CPU fallback)" % (device, rc, why.decode() if why else "?"))
        self.ctx = ctx

    def close(self):
        if self.ctx:
            self.lib.rhip_ctx_destroy(self.ctx)
            self.ctx = None

    def _check(self, rc):
        if rc != 0:
            raise EngineError("rabe_hip call failed (%d): %s" % (rc, self.lib.rhip_last_error(self.ctx).decode()))

    # ------------------------------------------------------------------ memory
    def alloc(self, nbytes):
        return DevBuf(self, max(int(nbytes), 4))

    def upload(self, data):
        data = bytes(data)
        b = self.alloc(len(data))
        if data:
            self._check(self.lib.rhip_upload(self.ctx, b.ptr, data, ctypes.c_size_t(len(data))))
        return b

    def upload_u32(self, values):
        import array
        return self.upload(array.array("I", values).tobytes())

    def download(self, buf, nbytes=None):
        n = buf.nbytes if nbytes is None else nbytes
        out = ctypes.create_string_buffer(n)
        if n:
            self._check(self.lib.rhip_download(self.ctx, out, buf.ptr, ctypes.c_size_t(n)))
        return out.raw

    def sync(self):
        self._check(self.lib.rhip_sync(self.ctx))

    def set_stream(self, hip_stream_ptr):
        self._check(self.lib.rhip_ctx_set_stream(self.ctx, ctypes.c_void_p(hip_stream_ptr)))

    def set_pairing_mode(self, mode):
        """0 auto, 1 one lane per pairing, 3 three cooperating lanes per pairing, 6 six lanes per accumulator, 29 reduced radix (same
        results); 99 cross-check: auto, then every family forced on the same inputs, compared on the device (a difference fails the call)."""
        self._check(self.lib.rhip_ctx_set_pairing_mode(self.ctx, ctypes.c_int32(mode)))

    def timing(self, enable):
        self._check(self.lib.rhip_ctx_timing(self.ctx, ctypes.c_int32(1 if enable else 0)))

    def timing_read(self):
        """{kernel_name: (total_ms, launches)} since the last read."""
        buf = ctypes.create_string_buffer(1 << 16)
        self._check(self.lib.rhip_ctx_timing_read(self.ctx, buf, ctypes.c_size_t(len(buf))))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt = line.split()
            out[name] = (float(ms), int(cnt))
        return out

    def device_info(self):
        n = ctypes.c_int32()
        name = ctypes.create_string_buffer(128)
        self._check(self.lib.rhip_device_info(self.ctx, ctypes.byref(n), name, ctypes.c_size_t(128)))
        return n.value, name.value.decode()

    # ------------------------------------------------------------------ Level E helpers (host bytes in, host bytes out)
    def _elem(self, fn, n, ins, out_elem_bytes, extra_first=()):
        bufs = [self.upload(b"".join(x)) if not isinstance(x, (bytes, bytearray)) else self.upload(x) for x in ins]
        out = self.alloc(n * out_elem_bytes)
        args = list(extra_first) + [ctypes.c_size_t(n)] + [b.ptr for b in bufs] + [out.ptr]
        self._check(getattr(self.lib, fn)(self.ctx, *args))
        raw = self.download(out, n * out_elem_bytes)
        return [raw[i * out_elem_bytes:(i + 1) * out_elem_bytes] for i in range(n)]

    def fr_op(self, op, a, b=None):
        n = len(a)
        bb = b if b is not None else a
        return self._elem("rhip_fr_op", n, [a, bb], FR, extra_first=(ctypes.c_int32(op),))

    def fr_from_be32_reduce(self, digests):
        return self._elem("rhip_fr_from_be32_reduce", len(digests), [digests], FR)

    def g1_add(self, a, b): return self._elem("rhip_g1_add", len(a), [a, b], G1)
    def g1_neg(self, a): return self._elem("rhip_g1_neg", len(a), [a], G1)
    def g1_mul(self, p, k): return self._elem("rhip_g1_mul", len(p), [p, k], G1)
    def g2_add(self, a, b): return self._elem("rhip_g2_add", len(a), [a, b], G2)
    def g2_neg(self, a): return self._elem("rhip_g2_neg", len(a), [a], G2)
    def g2_mul(self, p, k): return self._elem("rhip_g2_mul", len(p), [p, k], G2)
    def gt_mul(self, a, b): return self._elem("rhip_gt_mul", len(a), [a, b], GT)
    def gt_inv(self, a): return self._elem("rhip_gt_inv", len(a), [a], GT)
    def gt_pow(self, a, k): return self._elem("rhip_gt_pow", len(a), [a, k], GT)
    def pairing(self, p, q): return self._elem("rhip_pairing", len(p), [p, q], GT)

    def g1_on_curve(self, p):
        return [int.from_bytes(x, "little") for x in self._elem("rhip_g1_on_curve", len(p), [p], 4)]

    def g2_on_curve(self, p):
        return [int.from_bytes(x, "little") for x in self._elem("rhip_g2_on_curve", len(p), [p], 4)]

    def pairing_product(self, offsets, p, q):
        n_items = len(offsets) - 1
        off = self.upload_u32(offsets)
        bp, bq = self.upload(b"".join(p)), self.upload(b"".join(q))
        out = self.alloc(n_items * GT)
        self._check(self.lib.rhip_pairing_product(self.ctx, ctypes.c_size_t(n_items), off.ptr, ctypes.c_size_t(len(p)),
                                                  bp.ptr, bq.ptr, out.ptr))
        raw = self.download(out, n_items * GT)
        return [raw[i * GT:(i + 1) * GT] for i in range(n_items)]

    def pairing_jobs(self, offsets, p, q, scal=None, lead=None):
        """rhip_pairing_jobs without a summed pair: out[i] = lead[i] * FE(prod_j ML(scal[j] * p[j], q[j])) over item i's pairs"""
        n_items = len(offsets) - 1
        off = self.upload_u32(offsets)
        bp, bq = self.upload(b"".join(p) or b"\0"), self.upload(b"".join(q) or b"\0")
        bs = self.upload(b"".join(scal)) if scal else None
        bl = self.upload(b"".join(lead)) if lead else None
        out = self.alloc(n_items * GT)
        mx = max(offsets[i + 1] - offsets[i] for i in range(n_items))
        self._check(self.lib.rhip_pairing_jobs(self.ctx, ctypes.c_size_t(n_items), ctypes.c_size_t(mx), ctypes.c_size_t(len(p)), off.ptr, bp.ptr,
                                               bs.ptr if bs else None, bq.ptr, ctypes.c_size_t(0), ctypes.c_size_t(0), None, None, None, None,
                                               bl.ptr if bl else None, out.ptr))
        raw = self.download(out, n_items * GT)
        return [raw[i * GT:(i + 1) * GT] for i in range(n_items)]

    # ------------------------------------------------------------------ pinned host buffers / stream-ordered copies
    def host_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self.lib.rhip_host_alloc(self.ctx, ctypes.c_size_t(nbytes), ctypes.byref(p)))
        return p

    def host_free(self, p):
        self._check(self.lib.rhip_host_free(self.ctx, p))

    def release_before_final_exp(self, waiter):
        """one-shot: `waiter`'s stream is held until this context's next decrypt has issued its Miller loops (rhip_ctx_release_before_final_exp)"""
        self._check(self.lib.rhip_ctx_release_before_final_exp(self.ctx, waiter.ctx))

    def release_when_miller_resident(self, waiter):
        """one-shot: `waiter`'s stream goes on as soon as the blocks of this context's next Miller launch are resident
        (rhip_ctx_release_when_miller_resident); waiter None withdraws"""
        self._check(self.lib.rhip_ctx_release_when_miller_resident(self.ctx, waiter.ctx if waiter is not None else None))

    def wait_for(self, other):
        """order this context's future work after everything submitted to `other` so far (no host wait)"""
        self._check(self.lib.rhip_ctx_wait_for(self.ctx, other.ctx))

    def upload_async(self, dev, host_ptr, nbytes):
        self._check(self.lib.rhip_upload_async(self.ctx, dev.ptr, host_ptr, ctypes.c_size_t(nbytes)))

    def download_async(self, host_ptr, dev, nbytes):
        self._check(self.lib.rhip_download_async(self.ctx, host_ptr, dev.ptr, ctypes.c_size_t(nbytes)))

    # ------------------------------------------------------------------ tables
    def g1_table(self, base): return _Table(self, "g1", base)
    def g2_table(self, base): return _Table(self, "g2", base)
    def gt_table(self, base): return _Table(self, "gt", base)

    def calibrate(self, variant, iters):
        ms, ops = ctypes.c_double(), ctypes.c_double()
        self._check(self.lib.rhip_calibrate_mad(self.ctx, ctypes.c_int32(variant), ctypes.c_uint32(iters),
                                                ctypes.byref(ms), ctypes.byref(ops)))
        return ms.value, ops.value


class _Table:
    _OUT = {"g1": G1, "g2": G2, "gt": GT}

    def __init__(self, eng, kind, base):
        self.eng, self.kind = eng, kind
        self.h = ctypes.c_void_p()
        eng._check(getattr(eng.lib, "rhip_%s_table_create" % kind)(eng.ctx, bytes(base), ctypes.byref(self.h)))

    def mul(self, scalars):
        fn = "rhip_%s_table_%s" % (self.kind, "pow" if self.kind == "gt" else "mul")
        eng = self.eng
        n = len(scalars)
        k = eng.upload(b"".join(scalars))
        sz = self._OUT[self.kind]
        out = eng.alloc(n * sz)
        eng._check(getattr(eng.lib, fn)(eng.ctx, self.h, ctypes.c_size_t(n), k.ptr, out.ptr))
        raw = eng.download(out, n * sz)
        return [raw[i * sz:(i + 1) * sz] for i in range(n)]

    def add_w16(self):
        """16-bit windows beside the 8-bit ones (rhip_{g1,g2,gt}_table_add_w16)"""
        self.eng._check(getattr(self.eng.lib, "rhip_%s_table_add_w16" % self.kind)(self.eng.ctx, self.h))

    def add_wide(self, w_bits):
        """signed w_bits-wide windows (G1 only, rhip_g1_table_add_wide)"""
        assert self.kind == "g1"
        self.eng._check(self.eng.lib.rhip_g1_table_add_wide(self.eng.ctx, self.h, ctypes.c_int32(int(w_bits))))

    def destroy(self):
        if self.h:
            getattr(self.eng.lib, "rhip_%s_table_destroy" % self.kind)(self.h)
            self.h = None


# ---------------------------------------------------------------------- Level B: AC17 (device-buffer level)
class Ac17Pk:
    """Device tables of an Ac17PublicKey (g, h_a[3], e_gh_ka[2])."""

    def __init__(self, eng, g, h_a, e_gh_ka):
        self.eng = eng
        self.h = ctypes.c_void_p()
        eng._check(eng.lib.rhip_ac17_pk_create(eng.ctx, bytes(g), b"".join(h_a), b"".join(e_gh_ka), ctypes.byref(self.h)))

    def set_g_window(self, w_bits):
        """signed w_bits-wide windows for g (rhip_ac17_pk_set_g_window): fewer additions per row, more HBM"""
        self.eng._check(self.eng.lib.rhip_ac17_pk_set_g_window(self.eng.ctx, self.h, ctypes.c_int32(int(w_bits))))

    def destroy(self):
        if self.h:
            self.eng.lib.rhip_ac17_pk_destroy(self.h)
            self.h = None


def _sz(n):
    return ctypes.c_size_t(int(n))


def wide_windows(w):
    """number of signed w-bit windows of a 254-bit scalar (engine.hip: wide_windows)"""
    return (254 + w - 1) // w


def wide_count(w, i):
    """entries of window i of the signed w-bit table (engine.hip: wide_count)"""
    n = wide_windows(w)
    return (1 << (w - 1)) if i < n - 1 else (1 << (254 - w * (n - 1)))


def ac17_encrypt_dev(eng, pk, n_items, dA, ditem_A_off, dct_row_off, total_rows, ds, dmsg, dc0, dc, dcp):
    eng._check(eng.lib.rhip_ac17_cp_encrypt_batch(eng.ctx, pk.h, _sz(n_items), dA.ptr, ditem_A_off.ptr, dct_row_off.ptr,
                                                  _sz(total_rows), ds.ptr, dmsg.ptr, dc0.ptr, dc.ptr, dcp.ptr))


def ac17_keygen_dev(eng, g_table, h_table, dgk, dainv, db, n_items, n_attrs, dH, dH01, dr, dsigma, dsigmap, dk0, dk, dkp):
    eng._check(eng.lib.rhip_ac17_cp_keygen_batch(eng.ctx, g_table.h, h_table.h, dgk.ptr, dainv.ptr, db.ptr, _sz(n_items),
                                                 _sz(n_attrs), dH.ptr, dH01.ptr, dr.ptr, dsigma.ptr, dsigmap.ptr,
                                                 dk0.ptr, dk.ptr, dkp.ptr))


def ac17_decrypt_dev(eng, n_items, dct_c0, dct_c, dct_row_off, dct_cp, dsk_k0, dsk_k, dsk_row_off, dsk_kp, dsk_idx,
                     dct_sel, dct_sel_off, dsk_sel, dsk_sel_off, dout):
    eng._check(eng.lib.rhip_ac17_cp_decrypt_batch(eng.ctx, _sz(n_items), dct_c0.ptr, dct_c.ptr, dct_row_off.ptr, dct_cp.ptr,
                                                  dsk_k0.ptr, dsk_k.ptr, dsk_row_off.ptr, dsk_kp.ptr, dsk_idx.ptr,
                                                  dct_sel.ptr, dct_sel_off.ptr, dsk_sel.ptr, dsk_sel_off.ptr, dout.ptr))


class Ac17SkLines:
    """Prepared Miller-loop lines of n_sk secret keys' k_0 (rhip_ac17_sk_prepare)."""

    def __init__(self, eng, n_sk, dsk_k0):
        self.eng = eng
        self.h = ctypes.c_void_p()
        eng._check(eng.lib.rhip_ac17_sk_prepare(eng.ctx, _sz(n_sk), dsk_k0.ptr, ctypes.byref(self.h)))

    def destroy(self):
        if self.h:
            self.eng.lib.rhip_ac17_sk_lines_destroy(self.h)
            self.h = None


def ac17_decrypt_prepared_dev(eng, n_items, dct_c0, dct_c, dct_row_off, dct_cp, sk_lines, dsk_k, dsk_row_off, dsk_kp, dsk_idx,
                              dct_sel, dct_sel_off, dsk_sel, dsk_sel_off, dout):
    eng._check(eng.lib.rhip_ac17_cp_decrypt_batch_prepared(eng.ctx, _sz(n_items), dct_c0.ptr, dct_c.ptr, dct_row_off.ptr, dct_cp.ptr,
                                                           sk_lines.h, dsk_k.ptr, dsk_row_off.ptr, dsk_kp.ptr, dsk_idx.ptr,
                                                           dct_sel.ptr, dct_sel_off.ptr, dsk_sel.ptr, dsk_sel_off.ptr, dout.ptr))


# ---------------------------------------------------------------------- Level B: bsw / lsw / aw11 (device-buffer level)
def _p(x):
    """DevBuf / handle / None -> pointer argument"""
    if x is None:
        return ctypes.c_void_p(0)
    if hasattr(x, "ptr"):
        return x.ptr
    if hasattr(x, "h"):
        return x.h
    return x


class DevTreeTables:
    """hostprep.TreeTables uploaded once (flattened policy trees: include/rabe_hip.h)."""

    def __init__(self, eng, tt):
        self.tt = tt
        self.path_off = eng.upload_u32(tt.path_off)
        self.path_gate = eng.upload_u32(tt.path_gate or [0])
        self.path_x = eng.upload_u32(tt.path_x or [0])
        self.gate_k = eng.upload_u32(tt.gate_k or [0])
        self.gate_coef_off = eng.upload_u32(tt.gate_coef_off or [0])
        self.leaf_hash = eng.upload(b"".join(fr_bytes(h) for h in tt.leaf_hash))


class G2Lines:
    """Prepared Miller-loop lines of n G2 points (rhip_g2_lines_prepare)."""

    def __init__(self, eng, n, dq):
        self.eng = eng
        self.h = ctypes.c_void_p()
        eng._check(eng.lib.rhip_g2_lines_prepare(eng.ctx, _sz(n), dq.ptr, ctypes.byref(self.h)))

    def destroy(self):
        if self.h:
            self.eng.lib.rhip_g2_lines_destroy(self.h)
            self.h = None


class BswPk:
    """Device tables of a CpAbePublicKey (g1, g2, h, e_gg_alpha)."""

    def __init__(self, eng, g1, g2, h, e_gg_alpha):
        self.eng = eng
        self.h = ctypes.c_void_p()
        eng._check(eng.lib.rhip_bsw_pk_create(eng.ctx, bytes(g1), bytes(g2), bytes(h), bytes(e_gg_alpha), ctypes.byref(self.h)))

    def destroy(self):
        if self.h:
            self.eng.lib.rhip_bsw_pk_destroy(self.h)
            self.h = None


class BswSkLines:
    def __init__(self, eng, n_sk, total_attrs, dsk_d, dsk_dj_g2):
        self.eng = eng
        self.h = ctypes.c_void_p()
        eng._check(eng.lib.rhip_bsw_sk_prepare(eng.ctx, _sz(n_sk), _sz(total_attrs), dsk_d.ptr, dsk_dj_g2.ptr, ctypes.byref(self.h)))

    def destroy(self):
        if self.h:
            self.eng.lib.rhip_bsw_sk_lines_destroy(self.h)
            self.h = None


def bsw_encrypt_dev(eng, pk, n_items, total_leaves, d_item_leaf_off, d_item_tree_leaf, d_item_tree_gate, dtt, d_secret, d_coef,
                    d_item_coef_off, d_msg, d_c, d_cp, d_cy_g1, d_cy_g2):
    eng._check(eng.lib.rhip_bsw_encrypt_batch(eng.ctx, pk.h, _sz(n_items), _sz(total_leaves), _p(d_item_leaf_off), _p(d_item_tree_leaf),
                                              _p(d_item_tree_gate), _p(dtt.path_off), _p(dtt.path_gate), _p(dtt.path_x), _p(dtt.gate_k),
                                              _p(dtt.gate_coef_off), _p(dtt.leaf_hash), _p(d_secret), _p(d_coef), _p(d_item_coef_off),
                                              _p(d_msg), _p(d_c), _p(d_cp), _p(d_cy_g1), _p(d_cy_g2)))


def bsw_decrypt_dev(eng, n_items, max_pairs, total_pairs, d_pair_off, d_sel_start, d_sel_ct_leaf, d_sel_sk_attr, d_sel_coeff,
                    d_ct_c, d_ct_cp, d_ct_cy_g1, d_ct_cy_g2, d_ct_leaf_off, d_sk_d, d_sk_dj_g1, d_sk_dj_g2, d_sk_attr_off, d_sk_idx,
                    sk_lines, d_out):
    eng._check(eng.lib.rhip_bsw_decrypt_batch(eng.ctx, _sz(n_items), _sz(max_pairs), _sz(total_pairs), _p(d_pair_off), _p(d_sel_start),
                                              _p(d_sel_ct_leaf), _p(d_sel_sk_attr), _p(d_sel_coeff), _p(d_ct_c), _p(d_ct_cp), _p(d_ct_cy_g1),
                                              _p(d_ct_cy_g2), _p(d_ct_leaf_off), _p(d_sk_d), _p(d_sk_dj_g1), _p(d_sk_dj_g2),
                                              _p(d_sk_attr_off), _p(d_sk_idx), _p(sk_lines), _p(d_out)))


def bsw_decrypt_one_sk_dev(eng, n_items, max_pairs, total_pairs, n_sel, d_pair_off, d_sel_start, d_sel_ct_leaf, d_sel_sk_attr, d_sel_coeff,
                           d_ct_c, d_ct_cp, d_ct_cy_g1, d_ct_cy_g2, d_ct_leaf_off, d_sk_d, d_sk_dj_g1, d_sk_dj_g2, d_sk_attr_off, sk_lines, d_out):
    """every item is decrypted with the SAME key (rhip_bsw_decrypt_batch_one_sk): -z_e * Dj.g1 is computed once per selection entry"""
    eng._check(eng.lib.rhip_bsw_decrypt_batch_one_sk(eng.ctx, _sz(n_items), _sz(max_pairs), _sz(total_pairs), _sz(n_sel), _p(d_pair_off), _p(d_sel_start),
                                                     _p(d_sel_ct_leaf), _p(d_sel_sk_attr), _p(d_sel_coeff), _p(d_ct_c), _p(d_ct_cp), _p(d_ct_cy_g1),
                                                     _p(d_ct_cy_g2), _p(d_ct_leaf_off), _p(d_sk_d), _p(d_sk_dj_g1), _p(d_sk_dj_g2), _p(d_sk_attr_off),
                                                     _p(sk_lines), _p(d_out)))


class LswPk:
    def __init__(self, eng, g1, g2):
        self.eng = eng
        self.h = ctypes.c_void_p()
        eng._check(eng.lib.rhip_lsw_pk_create(eng.ctx, bytes(g1), bytes(g2), ctypes.byref(self.h)))

    def destroy(self):
        if self.h:
            self.eng.lib.rhip_lsw_pk_destroy(self.h)
            self.h = None


def lsw_keygen_dev(eng, pk, n_items, total_leaves, d_item_leaf_off, d_item_tree_leaf, d_item_tree_gate, dtt, d_alpha, d_coef, d_item_coef_off,
                   d_rand, d_d1, d_d2):
    eng._check(eng.lib.rhip_lsw_keygen_batch(eng.ctx, pk.h, _sz(n_items), _sz(total_leaves), _p(d_item_leaf_off), _p(d_item_tree_leaf),
                                             _p(d_item_tree_gate), _p(dtt.path_off), _p(dtt.path_gate), _p(dtt.path_x), _p(dtt.gate_k),
                                             _p(dtt.gate_coef_off), _p(dtt.leaf_hash), _p(d_alpha), _p(d_coef), _p(d_item_coef_off), _p(d_rand),
                                             _p(d_d1), _p(d_d2)))


def lsw_keygen_signed_dev(eng, pk, n_items, total_leaves, d_item_leaf_off, d_item_tree_leaf, d_item_tree_gate, dtt, d_leaf_neg, d_alpha, d_b, h_g1,
                          d_coef, d_item_coef_off, d_rand, d_d1, d_d2, d_d3, d_d4, d_d5):
    """rhip_lsw_keygen_batch_signed: policies with negative leaves ("!x"); h_g1 = the master key's h_g1 (64 host bytes)"""
    eng._check(eng.lib.rhip_lsw_keygen_batch_signed(eng.ctx, pk.h, _sz(n_items), _sz(total_leaves), _p(d_item_leaf_off), _p(d_item_tree_leaf),
                                                    _p(d_item_tree_gate), _p(dtt.path_off), _p(dtt.path_gate), _p(dtt.path_x), _p(dtt.gate_k),
                                                    _p(dtt.gate_coef_off), _p(dtt.leaf_hash), _p(d_leaf_neg), _p(d_alpha), _p(d_b), bytes(h_g1),
                                                    _p(d_coef), _p(d_item_coef_off), _p(d_rand), _p(d_d1), _p(d_d2), _p(d_d3), _p(d_d4), _p(d_d5)))


def lsw_decrypt_dev(eng, n_items, max_pairs, total_pairs, n_sel, d_pair_off, d_sel_start, d_sel_sk_leaf, d_sel_ct_attr, d_sel_coeff, d_ct_e1,
                    d_ct_e2, d_ct_e1j, d_ct_attr_off, d_ct_idx, d_sk_d1, d_sk_d2, d_sk_leaf_off, d_sk_idx, e2_lines, d_out):
    eng._check(eng.lib.rhip_lsw_decrypt_batch(eng.ctx, _sz(n_items), _sz(max_pairs), _sz(total_pairs), _sz(n_sel), _p(d_pair_off), _p(d_sel_start),
                                              _p(d_sel_sk_leaf), _p(d_sel_ct_attr), _p(d_sel_coeff), _p(d_ct_e1), _p(d_ct_e2), _p(d_ct_e1j),
                                              _p(d_ct_attr_off), _p(d_ct_idx), _p(d_sk_d1), _p(d_sk_d2), _p(d_sk_leaf_off), _p(d_sk_idx),
                                              _p(e2_lines), _p(d_out)))


def lsw_decrypt_one_ct_dev(eng, n_items, max_pairs, total_pairs, n_sel, d_pair_off, d_sel_start, d_sel_sk_leaf, d_sel_ct_attr, d_sel_coeff, d_ct_e1,
                           d_ct_e2, d_ct_e1j, d_sk_d1, d_sk_d2, d_sk_leaf_off, d_sk_idx, e2_lines, d_out):
    """every item decrypts the SAME ciphertext (rhip_lsw_decrypt_batch_one_ct): the scaled G1 arguments are computed once per selection entry"""
    eng._check(eng.lib.rhip_lsw_decrypt_batch_one_ct(eng.ctx, _sz(n_items), _sz(max_pairs), _sz(total_pairs), _sz(n_sel), _p(d_pair_off), _p(d_sel_start),
                                                     _p(d_sel_sk_leaf), _p(d_sel_ct_attr), _p(d_sel_coeff), _p(d_ct_e1), _p(d_ct_e2), _p(d_ct_e1j),
                                                     _p(d_sk_d1), _p(d_sk_d2), _p(d_sk_leaf_off), _p(d_sk_idx), _p(e2_lines), _p(d_out)))


class Aw11Pk:
    def __init__(self, eng, g1, g2, egg_alpha, g2_y):
        self.eng = eng
        self.h = ctypes.c_void_p()
        eng._check(eng.lib.rhip_aw11_pk_create(eng.ctx, bytes(g1), bytes(g2), _sz(len(egg_alpha)), b"".join(egg_alpha), b"".join(g2_y),
                                               ctypes.byref(self.h)))

    def destroy(self):
        if self.h:
            self.eng.lib.rhip_aw11_pk_destroy(self.h)
            self.h = None


def aw11_encrypt_dev(eng, pk, n_items, total_rows, d_item_row_off, d_item_tree_leaf, d_item_tree_gate, d_item_n_coef, dtt, d_leaf_attr, d_s,
                     d_coef, d_item_coef_off, d_rand, d_msg, d_c0, d_c1, d_c2, d_c3):
    eng._check(eng.lib.rhip_aw11_encrypt_batch(eng.ctx, pk.h, _sz(n_items), _sz(total_rows), _p(d_item_row_off), _p(d_item_tree_leaf),
                                               _p(d_item_tree_gate), _p(d_item_n_coef), _p(dtt.path_off), _p(dtt.path_gate), _p(dtt.path_x),
                                               _p(dtt.gate_k), _p(dtt.gate_coef_off), _p(d_leaf_attr), _p(d_s), _p(d_coef), _p(d_item_coef_off),
                                               _p(d_rand), _p(d_msg), _p(d_c0), _p(d_c1), _p(d_c2), _p(d_c3)))


def aw11_decrypt_dev(eng, n_items, max_pairs, total_pairs, n_sel, d_pair_off, d_sel_start, d_sel_ct_row, d_sel_sk_attr, d_sel_coeff, d_ct_c0,
                     d_ct_c1, d_ct_c2, d_ct_c3, d_ct_row_off, d_sk_hash, d_sk_k, d_sk_attr_off, d_sk_idx, d_out):
    eng._check(eng.lib.rhip_aw11_decrypt_batch(eng.ctx, _sz(n_items), _sz(max_pairs), _sz(total_pairs), _sz(n_sel), _p(d_pair_off), _p(d_sel_start),
                                               _p(d_sel_ct_row), _p(d_sel_sk_attr), _p(d_sel_coeff), _p(d_ct_c0), _p(d_ct_c1), _p(d_ct_c2),
                                               _p(d_ct_c3), _p(d_ct_row_off), _p(d_sk_hash), _p(d_sk_k), _p(d_sk_attr_off), _p(d_sk_idx), _p(d_out)))
