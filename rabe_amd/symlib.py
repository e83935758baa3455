"""ctypes plumbing of Level S (include/rabe_hip.h: SHA3-256, the Gt KDF, AES-256-GCM on the device; rabe_amd/csrc/engine_sym.hip).

Host bytes in, host bytes out -- for the tests and tools; the C++ host layer calls the same entry points on device-resident data."""
import ctypes
import struct

from .engine import FR


def _u64(values):
    return struct.pack("<%dQ" % len(values), *values)


def _u32(values):
    return struct.pack("<%dI" % len(values), *values)


def _offsets(items):
    off = [0]
    for x in items:
        off.append(off[-1] + len(x))
    return off


def shape(lengths):
    """(blk_off, seg_off): prefix sums of the 16-byte blocks and of the 64-block GHASH segments of every item"""
    blk, seg = [0], [0]
    for n in lengths:
        b = (n + 15) // 16
        blk.append(blk[-1] + b)
        seg.append(seg[-1] + (b + 63) // 64)
    return blk, seg


def sha3_256(eng, messages):
    n = len(messages)
    off = _offsets(messages)
    data, doff, out = eng.upload(b"".join(messages) or b"\0"), eng.upload(_u64(off)), eng.alloc(32 * n)
    eng._check(eng.lib.rhip_sha3_256_batch(eng.ctx, ctypes.c_size_t(n), data.ptr, doff.ptr, out.ptr))
    raw = eng.download(out, 32 * n)
    return [raw[32 * i:32 * i + 32] for i in range(n)]


def sha3_fr(eng, messages):
    n = len(messages)
    off = _offsets(messages)
    data, doff, out = eng.upload(b"".join(messages) or b"\0"), eng.upload(_u64(off)), eng.alloc(FR * n)
    eng._check(eng.lib.rhip_sha3_fr_batch(eng.ctx, ctypes.c_size_t(n), data.ptr, doff.ptr, out.ptr))
    raw = eng.download(out, FR * n)
    return [raw[FR * i:FR * i + FR] for i in range(n)]


def gt_kdf(eng, gts, idx=None):
    n = len(idx) if idx is not None else len(gts)
    d, out = eng.upload(b"".join(gts)), eng.alloc(32 * n)
    di = eng.upload_u32(idx) if idx is not None else None
    eng._check(eng.lib.rhip_gt_kdf_batch(eng.ctx, ctypes.c_size_t(n), d.ptr, di.ptr if di else None, out.ptr))
    raw = eng.download(out, 32 * n)
    return [raw[32 * i:32 * i + 32] for i in range(n)]


def aes256_blocks(eng, keys, blocks):
    n = len(keys)
    dk, di, out = eng.upload(b"".join(keys)), eng.upload(b"".join(blocks)), eng.alloc(16 * n)
    eng._check(eng.lib.rhip_aes256_encrypt_blocks(eng.ctx, ctypes.c_size_t(n), dk.ptr, di.ptr, out.ptr))
    raw = eng.download(out, 16 * n)
    return [raw[16 * i:16 * i + 16] for i in range(n)]


class _Keep:
    """device buffers of one call, kept alive until the results are back (a DevBuf frees its memory when collected)"""

    def __init__(self, eng):
        self.eng, self.bufs = eng, []

    def u64(self, values):
        b = self.eng.upload(_u64(values) or b"\0" * 8)
        self.bufs.append(b)
        return b.ptr

    def u32(self, values):
        b = self.eng.upload(_u32(values) or b"\0" * 4)
        self.bufs.append(b)
        return b.ptr


def _ws(eng, n, segs):
    eng.lib.rhip_seal_workspace_bytes.restype = ctypes.c_size_t
    return eng.alloc(eng.lib.rhip_seal_workspace_bytes(ctypes.c_size_t(n), ctypes.c_size_t(segs)))


def gcm_seal(eng, keys, nonces, plaintexts, len_prefix=False):
    """AES-256-GCM with explicit keys: [nonce || ct || tag] per item (with len_prefix: a u32 length in front, as in the records)"""
    n = len(keys)
    lens = [len(p) for p in plaintexts]
    blk, seg = shape(lens)
    pre = 4 if len_prefix else 0
    pt_off = _offsets(plaintexts)[:-1]
    out_off, pos = [], 0
    for ln in lens:
        out_off.append(pos + pre)
        pos += pre + ln + 28
    dk, dn, dp = eng.upload(b"".join(keys)), eng.upload(b"".join(nonces)), eng.upload(b"".join(plaintexts) or b"\0")
    out = eng.alloc(pos)
    ws = _ws(eng, n, seg[-1])
    k = _Keep(eng)
    eng._check(eng.lib.rhip_aes256_gcm_batch(eng.ctx, ctypes.c_int32(1), ctypes.c_size_t(n), dk.ptr, dn.ptr, dp.ptr, k.u64(pt_off),
                                             out.ptr, k.u64(out_off), k.u32(lens), k.u32(blk),
                                             ctypes.c_size_t(blk[-1]), k.u32(seg), ctypes.c_size_t(seg[-1]),
                                             ctypes.c_int32(1 if len_prefix else 0), None, ws.ptr))
    raw = eng.download(out, pos)
    return [raw[o - pre:o + ln + 28] for o, ln in zip(out_off, lens)]


def gcm_open(eng, keys, sealed):
    """-> (plaintexts, ok flags); sealed = nonce || ct || tag per item (>= 28 bytes each)"""
    n = len(keys)
    lens = [len(s) - 28 for s in sealed]
    assert min(lens) >= 0
    blk, seg = shape(lens)
    s_off = _offsets(sealed)[:-1]
    pt_off = [0]
    for ln in lens:
        pt_off.append(pt_off[-1] + ln)
    dk, dblob = eng.upload(b"".join(keys)), eng.upload(b"".join(sealed))
    out, ok = eng.alloc(max(pt_off[-1], 4)), eng.alloc(4 * n)
    ws = _ws(eng, n, seg[-1])
    k = _Keep(eng)
    eng._check(eng.lib.rhip_aes256_gcm_batch(eng.ctx, ctypes.c_int32(0), ctypes.c_size_t(n), dk.ptr, None, dblob.ptr, k.u64(s_off),
                                             out.ptr, k.u64(pt_off[:-1]), k.u32(lens), k.u32(blk),
                                             ctypes.c_size_t(blk[-1]), k.u32(seg), ctypes.c_size_t(seg[-1]), ctypes.c_int32(0), ok.ptr,
                                             ws.ptr))
    raw = eng.download(out, max(pt_off[-1], 4))
    flags = struct.unpack("<%dI" % n, eng.download(ok, 4 * n))
    return [raw[pt_off[i]:pt_off[i + 1]] for i in range(n)], list(flags)


def seal(eng, gts, nonces, plaintexts):
    """n calls of encrypt_symmetric (aes/mod.rs:10-26) with explicit nonces"""
    n = len(gts)
    lens = [len(p) for p in plaintexts]
    blk, seg = shape(lens)
    pt_off = _offsets(plaintexts)[:-1]
    out_off, pos = [], 0
    for ln in lens:
        out_off.append(pos)
        pos += ln + 28
    dg, dn, dp = eng.upload(b"".join(gts)), eng.upload(b"".join(nonces)), eng.upload(b"".join(plaintexts) or b"\0")
    out = eng.alloc(pos)
    ws = _ws(eng, n, seg[-1])
    k = _Keep(eng)
    eng._check(eng.lib.rhip_seal_batch(eng.ctx, ctypes.c_size_t(n), dg.ptr, dn.ptr, dp.ptr, k.u64(pt_off), out.ptr,
                                       k.u64(out_off), k.u32(lens), k.u32(blk), ctypes.c_size_t(blk[-1]),
                                       k.u32(seg), ctypes.c_size_t(seg[-1]), ctypes.c_int32(0), ws.ptr))
    raw = eng.download(out, pos)
    return [raw[o:o + ln + 28] for o, ln in zip(out_off, lens)]


def open_(eng, gts, sealed, idx=None):
    """n calls of decrypt_symmetric (aes/mod.rs:29-44) -> (plaintexts, ok flags)"""
    n = len(sealed)
    lens = [len(s) - 28 for s in sealed]
    blk, seg = shape(lens)
    s_off = _offsets(sealed)[:-1]
    pt_off = [0]
    for ln in lens:
        pt_off.append(pt_off[-1] + ln)
    dg, dblob = eng.upload(b"".join(gts)), eng.upload(b"".join(sealed))
    di = eng.upload_u32(idx) if idx is not None else None
    out, ok = eng.alloc(max(pt_off[-1], 4)), eng.alloc(4 * n)
    ws = _ws(eng, n, seg[-1])
    k = _Keep(eng)
    eng._check(eng.lib.rhip_open_batch(eng.ctx, ctypes.c_size_t(n), dg.ptr, di.ptr if di else None, dblob.ptr, k.u64(s_off), out.ptr,
                                       k.u64(pt_off[:-1]), k.u32(lens), k.u32(blk), ctypes.c_size_t(blk[-1]),
                                       k.u32(seg), ctypes.c_size_t(seg[-1]), ok.ptr, ws.ptr))
    raw = eng.download(out, max(pt_off[-1], 4))
    flags = struct.unpack("<%dI" % n, eng.download(ok, 4 * n))
    return [raw[pt_off[i]:pt_off[i + 1]] for i in range(n)], list(flags)
