"""rabe's serde form  <->  this engine's canonical records (wire compatibility with real rabe).

rabe derives Serialize / Deserialize on every key and ciphertext struct (e.g. /root/reference/src/schemes/ac17/mod.rs:58-135), and the
group elements inside are rabe-bn's own serde forms -- whose layout is unknown here (rabe-bn is not in /root/reference).  This module is
the converter that needs nothing but that layout: it walks rabe's struct shapes (field names and tuple positions as the reference
declares them) and hands every Fr / G1 / G2 / Gt it meets to an ELEMENT CODEC -- `dec(kind, serde_value) -> canonical bytes` and
`enc(kind, canonical bytes) -> serde_value`.  tests/refpin.py learns such a codec from the vectors integration/ref-harness dumps from
real rabe (`Source("serde")` after `learn`); tests/wire_codec.py wraps that into a codec.  The codec is always an ARGUMENT here: this
module imports neither the CPU checker nor the test suite.  So the day `tests/golden/ref_*.json` exist:

    src = refpin.Source("serde"); ...learn...                     (tests/test_ref_pin.py does this)
    dec, enc = tests.wire_codec.codec_from_source(src, samples)
    blob = wire_compat.to_canonical("ac17_cp_ct", json.loads(rabe_json), dec)      # -> hostlib.Obj.deserialize / *_decrypt_packed
    back = wire_compat.from_canonical("ac17_cp_ct", blob, enc)                     # -> serde_json::from_str on the rabe side

The same shapes drive rabe's BORSH form (`*_borsh` below; borsh = fields in declaration order, Vec / String with a u32 length, an enum
as its u8 variant index) -- what rabe-console writes by default: `-----BEGIN CT-----\n hex(deflate(borsh(struct))) \n-----END CT-----`
(/root/reference/rabe-console/src/mod.rs:82-105, 1623-1683; `read_envelope` / `write_envelope`).

The canonical side is the byte form of rabe_amd/csrc/host/host_abi.cpp (`ser` / `deser`).  No group arithmetic happens here."""
import struct
import zlib

FR, G1, G2, GT = "fr", "g1", "g2", "gt"
SIZE = {FR: 32, G1: 64, G2: 128, GT: 384}


# ------------------------------------------------------------------------------------------------ canonical byte form (host_abi.cpp: W / R)
class _W:
    def __init__(self):
        self.b = bytearray()

    def u8(self, v):
        self.b.append(v)

    def u32(self, v):
        self.b += struct.pack("<I", v)

    def s(self, text):
        raw = text.encode("utf-8")
        self.u32(len(raw))
        self.b += raw

    def raw(self, data):
        self.b += data

    def bytes_(self, data):
        self.u32(len(data))
        self.b += bytes(data)


class _R:
    def __init__(self, data):
        self.b, self.o = bytes(data), 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("canonical record truncated")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def u8(self):
        return self.take(1)[0]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def s(self):
        return self.take(self.u32()).decode("utf-8")

    def bytes_(self):
        return self.take(self.u32())


def _lang_to_json(v):
    return "HumanPolicy" if v else "JsonPolicy"


def _lang_from_json(v):
    if v not in ("JsonPolicy", "HumanPolicy"):
        raise ValueError("PolicyLanguage %r" % (v,))
    return 1 if v == "HumanPolicy" else 0


# ------------------------------------------------------------------------------------------------ struct shapes
# A shape is a tree of: an element kind ("fr" / "g1" / "g2" / "gt"), "str", "bytes", "policy" ((String, PolicyLanguage)),
# ("vec", shape), ("fixed", n, shape) (a Vec of exactly n entries on the canonical side: AC17's ASSUMPTION_SIZE vectors),
# ("tuple", [shapes]) (a Rust tuple: JSON array, positional), ("struct", [(field, shape)...]) (JSON object, declaration order),
# ("optel", kind) (an element slot that the canonical form always carries -- identity = zeros -- where JSON may hold null).
# Field names / tuple positions: /root/reference/src/schemes/{ac17,bsw,lsw,aw11}/mod.rs struct definitions; canonical order: host_abi.cpp ser().
def _struct(*fields):
    return ("struct", list(fields))


_AC17_SK = _struct(("k_0", ("fixed", 3, G2)), ("k", ("vec", ("tuple", ["str", ("fixed", 3, G1)]))), ("k_p", ("fixed", 3, G1)))
_AC17_KP_SK = _struct(("k_0", ("fixed", 3, G2)), ("k", ("vec", ("tuple", ["str", ("fixed", 3, G1)]))), ("k_p", ("vec", G1)))   # kp_keygen: k_p = Vec::new() (:540)
_AC17_CT = _struct(("c_0", ("fixed", 3, G2)), ("c", ("vec", ("tuple", ["str", ("fixed", 3, G1)]))), ("c_p", GT), ("ct", "bytes"))
_BSW_ATTR = _struct(("string", "str"), ("g1", G1), ("g2", G2))
SHAPES = {
    "ac17_pk": _struct(("g", G1), ("h_a", ("fixed", 3, G2)), ("e_gh_ka", ("fixed", 2, GT))),
    "ac17_msk": _struct(("g", G1), ("h", G2), ("g_k", ("fixed", 3, G1)), ("a", ("fixed", 2, FR)), ("b", ("fixed", 2, FR))),
    "ac17_cp_sk": _struct(("attr", ("vec", "str")), ("sk", _AC17_SK)),
    "ac17_cp_ct": _struct(("policy", "policy"), ("ct", _AC17_CT)),
    "ac17_kp_sk": _struct(("policy", "policy"), ("sk", _AC17_KP_SK)),
    "ac17_kp_ct": _struct(("attr", ("vec", "str")), ("ct", _AC17_CT)),
    "bsw_pk": _struct(("g1", G1), ("g2", G2), ("h", G1), ("f", G2), ("e_gg_alpha", GT)),
    "bsw_msk": _struct(("beta", FR), ("g2_alpha", G2)),
    "bsw_sk": _struct(("d", G2), ("d_j", ("vec", _BSW_ATTR))),
    "bsw_ct": _struct(("policy", "policy"), ("c", G1), ("c_p", GT), ("c_y", ("vec", _BSW_ATTR)), ("data", "bytes")),
    "lsw_pk": _struct(("g1", G1), ("g2", G2), ("g1_b", G1), ("g1_b2", G1), ("h_b", G1), ("e_gg_alpha", GT)),
    "lsw_msk": _struct(("alpha1", FR), ("alpha2", FR), ("b", FR), ("h_g1", G1), ("h_g2", G2)),
    "lsw_sk": _struct(("policy", "policy"), ("dj", ("vec", ("tuple", ["str", ("optel", G1), ("optel", G2), ("optel", G1), ("optel", G1), ("optel", G1)])))),
    "lsw_ct": _struct(("e1", GT), ("e2", G2), ("ej", ("vec", ("tuple", ["str", G1, G1, G1]))), ("ct", "bytes")),
    "aw11_gk": _struct(("g1", G1), ("g2", G2)),
    "aw11_pk": _struct(("attr", ("vec", ("tuple", ["str", GT, G2])))),
    "aw11_msk": _struct(("attr", ("vec", ("tuple", ["str", FR, FR])))),
    "aw11_sk": _struct(("gid", "str"), ("attr", ("vec", ("tuple", ["str", G1])))),
    "aw11_ct": _struct(("policy", "policy"), ("c_0", GT), ("c", ("vec", ("tuple", ["str", GT, G2, G2]))), ("ct", "bytes")),
}
# canonical vectors that carry NO length prefix... none: every Vec is u32-prefixed in host_abi.cpp, fixed-size ones included.
# The one exception to "declaration order": Ac17CpSecretKey / KpCiphertext write the attribute strings as u32 count + strings (a vec of str).


def _to_canon(shape, v, dec, w):
    if isinstance(shape, str) and shape in SIZE:
        w.raw(dec(shape, v))
    elif shape == "str":
        w.s(v)
    elif shape == "bytes":
        w.bytes_(bytes(v))
    elif shape == "policy":
        w.s(v[0])
        w.u8(_lang_from_json(v[1]))
    elif shape[0] == "vec":
        w.u32(len(v))
        for x in v:
            _to_canon(shape[1], x, dec, w)
    elif shape[0] == "fixed":
        if len(v) != shape[1]:
            raise ValueError("expected %d elements, got %d" % (shape[1], len(v)))
        w.u32(len(v))
        for x in v:
            _to_canon(shape[2], x, dec, w)
    elif shape[0] == "tuple":
        if len(v) != len(shape[1]):
            raise ValueError("tuple of %d, got %d" % (len(shape[1]), len(v)))
        for sh, x in zip(shape[1], v):
            _to_canon(sh, x, dec, w)
    elif shape[0] == "optel":
        w.raw(bytes(SIZE[shape[1]]) if v is None else dec(shape[1], v))
    elif shape[0] == "struct":
        for name, sh in shape[1]:
            _to_canon(sh, v[name], dec, w)
    else:
        raise ValueError(shape)


def _from_canon(shape, r, enc):
    if isinstance(shape, str) and shape in SIZE:
        return enc(shape, r.take(SIZE[shape]))
    if shape == "str":
        return r.s()
    if shape == "bytes":
        return list(r.bytes_())
    if shape == "policy":
        text = r.s()
        return [text, _lang_to_json(r.u8())]
    if shape[0] == "vec":
        return [_from_canon(shape[1], r, enc) for _ in range(r.u32())]
    if shape[0] == "fixed":
        n = r.u32()
        if n != shape[1]:
            raise ValueError("expected %d elements, got %d" % (shape[1], n))
        return [_from_canon(shape[2], r, enc) for _ in range(n)]
    if shape[0] == "tuple":
        return [_from_canon(sh, r, enc) for sh in shape[1]]
    if shape[0] == "optel":
        return enc(shape[1], r.take(SIZE[shape[1]]))          # rabe's tuple holds an element in every slot: the identity where unused
    if shape[0] == "struct":
        return {name: _from_canon(sh, r, enc) for name, sh in shape[1]}
    raise ValueError(shape)


def to_canonical(kind, obj, dec):
    """rabe's serde value of a struct (json.loads of serde_json::to_string) -> this engine's canonical record"""
    w = _W()
    _to_canon(SHAPES[kind], obj, dec, w)
    return bytes(w.b)


def from_canonical(kind, data, enc):
    """this engine's canonical record -> rabe's serde value (json.dumps of it is what serde_json::from_str takes)"""
    r = _R(data)
    out = _from_canon(SHAPES[kind], r, enc)
    if r.o != len(r.b):
        raise ValueError("trailing bytes after the canonical record")
    return out


# ------------------------------------------------------------------------------------------------ borsh form of the same structs
def _borsh_to_canon(shape, r, dec, size, w):
    if isinstance(shape, str) and shape in SIZE:
        w.raw(dec(shape, r.take(size[shape])))
    elif shape == "str":
        w.s(r.s())
    elif shape == "bytes":
        w.bytes_(r.bytes_())
    elif shape == "policy":                                   # (String, PolicyLanguage): the enum is its variant index (JsonPolicy = 0)
        w.s(r.s())
        lang = r.u8()
        if lang > 1:
            raise ValueError("PolicyLanguage variant %d" % lang)
        w.u8(lang)
    elif shape[0] in ("vec", "fixed"):
        n = r.u32()
        if shape[0] == "fixed" and n != shape[1]:
            raise ValueError("expected %d elements, got %d" % (shape[1], n))
        w.u32(n)
        for _ in range(n):
            _borsh_to_canon(shape[-1], r, dec, size, w)
    elif shape[0] == "tuple":
        for sh in shape[1]:
            _borsh_to_canon(sh, r, dec, size, w)
    elif shape[0] == "optel":
        w.raw(dec(shape[1], r.take(size[shape[1]])))
    elif shape[0] == "struct":
        for _name, sh in shape[1]:
            _borsh_to_canon(sh, r, dec, size, w)
    else:
        raise ValueError(shape)


def _canon_to_borsh(shape, r, enc, w):
    if isinstance(shape, str) and shape in SIZE:
        w.raw(enc(shape, r.take(SIZE[shape])))
    elif shape == "str":
        w.s(r.s())
    elif shape == "bytes":
        w.bytes_(r.bytes_())
    elif shape == "policy":
        w.s(r.s())
        w.u8(r.u8())
    elif shape[0] in ("vec", "fixed"):
        n = r.u32()
        if shape[0] == "fixed" and n != shape[1]:
            raise ValueError("expected %d elements, got %d" % (shape[1], n))
        w.u32(n)
        for _ in range(n):
            _canon_to_borsh(shape[-1], r, enc, w)
    elif shape[0] == "tuple":
        for sh in shape[1]:
            _canon_to_borsh(sh, r, enc, w)
    elif shape[0] == "optel":
        w.raw(enc(shape[1], r.take(SIZE[shape[1]])))
    elif shape[0] == "struct":
        for _name, sh in shape[1]:
            _canon_to_borsh(sh, r, enc, w)
    else:
        raise ValueError(shape)


def to_canonical_borsh(kind, data, codec):
    """borsh bytes of a rabe struct -> canonical record.  codec = (dec, enc, size), e.g. of tests/wire_codec.py `borsh_codec_from_source`."""
    dec, _enc, size = codec
    r, w = _R(data), _W()
    _borsh_to_canon(SHAPES[kind], r, dec, size, w)
    if r.o != len(r.b):
        raise ValueError("trailing bytes after the borsh record")
    return bytes(w.b)


def from_canonical_borsh(kind, data, codec):
    """canonical record -> the borsh bytes rabe's `try_from_slice` takes"""
    _dec, enc, _size = codec
    r, w = _R(data), _W()
    _canon_to_borsh(SHAPES[kind], r, enc, w)
    if r.o != len(r.b):
        raise ValueError("trailing bytes after the canonical record")
    return bytes(w.b)


# ------------------------------------------------------------------------------------------------ rabe-console's file envelope
# label -> what the struct is (rabe-console/src/mod.rs:82-105): GP, SK, MSK, PK, CT, SAK, PAK, PAUK, SAUK
def write_envelope(label, data):
    """`ser_enc` (:1623-1636): head, lower-case hex of the raw-DEFLATE stream of the serialised struct, tail"""
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    return "-----BEGIN %s-----\n%s\n-----END %s-----" % (label, (c.compress(bytes(data)) + c.flush()).hex(), label)


def read_envelope(text):
    """`ser_dec_bin` (:1663-1683): the SECOND line of the file (`read_raw`, src/utils/file/mod.rs:70-76), hex -> inflate.  Returns (label, bytes)."""
    lines = text.splitlines()
    if len(lines) < 2 or not lines[0].startswith("-----BEGIN ") or not lines[0].endswith("-----"):
        raise ValueError("not a rabe-console file")
    return lines[0][len("-----BEGIN "):-5], zlib.decompress(bytes.fromhex(lines[1].strip()), -15)
