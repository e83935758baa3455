"""Loader for the vectors integration/ref-harness/ dumps from REAL rabe / rabe-bn (tests/golden/ref_primitives.json,
ref_schemes.json), used by tests/test_ref_pin.py.

rabe-bn 0.4.23 is not in /root/reference, so the byte layout of its serde / borsh forms is unknown here.  The loader does
not assume one: every element arrives as {serde JSON value, borsh hex, Debug string}; `Layout.learn` searches a small,
explicit space of layouts per element type until an ANCHOR whose value is certain (k * generator for small k, Fr of a
small integer) decodes to that value, and every other vector is then read through the layout found.  The search space:

  * field element = 32 bytes read as one integer: little- / big-endian, or little-endian limbs of 8 / 16 bytes stored
    most significant limb first; plain or Montgomery (x R mod m with R = 2^256, as zcash `bn`'s Fq / Fr hold it);
  * G1 / G2: affine (x, y) or three coordinates (Jacobian X/Z^2, Y/Z^3 or homogeneous X/Z, Y/Z); Fq2 as (c0, c1) or (c1, c0);
  * Gt: 12 Fq, nested (Fq12 (Fq6 (Fq2))) in declaration order or reversed at any of the three levels;
  * containers: an optional 4-byte borsh length prefix; serde values are flattened to integers (bytes, or wider limbs).

What a mismatch means is reported by convention so that the one function holding that
assumption can be flipped.  This file is test infrastructure."""
import itertools
import json

from oracle import bn254 as bn

R256 = 1 << 256


# ------------------------------------------------------------------------------------------------ raw material
def flatten_ints(v):
    """serde value -> flat list of integers (strings of hex digits become bytes; decimal strings become one integer)"""
    out = []
    if isinstance(v, bool):
        out.append(int(v))
    elif isinstance(v, int):
        out.append(v)
    elif isinstance(v, str):
        s = v[2:] if v.startswith("0x") else v
        if s and all(c in "0123456789abcdefABCDEF" for c in s) and len(s) % 2 == 0 and len(s) >= 64:
            out.extend(bytes.fromhex(s))
        elif s.isdigit():
            out.append(int(s))
        else:
            raise ValueError("unreadable string in a serde element: %r" % v[:40])
    elif isinstance(v, list):
        for x in v:
            out.extend(flatten_ints(x))
    elif isinstance(v, dict):
        for k in v:               # serde_json keeps declaration order (preserve_order off sorts keys: both are tried via `orders`)
            out.extend(flatten_ints(v[k]))
    elif v is None:
        pass
    else:
        raise ValueError("unreadable serde element %r" % (v,))
    return out


def ints_to_bytes(ints, n_fe):
    """a flat integer list holding n_fe field elements -> 32 * n_fe bytes, little-endian inside every limb"""
    if not ints:
        return None
    if all(0 <= x < 256 for x in ints) and len(ints) in (32 * n_fe, 32 * n_fe + 4):
        return bytes(ints[-32 * n_fe:])
    if len(ints) % n_fe:
        return None
    per = len(ints) // n_fe
    if per not in (1, 2, 4, 8, 16):
        return None
    w = 32 // per
    if any(x < 0 or x >= 1 << (8 * w) for x in ints):
        return None
    return b"".join(x.to_bytes(w, "little") for x in ints)


BYTE_FORMS = ("le", "be", "limb8_msf", "limb16_msf")


def fe_from(chunk, form):
    if form == "le":
        return int.from_bytes(chunk, "little")
    if form == "be":
        return int.from_bytes(chunk, "big")
    w = 8 if form == "limb8_msf" else 16
    limbs = [chunk[i:i + w] for i in range(0, 32, w)]
    return int.from_bytes(b"".join(reversed(limbs)), "little")


def fe_to(x, form):
    if form == "le":
        return x.to_bytes(32, "little")
    if form == "be":
        return x.to_bytes(32, "big")
    w = 8 if form == "limb8_msf" else 16
    b = x.to_bytes(32, "little")
    return b"".join(reversed([b[i:i + w] for i in range(0, 32, w)]))


class FeLayout:
    def __init__(self, form, mont, mod):
        self.form, self.mont, self.mod = form, mont, mod

    def dec(self, chunk):
        x = fe_from(chunk, self.form)
        if self.mont:
            x = x * pow(R256, -1, self.mod) % self.mod
        return x

    def enc(self, x):
        return fe_to(x * R256 % self.mod if self.mont else x, self.form)

    def __repr__(self):
        return "%s%s" % (self.form, "+montgomery" if self.mont else "")


def fe_layouts(mod):
    return [FeLayout(f, m, mod) for f in BYTE_FORMS for m in (False, True)]


def gt_orders():
    """index permutations of the 12 Fq of an Fq12 relative to the engine's tower order c0.a0.c0, c0.a0.c1, c0.a1.c0, ..."""
    out = []
    for f12, f6, f2 in itertools.product((False, True), repeat=3):
        idx = []
        for h in ((1, 0) if f12 else (0, 1)):
            for a in ((2, 1, 0) if f6 else (0, 1, 2)):
                for c in ((1, 0) if f2 else (0, 1)):
                    idx.append(h * 6 + a * 2 + c)
        out.append((("fq12" if f12 else "") + ("fq6" if f6 else "") + ("fq2" if f2 else "") or "declared", idx))
    return out


# ------------------------------------------------------------------------------------------------ typed layouts
class ElementLayout:
    """how one element type (fr, g1, g2, gt) of one source (borsh / serde) is laid out"""

    def __init__(self, kind, n_fe, fe, shape, order):
        self.kind, self.n_fe, self.fe, self.shape, self.order = kind, n_fe, fe, shape, order

    def __repr__(self):
        return "%s: %d x Fq %r, %s, order %s" % (self.kind, self.n_fe, self.fe, self.shape, self.order[0] if self.order else "-")

    def coords(self, raw):
        return [self.fe.dec(raw[32 * i:32 * i + 32]) for i in range(self.n_fe)]

    def decode(self, raw):
        c = self.coords(raw)
        P = bn.P
        if self.kind == "fr":
            return c[0]
        if self.kind == "g1":
            if self.shape == "affine":
                return None if c[0] == 0 and c[1] == 0 else (c[0], c[1])
            x, y, z = c
            if z == 0:
                return None
            zi = pow(z, -1, P)
            if self.shape == "jacobian":
                return (x * zi * zi % P, y * zi * zi * zi % P)
            return (x * zi % P, y * zi % P)
        if self.kind == "g2":
            f2 = [(c[2 * i], c[2 * i + 1]) if self.order[1] == [0, 1] else (c[2 * i + 1], c[2 * i]) for i in range(self.n_fe // 2)]
            if self.shape == "affine":
                return None if f2[0] == (0, 0) and f2[1] == (0, 0) else (f2[0], f2[1])
            x, y, z = f2
            if z == (0, 0):
                return None
            zi = bn.fp2_inv(z)
            if self.shape == "jacobian":
                zi2 = bn.fp2_sqr(zi)
                return (bn.fp2_mul(x, zi2), bn.fp2_mul(y, bn.fp2_mul(zi2, zi)))
            return (bn.fp2_mul(x, zi), bn.fp2_mul(y, zi))
        # gt: order[1][i] = position in the engine's tower order of the i-th stored coefficient
        tower = [0] * 12
        for i, t in enumerate(self.order[1]):
            tower[t] = c[i]
        return bn.gt_from_le(b"".join(x.to_bytes(32, "little") for x in tower))


def candidates(kind):
    mod = bn.R if kind == "fr" else bn.P
    fes = fe_layouts(mod)
    if kind == "fr":
        return [ElementLayout(kind, 1, fe, "scalar", None) for fe in fes]
    if kind == "g1":
        return [ElementLayout(kind, n, fe, shape, None) for fe in fes for n, shape in ((2, "affine"), (3, "jacobian"), (3, "homogeneous"))]
    if kind == "g2":
        return [ElementLayout(kind, n, fe, shape, (name, o)) for fe in fes for n, shape in ((4, "affine"), (6, "jacobian"), (6, "homogeneous"))
                for name, o in (("c0,c1", [0, 1]), ("c1,c0", [1, 0]))]
    return [ElementLayout(kind, 12, fe, "fq12", o) for fe in fes for o in gt_orders()]


def same(kind, a, b):
    if kind == "gt":
        return bn.gt_to_le(a) == bn.gt_to_le(b)
    return a == b


class Source:
    """one encoding of the elements (borsh or serde) with the layout learnt per type"""

    def __init__(self, name):
        self.name = name
        self.layout = {}

    def raw(self, el, n_fe):
        if self.name == "borsh":
            b = bytes.fromhex(el["borsh"])
            if len(b) == 32 * n_fe + 4:
                b = b[4:]
            return b if len(b) == 32 * n_fe else None
        try:
            return ints_to_bytes(flatten_ints(el["serde"]), n_fe)
        except ValueError:
            return None

    def learn(self, kind, anchors):
        """anchors: [(element, expected value)] -- the first layout under which ALL anchors decode to their values"""
        for lay in candidates(kind):
            ok = True
            for el, want in anchors:
                raw = self.raw(el, lay.n_fe)
                if raw is None:
                    ok = False
                    break
                try:
                    got = lay.decode(raw)
                except (ValueError, ZeroDivisionError):
                    ok = False
                    break
                if got is None or not same(kind, got, want):
                    ok = False
                    break
            if ok:
                self.layout[kind] = lay
                return lay
        return None

    def decode(self, kind, el):
        lay = self.layout[kind]
        raw = self.raw(el, lay.n_fe)
        assert raw is not None, "%s element of kind %s has an unexpected size" % (self.name, kind)
        return lay.decode(raw)


# ------------------------------------------------------------------------------------------------ writing a reference file (self-test)
def encode_element(kind, value, fe, shape="affine", order=None, z=1, serde_limb=None, prefix=False):
    """the inverse of decode for ONE layout: used by the self-test to synthesise files in layouts a real crate might use"""
    P = bn.P
    if kind == "fr":
        fes = [value % bn.R]
    elif kind == "g1":
        if shape == "affine":
            fes = [0, 0] if value is None else list(value)
        else:
            x, y = value
            fes = [x * z * z % P, y * z * z * z % P, z] if shape == "jacobian" else [x * z % P, y * z % P, z]
    elif kind == "g2":
        if shape == "affine":
            pts = list(value)
        else:
            zz = (z, 0)
            x, y = value
            z2 = bn.fp2_sqr(zz)
            pts = [bn.fp2_mul(x, z2), bn.fp2_mul(y, bn.fp2_mul(z2, zz)), zz] if shape == "jacobian" else [bn.fp2_mul(x, zz), bn.fp2_mul(y, zz), zz]
        fes = []
        for c in pts:
            fes += [c[0], c[1]] if (order or [0, 1]) == [0, 1] else [c[1], c[0]]
    else:
        tower = [int.from_bytes(bn.gt_to_le(value)[32 * i:32 * i + 32], "little") for i in range(12)]
        fes = [tower[t] for t in (order or list(range(12)))]
    raw = b"".join(fe.enc(x) for x in fes)
    if serde_limb:
        serde = [[int.from_bytes(raw[32 * i + j:32 * i + j + serde_limb], "little") for j in range(0, 32, serde_limb)] for i in range(len(fes))]
    else:
        serde = list(raw)
    return {"serde": serde, "borsh": ((len(raw)).to_bytes(4, "little") if prefix else b"").hex() + raw.hex(), "debug": "synthetic"}


def load(path):
    with open(path) as f:
        return json.load(f)
