"""Element codecs for rabe_amd/wire_compat.py built from a LEARNT reference layout (tests/refpin.py `Source` after `learn`).

Test infrastructure: the learner needs the oracle's element arithmetic, so it lives under tests/ and the product module takes the codec
as an argument."""
from oracle import bn254 as bn
from tests import refpin as rp
from rabe_amd.wire_compat import FR, G1, G2, GT


def _shape_of(v):
    """nesting of a serde element value, for re-filling: ints become None"""
    if isinstance(v, list):
        return [_shape_of(x) for x in v]
    if isinstance(v, dict):
        return {k: _shape_of(x) for k, x in v.items()}
    return None


def _fill(shape, it):
    if isinstance(shape, list):
        return [_fill(x, it) for x in shape]
    if isinstance(shape, dict):
        return {k: _fill(x, it) for k, x in shape.items()}
    return next(it)


def codec_from_source(src, samples, zeros=None):
    """(dec, enc) over a tests/refpin.py Source whose layouts have been learnt.  `samples`: {kind: one serde element value of that kind}
    (any element of the dumped vectors) -- its nesting and integer width are the template `enc` fills.  `zeros`: {kind: the serde value
    of the group's identity} (ref_primitives.json: group_ops.g1_zero / g2_zero), emitted verbatim for an all-zero canonical element."""
    to_le = {FR: lambda v: int(v).to_bytes(32, "little"), G1: bn.g1_to_le, G2: bn.g2_to_le, GT: bn.gt_to_le}
    from_le = {FR: lambda b: int.from_bytes(b, "little"), G1: bn.g1_from_le, G2: bn.g2_from_le, GT: bn.gt_from_le}
    templates = {}
    for kind, sample in samples.items():
        ints = rp.flatten_ints(sample)
        lay = src.layout[kind]
        templates[kind] = (_shape_of(sample), (32 * lay.n_fe) // len(ints))          # bytes per integer of the serde form

    def dec(kind, value):
        return to_le[kind](src.decode(kind, {"serde": value, "borsh": ""}))

    def enc(kind, canon):
        if kind in (G1, G2) and canon == bytes(len(canon)):
            if not zeros or kind not in zeros:
                raise ValueError("identity element of %s: pass its serde form in `zeros`" % kind)
            return zeros[kind]
        lay = src.layout[kind]
        el = rp.encode_element(kind, from_le[kind](canon), lay.fe, lay.shape if kind in (G1, G2) else "affine",
                               order=(lay.order[1] if lay.order else None))
        raw = bytes.fromhex(el["borsh"])
        shape, width = templates[kind]
        ints = [int.from_bytes(raw[i:i + width], "little") for i in range(0, len(raw), width)]
        return _fill(shape, iter(ints))
    return dec, enc


def borsh_codec_from_source(src, samples, zeros=None):
    """(dec, enc, size) over a tests/refpin.py Source("borsh") whose layouts have been learnt.  `samples`: {kind: hex of one borsh element}
    (its length is the kind's size on the wire, a length prefix included if the crate writes one); `zeros`: {kind: hex of the identity}."""
    to_le = {FR: lambda v: int(v).to_bytes(32, "little"), G1: bn.g1_to_le, G2: bn.g2_to_le, GT: bn.gt_to_le}
    from_le = {FR: lambda b: int.from_bytes(b, "little"), G1: bn.g1_from_le, G2: bn.g2_from_le, GT: bn.gt_from_le}
    size = {k: len(bytes.fromhex(v)) for k, v in samples.items()}
    prefixed = {k: size[k] == 32 * src.layout[k].n_fe + 4 for k in size}

    def dec(kind, raw):
        return to_le[kind](src.decode(kind, {"borsh": bytes(raw).hex(), "serde": None}))

    def enc(kind, canon):
        if kind in (G1, G2) and canon == bytes(len(canon)):
            if not zeros or kind not in zeros:
                raise ValueError("identity element of %s: pass its borsh form in `zeros`" % kind)
            return bytes.fromhex(zeros[kind])
        lay = src.layout[kind]
        el = rp.encode_element(kind, from_le[kind](canon), lay.fe, lay.shape if kind in (G1, G2) else "affine",
                               order=(lay.order[1] if lay.order else None), prefix=prefixed[kind])
        return bytes.fromhex(el["borsh"])
    return dec, enc, size
