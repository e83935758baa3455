"""rabe_amd/wire_compat.py: rabe's serde structs <-> this engine's canonical records, through an element codec learnt from reference
vectors (tests/refpin.py).  Real vectors are absent here (no Rust), so the codec is learnt from reference files SYNTHESISED in a layout a
zcash-bn-style crate would plausibly write (Montgomery limbs, Jacobian points, u128 limbs in serde, c1 before c0: tests/test_ref_pin.py
LAYOUT_B): a transcript of every scheme in that form is imported, must be the canonical record the host layer itself accepts and
re-serialises byte for byte, and exported back to the identical JSON.  The GPU test decrypts the imported ciphertexts with the imported
keys on the engine."""
import json

import pytest

from oracle import bn254 as bn
from oracle import policy as pol
from oracle import schemes as sch
from oracle.tape import SeededRng
from rabe_amd import hostlib as hl
from rabe_amd import wire_compat as wc
from tests import wire_codec as wcodec
from tests import refpin as rp
from tests import test_ref_pin as trp

PLAINTEXT = b"dance like no one's watching, encrypt like everyone is!"


@pytest.fixture(scope="module", params=["montgomery-jacobian", "canonical-affine"])
def world(request):
    """(codec, {kind: rabe-JSON object}, points are unique) in a synthetic rabe-bn layout"""
    layout = trp.LAYOUT_B if request.param == "montgomery-jacobian" else trp.LAYOUT_A
    prim, e = trp.synth_primitives(**layout)
    prim = json.loads(json.dumps(prim))
    srcs, report = trp.check_primitives(prim)
    src = srcs["serde"]
    samples = {"fr": prim["fr_from_str"][1]["out"]["serde"], "g1": prim["g1_mul"][1]["out"]["serde"], "g2": prim["g2_mul"][1]["out"]["serde"],
               "gt": prim["gt_pow"][1]["out"]["serde"]}
    zeros = {"g1": prim["group_ops"]["g1_zero"]["serde"], "g2": prim["group_ops"]["g2_zero"]["serde"]}
    dec, enc = wcodec.codec_from_source(src, samples, zeros)
    S = lambda kind, v: e[kind](v)["serde"]
    Z1, Z2 = zeros["g1"], zeros["g2"]
    from tests.test_host_kats import aes256_gcm
    nonce = bytes(range(12))

    def seal(g):
        ct, tag = aes256_gcm(trp.kdf_key(g, report["into_vec_u8"]), nonce, PLAINTEXT)
        return list(nonce + ct + tag)
    rng = SeededRng(91)
    e_base = bn.pairing(bn.G1_GEN, bn.G2_GEN)
    objs = {}
    # ---- AC17
    pk, msk = sch.ac17_setup(rng)
    policy = '{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "B"}, {"name": "C"}]}]}'
    msg = bn.gt_pow(e_base, rng.fr_nonzero())
    ct = sch.ac17_cp_encrypt(pk, policy, pol.JSON, rng, msg)
    sk = sch.ac17_cp_keygen(msk, ["A", "B"], rng)
    skj = lambda s: {"k_0": [S("g2", x) for x in s["k_0"]], "k": [[n, [S("g1", p) for p in v]] for n, v in s["k"]], "k_p": [S("g1", x) for x in s["k_p"]]}
    objs["ac17_pk"] = {"g": S("g1", pk["g"]), "h_a": [S("g2", x) for x in pk["h_a"]], "e_gh_ka": [S("gt", x) for x in pk["e_gh_ka"]]}
    objs["ac17_msk"] = {"g": S("g1", msk["g"]), "h": S("g2", msk["h"]), "g_k": [S("g1", x) for x in msk["g_k"]], "a": [S("fr", x) for x in msk["a"]],
                        "b": [S("fr", x) for x in msk["b"]]}
    objs["ac17_cp_sk"] = {"attr": ["A", "B"], "sk": skj(sk["sk"])}
    objs["ac17_cp_ct"] = {"policy": [policy, "JsonPolicy"], "ct": {"c_0": [S("g2", x) for x in ct["ct"]["c_0"]],
                                                                     "c": [[n, [S("g1", p) for p in v]] for n, v in ct["ct"]["c"]],
                                                                     "c_p": S("gt", ct["ct"]["c_p"]), "ct": seal(msg)}}
    # ---- BSW
    pk, msk = sch.bsw_setup(rng)
    policy = '{"name": "and", "children": [{"name": "A"}, {"name": "B"}, {"name": "or", "children": [{"name": "C"}, {"name": "D"}]}]}'
    msg = bn.gt_pow(e_base, rng.fr_nonzero())
    ct = sch.bsw_encrypt(pk, policy, pol.JSON, rng, msg)
    sk = sch.bsw_keygen(pk, msk, ["A", "B", "D"], rng)
    at = lambda x: {"string": x["string"], "g1": S("g1", x["g1"]), "g2": S("g2", x["g2"])}
    objs["bsw_pk"] = {"g1": S("g1", pk["g1"]), "g2": S("g2", pk["g2"]), "h": S("g1", pk["h"]), "f": S("g2", pk["f"]), "e_gg_alpha": S("gt", pk["e_gg_alpha"])}
    objs["bsw_sk"] = {"d": S("g2", sk["d"]), "d_j": [at(x) for x in sk["d_j"]]}
    objs["bsw_ct"] = {"policy": [policy, "JsonPolicy"], "c": S("g1", ct["c"]), "c_p": S("gt", ct["c_p"]), "c_y": [at(x) for x in ct["c_y"]], "data": seal(msg)}
    # ---- LSW (rabe's key rows hold an element in every slot: the identity where the oracle says None)
    pk, msk = sch.lsw_setup(rng)
    policy = '{"name": "or", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}'
    sk = sch.lsw_keygen(pk, msk, policy, pol.JSON, rng)
    msg = bn.gt_pow(e_base, rng.fr_nonzero())
    ct = sch.lsw_encrypt(pk, ["B", "C"], rng, msg)
    o1 = lambda v: Z1 if v is None else S("g1", v)
    o2 = lambda v: Z2 if v is None else S("g2", v)
    objs["lsw_sk"] = {"policy": [policy, "JsonPolicy"], "dj": [[t[0], o1(t[1]), o2(t[2]), o1(t[3]), o1(t[4]), o1(t[5])] for t in sk["dj"]]}
    objs["lsw_ct"] = {"e1": S("gt", ct["e1"]), "e2": S("g2", ct["e2"]), "ej": [[t[0], S("g1", t[1]), S("g1", t[2]), S("g1", t[3])] for t in ct["ej"]],
                      "ct": seal(msg)}
    # ---- AW11
    gk = sch.aw11_setup(rng)
    pk1, msk1 = sch.aw11_authgen(gk, ["A", "B"], rng)
    pk2, msk2 = sch.aw11_authgen(gk, ["C", "D"], rng)
    policy = '{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "C"}, {"name": "B"}]}]}'
    sk = sch.aw11_keygen(gk, msk1, "bob", ["A"])
    sk["attr"] += sch.aw11_keygen(gk, msk2, "bob", ["C"])["attr"]
    msg = bn.gt_pow(e_base, rng.fr_nonzero())
    ct = sch.aw11_encrypt(gk, [pk1, pk2], policy, pol.JSON, rng, msg)
    objs["aw11_gk"] = {"g1": S("g1", gk["g1"]), "g2": S("g2", gk["g2"])}
    objs["aw11_pk"] = {"attr": [[n, S("gt", g), S("g2", y)] for n, g, y in pk1["attr"]]}
    objs["aw11_msk"] = {"attr": [[n, S("fr", a), S("fr", y)] for n, a, y in msk1["attr"]]}
    objs["aw11_sk"] = {"gid": "bob", "attr": [[n, S("g1", p)] for n, p in sk["attr"]]}
    objs["aw11_ct"] = {"policy": [policy, "JsonPolicy"], "c_0": S("gt", ct["c_0"]),
                       "c": [[t[0], S("gt", t[1]), S("g2", t[2]), S("g2", t[3])] for t in ct["c"]], "ct": seal(msg)}
    bsamples = {"fr": prim["fr_from_str"][1]["out"]["borsh"], "g1": prim["g1_mul"][1]["out"]["borsh"], "g2": prim["g2_mul"][1]["out"]["borsh"],
                "gt": prim["gt_pow"][1]["out"]["borsh"]}
    bzeros = {"g1": prim["group_ops"]["g1_zero"]["borsh"], "g2": prim["group_ops"]["g2_zero"]["borsh"]}
    borsh = wcodec.borsh_codec_from_source(srcs["borsh"], bsamples, bzeros)
    return (dec, enc), json.loads(json.dumps(objs)), layout["g_shape"] == "affine", borsh


def test_import_is_the_host_layers_canonical_record_and_export_restores_the_json(world):
    (dec, enc), objs, unique, _ = world
    for kind, obj in objs.items():
        blob = wc.to_canonical(kind, obj, dec)
        o = hl.Obj.deserialize(kind, blob)                      # the C++ reader accepts it (structure, ranges) ...
        assert o.serialize() == blob, kind                      # ... and writes the same bytes
        back = wc.from_canonical(kind, blob, enc)
        assert wc.to_canonical(kind, back, dec) == blob, kind   # the way back names the same elements ...
        if unique:
            assert back == obj, kind                            # ... and is the identical serde value where a point has one form
        else:
            assert back.keys() == obj.keys()                    # (Jacobian triples: export writes z = 1)


def test_malformed_serde_is_rejected(world):
    (dec, enc), objs, _, _ = world
    bad = json.loads(json.dumps(objs["ac17_cp_ct"]))
    bad["ct"]["c_0"] = bad["ct"]["c_0"][:2]                     # ASSUMPTION_SIZE + 1 elements expected
    with pytest.raises(ValueError):
        wc.to_canonical("ac17_cp_ct", bad, dec)
    bad = json.loads(json.dumps(objs["bsw_ct"]))
    bad["policy"][1] = "YamlPolicy"
    with pytest.raises(ValueError):
        wc.to_canonical("bsw_ct", bad, dec)
    with pytest.raises(ValueError):
        wc.from_canonical("aw11_gk", wc.to_canonical("aw11_gk", objs["aw11_gk"], dec) + b"\0", enc)


def test_borsh_form_and_the_console_envelope_round_trip(world):
    """rabe-console's default on-disk form: borsh of the struct, raw deflate, hex between BEGIN / END lines"""
    (dec, enc), objs, unique, codec = world
    labels = {"ac17_cp_ct": "CT", "bsw_sk": "SK", "aw11_gk": "GP", "lsw_sk": "SK", "ac17_msk": "MSK", "bsw_pk": "PK"}
    for kind, obj in objs.items():
        blob = wc.to_canonical(kind, obj, dec)
        b = wc.from_canonical_borsh(kind, blob, codec)
        assert wc.to_canonical_borsh(kind, b, codec) == blob, kind            # borsh -> canonical is the inverse
        if kind in labels:
            text = wc.write_envelope(labels[kind], b)
            assert text.count("\n") == 2 and text.splitlines()[1] == text.splitlines()[1].lower()
            label, back = wc.read_envelope(text)
            assert (label, back) == (labels[kind], b)
    with pytest.raises(ValueError):
        wc.to_canonical_borsh("aw11_gk", wc.from_canonical_borsh("aw11_gk", wc.to_canonical("aw11_gk", objs["aw11_gk"], dec), codec) + b"\0", codec)
    with pytest.raises(ValueError):
        wc.read_envelope("no envelope here")


@pytest.mark.gpu
def test_imported_rabe_form_ciphertexts_decrypt_on_the_engine(world):
    from rabe_amd.schemes import ac17, aw11, bsw, lsw
    (dec, enc), objs, _, _ = world
    host = hl.Host(0)
    imp = lambda kind: hl.Obj.deserialize(kind, wc.to_canonical(kind, objs[kind], dec), host)          # with the GPU membership checks
    assert ac17.cp_decrypt(host, imp("ac17_cp_sk"), imp("ac17_cp_ct")) == PLAINTEXT
    assert bsw.decrypt(host, imp("bsw_sk"), imp("bsw_ct")) == PLAINTEXT
    assert lsw.decrypt(host, imp("lsw_sk"), imp("lsw_ct")) == PLAINTEXT
    assert aw11.decrypt(host, imp("aw11_gk"), imp("aw11_sk"), imp("aw11_ct")) == PLAINTEXT
    host.close()
