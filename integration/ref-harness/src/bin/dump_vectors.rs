//! Known-answer vectors from REAL rabe / rabe-bn, in the shape tests/test_ref_pin.py loads.
//!
//! Only API that rabe itself calls is used (so this compiles against whatever rabe-bn 0.4.23 exports):
//! `Fr::from_str`, `Fr::from_slice`, `G1::one()`, `G2::one()`, `* Fr`, `+`, `pairing`, `Gt::pow`, `Into<Vec<u8>> for Gt`,
//! and the serde / borsh derives.  Every element is written three ways -- serde_json value, borsh bytes (hex), `{:?}` --
//! and the loader works out the layout (limb width / order, Montgomery or plain, affine or Jacobian) by comparing with
//! values it knows (k * generator for small k), then checks everything else through that layout.
//!
//! usage: dump_vectors <output dir>     (writes ref_primitives.json and ref_schemes.json there)
extern crate borsh;
extern crate rabe;
extern crate rabe_bn;
extern crate serde;
extern crate serde_json;
extern crate sha3;

use rabe::schemes::{ac17, aw11, bsw, lsw};
use rabe::utils::policy::pest::PolicyLanguage;
use rabe_bn::{pairing, Fr, Group, Gt, G1, G2};
use serde::Serialize;
use serde_json::{json, Value};
use sha3::{Digest, Sha3_256};
use std::fs::File;
use std::io::Write;

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
}

/// one element, three encodings
fn el<T: Serialize + borsh::BorshSerialize + std::fmt::Debug>(x: &T) -> Value {
    json!({
        "serde": serde_json::to_value(x).unwrap(),
        "borsh": hex(&borsh::to_vec(x).unwrap()),
        "debug": format!("{:?}", x),
    })
}

fn fr(dec: &str) -> Fr {
    Fr::from_str(dec).expect("decimal scalar")
}

// scalars as decimal strings (rabe's own `usize_to_fr` goes through `Fr::from_str`, src/utils/tools/mod.rs:11-13)
const SCALARS: [&str; 8] = [
    "1",
    "2",
    "3",
    "65537",
    "4294967296",                                                                    // 2^32: limb order
    "340282366920938463463374607431768211456",                                       // 2^128
    "21888242871839275222246405745257275088548364400416034343698204186575808495616", // r - 1
    "12345678901234567890123456789012345678901234567890123456789012345678",
];
// labels hashed the way rabe does it (src/utils/hash/mod.rs:10-31); the first four are tests/golden/bn254_primitives.json's
const LABELS: [&str; 8] = ["A00", "B10", "01", "", "a1", "a10", "AUTH1:A1", "dance like no one's watching"];

fn main() {
    let out_dir = std::env::args().nth(1).unwrap_or_else(|| ".".to_string());

    // ---------------------------------------------------------------- primitives
    let mut fr_from_str = Vec::new();
    let mut g1_mul = Vec::new();
    let mut g2_mul = Vec::new();
    let mut gt_pow = Vec::new();
    let e11 = pairing(G1::one(), G2::one());
    for s in SCALARS.iter() {
        let k = fr(s);
        fr_from_str.push(json!({"k": s, "out": el(&k)}));
        g1_mul.push(json!({"k": s, "out": el(&(G1::one() * k))}));
        g2_mul.push(json!({"k": s, "out": el(&(G2::one() * k))}));
        gt_pow.push(json!({"k": s, "out": el(&e11.pow(k))}));
    }
    let mut fr_from_digest = Vec::new();
    for l in LABELS.iter() {
        let mut h = Sha3_256::new();
        h.update(l.as_bytes());
        let d = h.finalize();
        let f = Fr::from_slice(&d).expect("Fr::from_slice on a SHA3-256 digest");
        // the same value as a group element, so that the reduction is pinned even if Fr's own encoding cannot be read
        fr_from_digest.push(json!({"label": l, "digest_be": hex(&d), "out": el(&f), "g1": el(&(G1::one() * f))}));
    }
    // Fr arithmetic that rabe uses: inverse (secretsharing/mod.rs:66), pow with an Fr exponent (:218)
    let a = fr(SCALARS[7]);
    let fr_ops = json!({
        "a": SCALARS[7],
        "inverse": el(&a.inverse().unwrap()),
        "pow_3": el(&a.pow(fr("3"))),
        "neg": el(&(Fr::zero() - a)),
        "a_times_65537": el(&(a * fr("65537"))),
    });
    let mut pairings = Vec::new();
    for (x, y) in [("1", "1"), ("2", "3"), ("65537", SCALARS[7])].iter() {
        let p = G1::one() * fr(x);
        let q = G2::one() * fr(y);
        let e = pairing(p, q);
        let bytes: Vec<u8> = e.into();
        pairings.push(json!({"a": x, "b": y, "p": el(&p), "q": el(&q), "out": el(&e), "into_vec_u8": hex(&bytes)}));
    }
    let group_ops = json!({
        "g1_zero": el(&G1::zero()),
        "g2_zero": el(&G2::zero()),
        "gt_one": el(&Gt::one()),
        "g1_2_plus_3": el(&(G1::one() * fr("2") + G1::one() * fr("3"))),
        "g1_neg_2": el(&(G1::one() * (Fr::zero() - fr("2")))),
        "gt_inverse_e11": el(&e11.inverse()),
        "gt_mul": el(&(e11 * e11.pow(fr("3")))),
    });
    let prim = json!({
        "source": "rabe 0.4.2 / rabe-bn 0.4.23 (integration/ref-harness/src/bin/dump_vectors.rs)",
        "fr_from_str": fr_from_str, "fr_from_digest": fr_from_digest, "fr_ops": fr_ops,
        "g1_mul": g1_mul, "g2_mul": g2_mul, "gt_pow": gt_pow, "pairing": pairings, "group_ops": group_ops,
    });
    let mut f = File::create(format!("{}/ref_primitives.json", out_dir)).unwrap();
    f.write_all(serde_json::to_string_pretty(&prim).unwrap().as_bytes()).unwrap();

    // ---------------------------------------------------------------- whole-scheme transcripts (randomness is rabe's own
    // thread_rng, so these pin DECRYPTION: key + ciphertext -> plaintext, and the algebraic relations between the
    // elements of a key / ciphertext; encryption parity under explicit randomness follows from the primitives above)
    let plaintext = String::from("dance like no one's watching, encrypt like everyone is!").into_bytes();
    let mut schemes = serde_json::Map::new();
    schemes.insert("plaintext_hex".to_string(), json!(hex(&plaintext)));
    {
        let (pk, msk) = ac17::setup();
        let policy = String::from(r#"{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "B"}, {"name": "C"}]}]}"#);
        let ct = ac17::cp_encrypt(&pk, &policy, &plaintext, PolicyLanguage::JsonPolicy).unwrap();
        let sk = ac17::cp_keygen(&msk, &["A", "B"]).unwrap();
        assert_eq!(ac17::cp_decrypt(&sk, &ct).unwrap(), plaintext);
        let kp_sk = ac17::kp_keygen(&msk, &policy, PolicyLanguage::JsonPolicy).unwrap();
        let kp_ct = ac17::kp_encrypt(&pk, &["A", "B"], &plaintext).unwrap();
        assert_eq!(ac17::kp_decrypt(&kp_sk, &kp_ct).unwrap(), plaintext);
        schemes.insert("ac17".to_string(), json!({
            "policy": policy, "attributes": ["A", "B"],
            "pk": serde_json::to_value(&pk).unwrap(), "msk": serde_json::to_value(&msk).unwrap(),
            "cp_sk": serde_json::to_value(&sk).unwrap(), "cp_ct": serde_json::to_value(&ct).unwrap(),
            "kp_sk": serde_json::to_value(&kp_sk).unwrap(), "kp_ct": serde_json::to_value(&kp_ct).unwrap(),
            "cp_ct_borsh": hex(&borsh::to_vec(&ct).unwrap()), "cp_sk_borsh": hex(&borsh::to_vec(&sk).unwrap()),
            "pk_borsh": hex(&borsh::to_vec(&pk).unwrap()),
        }));
    }
    {
        let (pk, msk) = bsw::setup();
        let policy = String::from(r#"{"name": "and", "children": [{"name": "A"}, {"name": "B"}, {"name": "or", "children": [{"name": "C"}, {"name": "D"}]}]}"#);
        let ct = bsw::encrypt(&pk, &policy, PolicyLanguage::JsonPolicy, &plaintext).unwrap();
        let sk = bsw::keygen(&pk, &msk, &["A", "B", "D"]).unwrap();
        assert_eq!(bsw::decrypt(&sk, &ct).unwrap(), plaintext);
        schemes.insert("bsw".to_string(), json!({
            "policy": policy, "attributes": ["A", "B", "D"],
            "pk": serde_json::to_value(&pk).unwrap(), "msk": serde_json::to_value(&msk).unwrap(),
            "sk": serde_json::to_value(&sk).unwrap(), "ct": serde_json::to_value(&ct).unwrap(),
            "ct_borsh": hex(&borsh::to_vec(&ct).unwrap()), "sk_borsh": hex(&borsh::to_vec(&sk).unwrap()),
        }));
    }
    {
        let (pk, msk) = lsw::setup();
        let policy = String::from(r#"{"name": "or", "children": [{"name": "A"}, {"name": "and", "children": [{"name": "B"}, {"name": "C"}]}]}"#);
        let sk = lsw::keygen(&pk, &msk, &policy, PolicyLanguage::JsonPolicy).unwrap();
        let ct = lsw::encrypt(&pk, &["B", "C"], &plaintext).unwrap();
        assert_eq!(lsw::decrypt(&sk, &ct).unwrap(), plaintext);
        schemes.insert("lsw".to_string(), json!({
            "policy": policy, "attributes": ["B", "C"],
            "pk": serde_json::to_value(&pk).unwrap(), "msk": serde_json::to_value(&msk).unwrap(),
            "sk": serde_json::to_value(&sk).unwrap(), "ct": serde_json::to_value(&ct).unwrap(),
            "ct_borsh": hex(&borsh::to_vec(&ct).unwrap()), "sk_borsh": hex(&borsh::to_vec(&sk).unwrap()),
        }));
    }
    {
        let gk = aw11::setup();
        let (pk1, msk1) = aw11::authgen(&gk, &["A", "B"]).unwrap();
        let (pk2, msk2) = aw11::authgen(&gk, &["C", "D"]).unwrap();
        let policy = String::from(r#"{"name": "and", "children": [{"name": "A"}, {"name": "or", "children": [{"name": "C"}, {"name": "B"}]}]}"#);
        let mut sk = aw11::keygen(&gk, &msk1, "bob", &["A"]).unwrap();
        aw11::add_to_attribute(&gk, &msk2, "C", &mut sk).unwrap();
        let ct = aw11::encrypt(&gk, &[&pk1, &pk2], &policy, PolicyLanguage::JsonPolicy, &plaintext).unwrap();
        assert_eq!(aw11::decrypt(&gk, &sk, &ct).unwrap(), plaintext);
        schemes.insert("aw11".to_string(), json!({
            "policy": policy, "gid": "bob",
            "gk": serde_json::to_value(&gk).unwrap(),
            "pks": [serde_json::to_value(&pk1).unwrap(), serde_json::to_value(&pk2).unwrap()],
            "msks": [serde_json::to_value(&msk1).unwrap(), serde_json::to_value(&msk2).unwrap()],
            "sk": serde_json::to_value(&sk).unwrap(), "ct": serde_json::to_value(&ct).unwrap(),
            "ct_borsh": hex(&borsh::to_vec(&ct).unwrap()), "sk_borsh": hex(&borsh::to_vec(&sk).unwrap()),
        }));
    }
    let mut f = File::create(format!("{}/ref_schemes.json", out_dir)).unwrap();
    f.write_all(serde_json::to_string_pretty(&Value::Object(schemes)).unwrap().as_bytes()).unwrap();
    eprintln!("wrote {0}/ref_primitives.json and {0}/ref_schemes.json", out_dir);
}
