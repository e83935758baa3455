//! Decrypt a key / ciphertext pair given in rabe's serde-JSON form with REAL rabe and print the plaintext (hex).
//! The files are what `serde_json::to_string(&sk)` / `(&ct)` produce -- e.g. the "cp_sk" / "cp_ct" members of
//! ref_schemes.json, or the output of tools that re-encode this repository's canonical records into rabe's layout
//! once tests/test_ref_pin.py has established it.
//!
//! usage: decrypt_file <ac17cp|ac17kp|bsw|lsw|aw11> <sk.json> <ct.json> [gk.json (aw11 only)]
extern crate rabe;
extern crate serde_json;

use rabe::schemes::{ac17, aw11, bsw, lsw};
use std::fs;

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
}

fn main() {
    let a: Vec<String> = std::env::args().collect();
    if a.len() < 4 {
        eprintln!("usage: decrypt_file <ac17cp|ac17kp|bsw|lsw|aw11> <sk.json> <ct.json> [gk.json]");
        std::process::exit(2);
    }
    let sk = fs::read_to_string(&a[2]).expect("sk file");
    let ct = fs::read_to_string(&a[3]).expect("ct file");
    let res = match a[1].as_str() {
        "ac17cp" => ac17::cp_decrypt(&serde_json::from_str(&sk).unwrap(), &serde_json::from_str(&ct).unwrap()),
        "ac17kp" => ac17::kp_decrypt(&serde_json::from_str(&sk).unwrap(), &serde_json::from_str(&ct).unwrap()),
        "bsw" => bsw::decrypt(&serde_json::from_str(&sk).unwrap(), &serde_json::from_str(&ct).unwrap()),
        "lsw" => lsw::decrypt(&serde_json::from_str(&sk).unwrap(), &serde_json::from_str(&ct).unwrap()),
        "aw11" => {
            let gk = fs::read_to_string(a.get(4).expect("aw11 needs gk.json")).expect("gk file");
            aw11::decrypt(&serde_json::from_str(&gk).unwrap(), &serde_json::from_str(&sk).unwrap(), &serde_json::from_str(&ct).unwrap())
        }
        other => {
            eprintln!("unknown scheme {}", other);
            std::process::exit(2);
        }
    };
    match res {
        Ok(pt) => println!("{}", hex(&pt)),
        Err(e) => {
            eprintln!("decrypt failed: {}", e.to_string());
            std::process::exit(1);
        }
    }
}
