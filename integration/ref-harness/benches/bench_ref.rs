//! criterion timings of REAL rabe at the BASELINE.json configurations (SURVEY.md 8d plan (1)), single thread:
//!   config 1/2  ac17::cp_encrypt + cp_decrypt, 5 and 50 attributes (right-nested binary ANDs: msp.rs:132-134 needs binary gates)
//!   config 3    bsw::encrypt + decrypt, flat 100-leaf AND
//!   config 4    lsw::keygen + decrypt, flat 200-leaf AND
//!   config 5    aw11::encrypt + decrypt, 10 authorities x 20 attributes, binary ANDs over all 200
//! One "op" = the pair the metric counts.  ops/s = 1 / (mean time of the pair).  Report the CPU model and that this is ONE core.
extern crate criterion;
extern crate rabe;

use criterion::{criterion_group, criterion_main, Criterion};
use rabe::schemes::{ac17, aw11, bsw, lsw};
use rabe::utils::policy::pest::PolicyLanguage;

fn nested_and(names: &[String]) -> String {
    if names.len() == 1 {
        return format!(r#"{{"name": "{}"}}"#, names[0]);
    }
    format!(r#"{{"name": "and", "children": [{{"name": "{}"}}, {}]}}"#, names[0], nested_and(&names[1..]))
}
fn flat_and(names: &[String]) -> String {
    let kids: Vec<String> = names.iter().map(|n| format!(r#"{{"name": "{}"}}"#, n)).collect();
    format!(r#"{{"name": "and", "children": [{}]}}"#, kids.join(", "))
}
fn names(prefix: &str, n: usize) -> Vec<String> {
    (1..=n).map(|i| format!("{}{}", prefix, i)).collect()
}

fn bench(c: &mut Criterion) {
    let pt = String::from("dance like no one's watching, encrypt like everyone is!").into_bytes();
    let mut g = c.benchmark_group("rabe_ref");
    g.sample_size(10);
    for &n in [5usize, 50].iter() {
        let attrs = names("a", n);
        let refs: Vec<&str> = attrs.iter().map(|s| s.as_str()).collect();
        let policy = nested_and(&attrs);
        let (pk, msk) = ac17::setup();
        let sk = ac17::cp_keygen(&msk, &refs).unwrap();
        g.bench_function(format!("ac17_cp_encrypt+decrypt_{}", n), |b| {
            b.iter(|| {
                let ct = ac17::cp_encrypt(&pk, &policy, &pt, PolicyLanguage::JsonPolicy).unwrap();
                ac17::cp_decrypt(&sk, &ct).unwrap()
            })
        });
    }
    {
        let attrs = names("a", 100);
        let refs: Vec<&str> = attrs.iter().map(|s| s.as_str()).collect();
        let policy = flat_and(&attrs);
        let (pk, msk) = bsw::setup();
        let sk = bsw::keygen(&pk, &msk, &refs).unwrap();
        g.bench_function("bsw_encrypt+decrypt_100", |b| {
            b.iter(|| {
                let ct = bsw::encrypt(&pk, &policy, PolicyLanguage::JsonPolicy, &pt).unwrap();
                bsw::decrypt(&sk, &ct).unwrap()
            })
        });
    }
    {
        let attrs = names("a", 200);
        let refs: Vec<&str> = attrs.iter().map(|s| s.as_str()).collect();
        let policy = flat_and(&attrs);
        let (pk, msk) = lsw::setup();
        let ct = lsw::encrypt(&pk, &refs, &pt).unwrap();
        g.bench_function("lsw_keygen+decrypt_200", |b| {
            b.iter(|| {
                let sk = lsw::keygen(&pk, &msk, &policy, PolicyLanguage::JsonPolicy).unwrap();
                lsw::decrypt(&sk, &ct).unwrap()
            })
        });
    }
    {
        let gk = aw11::setup();
        let mut pks = Vec::new();
        let mut msks = Vec::new();
        let mut all: Vec<String> = Vec::new();
        for a in 0..10 {
            let attrs = names(&format!("AUTH{}A", a), 20);        // upper case: aw11/mod.rs:137,303-316
            let refs: Vec<&str> = attrs.iter().map(|s| s.as_str()).collect();
            let (pk, msk) = aw11::authgen(&gk, &refs).unwrap();
            pks.push(pk);
            msks.push(msk);
            all.extend(attrs);
        }
        let first: Vec<&str> = all[..20].iter().map(|s| s.as_str()).collect();
        let mut sk = aw11::keygen(&gk, &msks[0], "bob", &first).unwrap();
        for (i, name) in all.iter().enumerate().skip(20) {
            aw11::add_to_attribute(&gk, &msks[i / 20], name, &mut sk).unwrap();
        }
        let policy = nested_and(&all);
        let pk_refs: Vec<&aw11::Aw11PublicKey> = pks.iter().collect();
        g.bench_function("aw11_encrypt+decrypt_10x20", |b| {
            b.iter(|| {
                let ct = aw11::encrypt(&gk, pk_refs.as_slice(), &policy, PolicyLanguage::JsonPolicy, &pt).unwrap();
                aw11::decrypt(&gk, &sk, &ct).unwrap()
            })
        });
    }
    g.finish();
}

criterion_group!(benches, bench);
criterion_main!(benches);
