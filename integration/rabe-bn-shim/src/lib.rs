//! `rabe_bn` surface used by rabe, implemented over the host-value Level E functions of the HIP engine (`rhip_host_*`,
//! include/rabe_hip.h).  Audited against every `use rabe_bn::` site of rabe 0.4.2:
//!
//! | rabe call site                                                        | what it needs                                    |
//! |-----------------------------------------------------------------------|--------------------------------------------------|
//! | `src/schemes/{ac17,bsw,lsw,aw11,ghw11}/mod.rs` (`Group, Gt, G1, G2, Fr, pairing`) | `+ - neg`, `* Fr`, `G::zero/one`, `Gt::one`, `Gt::pow`, `Gt::inverse`, `Gt * Gt`, `Fr::inverse() -> Option`, `pairing`, `rng.gen::<Fr / G1 / G2 / Gt>()`, `Copy + Clone + PartialEq + Debug` (the key / ciphertext structs derive them, ac17/mod.rs:59) |
//! | `src/schemes/*` struct derives behind `serde` / `borsh` (ac17/mod.rs:60-61, Cargo.toml:19-20 forwards the features) | `Serialize + Deserialize`, `BorshSerialize + BorshDeserialize` for `Fr, G1, G2, Gt` |
//! | `src/utils/hash/mod.rs:16,27`                                         | `Fr::from_slice(&[u8]) -> Result<Fr, FieldError>` |
//! | `src/utils/secretsharing/mod.rs:25-28,66,133,218`                     | `Fr::one/zero`, `+ - *`, `inverse`, `rng.gen()`, `Fr::pow(Fr)` |
//! | `src/utils/tools/mod.rs:12`                                           | `Fr::from_str(&str) -> Option<Fr>`               |
//! | `src/utils/aes/mod.rs:10,29,47`                                       | `Into<Vec<u8>> for Gt`                           |
//! | `src/error.rs:10,60-69`                                               | `enum FieldError { InvalidSliceLength, InvalidU512Encoding, NotMember }` |
//! | `src/utils/policy/dnf.rs:2` (out-of-scope schemes)                    | `Group, Gt, G1, G2` as above                      |
//!
//! Every value is the engine's wire record: canonical little-endian, affine, infinity = zeros -- so `==` is byte equality
//! and `normalize` is the identity.  One launch per operator: this crate exists to run unmodified rabe against the engine
//! for whole-program parity, not for throughput (the batch path of INTEGRATION.md section 1 is the fast one).
//!
//! NOT COMPILED in this repository (no Rust toolchain in the build container).  Conventions that cannot be checked without
//! the real crate's source are isolated and marked `ASSUMPTION`: byte layouts of the serde / borsh
//! forms, `Into<Vec<u8>> for Gt`, how random group elements are sampled.
use std::ops::{Add, Mul, Neg, Sub};
use std::os::raw::c_char;
use std::sync::Once;

#[repr(C)] pub struct RhipCtx { _p: [u8; 0] }
#[repr(C)] #[derive(Copy, Clone, PartialEq, Eq, Debug)] pub struct Fr(pub [u32; 8]);
#[repr(C)] #[derive(Copy, Clone, PartialEq, Eq, Debug)] pub struct G1(pub [u32; 16]);
#[repr(C)] #[derive(Copy, Clone, PartialEq, Eq, Debug)] pub struct G2(pub [u32; 32]);
#[repr(C)] #[derive(Copy, Clone, PartialEq, Eq, Debug)] pub struct Gt(pub [u32; 96]);

#[link(name = "rabe_hip")]
extern "C" {
    fn rhip_ctx_create(device: i32, out: *mut *mut RhipCtx) -> i32;
    fn rhip_last_error(ctx: *mut RhipCtx) -> *const c_char;
    fn rhip_host_fr_op(ctx: *mut RhipCtx, op: i32, a: *const Fr, b: *const Fr, out: *mut Fr) -> i32;
    fn rhip_host_fr_pow(ctx: *mut RhipCtx, a: *const Fr, e: *const Fr, out: *mut Fr) -> i32;
    fn rhip_host_fr_from_be32_reduce(ctx: *mut RhipCtx, digest: *const u8, out: *mut Fr) -> i32;
    fn rhip_host_g1_add(ctx: *mut RhipCtx, a: *const G1, b: *const G1, out: *mut G1) -> i32;
    fn rhip_host_g1_neg(ctx: *mut RhipCtx, a: *const G1, out: *mut G1) -> i32;
    fn rhip_host_g1_mul(ctx: *mut RhipCtx, p: *const G1, k: *const Fr, out: *mut G1) -> i32;
    fn rhip_host_g1_on_curve(ctx: *mut RhipCtx, p: *const G1, ok: *mut i32) -> i32;
    fn rhip_host_g2_add(ctx: *mut RhipCtx, a: *const G2, b: *const G2, out: *mut G2) -> i32;
    fn rhip_host_g2_neg(ctx: *mut RhipCtx, a: *const G2, out: *mut G2) -> i32;
    fn rhip_host_g2_mul(ctx: *mut RhipCtx, p: *const G2, k: *const Fr, out: *mut G2) -> i32;
    fn rhip_host_g2_on_curve(ctx: *mut RhipCtx, p: *const G2, ok: *mut i32) -> i32;
    fn rhip_host_g2_in_subgroup(ctx: *mut RhipCtx, p: *const G2, ok: *mut i32) -> i32;
    fn rhip_host_gt_is_member(ctx: *mut RhipCtx, a: *const Gt, ok: *mut i32) -> i32;
    fn rhip_host_gt_mul(ctx: *mut RhipCtx, a: *const Gt, b: *const Gt, out: *mut Gt) -> i32;
    fn rhip_host_gt_inv(ctx: *mut RhipCtx, a: *const Gt, out: *mut Gt) -> i32;
    fn rhip_host_gt_pow(ctx: *mut RhipCtx, a: *const Gt, k: *const Fr, out: *mut Gt) -> i32;
    fn rhip_host_pairing(ctx: *mut RhipCtx, p: *const G1, q: *const G2, out: *mut Gt) -> i32;
}
const FR_ADD: i32 = 0; const FR_SUB: i32 = 1; const FR_MUL: i32 = 2; const FR_NEG: i32 = 3; const FR_INV: i32 = 4;

// One engine context per process (rabe's functions are free functions without a context argument).
static INIT: Once = Once::new();
static mut CTX: *mut RhipCtx = std::ptr::null_mut();
fn ctx() -> *mut RhipCtx {
    unsafe {
        INIT.call_once(|| {
            let mut c: *mut RhipCtx = std::ptr::null_mut();
            let dev = std::env::var("RABE_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            if rhip_ctx_create(dev, &mut c) != 0 { panic!("rabe-bn (HIP): no usable device: the engine has no CPU fallback"); }
            CTX = c;
        });
        CTX
    }
}
fn ok(rc: i32) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(rhip_last_error(ctx())).to_string_lossy().into_owned() };
        panic!("rabe-bn (HIP): engine call failed ({}): {}", rc, msg);
    }
}

/// `rabe_bn::FieldError` exactly as `impl From<FieldError> for RabeError` matches it (src/error.rs:60-69).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum FieldError {
    InvalidSliceLength,
    InvalidU512Encoding,
    NotMember,
}

/// `rabe_bn::Group` as rabe uses it: `zero`, `one`, `random`, `is_zero`, `normalize`.
pub trait Group: Sized + Copy + PartialEq + Add<Output = Self> + Sub<Output = Self> + Neg<Output = Self> + Mul<Fr, Output = Self> {
    fn zero() -> Self;
    fn one() -> Self;
    fn random<R: rand::Rng + ?Sized>(rng: &mut R) -> Self { Self::one() * Fr::random(rng) }   // ASSUMPTION: generator * Fr::random
    fn is_zero(&self) -> bool { *self == Self::zero() }
    fn normalize(&mut self) {}
}

// ------------------------------------------------------------------------------------------------ Fr
// r, little-endian limbs (for the canonical-range check of decoded scalars)
const R_MOD: [u32; 8] = [0xf0000001, 0x43e1f593, 0x79b97091, 0x2833e848, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72];
fn below_r(l: &[u32; 8]) -> bool {
    for i in (0..8).rev() {
        if l[i] != R_MOD[i] { return l[i] < R_MOD[i]; }
    }
    false
}
impl Fr {
    pub fn zero() -> Fr { Fr([0; 8]) }
    pub fn one() -> Fr { let mut l = [0u32; 8]; l[0] = 1; Fr(l) }
    /// ASSUMPTION: 512 random bits reduced mod r (two reductions of 256-bit halves combined with 2^256 mod r).
    pub fn random<R: rand::Rng + ?Sized>(rng: &mut R) -> Fr {
        let (mut hi, mut lo) = ([0u8; 32], [0u8; 32]);
        rng.fill_bytes(&mut hi);
        rng.fill_bytes(&mut lo);
        let two256 = Fr::from_slice(&[0xffu8; 32]).unwrap() + Fr::one();      // 2^256 mod r
        Fr::from_slice(&hi).unwrap() * two256 + Fr::from_slice(&lo).unwrap()
    }
    /// ASSUMPTION: 32 big-endian bytes, reduced mod r (`sha3_hash`, src/utils/hash/mod.rs:16,27, feeds digests here).
    pub fn from_slice(b: &[u8]) -> Result<Fr, FieldError> {
        if b.len() != 32 { return Err(FieldError::InvalidSliceLength); }
        let mut o = Fr::zero();
        ok(unsafe { rhip_host_fr_from_be32_reduce(ctx(), b.as_ptr(), &mut o) });
        Ok(o)
    }
    /// decimal string (`usize_to_fr`, src/utils/tools/mod.rs:11-13)
    pub fn from_str(s: &str) -> Option<Fr> {
        let ten = Fr([10, 0, 0, 0, 0, 0, 0, 0]);
        let mut acc = Fr::zero();
        if s.is_empty() { return None; }
        for c in s.bytes() {
            if !c.is_ascii_digit() { return None; }
            acc = acc * ten + Fr([(c - b'0') as u32, 0, 0, 0, 0, 0, 0, 0]);
        }
        Some(acc)
    }
    pub fn inverse(&self) -> Option<Fr> {
        if *self == Fr::zero() { return None; }
        let mut o = Fr::zero();
        ok(unsafe { rhip_host_fr_op(ctx(), FR_INV, self, std::ptr::null(), &mut o) });
        Some(o)
    }
    /// `x.pow(usize_to_fr(j))` in `polynomial` (src/utils/secretsharing/mod.rs:218): self^exp, exp read as an integer
    pub fn pow(&self, exp: Fr) -> Fr {
        let mut o = Fr::zero();
        ok(unsafe { rhip_host_fr_pow(ctx(), self, &exp, &mut o) });
        o
    }
    pub fn is_zero(&self) -> bool { *self == Fr::zero() }
}
macro_rules! fr_binop { ($tr:ident, $f:ident, $op:expr) => {
    impl $tr for Fr { type Output = Fr; fn $f(self, b: Fr) -> Fr { let mut o = Fr::zero(); ok(unsafe { rhip_host_fr_op(ctx(), $op, &self, &b, &mut o) }); o } }
} }
fr_binop!(Add, add, FR_ADD);
fr_binop!(Sub, sub, FR_SUB);
fr_binop!(Mul, mul, FR_MUL);
impl Neg for Fr { type Output = Fr; fn neg(self) -> Fr { let mut o = Fr::zero(); ok(unsafe { rhip_host_fr_op(ctx(), FR_NEG, &self, std::ptr::null(), &mut o) }); o } }

// ------------------------------------------------------------------------------------------------ G1 / G2
macro_rules! group_impl { ($t:ident, $n:expr, $add:ident, $neg:ident, $mul:ident, $one:expr) => {
    impl Add for $t { type Output = $t; fn add(self, b: $t) -> $t { let mut o = $t([0; $n]); ok(unsafe { $add(ctx(), &self, &b, &mut o) }); o } }
    impl Neg for $t { type Output = $t; fn neg(self) -> $t { let mut o = $t([0; $n]); ok(unsafe { $neg(ctx(), &self, &mut o) }); o } }
    impl Sub for $t { type Output = $t; fn sub(self, b: $t) -> $t { self + (-b) } }
    impl Mul<Fr> for $t { type Output = $t; fn mul(self, k: Fr) -> $t { let mut o = $t([0; $n]); ok(unsafe { $mul(ctx(), &self, &k, &mut o) }); o } }
    impl Group for $t { fn zero() -> $t { $t([0; $n]) } fn one() -> $t { $one } }
} }
fn limbs<const N: usize>(words: &[[u32; 8]]) -> [u32; N] { let mut o = [0u32; N]; for (i, w) in words.iter().enumerate() { o[8 * i..8 * i + 8].copy_from_slice(w); } o }
// generators of alt_bn128: G1 = (1, 2); G2 = the standard twist generator (x.c0, x.c1, y.c0, y.c1), little-endian limbs
const G2X0: [u32; 8] = [0xd992f6ed, 0x46debd5c, 0xf75edadd, 0x674322d4, 0x5e5c4479, 0x426a0066, 0x121f1e76, 0x1800deef];
const G2X1: [u32; 8] = [0xaef312c2, 0x97e485b7, 0x35a9e712, 0xf1aa4933, 0x31fb5d25, 0x7260bfb7, 0x920d483a, 0x198e9393];
const G2Y0: [u32; 8] = [0x66fa7daa, 0x4ce6cc01, 0x0c43d37b, 0xe3d1e769, 0x8dcb408f, 0x4aab7180, 0xdb8c6deb, 0x12c85ea5];
const G2Y1: [u32; 8] = [0xd122975b, 0x55acdadc, 0x70b38ef3, 0xbc4b3133, 0x690c3395, 0xec9e99ad, 0x585ff075, 0x090689d0];
group_impl!(G1, 16, rhip_host_g1_add, rhip_host_g1_neg, rhip_host_g1_mul,
            G1(limbs::<16>(&[[1, 0, 0, 0, 0, 0, 0, 0], [2, 0, 0, 0, 0, 0, 0, 0]])));
group_impl!(G2, 32, rhip_host_g2_add, rhip_host_g2_neg, rhip_host_g2_mul, G2(limbs::<32>(&[G2X0, G2X1, G2Y0, G2Y1])));

// ------------------------------------------------------------------------------------------------ Gt / pairing
impl Gt {
    pub fn one() -> Gt { let mut l = [0u32; 96]; l[0] = 1; Gt(l) }
    pub fn pow(&self, k: Fr) -> Gt { let mut o = Gt::one(); ok(unsafe { rhip_host_gt_pow(ctx(), self, &k, &mut o) }); o }
    pub fn inverse(&self) -> Gt { let mut o = Gt::one(); ok(unsafe { rhip_host_gt_inv(ctx(), self, &mut o) }); o }
}
impl Mul for Gt { type Output = Gt; fn mul(self, b: Gt) -> Gt { let mut o = Gt::one(); ok(unsafe { rhip_host_gt_mul(ctx(), &self, &b, &mut o) }); o } }
/// ASSUMPTION: the AES key derivation hashes the 12 Fq coefficients, 32 big-endian bytes each, in tower order.
impl From<Gt> for Vec<u8> {
    fn from(g: Gt) -> Vec<u8> {
        let mut v = Vec::with_capacity(384);
        for c in 0..12 { for w in (0..8).rev() { v.extend_from_slice(&g.0[8 * c + w].to_be_bytes()); } }
        v
    }
}
/// `rng.gen::<Gt>()` -- ASSUMPTION: e(G1::one(), G2::one()) ^ Fr::random
impl rand::distributions::Distribution<Gt> for rand::distributions::Standard {
    fn sample<R: rand::Rng + ?Sized>(&self, rng: &mut R) -> Gt { pairing(G1::one(), G2::one()).pow(Fr::random(rng)) }
}
impl rand::distributions::Distribution<Fr> for rand::distributions::Standard { fn sample<R: rand::Rng + ?Sized>(&self, rng: &mut R) -> Fr { Fr::random(rng) } }
impl rand::distributions::Distribution<G1> for rand::distributions::Standard { fn sample<R: rand::Rng + ?Sized>(&self, rng: &mut R) -> G1 { G1::random(rng) } }
impl rand::distributions::Distribution<G2> for rand::distributions::Standard { fn sample<R: rand::Rng + ?Sized>(&self, rng: &mut R) -> G2 { G2::random(rng) } }

/// optimal-ate pairing with the libff / zcash-bn final-exponentiation chain
pub fn pairing(p: G1, q: G2) -> Gt { let mut o = Gt::one(); ok(unsafe { rhip_host_pairing(ctx(), &p, &q, &mut o) }); o }

// ------------------------------------------------------------------------------------------------ byte forms (serde / borsh)
// ASSUMPTION: the byte layout.  rabe-bn's own encodings are unknown here; this crate encodes every
// element as its canonical wire record (little-endian limbs, the layouts of include/rabe_hip.h), so rabe built against it
// round-trips its own files; interoperability with files written by the real rabe-bn is the open item of DESIGN.md 7.
// Decoding validates what rabe-bn's decoding validates: scalars are < r, coordinates < p (one encoding per element, so `==` is byte
// equality), points on the curve / in the subgroup (FieldError::NotMember).  rhip_host_g1_on_curve rejects coordinates >= p itself.
trait Wire: Sized {
    const BYTES: usize;
    fn to_wire(&self, out: &mut [u8]);
    fn from_wire(b: &[u8]) -> Result<Self, FieldError>;
}
fn words_to_bytes(w: &[u32], out: &mut [u8]) { for (i, x) in w.iter().enumerate() { out[4 * i..4 * i + 4].copy_from_slice(&x.to_le_bytes()); } }
fn bytes_to_words<const N: usize>(b: &[u8]) -> Result<[u32; N], FieldError> {
    if b.len() != 4 * N { return Err(FieldError::InvalidSliceLength); }
    let mut w = [0u32; N];
    for i in 0..N { w[i] = u32::from_le_bytes([b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]]); }
    Ok(w)
}
impl Wire for Fr {
    const BYTES: usize = 32;
    fn to_wire(&self, out: &mut [u8]) { words_to_bytes(&self.0, out) }
    fn from_wire(b: &[u8]) -> Result<Fr, FieldError> {
        let w = bytes_to_words::<8>(b)?;
        if !below_r(&w) { return Err(FieldError::NotMember); }
        Ok(Fr(w))
    }
}
impl Wire for G1 {
    const BYTES: usize = 64;
    fn to_wire(&self, out: &mut [u8]) { words_to_bytes(&self.0, out) }
    fn from_wire(b: &[u8]) -> Result<G1, FieldError> {
        let p = G1(bytes_to_words::<16>(b)?);
        let mut on = 0i32;
        ok(unsafe { rhip_host_g1_on_curve(ctx(), &p, &mut on) });
        if on == 1 { Ok(p) } else { Err(FieldError::NotMember) }
    }
}
impl Wire for G2 {
    const BYTES: usize = 128;
    fn to_wire(&self, out: &mut [u8]) { words_to_bytes(&self.0, out) }
    fn from_wire(b: &[u8]) -> Result<G2, FieldError> {
        let p = G2(bytes_to_words::<32>(b)?);
        let mut on = 0i32;
        // canonical coordinates (< p), on the twist, and in its r-torsion (the twist has cofactor > 1): one engine call
        ok(unsafe { rhip_host_g2_in_subgroup(ctx(), &p, &mut on) });
        if on == 1 { Ok(p) } else { Err(FieldError::NotMember) }
    }
}
impl Wire for Gt {
    const BYTES: usize = 384;
    fn to_wire(&self, out: &mut [u8]) { words_to_bytes(&self.0, out) }
    fn from_wire(b: &[u8]) -> Result<Gt, FieldError> {
        let g = Gt(bytes_to_words::<96>(b)?);
        // canonical coefficients, cyclotomic (Frobenius test) FIRST, then order r: Gt::pow's cyclotomic squarings are only valid
        // once the first test has passed, which is why this is the engine's k_gt_is_member and not a pow here
        let mut on = 0i32;
        ok(unsafe { rhip_host_gt_is_member(ctx(), &g, &mut on) });
        if on == 1 { Ok(g) } else { Err(FieldError::NotMember) }
    }
}

#[cfg(feature = "serde")]
mod serde_impl {
    use super::*;
    use serde::de::{Error, SeqAccess, Visitor};
    use serde::ser::SerializeTuple;
    use serde::{Deserialize, Deserializer, Serialize, Serializer};
    use std::marker::PhantomData;
    // a fixed-length tuple of bytes: the same data in binary and self-describing formats (JSON: an array of numbers)
    struct WireVisitor<T>(PhantomData<T>);
    impl<'de, T: Wire> Visitor<'de> for WireVisitor<T> {
        type Value = T;
        fn expecting(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result { write!(f, "{} bytes", T::BYTES) }
        fn visit_seq<A: SeqAccess<'de>>(self, mut seq: A) -> Result<T, A::Error> {
            let mut b = vec![0u8; T::BYTES];
            for (i, slot) in b.iter_mut().enumerate() {
                *slot = seq.next_element()?.ok_or_else(|| A::Error::invalid_length(i, &self))?;
            }
            T::from_wire(&b).map_err(|e| A::Error::custom(format!("{:?}", e)))
        }
    }
    macro_rules! serde_wire { ($t:ident) => {
        impl Serialize for $t {
            fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
                let mut b = vec![0u8; <$t as Wire>::BYTES];
                self.to_wire(&mut b);
                let mut t = s.serialize_tuple(b.len())?;
                for x in &b { t.serialize_element(x)?; }
                t.end()
            }
        }
        impl<'de> Deserialize<'de> for $t {
            fn deserialize<D: Deserializer<'de>>(d: D) -> Result<$t, D::Error> { d.deserialize_tuple(<$t as Wire>::BYTES, WireVisitor::<$t>(PhantomData)) }
        }
    } }
    serde_wire!(Fr);
    serde_wire!(G1);
    serde_wire!(G2);
    serde_wire!(Gt);
}

#[cfg(feature = "borsh")]
mod borsh_impl {
    use super::*;
    use borsh::io::{Error, ErrorKind, Read, Result, Write};
    use borsh::{BorshDeserialize, BorshSerialize};
    macro_rules! borsh_wire { ($t:ident) => {
        impl BorshSerialize for $t {
            fn serialize<W: Write>(&self, w: &mut W) -> Result<()> {
                let mut b = vec![0u8; <$t as Wire>::BYTES];
                self.to_wire(&mut b);
                w.write_all(&b)
            }
        }
        impl BorshDeserialize for $t {
            fn deserialize_reader<R: Read>(r: &mut R) -> Result<$t> {
                let mut b = vec![0u8; <$t as Wire>::BYTES];
                r.read_exact(&mut b)?;
                <$t as Wire>::from_wire(&b).map_err(|e| Error::new(ErrorKind::InvalidData, format!("{:?}", e)))
            }
        }
    } }
    borsh_wire!(Fr);
    borsh_wire!(G1);
    borsh_wire!(G2);
    borsh_wire!(Gt);
}
