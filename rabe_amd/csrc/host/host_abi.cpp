// C interface of the host layer (include/rabe_host.h): opaque handles, status codes, canonical serialisation.
//
// Serialised layout ("canonical byte form"; the reference's serde/borsh layouts of rabe-bn types are not
// visible in /root/reference -- SURVEY.md 8f item 2 -- so this is the engine's own, borsh-shaped one):
//   u32 little-endian lengths; String = u32 len + UTF-8; Vec<T> = u32 count + items; PolicyLanguage = u8;
//   Fr 32 B, G1 64 B, G2 128 B, Gt 384 B in the wire format of include/rabe_hip.h; struct fields in the
//   reference's declaration order (ac17/mod.rs:58-135, bsw/mod.rs:39-89, lsw/mod.rs:40-83, aw11/mod.rs:46-97).
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

#include "../../../include/rabe_host.h"
#include "schemes.h"

using namespace rabe;
using namespace rabe::host;
using namespace rabe::schemes;

// One queued call of the submission queue (below: "submission queue"); the blocking one-call entry points build one on their stack.
struct rabe_ticket {
  enum Op { AC17_ENC, AC17_DEC, BSW_ENC, BSW_DEC, LSW_ENC, LSW_DEC, AW11_ENC, AW11_DEC };
  Op op;
  const void* a = nullptr;                 // pk (encrypt) / sk (decrypt) / gk (aw11)
  const void* b = nullptr;                 // aw11 decrypt: sk
  std::vector<const void*> pks;            // aw11 encrypt: the authorities' public keys
  std::string policy;
  int32_t language = 0;
  std::vector<std::string> attrs;          // lsw encrypt
  Bytes pt;                                // plaintext (a copy: an asynchronous caller may reuse its buffer)
  const void* ct = nullptr;                // decrypt: the ciphertext object (the caller keeps it alive until the wait)
  bool done = false;
  bool has_verdict = false;                // set wherever rc is decided: a ticket without one is what the leader's catch fails (an empty plaintext is a verdict too)
  int32_t rc = 0;
  std::string err;
  void* obj = nullptr;                     // encrypt: the ciphertext object
  Bytes out;                               // decrypt: the plaintext
};
struct rabe_host {
  Engine eng;
  OsRng os;
  std::unique_ptr<TapeRng> tape;
  std::string err;
  // submission queue: concurrent one-at-a-time calls collected into packed batches (rabe_host_set_coalescing)
  bool coalesce = false;
  uint32_t window_us = 0;
  std::mutex q_mu;
  std::condition_variable q_cv;             // "a batch has finished"
  std::condition_variable q_arrival;        // "something was queued" (the leader's window)
  std::deque<rabe_ticket*> q;
  // up to Q_LANES batches run at once, each on its own engine lane (stream, staging, arena) with its own OS randomness source: a small
  // batch is bound by the latency of its launch sets, not by the chip, so batches side by side multiply the rate.  On a tape: one.
  enum { Q_LANES = 8 };                       // capacity; q_lanes of them are used (RABE_QUEUE_LANES, default 3)
  int q_lanes = 3;
  int q_min_extra = 256;                      // queued requests before a lane other than 0 opens (RABE_QUEUE_MIN_EXTRA)
  bool q_lane_busy[Q_LANES] = {false, false, false, false, false, false, false, false};
  enum { Q_SUB = 3 };                         // groups of ONE batch that run side by side (sub-lane s of lane l = engine lane Q_LANES s + l)
  OsRng q_rng[Q_LANES * Q_SUB];
  uint64_t q_stats[6] = {0, 0, 0, 0, 0, 0};  // batches, requests, groups, requests run singly, microseconds inside batches, largest batch
  // a device GROUP (rabe_host_open_group): the host's own engine + one more per further entry of the device list.  The packed entry
  // points of ac17 / bsw / lsw / aw11 split their items into one contiguous block per engine (pipeline.cpp); everything else runs on `eng`.
  std::vector<std::unique_ptr<Engine>> peers;
  explicit rabe_host(int device) : eng(device) {}
  std::vector<Engine*> engines() {
    std::vector<Engine*> v{&eng};
    for (auto& p : peers) v.push_back(p.get());
    return v;
  }
  Rng& rng() { return tape ? (Rng&)*tape : (Rng&)os; }
};
static thread_local std::string g_err;

// fewest items per chunk of a pipelined packed call (pipeline.cpp).  Measured, not assumed (tools/bench_packed_pipeline.py, tools/exp_r03x.sh):
// once the per-buffer hipMalloc / hipFree, the per-call line preparation and the 79 MB of per-element verdicts were gone from the
// unchunked call, two half-size chunks on two lanes are no faster for ac17 (20 480 items 339 k -> 311 k ops/s, 65 536 458 k -> 433 k,
// 131 072 406 k -> 402 k) and slower for the GPU-bound bsw / lsw / aw11 (-6 ... -10 %): no entry point is chunked unless
// RABE_PACKED_CHUNK asks for it.
static const size_t CHUNK_AC17 = (size_t)1 << 40, CHUNK_BSW = (size_t)1 << 40, CHUNK_LSW = (size_t)1 << 40, CHUNK_AW11 = (size_t)1 << 40;
#define GUARD_BEGIN try {
#define GUARD_END(h)                                                         \
  }                                                                          \
  catch (const RabeError& e) { set_err(h, e.what()); return -1; }            \
  catch (const PolicyError& e) { set_err(h, e.what()); return -1; }          \
  catch (const std::exception& e) { set_err(h, std::string("panic: ") + e.what()); return -2; }
static void set_err(rabe_host* h, const std::string& s) {
  g_err = s;
  if (h) {                                       // several threads may share a host when its submission queue is on
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    h->err = s;
  }
}

// ------------------------------------------------------------------------------------------------ serialisation
struct W {
  Bytes b;
  void u8(uint8_t v) { b.push_back(v); }
  void u32(uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
  void raw(const uint8_t* p, size_t n) { b.insert(b.end(), p, p + n); }
  void str(const std::string& s) { u32((uint32_t)s.size()); raw((const uint8_t*)s.data(), s.size()); }
  void bytes(const Bytes& v) { u32((uint32_t)v.size()); raw(v.data(), v.size()); }
  void fr(const Fr& f) { raw((const uint8_t*)f.l, 32); }
  template <size_t N> void el(const std::array<uint8_t, N>& e) { raw(e.data(), N); }
  template <size_t N> void vec(const std::vector<std::array<uint8_t, N>>& v) { u32((uint32_t)v.size()); for (auto& e : v) el(e); }
  void vfr(const std::vector<Fr>& v) { u32((uint32_t)v.size()); for (auto& e : v) fr(e); }
  void pol(const PolicyRef& p) { str(p.first); u8((uint8_t)p.second); }
};
// r and p, little-endian 32-bit limbs: canonical-range checks of decoded scalars / coordinates
static const uint32_t R_MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
static const uint32_t P_MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
static bool below(const uint8_t* le32, const uint32_t m[8]) {
  for (int i = 7; i >= 0; i--) {
    uint32_t w;
    memcpy(&w, le32 + 4 * i, 4);
    if (w != m[i]) return w < m[i];
  }
  return false;
}
struct R {
  const uint8_t* p;
  size_t n, o = 0;
  // every group element decoded so far (offsets into p), for the membership checks of rabe_obj_deserialize_checked
  std::vector<size_t> g1s, g2s, gts;
  R(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  void need(size_t k) { if (o + k > n) throw RabeError("deserialize: truncated input"); }
  uint8_t u8() { need(1); return p[o++]; }
  uint32_t u32() { need(4); uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)p[o + i] << (8 * i); o += 4; return v; }
  std::string str() { uint32_t l = u32(); need(l); std::string s((const char*)p + o, l); o += l; return s; }
  Bytes bytes() { uint32_t l = u32(); need(l); Bytes b(p + o, p + o + l); o += l; return b; }
  Fr fr() {
    need(32);
    if (!below(p + o, R_MOD)) throw RabeError("deserialize: scalar not below r (FieldError::NotMember)");
    Fr f; memcpy(f.l, p + o, 32); o += 32; return f;
  }
  template <size_t N> std::array<uint8_t, N> el() {
    need(N);
    for (size_t c = 0; c < N; c += 32)
      if (!below(p + o + c, P_MOD)) throw RabeError("deserialize: coordinate not below p (FieldError::NotMember)");
    (N == 64 ? g1s : N == 128 ? g2s : gts).push_back(o);
    std::array<uint8_t, N> e; memcpy(e.data(), p + o, N); o += N; return e;
  }
  template <size_t N> std::vector<std::array<uint8_t, N>> vec() { uint32_t c = u32(); need((size_t)c * N); std::vector<std::array<uint8_t, N>> v; for (uint32_t i = 0; i < c; i++) v.push_back(el<N>()); return v; }
  // fixed-size vectors of the AC17 structs (ASSUMPTION_SIZE = 2: ac17/mod.rs:138): anything else is malformed
  template <size_t N> std::vector<std::array<uint8_t, N>> vec_of(size_t want, const char* what) {
    auto v = vec<N>();
    if (v.size() != want) throw RabeError(std::string("deserialize: ") + what + " has " + std::to_string(v.size()) + " elements, expected " + std::to_string(want));
    return v;
  }
  std::vector<Fr> vfr_of(size_t want, const char* what) {
    auto v = vfr();
    if (v.size() != want) throw RabeError(std::string("deserialize: ") + what + " has " + std::to_string(v.size()) + " elements, expected " + std::to_string(want));
    return v;
  }
  std::vector<Fr> vfr() { uint32_t c = u32(); std::vector<Fr> v; for (uint32_t i = 0; i < c; i++) v.push_back(fr()); return v; }
  PolicyRef pol() { std::string s = str(); uint8_t l = u8(); return {s, l ? PolicyLanguage::HumanPolicy : PolicyLanguage::JsonPolicy}; }
};
typedef std::vector<std::pair<std::string, std::vector<G1>>> NamedG1Vec;
static void w_named(W& w, const NamedG1Vec& v) { w.u32((uint32_t)v.size()); for (auto& e : v) { w.str(e.first); w.vec(e.second); } }
static NamedG1Vec r_named(R& r) { uint32_t c = r.u32(); NamedG1Vec v; for (uint32_t i = 0; i < c; i++) { std::string s = r.str(); v.push_back({s, r.vec_of<64>(3, "an AC17 row")}); } return v; }

static void ser(W& w, int32_t kind, const void* o) {
  switch (kind) {
    case RABE_AC17_PK: { auto& x = *(const ac17::Ac17PublicKey*)o; w.el(x.g); w.vec(x.h_a); w.vec(x.e_gh_ka); break; }
    case RABE_AC17_MSK: { auto& x = *(const ac17::Ac17MasterKey*)o; w.el(x.g); w.el(x.h); w.vec(x.g_k); w.vfr(x.a); w.vfr(x.b); break; }
    case RABE_AC17_CP_SK: { auto& x = *(const ac17::Ac17CpSecretKey*)o; w.u32((uint32_t)x.attr.size()); for (auto& a : x.attr) w.str(a);
                            w.vec(x.sk.k_0); w_named(w, x.sk.k); w.vec(x.sk.k_p); break; }
    case RABE_AC17_CP_CT: { auto& x = *(const ac17::Ac17CpCiphertext*)o; w.pol(x.policy); w.vec(x.ct.c_0); w_named(w, x.ct.c); w.el(x.ct.c_p);
                            w.bytes(x.ct.ct); break; }
    case RABE_AC17_KP_SK: { auto& x = *(const ac17::Ac17KpSecretKey*)o; w.pol(x.policy); w.vec(x.sk.k_0); w_named(w, x.sk.k); w.vec(x.sk.k_p); break; }
    case RABE_AC17_KP_CT: { auto& x = *(const ac17::Ac17KpCiphertext*)o; w.u32((uint32_t)x.attr.size()); for (auto& a : x.attr) w.str(a);
                            w.vec(x.ct.c_0); w_named(w, x.ct.c); w.el(x.ct.c_p); w.bytes(x.ct.ct); break; }
    case RABE_BSW_PK: { auto& x = *(const bsw::CpAbePublicKey*)o; w.el(x.g1); w.el(x.g2); w.el(x.h); w.el(x.f); w.el(x.e_gg_alpha); break; }
    case RABE_BSW_MSK: { auto& x = *(const bsw::CpAbeMasterKey*)o; w.fr(x.beta); w.el(x.g2_alpha); break; }
    case RABE_BSW_SK: { auto& x = *(const bsw::CpAbeSecretKey*)o; w.el(x.d); w.u32((uint32_t)x.d_j.size());
                        for (auto& a : x.d_j) { w.str(a.string); w.el(a.g1); w.el(a.g2); } break; }
    case RABE_BSW_CT: { auto& x = *(const bsw::CpAbeCiphertext*)o; w.pol(x.policy); w.el(x.c); w.el(x.c_p); w.u32((uint32_t)x.c_y.size());
                        for (auto& a : x.c_y) { w.str(a.string); w.el(a.g1); w.el(a.g2); } w.bytes(x.data); break; }
    case RABE_LSW_PK: { auto& x = *(const lsw::KpAbePublicKey*)o; w.el(x.g1); w.el(x.g2); w.el(x.g1_b); w.el(x.g1_b2); w.el(x.h_b); w.el(x.e_gg_alpha); break; }
    case RABE_LSW_MSK: { auto& x = *(const lsw::KpAbeMasterKey*)o; w.fr(x.alpha1); w.fr(x.alpha2); w.fr(x.b); w.el(x.h_g1); w.el(x.h_g2); break; }
    case RABE_LSW_SK: { auto& x = *(const lsw::KpAbeSecretKey*)o; w.pol(x.policy); w.u32((uint32_t)x.dj.size());
                        for (auto& d : x.dj) { w.str(d.name); w.el(d.d1); w.el(d.d2); w.el(d.d3); w.el(d.d4); w.el(d.d5); } break; }
    case RABE_LSW_CT: { auto& x = *(const lsw::KpAbeCiphertext*)o; w.el(x.e1); w.el(x.e2); w.u32((uint32_t)x.ej.size());
                        for (auto& e : x.ej) { w.str(e.name); w.el(e.e1); w.el(e.e2); w.el(e.e3); } w.bytes(x.ct); break; }
    case RABE_AW11_GK: { auto& x = *(const aw11::Aw11GlobalKey*)o; w.el(x.g1); w.el(x.g2); break; }
    case RABE_AW11_PK: { auto& x = *(const aw11::Aw11PublicKey*)o; w.u32((uint32_t)x.attr.size());
                         for (auto& a : x.attr) { w.str(a.name); w.el(a.egg_alpha); w.el(a.g2_y); } break; }
    case RABE_AW11_MSK: { auto& x = *(const aw11::Aw11MasterKey*)o; w.u32((uint32_t)x.attr.size());
                          for (auto& a : x.attr) { w.str(a.name); w.fr(a.alpha); w.fr(a.y); } break; }
    case RABE_AW11_SK: { auto& x = *(const aw11::Aw11SecretKey*)o; w.str(x.gid); w.u32((uint32_t)x.attr.size());
                         for (auto& a : x.attr) { w.str(a.first); w.el(a.second); } break; }
    case RABE_AW11_CT: { auto& x = *(const aw11::Aw11Ciphertext*)o; w.pol(x.policy); w.el(x.c_0); w.u32((uint32_t)x.c.size());
                         for (auto& c : x.c) { w.str(c.name); w.el(c.c1); w.el(c.c2); w.el(c.c3); } w.bytes(x.ct); break; }
    case RABE_GHW11_PK: { auto& x = *(const ghw11::Ghw11PublicKey*)o; w.el(x.g1); w.el(x.g2); w.el(x.g1_a); w.el(x.g2_a); w.el(x.e_gg_alpha); break; }
    case RABE_GHW11_MSK: { auto& x = *(const ghw11::Ghw11MasterKey*)o; w.el(x.g2_alpha); ser(w, RABE_GHW11_PK, &x.pk); break; }
    case RABE_GHW11_SK: { auto& x = *(const ghw11::Ghw11SecretKey*)o; w.el(x.k); w.el(x.l); w.u32((uint32_t)x.attr_key.size());
                          for (auto& a : x.attr_key) { w.str(a.string); w.el(a.k_x); } break; }
    case RABE_GHW11_TK: { auto& x = *(const ghw11::Ghw11TransformKey*)o; w.el(x.k_z); w.el(x.l_z); w.u32((uint32_t)x.attr_key_z.size());
                          for (auto& a : x.attr_key_z) { w.str(a.string); w.el(a.k_x); } break; }
    case RABE_GHW11_RK: { auto& x = *(const ghw11::Ghw11RetrieveKey*)o; w.fr(x.z); break; }
    case RABE_GHW11_CT: { auto& x = *(const ghw11::Ghw11Ciphertext*)o; w.pol(x.policy); w.el(x.c); w.el(x.c1); w.u32((uint32_t)x.ci_di.size());
                          for (auto& r : x.ci_di) { w.str(r.name); w.el(r.c); w.el(r.d); } w.bytes(x.data); break; }
    case RABE_GHW11_TCT: { auto& x = *(const ghw11::Ghw11TransformCiphertext*)o; w.el(x.c); w.el(x.t); break; }
    case RABE_BDABE_PK: { auto& x = *(const bdabe::BdabePublicKey*)o; w.el(x.g1); w.el(x.g2); w.el(x.p1); w.el(x.p2); w.el(x.e_gg_y); break; }
    case RABE_BDABE_MSK: { auto& x = *(const bdabe::BdabeMasterKey*)o; w.fr(x.y); break; }
    case RABE_BDABE_SKA: { auto& x = *(const bdabe::BdabeSecretAuthorityKey*)o; w.str(x.name); w.el(x.a1); w.el(x.a2); w.fr(x.a3); break; }
    case RABE_BDABE_UK: { auto& x = *(const bdabe::BdabeUserKey*)o; w.el(x.sk.u1); w.el(x.sk.u2); w.str(x.pk.u); w.el(x.pk.u1); w.el(x.pk.u2);
                          w.u32((uint32_t)x.sk_a.size()); for (auto& a : x.sk_a) { w.str(a.attr); w.el(a.au1); w.el(a.au2); } break; }
    case RABE_BDABE_PKA: { auto& x = *(const bdabe::BdabePublicAttributeKey*)o; w.str(x.attr); w.el(x.a1); w.el(x.a2); w.el(x.a3); break; }
    case RABE_BDABE_CT: { auto& x = *(const bdabe::BdabeCiphertext*)o; w.pol(x.policy); w.u32((uint32_t)x.j.size());
                          for (auto& t : x.j) { w.u32((uint32_t)t.attr.size()); for (auto& a : t.attr) w.str(a); w.el(t.e1); w.el(t.e2); w.el(t.e3); w.el(t.e4); w.el(t.e5); }
                          w.bytes(x.ct); break; }
    case RABE_MKE08_PK: { auto& x = *(const mke08::Mke08PublicKey*)o; w.el(x.g1); w.el(x.g2); w.el(x.p1); w.el(x.p2); w.el(x.e_gg_y1); w.el(x.e_gg_y2); break; }
    case RABE_MKE08_MSK: { auto& x = *(const mke08::Mke08MasterKey*)o; w.el(x.g1); w.el(x.g2); break; }
    case RABE_MKE08_SKA: { auto& x = *(const mke08::Mke08SecretAuthorityKey*)o; w.str(x.name); w.fr(x.r); break; }
    case RABE_MKE08_UK: { auto& x = *(const mke08::Mke08UserKey*)o; w.el(x.sk.g1); w.el(x.sk.g2); w.str(x.pk.name); w.el(x.pk.g1); w.el(x.pk.g2);
                          w.u32((uint32_t)x.sk_a.size()); for (auto& a : x.sk_a) { w.str(a.attr); w.el(a.g1); w.el(a.g2); } break; }
    case RABE_MKE08_PKA: { auto& x = *(const mke08::Mke08PublicAttributeKey*)o; w.str(x.attr); w.el(x.g1); w.el(x.g2); w.el(x.gt1); w.el(x.gt2); break; }
    case RABE_MKE08_CT: { auto& x = *(const mke08::Mke08Ciphertext*)o; w.pol(x.policy); w.u32((uint32_t)x.e.size());
                          for (auto& t : x.e) { w.u32((uint32_t)t.str.size()); for (auto& a : t.str) w.str(a); w.el(t.j1); w.el(t.j2); w.el(t.j3); w.el(t.j4); w.el(t.j5); w.el(t.j6); }
                          w.bytes(x.ct); break; }
    default: throw RabeError("serialize: unknown object kind");
  }
}
// objects are built in a unique_ptr and released on success: a truncated / malformed input leaks nothing
#define NEW_OBJ(T) std::unique_ptr<T> up(new T()); T* x = up.get()
static void* deser(R& r, int32_t kind) {
  switch (kind) {
    case RABE_AC17_PK: { NEW_OBJ(ac17::Ac17PublicKey); x->g = r.el<64>(); x->h_a = r.vec_of<128>(3, "h_a"); x->e_gh_ka = r.vec_of<384>(2, "e_gh_ka"); return up.release(); }
    case RABE_AC17_MSK: { NEW_OBJ(ac17::Ac17MasterKey); x->g = r.el<64>(); x->h = r.el<128>(); x->g_k = r.vec_of<64>(3, "g_k"); x->a = r.vfr_of(2, "a");
                          x->b = r.vfr_of(2, "b"); return up.release(); }
    case RABE_AC17_CP_SK: { NEW_OBJ(ac17::Ac17CpSecretKey); uint32_t c = r.u32(); for (uint32_t i = 0; i < c; i++) x->attr.push_back(r.str());
                            x->sk.k_0 = r.vec_of<128>(3, "k_0"); x->sk.k = r_named(r); x->sk.k_p = r.vec_of<64>(3, "k_p"); return up.release(); }
    case RABE_AC17_CP_CT: { NEW_OBJ(ac17::Ac17CpCiphertext); x->policy = r.pol(); x->ct.c_0 = r.vec_of<128>(3, "c_0"); x->ct.c = r_named(r);
                            x->ct.c_p = r.el<384>(); x->ct.ct = r.bytes(); return up.release(); }
    case RABE_AC17_KP_SK: { NEW_OBJ(ac17::Ac17KpSecretKey); x->policy = r.pol(); x->sk.k_0 = r.vec_of<128>(3, "k_0"); x->sk.k = r_named(r);
                            x->sk.k_p = r.vec<64>(); if (!x->sk.k_p.empty() && x->sk.k_p.size() != 3) throw RabeError("deserialize: k_p is neither empty nor 3 elements");
                            return up.release(); }
    case RABE_AC17_KP_CT: { NEW_OBJ(ac17::Ac17KpCiphertext); uint32_t c = r.u32(); for (uint32_t i = 0; i < c; i++) x->attr.push_back(r.str());
                            x->ct.c_0 = r.vec_of<128>(3, "c_0"); x->ct.c = r_named(r); x->ct.c_p = r.el<384>(); x->ct.ct = r.bytes(); return up.release(); }
    case RABE_BSW_PK: { NEW_OBJ(bsw::CpAbePublicKey); x->g1 = r.el<64>(); x->g2 = r.el<128>(); x->h = r.el<64>(); x->f = r.el<128>(); x->e_gg_alpha = r.el<384>(); return up.release(); }
    case RABE_BSW_MSK: { NEW_OBJ(bsw::CpAbeMasterKey); x->beta = r.fr(); x->g2_alpha = r.el<128>(); return up.release(); }
    case RABE_BSW_SK: { NEW_OBJ(bsw::CpAbeSecretKey); x->d = r.el<128>(); uint32_t c = r.u32();
                        for (uint32_t i = 0; i < c; i++) { bsw::CpAbeAttribute a; a.string = r.str(); a.g1 = r.el<64>(); a.g2 = r.el<128>(); x->d_j.push_back(a); } return up.release(); }
    case RABE_BSW_CT: { NEW_OBJ(bsw::CpAbeCiphertext); x->policy = r.pol(); x->c = r.el<64>(); x->c_p = r.el<384>(); uint32_t c = r.u32();
                        for (uint32_t i = 0; i < c; i++) { bsw::CpAbeAttribute a; a.string = r.str(); a.g1 = r.el<64>(); a.g2 = r.el<128>(); x->c_y.push_back(a); }
                        x->data = r.bytes(); return up.release(); }
    case RABE_LSW_PK: { NEW_OBJ(lsw::KpAbePublicKey); x->g1 = r.el<64>(); x->g2 = r.el<128>(); x->g1_b = r.el<64>(); x->g1_b2 = r.el<64>(); x->h_b = r.el<64>();
                        x->e_gg_alpha = r.el<384>(); return up.release(); }
    case RABE_LSW_MSK: { NEW_OBJ(lsw::KpAbeMasterKey); x->alpha1 = r.fr(); x->alpha2 = r.fr(); x->b = r.fr(); x->h_g1 = r.el<64>(); x->h_g2 = r.el<128>(); return up.release(); }
    case RABE_LSW_SK: { NEW_OBJ(lsw::KpAbeSecretKey); x->policy = r.pol(); uint32_t c = r.u32();
                        for (uint32_t i = 0; i < c; i++) { lsw::KpAbeKeyRow d; d.name = r.str(); d.d1 = r.el<64>(); d.d2 = r.el<128>(); d.d3 = r.el<64>(); d.d4 = r.el<64>(); d.d5 = r.el<64>(); x->dj.push_back(d); }
                        return up.release(); }
    case RABE_LSW_CT: { NEW_OBJ(lsw::KpAbeCiphertext); x->e1 = r.el<384>(); x->e2 = r.el<128>(); uint32_t c = r.u32();
                        for (uint32_t i = 0; i < c; i++) { lsw::KpAbeCtRow e; e.name = r.str(); e.e1 = r.el<64>(); e.e2 = r.el<64>(); e.e3 = r.el<64>(); x->ej.push_back(e); }
                        x->ct = r.bytes(); return up.release(); }
    case RABE_AW11_GK: { NEW_OBJ(aw11::Aw11GlobalKey); x->g1 = r.el<64>(); x->g2 = r.el<128>(); return up.release(); }
    case RABE_AW11_PK: { NEW_OBJ(aw11::Aw11PublicKey); uint32_t c = r.u32();
                         for (uint32_t i = 0; i < c; i++) { aw11::Aw11PkAttr a; a.name = r.str(); a.egg_alpha = r.el<384>(); a.g2_y = r.el<128>(); x->attr.push_back(a); } return up.release(); }
    case RABE_AW11_MSK: { NEW_OBJ(aw11::Aw11MasterKey); uint32_t c = r.u32();
                          for (uint32_t i = 0; i < c; i++) { aw11::Aw11MkAttr a; a.name = r.str(); a.alpha = r.fr(); a.y = r.fr(); x->attr.push_back(a); } return up.release(); }
    case RABE_AW11_SK: { NEW_OBJ(aw11::Aw11SecretKey); x->gid = r.str(); uint32_t c = r.u32();
                         for (uint32_t i = 0; i < c; i++) { std::string s = r.str(); x->attr.push_back({s, r.el<64>()}); } return up.release(); }
    case RABE_AW11_CT: { NEW_OBJ(aw11::Aw11Ciphertext); x->policy = r.pol(); x->c_0 = r.el<384>(); uint32_t c = r.u32();
                         for (uint32_t i = 0; i < c; i++) { aw11::Aw11CtRow t; t.name = r.str(); t.c1 = r.el<384>(); t.c2 = r.el<128>(); t.c3 = r.el<128>(); x->c.push_back(t); }
                         x->ct = r.bytes(); return up.release(); }
    case RABE_GHW11_PK: { NEW_OBJ(ghw11::Ghw11PublicKey); x->g1 = r.el<64>(); x->g2 = r.el<128>(); x->g1_a = r.el<64>(); x->g2_a = r.el<128>();
                          x->e_gg_alpha = r.el<384>(); return up.release(); }
    case RABE_GHW11_MSK: { NEW_OBJ(ghw11::Ghw11MasterKey); x->g2_alpha = r.el<128>(); auto* pk = (ghw11::Ghw11PublicKey*)deser(r, RABE_GHW11_PK);
                           x->pk = *pk; delete pk; return up.release(); }
    case RABE_GHW11_SK: { NEW_OBJ(ghw11::Ghw11SecretKey); x->k = r.el<128>(); x->l = r.el<128>(); uint32_t c = r.u32();
                          for (uint32_t i = 0; i < c; i++) { ghw11::Ghw11Attribute a; a.string = r.str(); a.k_x = r.el<128>(); x->attr_key.push_back(a); } return up.release(); }
    case RABE_GHW11_TK: { NEW_OBJ(ghw11::Ghw11TransformKey); x->k_z = r.el<128>(); x->l_z = r.el<128>(); uint32_t c = r.u32();
                          for (uint32_t i = 0; i < c; i++) { ghw11::Ghw11Attribute a; a.string = r.str(); a.k_x = r.el<128>(); x->attr_key_z.push_back(a); } return up.release(); }
    case RABE_GHW11_RK: { NEW_OBJ(ghw11::Ghw11RetrieveKey); x->z = r.fr(); return up.release(); }
    case RABE_GHW11_CT: { NEW_OBJ(ghw11::Ghw11Ciphertext); x->policy = r.pol(); x->c = r.el<384>(); x->c1 = r.el<64>(); uint32_t c = r.u32();
                          for (uint32_t i = 0; i < c; i++) { ghw11::Ghw11CtRow t; t.name = r.str(); t.c = r.el<64>(); t.d = r.el<64>(); x->ci_di.push_back(t); }
                          x->data = r.bytes(); return up.release(); }
    case RABE_GHW11_TCT: { NEW_OBJ(ghw11::Ghw11TransformCiphertext); x->c = r.el<384>(); x->t = r.el<384>(); return up.release(); }
    case RABE_BDABE_PK: { NEW_OBJ(bdabe::BdabePublicKey); x->g1 = r.el<64>(); x->g2 = r.el<128>(); x->p1 = r.el<64>(); x->p2 = r.el<128>(); x->e_gg_y = r.el<384>(); return up.release(); }
    case RABE_BDABE_MSK: { NEW_OBJ(bdabe::BdabeMasterKey); x->y = r.fr(); return up.release(); }
    case RABE_BDABE_SKA: { NEW_OBJ(bdabe::BdabeSecretAuthorityKey); x->name = r.str(); x->a1 = r.el<64>(); x->a2 = r.el<128>(); x->a3 = r.fr(); return up.release(); }
    case RABE_BDABE_UK: { NEW_OBJ(bdabe::BdabeUserKey); x->sk.u1 = r.el<64>(); x->sk.u2 = r.el<128>(); x->pk.u = r.str(); x->pk.u1 = r.el<64>(); x->pk.u2 = r.el<128>();
                          uint32_t c = r.u32();
                          for (uint32_t i = 0; i < c; i++) { bdabe::BdabeSecretAttributeKey a; a.attr = r.str(); a.au1 = r.el<64>(); a.au2 = r.el<128>(); x->sk_a.push_back(a); }
                          return up.release(); }
    case RABE_BDABE_PKA: { NEW_OBJ(bdabe::BdabePublicAttributeKey); x->attr = r.str(); x->a1 = r.el<64>(); x->a2 = r.el<128>(); x->a3 = r.el<384>(); return up.release(); }
    case RABE_BDABE_CT: { NEW_OBJ(bdabe::BdabeCiphertext); x->policy = r.pol(); uint32_t c = r.u32();
                          for (uint32_t i = 0; i < c; i++) { bdabe::BdabeCiphertextTuple t; uint32_t na = r.u32(); for (uint32_t k = 0; k < na; k++) t.attr.push_back(r.str());
                            t.e1 = r.el<384>(); t.e2 = r.el<64>(); t.e3 = r.el<128>(); t.e4 = r.el<64>(); t.e5 = r.el<128>(); x->j.push_back(t); }
                          x->ct = r.bytes(); return up.release(); }
    case RABE_MKE08_PK: { NEW_OBJ(mke08::Mke08PublicKey); x->g1 = r.el<64>(); x->g2 = r.el<128>(); x->p1 = r.el<64>(); x->p2 = r.el<128>(); x->e_gg_y1 = r.el<384>();
                          x->e_gg_y2 = r.el<384>(); return up.release(); }
    case RABE_MKE08_MSK: { NEW_OBJ(mke08::Mke08MasterKey); x->g1 = r.el<64>(); x->g2 = r.el<128>(); return up.release(); }
    case RABE_MKE08_SKA: { NEW_OBJ(mke08::Mke08SecretAuthorityKey); x->name = r.str(); x->r = r.fr(); return up.release(); }
    case RABE_MKE08_UK: { NEW_OBJ(mke08::Mke08UserKey); x->sk.g1 = r.el<64>(); x->sk.g2 = r.el<128>(); x->pk.name = r.str(); x->pk.g1 = r.el<64>(); x->pk.g2 = r.el<128>();
                          uint32_t c = r.u32();
                          for (uint32_t i = 0; i < c; i++) { mke08::Mke08SecretAttributeKey a; a.attr = r.str(); a.g1 = r.el<64>(); a.g2 = r.el<128>(); x->sk_a.push_back(a); }
                          return up.release(); }
    case RABE_MKE08_PKA: { NEW_OBJ(mke08::Mke08PublicAttributeKey); x->attr = r.str(); x->g1 = r.el<64>(); x->g2 = r.el<128>(); x->gt1 = r.el<384>(); x->gt2 = r.el<384>();
                           return up.release(); }
    case RABE_MKE08_CT: { NEW_OBJ(mke08::Mke08Ciphertext); x->policy = r.pol(); uint32_t c = r.u32();
                          for (uint32_t i = 0; i < c; i++) { mke08::Mke08CTConjunction t; uint32_t na = r.u32(); for (uint32_t k = 0; k < na; k++) t.str.push_back(r.str());
                            t.j1 = r.el<384>(); t.j2 = r.el<384>(); t.j3 = r.el<64>(); t.j4 = r.el<128>(); t.j5 = r.el<64>(); t.j6 = r.el<128>(); x->e.push_back(t); }
                          x->ct = r.bytes(); return up.release(); }
    default: throw RabeError("deserialize: unknown object kind");
  }
}

static int32_t give_bytes(const Bytes& b, uint8_t** out, size_t* len) {
  *out = (uint8_t*)malloc(b.size() ? b.size() : 1);
  if (!*out) return -1;
  memcpy(*out, b.data(), b.size());
  *len = b.size();
  return 0;
}

template <class R>
static void give_results(rabe_host* h, const std::vector<R>& r, int32_t* status, uint8_t** plaintexts, size_t* lens) {
  for (size_t i = 0; i < r.size(); i++) {
    status[i] = r[i].ok ? 0 : -1;
    plaintexts[i] = nullptr;
    lens[i] = 0;
    if (r[i].ok) give_bytes(r[i].plaintext, &plaintexts[i], &lens[i]);
    else set_err(h, r[i].error);
  }
}
static std::vector<Bytes> byte_items(const uint8_t* const* data, const size_t* lens, size_t n) {
  std::vector<Bytes> v;
  for (size_t i = 0; i < n; i++) v.push_back(Bytes(data[i], data[i] + lens[i]));
  return v;
}
static int32_t give_text(const std::string& s, char** out) {
  *out = (char*)malloc(s.size() + 1);
  if (!*out) return -1;
  memcpy(*out, s.c_str(), s.size() + 1);
  return 0;
}
static std::vector<std::string> strs(const char* const* a, size_t n) {
  std::vector<std::string> v;
  for (size_t i = 0; i < n; i++) v.push_back(a[i]);
  return v;
}
static PolicyLanguage lang_of(int32_t l) { return l == RABE_HUMAN_POLICY ? PolicyLanguage::HumanPolicy : PolicyLanguage::JsonPolicy; }

extern "C" void rabe_obj_free(int32_t kind, void* o);

// ------------------------------------------------------------------------------------------------ submission queue
// The reference's API is one call per ciphertext (ac17/mod.rs:274-279, :385-388; rabe-console/src/mod.rs:1098, 1310), and one call is
// one small launch set: milliseconds of latency for microseconds of chip time.  With the queue on (rabe_host_set_coalescing), calls that
// arrive while a batch is running are collected and go through the PACKED entry points as one batch (group commit: the batch size
// adapts to the load, no timer is needed; window_us > 0 additionally holds a batch open for that long).  Whoever waits first on an
// unfinished ticket runs the queue: no service thread.  Randomness is drawn in arrival order, so a single caller on a fixed tape gets
// the bytes of the unqueued path.
namespace rabe { void parallel_for(size_t n, const std::function<void(size_t)>& fn); }
namespace {
typedef rabe_ticket T;
thread_local Rng* tl_queue_rng = nullptr;          // the running batch's randomness source (a lane's own, or the host's tape)
Rng& qrng(rabe_host* h) { return tl_queue_rng ? *tl_queue_rng : h->rng(); }
void fail(T* t, const std::string& why, int32_t rc = -1) { t->rc = rc; t->err = why; t->has_verdict = true; }
// one call on its own, exactly as the unqueued entry points run it: the fallback when a batch as a whole throws (e.g. one request's
// policy does not parse: it fails alone)
void run_single(rabe_host* h, T* t) {
  try {
    switch (t->op) {
      case T::AC17_ENC: t->obj = new ac17::Ac17CpCiphertext(ac17::cp_encrypt(h->eng, qrng(h), *(const ac17::Ac17PublicKey*)t->a, t->policy, t->pt, lang_of(t->language))); break;
      case T::AC17_DEC: t->out = ac17::cp_decrypt(h->eng, *(const ac17::Ac17CpSecretKey*)t->a, *(const ac17::Ac17CpCiphertext*)t->ct); break;
      case T::BSW_ENC: t->obj = new bsw::CpAbeCiphertext(bsw::encrypt(h->eng, qrng(h), *(const bsw::CpAbePublicKey*)t->a, t->policy, lang_of(t->language), t->pt)); break;
      case T::BSW_DEC: t->out = bsw::decrypt(h->eng, *(const bsw::CpAbeSecretKey*)t->a, *(const bsw::CpAbeCiphertext*)t->ct); break;
      case T::LSW_ENC: t->obj = new lsw::KpAbeCiphertext(lsw::encrypt(h->eng, qrng(h), *(const lsw::KpAbePublicKey*)t->a, t->attrs, t->pt)); break;
      case T::LSW_DEC: t->out = lsw::decrypt(h->eng, *(const lsw::KpAbeSecretKey*)t->a, *(const lsw::KpAbeCiphertext*)t->ct); break;
      case T::AW11_ENC: {
        std::vector<const aw11::Aw11PublicKey*> v;
        for (const void* p : t->pks) v.push_back((const aw11::Aw11PublicKey*)p);
        t->obj = new aw11::Aw11Ciphertext(aw11::encrypt(h->eng, qrng(h), *(const aw11::Aw11GlobalKey*)t->a, v, t->policy, lang_of(t->language), t->pt));
        break;
      }
      case T::AW11_DEC: t->out = aw11::decrypt(h->eng, *(const aw11::Aw11GlobalKey*)t->a, *(const aw11::Aw11SecretKey*)t->b, *(const aw11::Aw11Ciphertext*)t->ct); break;
    }
    t->rc = 0; t->has_verdict = true;
  } catch (const RabeError& e) { fail(t, e.what());
  } catch (const PolicyError& e) { fail(t, e.what());
  } catch (const std::exception& e) { fail(t, std::string("panic: ") + e.what(), -2); }
}
// encrypt requests of one group -> one packed call -> one object per request
void run_encrypt_group(rabe_host* h, const std::vector<T*>& g, int32_t kind,
                       const std::function<bool(const std::vector<std::string>&, const uint32_t*, const uint8_t*, const uint64_t*, uint8_t*, size_t, uint64_t*)>& packed) {
  const size_t n = g.size();
  std::vector<std::string> pols;
  std::map<std::string, uint32_t> seen;
  std::vector<uint32_t> item(n);
  std::vector<uint64_t> pt_off(n + 1, 0), out_off(n + 1, 0);
  Bytes pt;
  for (size_t i = 0; i < n; i++) {
    auto it = seen.find(g[i]->policy);
    if (it == seen.end()) { it = seen.insert({g[i]->policy, (uint32_t)pols.size()}).first; pols.push_back(g[i]->policy); }
    item[i] = it->second;
    pt.insert(pt.end(), g[i]->pt.begin(), g[i]->pt.end());
    pt_off[i + 1] = pt.size();
  }
  if (pt.empty()) pt.push_back(0);
  (void)packed(pols, item.data(), pt.data(), pt_off.data(), nullptr, 0, out_off.data());          // sizes only: nothing is drawn or computed
  Bytes buf((size_t)out_off[n] + 1);
  if (!packed(pols, item.data(), pt.data(), pt_off.data(), buf.data(), buf.size(), out_off.data())) throw RabeError("submission queue: packed encrypt refused its buffer");
  rabe::parallel_for(n, [&](size_t i) {
    R r(buf.data() + out_off[i], (size_t)(out_off[i + 1] - out_off[i]));
    g[i]->obj = deser(r, kind);
    g[i]->rc = 0; g[i]->has_verdict = true;
  });
}
// decrypt requests of one group (one key) -> records -> one packed call (objects of this process: no membership pass, like the
// one-call path) -> one plaintext per request
void run_decrypt_group(rabe_host* h, const std::vector<T*>& g, int32_t ct_kind,
                       const std::function<bool(size_t, const uint8_t*, size_t, const uint64_t*, int32_t*, uint8_t*, size_t, uint64_t*, std::vector<std::string>*)>& packed) {
  const size_t n = g.size();
  std::vector<Bytes> rec(n);
  rabe::parallel_for(n, [&](size_t i) { W w; ser(w, ct_kind, g[i]->ct); rec[i].swap(w.b); });
  std::vector<uint64_t> off(n + 1, 0), pt_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + rec[i].size();
  Bytes blob((size_t)off[n] + 1), out((size_t)off[n] + 1);
  rabe::parallel_for(n, [&](size_t i) { memcpy(blob.data() + off[i], rec[i].data(), rec[i].size()); });
  std::vector<int32_t> status(n, -1);
  std::vector<std::string> errors;
  if (!packed(n, blob.data(), (size_t)off[n], off.data(), status.data(), out.data(), out.size(), pt_off.data(), &errors))
    throw RabeError("submission queue: packed decrypt refused its buffer");
  for (size_t i = 0; i < n; i++) {
    if (status[i] == 0) { g[i]->out.assign(out.begin() + pt_off[i], out.begin() + pt_off[i + 1]); g[i]->rc = 0; g[i]->has_verdict = true; }
    else fail(g[i], i < errors.size() && !errors[i].empty() ? errors[i] : "decryption failed");
  }
}
void run_group(rabe_host* h, const std::vector<T*>& g) {
  T* f = g[0];
  Engine& eng = h->eng;
  const PolicyLanguage lang = lang_of(f->language);
  switch (f->op) {
    case T::AC17_ENC:
      run_encrypt_group(h, g, RABE_AC17_CP_CT, [&](const std::vector<std::string>& pols, const uint32_t* item, const uint8_t* pt, const uint64_t* po, uint8_t* ob, size_t oc, uint64_t* oo) {
        return ac17::cp_encrypt_packed(eng, qrng(h), *(const ac17::Ac17PublicKey*)f->a, pols, lang, g.size(), item, pt, po, ob, oc, oo); });
      break;
    case T::AC17_DEC:
      run_decrypt_group(h, g, RABE_AC17_CP_CT, [&](size_t n, const uint8_t* b, size_t bl, const uint64_t* o, int32_t* st, uint8_t* pb, size_t pc, uint64_t* po, std::vector<std::string>* er) {
        return ac17::cp_decrypt_packed(eng, *(const ac17::Ac17CpSecretKey*)f->a, n, b, bl, o, true, st, pb, pc, po, er); });
      break;
    case T::BSW_ENC:
      run_encrypt_group(h, g, RABE_BSW_CT, [&](const std::vector<std::string>& pols, const uint32_t* item, const uint8_t* pt, const uint64_t* po, uint8_t* ob, size_t oc, uint64_t* oo) {
        return bsw::encrypt_packed(eng, qrng(h), *(const bsw::CpAbePublicKey*)f->a, pols, lang, g.size(), item, pt, po, ob, oc, oo); });
      break;
    case T::BSW_DEC:
      run_decrypt_group(h, g, RABE_BSW_CT, [&](size_t n, const uint8_t* b, size_t bl, const uint64_t* o, int32_t* st, uint8_t* pb, size_t pc, uint64_t* po, std::vector<std::string>* er) {
        return bsw::decrypt_packed(eng, *(const bsw::CpAbeSecretKey*)f->a, n, b, bl, o, true, st, pb, pc, po, er); });
      break;
    case T::AW11_ENC: {
      std::vector<const aw11::Aw11PublicKey*> v;
      for (const void* p : f->pks) v.push_back((const aw11::Aw11PublicKey*)p);
      run_encrypt_group(h, g, RABE_AW11_CT, [&](const std::vector<std::string>& pols, const uint32_t* item, const uint8_t* pt, const uint64_t* po, uint8_t* ob, size_t oc, uint64_t* oo) {
        return aw11::encrypt_packed(eng, qrng(h), *(const aw11::Aw11GlobalKey*)f->a, v, pols, lang, g.size(), item, pt, po, ob, oc, oo); });
      break;
    }
    case T::AW11_DEC:
      run_decrypt_group(h, g, RABE_AW11_CT, [&](size_t n, const uint8_t* b, size_t bl, const uint64_t* o, int32_t* st, uint8_t* pb, size_t pc, uint64_t* po, std::vector<std::string>* er) {
        return aw11::decrypt_packed(eng, *(const aw11::Aw11GlobalKey*)f->a, *(const aw11::Aw11SecretKey*)f->b, n, b, bl, o, true, st, pb, pc, po, er); });
      break;
    case T::LSW_ENC: {                         // the packed form is keyed by attribute LISTS: the requests' lists are the sets
      const size_t n = g.size();
      std::vector<std::vector<std::string>> sets;
      std::map<std::vector<std::string>, uint32_t> seen;
      std::vector<uint32_t> item(n);
      std::vector<uint64_t> pt_off(n + 1, 0), out_off(n + 1, 0);
      Bytes pt;
      for (size_t i = 0; i < n; i++) {
        auto it = seen.find(g[i]->attrs);
        if (it == seen.end()) { it = seen.insert({g[i]->attrs, (uint32_t)sets.size()}).first; sets.push_back(g[i]->attrs); }
        item[i] = it->second;
        pt.insert(pt.end(), g[i]->pt.begin(), g[i]->pt.end());
        pt_off[i + 1] = pt.size();
      }
      if (pt.empty()) pt.push_back(0);
      const lsw::KpAbePublicKey& pk = *(const lsw::KpAbePublicKey*)f->a;
      (void)lsw::encrypt_packed(eng, qrng(h), pk, sets, n, item.data(), pt.data(), pt_off.data(), nullptr, 0, out_off.data());
      Bytes buf((size_t)out_off[n] + 1);
      if (!lsw::encrypt_packed(eng, qrng(h), pk, sets, n, item.data(), pt.data(), pt_off.data(), buf.data(), buf.size(), out_off.data()))
        throw RabeError("submission queue: packed encrypt refused its buffer");
      rabe::parallel_for(n, [&](size_t i) {
        R r(buf.data() + out_off[i], (size_t)(out_off[i + 1] - out_off[i]));
        g[i]->obj = deser(r, RABE_LSW_CT);
        g[i]->rc = 0; g[i]->has_verdict = true;
      });
      break;
    }
    case T::LSW_DEC: {                         // arbitrary (key, ciphertext) pairs: the object-level batch (general pairing jobs)
      std::vector<const lsw::KpAbeSecretKey*> sks;
      std::vector<const lsw::KpAbeCiphertext*> cts;
      for (T* t : g) { sks.push_back((const lsw::KpAbeSecretKey*)t->a); cts.push_back((const lsw::KpAbeCiphertext*)t->ct); }
      auto r = lsw::decrypt_batch(eng, sks, cts);
      for (size_t i = 0; i < g.size(); i++) {
        if (r[i].ok) { g[i]->out = r[i].plaintext; g[i]->rc = 0; g[i]->has_verdict = true; }
        else fail(g[i], r[i].error);
      }
      break;
    }
  }
}
// everything queued, in arrival order, grouped by what one packed call can take (operation, key object(s), language)
void run_queue_batch(rabe_host* h, const std::vector<T*>& batch) {
  std::vector<std::vector<T*>> groups;
  for (T* t : batch) {
    bool placed = false;
    for (auto& g : groups) {
      T* f = g[0];
      if (f->op == t->op && (t->op == T::LSW_DEC || (f->a == t->a && f->b == t->b && f->pks == t->pks)) &&
          ((t->op != T::AC17_ENC && t->op != T::BSW_ENC && t->op != T::AW11_ENC) || f->language == t->language)) {
        g.push_back(t);
        placed = true;
        break;
      }
    }
    if (!placed) groups.push_back({t});
  }
  size_t singles = 0;
  for (auto& g : groups) if (g.size() == 1) singles++;
  {
    std::lock_guard<std::mutex> lk(h->q_mu);
    h->q_stats[2] += groups.size();
    h->q_stats[3] += singles;
  }
  auto run_one = [h](std::vector<T*>& g) {
    if (g.size() == 1) { run_single(h, g[0]); return; }
    try {
      run_group(h, g);
    } catch (const std::exception&) {          // e.g. one request's policy does not parse: every request gets its own verdict
      for (T* t : g) {
        if (t->obj) { rabe_obj_free(t->op == T::AC17_ENC ? RABE_AC17_CP_CT : t->op == T::BSW_ENC ? RABE_BSW_CT : t->op == T::LSW_ENC ? RABE_LSW_CT : RABE_AW11_CT, t->obj); t->obj = nullptr; }
        t->out.clear();
        t->has_verdict = false;
        run_single(h, t);
      }
    }
  };
  // The groups of a batch are independent packed calls (blocking callers are half encrypting, half decrypting at any moment: an encrypt
  // group of ~2 ms and a decrypt group of ~4.5 ms per batch): they run side by side, each on its own sub-lane of the batch's lane
  // (stream, staging, arena, randomness source) and its own thread.  On a tape the draws follow the arrival order: one after the other.
  const int lane = Engine::current_lane();
  const size_t SUB = rabe_host::Q_SUB;
  // (only the small batches of lightly loaded queues: under heavy load the lanes already run batches side by side, and more packed
  // calls at once only contend -- 64 x 64 calls in flight: 66 k ops/s with serial groups, 54 k with concurrent ones)
  // The sub-lanes must EXIST (rabe_host_set_coalescing creates them): a lane index beyond the engine's lanes falls back to lane 0
  // (Engine::cur_lane), and two groups on one lane would share its stream, arena, staging and the context's work buffers -- a batch call
  // on a host without the queue (rabe_*_batch straight from the caller) runs its groups one after the other.
  if (groups.size() < 2 || h->tape || lane < 0 || (size_t)lane >= (size_t)rabe_host::Q_LANES || batch.size() >= (size_t)h->q_min_extra ||
      h->eng.lane_count() < (size_t)rabe_host::Q_LANES * SUB) {
    for (auto& g : groups) run_one(g);
    return;
  }
  std::atomic<size_t> next{0};
  std::exception_ptr first;
  std::mutex emu;
  auto worker = [&](size_t sub) {
    try {
      Engine::Busy working(h->eng);
      Engine::LaneScope on_lane((int)(rabe_host::Q_LANES * sub + (size_t)lane));      // sub-lane 0 is the batch's own lane
      tl_queue_rng = (Rng*)&h->q_rng[rabe_host::Q_LANES * sub + (size_t)lane];
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= groups.size()) break;
        run_one(groups[k]);
      }
    } catch (...) {
      std::lock_guard<std::mutex> g(emu);
      if (!first) first = std::current_exception();
    }
    tl_queue_rng = nullptr;
  };
  Rng* const mine = tl_queue_rng;
  std::vector<std::thread> th;
  for (size_t sub = 1; sub < SUB && sub < groups.size(); sub++) th.emplace_back(worker, sub);
  worker(0);
  for (auto& t : th) t.join();
  tl_queue_rng = mine;
  if (first) std::rethrow_exception(first);
}
void queue_submit(rabe_host* h, T* t) {
  {
    std::lock_guard<std::mutex> g(h->q_mu);
    h->q.push_back(t);
  }
  h->q_arrival.notify_one();                   // only a leader inside its window listens; finished batches are announced on q_cv
}
// blocks until `t` is done; the first waiter that finds no leader becomes one and runs whatever is queued
void queue_wait(rabe_host* h, T* t) {
  std::unique_lock<std::mutex> lk(h->q_mu);
  while (!t->done) {
    int lane = -1;
    const int max_lanes = h->tape ? 1 : h->q_lanes;          // a tape is drawn in arrival order: one batch at a time
    // lane 0 takes whatever is queued; a further lane only opens for a queue that is worth a batch of its own.  With a few dozen blocking
    // callers every extra lane just cuts their requests into smaller batches that contend for the GPU and the runtime (64 blocking threads:
    // 4.8 k ops/s on one lane in batches of 63, 2.6 k on three in batches of 18) -- they wait for lane 0's next batch instead; callers
    // that keep hundreds of calls in flight fill several lanes (64 x 64 in flight: 49 k on one lane, 56-84 k on three).
    if (!h->q.empty())
      for (int k = 0; k < max_lanes && lane < 0; k++) if (!h->q_lane_busy[k] && (k == 0 || h->q.size() >= (size_t)h->q_min_extra)) lane = k;
    if (lane < 0) { h->q_cv.wait(lk); continue; }
    h->q_lane_busy[lane] = true;
    if (h->window_us) {                        // hold the batch open: arrivals inside the window join it
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(h->window_us);
      while (h->q_arrival.wait_until(lk, deadline) != std::cv_status::timeout) {}
    }
    std::vector<T*> batch(h->q.begin(), h->q.end());
    h->q.clear();
    lk.unlock();
    const auto b0 = std::chrono::steady_clock::now();
    try {
      Engine::Busy working(h->eng);          // handles evicted from the engine's caches meanwhile are parked until every lane is done
      Engine::LaneScope on_lane(lane);
      tl_queue_rng = h->tape ? nullptr : (Rng*)&h->q_rng[lane];
      run_queue_batch(h, batch);
      tl_queue_rng = nullptr;
    } catch (const std::exception& e) {      // e.g. bad_alloc while grouping: every ticket of the batch that has no verdict yet fails,
      tl_queue_rng = nullptr;                // the lane is released and the waiters are woken -- nothing crosses the C boundary
      for (T* x : batch) if (!x->has_verdict) { x->has_verdict = true; x->rc = -2; x->err = std::string("panic: ") + e.what(); }
    } catch (...) {
      tl_queue_rng = nullptr;
      for (T* x : batch) if (!x->has_verdict) { x->has_verdict = true; x->rc = -2; x->err = "panic: unknown exception in the submission queue"; }
    }
    const uint64_t us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - b0).count();
    lk.lock();
    for (T* x : batch) x->done = true;
    h->q_stats[0] += 1;
    h->q_stats[1] += batch.size();
    h->q_stats[4] += us;
    if (batch.size() > h->q_stats[5]) h->q_stats[5] = batch.size();
    h->q_lane_busy[lane] = false;
    h->q_cv.notify_all();
  }
}
// The object-level batch entry points (rabe_*_encrypt_batch / _decrypt_batch: arrays of handles in and out) as ONE queue batch run on
// the calling thread: grouped by key, each group through the packed path (device record assembly, KDF + AES-GCM on the device), records
// <-> objects on all host cores.  Round 3 ran them through per-object host assembly: 17 k ops/s where the packed path did 350 k.
void run_tickets_now(rabe_host* h, std::vector<T>& ts) {
  std::vector<T*> p;
  for (auto& t : ts) p.push_back(&t);
  run_queue_batch(h, p);
}
// encrypt batches are all-or-nothing like the functions they replace: the first failure is the call's failure, nothing is handed out
int32_t give_objects(rabe_host* h, std::vector<T>& ts, int32_t kind, void** out) {
  for (auto& t : ts)
    if (t.rc != 0) {
      for (auto& u : ts) if (u.obj) { rabe_obj_free(kind, u.obj); u.obj = nullptr; }
      set_err(h, t.err);
      return t.rc;
    }
  for (size_t i = 0; i < ts.size(); i++) out[i] = ts[i].obj;
  return 0;
}
void give_plaintexts(rabe_host* h, std::vector<T>& ts, int32_t* status, uint8_t** plaintexts, size_t* lens) {
  for (size_t i = 0; i < ts.size(); i++) {
    status[i] = ts[i].rc == 0 ? 0 : -1;
    plaintexts[i] = nullptr;
    lens[i] = 0;
    if (ts[i].rc == 0) give_bytes(ts[i].out, &plaintexts[i], &lens[i]);
    else set_err(h, ts[i].err);
  }
}
// a blocking one-call entry point through the queue
int32_t queue_call(rabe_host* h, T* t, void** obj, uint8_t** out, size_t* len) {
  queue_submit(h, t);
  queue_wait(h, t);
  if (t->rc != 0) { set_err(h, t->err); return t->rc; }
  if (obj) *obj = t->obj;
  if (out) return give_bytes(t->out, out, len);
  return 0;
}
}  // namespace

// ---- packed decrypts of the DNF schemes: the records of a blob become objects (all cores), the membership pass runs once per group over
// every element of the blob, the objects go through the schemes' batch decrypt (one pairing-job launch set, sealed parts opened on the device)
template <class CT>
static std::vector<std::unique_ptr<CT>> parse_blob(rabe_host* h, int32_t kind, size_t n, const uint8_t* blob, size_t len, const uint64_t* off, bool trusted,
                                                   std::vector<std::string>* errors) {
  std::vector<std::unique_ptr<CT>> objs(n);
  errors->assign(n, std::string());
  struct Els { std::vector<size_t> g1, g2, gt; };
  std::vector<Els> els(n);
  for (size_t i = 0; i < n; i++)
    if (off[i] > off[i + 1] || off[i + 1] > len) (*errors)[i] = "deserialize: record offsets outside the blob";
  rabe::parallel_for(n, [&](size_t i) {
    if (!(*errors)[i].empty()) return;
    try {
      R r(blob + off[i], (size_t)(off[i + 1] - off[i]));
      objs[i].reset((CT*)deser(r, kind));
      for (size_t o : r.g1s) els[i].g1.push_back((size_t)off[i] + o);
      for (size_t o : r.g2s) els[i].g2.push_back((size_t)off[i] + o);
      for (size_t o : r.gts) els[i].gt.push_back((size_t)off[i] + o);
    } catch (const std::exception& ex) {
      objs[i].reset();
      (*errors)[i] = ex.what();
      if ((*errors)[i].empty()) (*errors)[i] = "malformed record";
    }
  });
  if (trusted) return objs;
  Engine& e = h->eng;
  Engine::ArenaScope scope(e);
  auto pass = [&](int which, size_t sz, const char* what) {
    std::vector<uint8_t> flat;
    std::vector<uint32_t> owner;
    for (size_t i = 0; i < n; i++) {
      if (!objs[i]) continue;
      const auto& v = which == 1 ? els[i].g1 : which == 2 ? els[i].g2 : els[i].gt;
      for (size_t o : v) { flat.insert(flat.end(), blob + o, blob + o + sz); owner.push_back((uint32_t)i); }
    }
    if (owner.empty()) return;
    DBuf din(&e, flat.data(), flat.size()), dok(&e, owner.size() * 4);
    const int32_t rc = which == 1 ? rhip_g1_on_curve(e.ctx(), owner.size(), din.as<rhip_g1>(), dok.as<uint32_t>())
                     : which == 2 ? rhip_g2_in_subgroup(e.ctx(), owner.size(), din.as<rhip_g2>(), dok.as<uint32_t>())
                                  : rhip_gt_is_member(e.ctx(), owner.size(), din.as<rhip_gt>(), dok.as<uint32_t>());
    e.check(rc, what);
    std::vector<uint32_t> ok(owner.size());
    dok.download(ok.data(), ok.size() * 4);
    for (size_t t = 0; t < ok.size(); t++)
      if (!ok[t] && (*errors)[owner[t]].empty())
        (*errors)[owner[t]] = std::string("deserialize: a ") + what + " element is not a group member (FieldError::NotMember)";
  };
  pass(1, 64, "G1");
  pass(2, 128, "G2");
  pass(3, 384, "Gt");
  for (size_t i = 0; i < n; i++) if (!(*errors)[i].empty()) objs[i].reset();
  return objs;
}
template <class CT, class UK, class BATCH>
static int32_t dnf_decrypt_packed(rabe_host* h, int32_t kind, const void* uk, size_t n, const uint8_t* blob, size_t len, const uint64_t* off, uint32_t flags,
                                  int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, BATCH batch) {
  std::vector<std::string> errors;
  auto objs = parse_blob<CT>(h, kind, n, blob, len, off, (flags & RABE_PACKED_TRUSTED) != 0, &errors);
  std::vector<const UK*> sks;
  std::vector<const CT*> cts;
  std::vector<size_t> who;
  for (size_t i = 0; i < n; i++) if (objs[i]) { sks.push_back((const UK*)uk); cts.push_back(objs[i].get()); who.push_back(i); }
  std::vector<DecryptResult> res;
  if (!who.empty()) res = batch(sks, cts);
  std::vector<const Bytes*> pt(n, nullptr);
  for (size_t t = 0; t < who.size(); t++) {
    if (res[t].ok) pt[who[t]] = &res[t].plaintext;
    else errors[who[t]] = res[t].error.empty() ? std::string("decryption failed") : res[t].error;
  }
  pt_off[0] = 0;
  for (size_t i = 0; i < n; i++) pt_off[i + 1] = pt_off[i] + (pt[i] ? pt[i]->size() : 0);
  if (pt_off[n] > pt_cap || (pt_off[n] && !pt_buf)) return 1;
  for (size_t i = 0; i < n; i++) {
    status[i] = pt[i] ? 0 : -1;
    if (pt[i] && !pt[i]->empty()) memcpy(pt_buf + pt_off[i], pt[i]->data(), pt[i]->size());
  }
  for (const auto& e : errors) if (!e.empty()) { set_err(h, e); break; }
  return 0;
}
extern "C" {

int32_t rabe_host_create(int32_t device, rabe_host** out) {
  if (!out) return -1;
  *out = nullptr;
  GUARD_BEGIN
  *out = new rabe_host(device);
  return 0;
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_host_abi_version(void) { return RABE_HOST_ABI_VERSION; }
int32_t rabe_host_create_checked(int32_t abi_version, int32_t device, rabe_host** out) {
  if (out) *out = nullptr;
  if (abi_version != RABE_HOST_ABI_VERSION) {
    g_err = "rabe_host.h revision " + std::to_string(abi_version) + " does not match the library's (" + std::to_string(RABE_HOST_ABI_VERSION) + ")";
    return -3;
  }
  return rabe_host_create(device, out);
}
int32_t rabe_host_open_group_checked(int32_t abi_version, size_t n_devices, const int32_t* devices, rabe_host** out) {
  if (out) *out = nullptr;
  if (abi_version != RABE_HOST_ABI_VERSION) {
    g_err = "rabe_host.h revision " + std::to_string(abi_version) + " does not match the library's (" + std::to_string(RABE_HOST_ABI_VERSION) + ")";
    return -3;
  }
  if (!out || !n_devices || !devices || n_devices > 64) { g_err = "rabe_host_open_group: 1 .. 64 devices"; return -1; }
  GUARD_BEGIN
  std::unique_ptr<rabe_host> h(new rabe_host(devices[0]));
  for (size_t i = 1; i < n_devices; i++) h->peers.emplace_back(new Engine(devices[i]));
  h->eng.make_current();
  *out = h.release();
  return 0;
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_host_group_size(rabe_host* h) { return h ? (int32_t)(1 + h->peers.size()) : -1; }
void rabe_host_destroy(rabe_host* h) { delete h; }
const char* rabe_host_last_error(rabe_host* h) { return h ? h->err.c_str() : g_err.c_str(); }
int32_t rabe_host_set_fixed_base_min(rabe_host* h, size_t n) {
  if (!h) return -1;
  h->eng.fixed_base_min = n ? n : 1;
  return 0;
}
int32_t rabe_host_set_tape(rabe_host* h, const uint8_t* fr_le32, size_t n) {
  if (!h) return -1;
  if (!n) { h->tape.reset(); return 0; }
  std::vector<Fr> t(n);
  for (size_t i = 0; i < n; i++) memcpy(t[i].l, fr_le32 + 32 * i, 32);
  h->tape.reset(new TapeRng(t));
  return 0;
}
void rabe_bytes_free(void* p) { free(p); }

// ---- submission queue (see above): switch, asynchronous submits, wait
int32_t rabe_host_set_coalescing(rabe_host* h, int32_t on, uint32_t window_us) {
  if (!h) return -1;
  GUARD_BEGIN
  if (on) {
    if (const char* e = getenv("RABE_QUEUE_LANES")) { const int v = atoi(e); if (v >= 1 && v <= (int)rabe_host::Q_LANES) h->q_lanes = v; }
    if (const char* e = getenv("RABE_QUEUE_MIN_EXTRA")) { const int v = atoi(e); if (v >= 1) h->q_min_extra = v; }
    h->eng.ensure_lanes((size_t)rabe_host::Q_LANES * rabe_host::Q_SUB);          // before any thread is handed one (lanes are cheap until used)
  }
  std::lock_guard<std::mutex> g(h->q_mu);
  h->coalesce = on != 0;
  h->window_us = window_us;
  return 0;
  GUARD_END(h)
}
int32_t rabe_host_queue_stats(rabe_host* h, uint64_t out[6]) {
  if (!h || !out) return -1;
  std::lock_guard<std::mutex> g(h->q_mu);
  for (int i = 0; i < 6; i++) out[i] = h->q_stats[i];
  return 0;
}
static int32_t submit_new(rabe_host* h, rabe_ticket* t, rabe_ticket** out) {
  *out = t;
  queue_submit(h, t);
  return 0;
}
// every *_submit validates its pointers before it touches them and never lets an exception (bad_alloc of the copies) out
#define SUBMIT_BEGIN(cond)                                     \
  if (!h || !ticket) return -1;                                \
  *ticket = nullptr;                                           \
  if (!(cond)) { set_err(h, "null argument"); return -1; }     \
  rabe_ticket* t = nullptr;                                    \
  try {                                                        \
    t = new rabe_ticket;
#define SUBMIT_END                                             \
    return submit_new(h, t, ticket);                           \
  } catch (const std::exception& e) { delete t; *ticket = nullptr; set_err(h, std::string("panic: ") + e.what()); return -2; }
int32_t rabe_ac17_cp_encrypt_submit(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* pt, size_t len, rabe_ticket** ticket) {
  SUBMIT_BEGIN(pk && policy && (pt || !len))
  t->op = rabe_ticket::AC17_ENC; t->a = pk; t->policy = policy; t->language = language; t->pt.assign(pt, pt + len);
  SUBMIT_END
}
int32_t rabe_ac17_cp_decrypt_submit(rabe_host* h, const void* sk, const void* ct, rabe_ticket** ticket) {
  SUBMIT_BEGIN(sk && ct)
  t->op = rabe_ticket::AC17_DEC; t->a = sk; t->ct = ct;
  SUBMIT_END
}
int32_t rabe_bsw_encrypt_submit(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* pt, size_t len, rabe_ticket** ticket) {
  SUBMIT_BEGIN(pk && policy && (pt || !len))
  t->op = rabe_ticket::BSW_ENC; t->a = pk; t->policy = policy; t->language = language; t->pt.assign(pt, pt + len);
  SUBMIT_END
}
int32_t rabe_bsw_decrypt_submit(rabe_host* h, const void* sk, const void* ct, rabe_ticket** ticket) {
  SUBMIT_BEGIN(sk && ct)
  t->op = rabe_ticket::BSW_DEC; t->a = sk; t->ct = ct;
  SUBMIT_END
}
int32_t rabe_lsw_encrypt_submit(rabe_host* h, const void* pk, const char* const* attributes, size_t n, const uint8_t* pt, size_t len, rabe_ticket** ticket) {
  SUBMIT_BEGIN(pk && (attributes || !n) && (pt || !len))
  for (size_t i = 0; i < n; i++) if (!attributes[i]) { delete t; set_err(h, "null attribute"); return -1; }
  t->op = rabe_ticket::LSW_ENC; t->a = pk; t->attrs = strs(attributes, n); t->pt.assign(pt, pt + len);
  SUBMIT_END
}
int32_t rabe_lsw_decrypt_submit(rabe_host* h, const void* sk, const void* ct, rabe_ticket** ticket) {
  SUBMIT_BEGIN(sk && ct)
  t->op = rabe_ticket::LSW_DEC; t->a = sk; t->ct = ct;
  SUBMIT_END
}
int32_t rabe_aw11_encrypt_submit(rabe_host* h, const void* gk, const void* const* pks, size_t n_pks, const char* policy, int32_t language,
                                 const uint8_t* data, size_t len, rabe_ticket** ticket) {
  SUBMIT_BEGIN(gk && (pks || !n_pks) && policy && (data || !len))
  for (size_t i = 0; i < n_pks; i++) if (!pks[i]) { delete t; set_err(h, "null authority key"); return -1; }
  t->op = rabe_ticket::AW11_ENC; t->a = gk; t->pks.assign(pks, pks + n_pks); t->policy = policy; t->language = language;
  t->pt.assign(data, data + len);
  SUBMIT_END
}
int32_t rabe_aw11_decrypt_submit(rabe_host* h, const void* gk, const void* sk, const void* ct, rabe_ticket** ticket) {
  SUBMIT_BEGIN(gk && sk && ct)
  t->op = rabe_ticket::AW11_DEC; t->a = gk; t->b = sk; t->ct = ct;
  SUBMIT_END
}
#undef SUBMIT_BEGIN
#undef SUBMIT_END
// Measurement helper (bench.py: object_api.threads): `threads` native host threads call the PUBLIC one-call entry points on one host --
// depth 1: rabe_ac17_cp_encrypt then rabe_ac17_cp_decrypt, blocking; depth > 1: that many rabe_*_submit calls in flight per thread -- for
// `seconds`, checking every plaintext.  Native threads: an interpreter's global lock would be what is measured otherwise.
int32_t rabe_bench_ac17_threads(rabe_host* h, const void* pk, const void* sk, const char* const* policies, size_t n_policies, int32_t language,
                                uint32_t threads, uint32_t depth, double seconds, uint64_t* ops, uint64_t* bad) {
  if (!h || !pk || !sk || !policies || !n_policies || !threads || !depth || !ops || !bad) return -1;
  std::atomic<uint64_t> n_ok{0}, n_bad{0};
  const auto stop = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(seconds));
  std::vector<std::thread> th;
  for (uint32_t tid = 0; tid < threads; tid++) {
    th.emplace_back([&, tid]() {
      uint64_t k = 0;
      while (std::chrono::steady_clock::now() < stop) {
        std::vector<Bytes> pts(depth);
        for (uint32_t j = 0; j < depth; j++) {
          const char* msg = "dance like no one's watching, encrypt like everyone is!";
          pts[j].assign(msg, msg + strlen(msg));
          pts[j].push_back((uint8_t)tid); pts[j].push_back((uint8_t)(k + j)); pts[j].push_back((uint8_t)((k + j) >> 8));
        }
        if (depth == 1) {
          void* ct = nullptr;
          uint8_t* out = nullptr;
          size_t len = 0;
          bool ok = rabe_ac17_cp_encrypt(h, pk, policies[(tid + k) % n_policies], language, pts[0].data(), pts[0].size(), &ct) == 0 &&
                    rabe_ac17_cp_decrypt(h, sk, ct, &out, &len) == 0 && len == pts[0].size() && memcmp(out, pts[0].data(), len) == 0;
          if (out) free(out);
          if (ct) rabe_obj_free(RABE_AC17_CP_CT, ct);
          (ok ? n_ok : n_bad)++;
        } else {
          std::vector<rabe_ticket*> tk(depth, nullptr);
          std::vector<void*> cts(depth, nullptr);
          for (uint32_t j = 0; j < depth; j++)
            rabe_ac17_cp_encrypt_submit(h, pk, policies[(tid + k + j) % n_policies], language, pts[j].data(), pts[j].size(), &tk[j]);
          for (uint32_t j = 0; j < depth; j++) if (tk[j]) rabe_ticket_wait(h, tk[j], &cts[j], nullptr, nullptr);
          for (uint32_t j = 0; j < depth; j++) { tk[j] = nullptr; if (cts[j]) rabe_ac17_cp_decrypt_submit(h, sk, cts[j], &tk[j]); }
          for (uint32_t j = 0; j < depth; j++) {
            uint8_t* out = nullptr;
            size_t len = 0;
            const bool ok = tk[j] && rabe_ticket_wait(h, tk[j], nullptr, &out, &len) == 0 && len == pts[j].size() && memcmp(out, pts[j].data(), len) == 0;
            if (out) free(out);
            if (cts[j]) rabe_obj_free(RABE_AC17_CP_CT, cts[j]);
            (ok ? n_ok : n_bad)++;
          }
        }
        k += depth;
      }
    });
  }
  for (auto& t : th) t.join();
  *ops = n_ok.load();
  *bad = n_bad.load();
  return 0;
}
int32_t rabe_ticket_wait(rabe_host* h, rabe_ticket* t, void** obj, uint8_t** out, size_t* len) {
  if (!h || !t) return -1;
  try {
    queue_wait(h, t);
  } catch (const std::exception& e) {          // the leader section guards itself; what is left is the mutex / condition variable
    set_err(h, std::string("panic: ") + e.what());
    return -2;                                 // the ticket stays with the queue: freeing it here could race with a leader
  }
  int32_t rc = t->rc;
  if (rc != 0) set_err(h, t->err);
  else {
    const bool enc = t->op == rabe_ticket::AC17_ENC || t->op == rabe_ticket::BSW_ENC || t->op == rabe_ticket::LSW_ENC || t->op == rabe_ticket::AW11_ENC;
    if (enc) {
      if (obj) { *obj = t->obj; t->obj = nullptr; }
    } else if (out && len) {
      rc = give_bytes(t->out, out, len);
    }
  }
  if (t->obj) rabe_obj_free(t->op == rabe_ticket::AC17_ENC ? RABE_AC17_CP_CT : t->op == rabe_ticket::BSW_ENC ? RABE_BSW_CT : t->op == rabe_ticket::LSW_ENC ? RABE_LSW_CT : RABE_AW11_CT, t->obj);
  delete t;
  return rc;
}

void rabe_obj_free(int32_t kind, void* o) {
  switch (kind) {
    case RABE_AC17_PK: delete (ac17::Ac17PublicKey*)o; break;
    case RABE_AC17_MSK: delete (ac17::Ac17MasterKey*)o; break;
    case RABE_AC17_CP_SK: delete (ac17::Ac17CpSecretKey*)o; break;
    case RABE_AC17_CP_CT: delete (ac17::Ac17CpCiphertext*)o; break;
    case RABE_AC17_KP_SK: delete (ac17::Ac17KpSecretKey*)o; break;
    case RABE_AC17_KP_CT: delete (ac17::Ac17KpCiphertext*)o; break;
    case RABE_BSW_PK: delete (bsw::CpAbePublicKey*)o; break;
    case RABE_BSW_MSK: delete (bsw::CpAbeMasterKey*)o; break;
    case RABE_BSW_SK: delete (bsw::CpAbeSecretKey*)o; break;
    case RABE_BSW_CT: delete (bsw::CpAbeCiphertext*)o; break;
    case RABE_LSW_PK: delete (lsw::KpAbePublicKey*)o; break;
    case RABE_LSW_MSK: delete (lsw::KpAbeMasterKey*)o; break;
    case RABE_LSW_SK: delete (lsw::KpAbeSecretKey*)o; break;
    case RABE_LSW_CT: delete (lsw::KpAbeCiphertext*)o; break;
    case RABE_AW11_GK: delete (aw11::Aw11GlobalKey*)o; break;
    case RABE_AW11_PK: delete (aw11::Aw11PublicKey*)o; break;
    case RABE_AW11_MSK: delete (aw11::Aw11MasterKey*)o; break;
    case RABE_AW11_SK: delete (aw11::Aw11SecretKey*)o; break;
    case RABE_AW11_CT: delete (aw11::Aw11Ciphertext*)o; break;
    case RABE_BDABE_PK: delete (bdabe::BdabePublicKey*)o; break;
    case RABE_BDABE_MSK: delete (bdabe::BdabeMasterKey*)o; break;
    case RABE_BDABE_SKA: delete (bdabe::BdabeSecretAuthorityKey*)o; break;
    case RABE_BDABE_UK: delete (bdabe::BdabeUserKey*)o; break;
    case RABE_BDABE_PKA: delete (bdabe::BdabePublicAttributeKey*)o; break;
    case RABE_BDABE_CT: delete (bdabe::BdabeCiphertext*)o; break;
    case RABE_MKE08_PK: delete (mke08::Mke08PublicKey*)o; break;
    case RABE_MKE08_MSK: delete (mke08::Mke08MasterKey*)o; break;
    case RABE_MKE08_SKA: delete (mke08::Mke08SecretAuthorityKey*)o; break;
    case RABE_MKE08_UK: delete (mke08::Mke08UserKey*)o; break;
    case RABE_MKE08_PKA: delete (mke08::Mke08PublicAttributeKey*)o; break;
    case RABE_MKE08_CT: delete (mke08::Mke08Ciphertext*)o; break;
    case RABE_GHW11_PK: delete (ghw11::Ghw11PublicKey*)o; break;
    case RABE_GHW11_MSK: delete (ghw11::Ghw11MasterKey*)o; break;
    case RABE_GHW11_SK: delete (ghw11::Ghw11SecretKey*)o; break;
    case RABE_GHW11_TK: delete (ghw11::Ghw11TransformKey*)o; break;
    case RABE_GHW11_RK: delete (ghw11::Ghw11RetrieveKey*)o; break;
    case RABE_GHW11_CT: delete (ghw11::Ghw11Ciphertext*)o; break;
    case RABE_GHW11_TCT: delete (ghw11::Ghw11TransformCiphertext*)o; break;
    default: break;
  }
}
int32_t rabe_obj_serialize(int32_t kind, const void* obj, uint8_t** out, size_t* len) {
  GUARD_BEGIN
  W w;
  ser(w, kind, obj);
  return give_bytes(w.b, out, len);
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_obj_deserialize(int32_t kind, const uint8_t* data, size_t len, void** obj) {
  GUARD_BEGIN
  R r(data, len);
  *obj = deser(r, kind);
  return 0;
  GUARD_END((rabe_host*)nullptr)
}
// The same, plus group membership of every decoded element on the GPU (one batched launch per group): G1 on the curve, G2
// on the twist AND in its r-torsion, Gt in the order-r subgroup -- what rabe-bn's decoding establishes (FieldError::NotMember).
// Use it for anything that arrives from outside; the unchecked form is for objects this process serialised itself.
int32_t rabe_obj_deserialize_checked(rabe_host* h, int32_t kind, const uint8_t* data, size_t len, void** obj) {
  GUARD_BEGIN
  if (!h) throw RabeError("rabe_obj_deserialize_checked: no host");
  R r(data, len);
  void* o = deser(r, kind);
  try {
    Engine& e = h->eng;
    auto run = [&](const std::vector<size_t>& offs, size_t sz, int which, const char* what) {
      if (offs.empty()) return;
      std::vector<uint8_t> flat(offs.size() * sz);
      for (size_t i = 0; i < offs.size(); i++) memcpy(flat.data() + i * sz, data + offs[i], sz);
      DBuf din(&e, flat.data(), flat.size()), dok(&e, offs.size() * 4);
      int32_t rc = which == 1 ? rhip_g1_on_curve(e.ctx(), offs.size(), din.as<rhip_g1>(), dok.as<uint32_t>())
                 : which == 2 ? rhip_g2_in_subgroup(e.ctx(), offs.size(), din.as<rhip_g2>(), dok.as<uint32_t>())
                              : rhip_gt_is_member(e.ctx(), offs.size(), din.as<rhip_gt>(), dok.as<uint32_t>());
      e.check(rc, what);
      std::vector<uint32_t> ok(offs.size());
      dok.download(ok.data(), ok.size() * 4);
      for (size_t i = 0; i < ok.size(); i++)
        if (!ok[i]) throw RabeError(std::string("deserialize: ") + what + " element " + std::to_string(i) + " is not a group member (FieldError::NotMember)");
    };
    run(r.g1s, 64, 1, "G1");
    run(r.g2s, 128, 2, "G2");
    run(r.gts, 384, 3, "Gt");
  } catch (...) {
    rabe_obj_free(kind, o);
    throw;
  }
  *obj = o;
  return 0;
  GUARD_END(h)
}

// ---------------------------------------------------------------- ac17
int32_t rabe_ac17_setup(rabe_host* h, void** pk, void** msk) {
  GUARD_BEGIN
  auto r = ac17::setup(h->eng, h->rng());
  *pk = new ac17::Ac17PublicKey(r.first);
  *msk = new ac17::Ac17MasterKey(r.second);
  return 0;
  GUARD_END(h)
}
int32_t rabe_ac17_cp_keygen(rabe_host* h, const void* msk, const char* const* attributes, size_t n, void** sk) {
  GUARD_BEGIN
  *sk = new ac17::Ac17CpSecretKey(ac17::cp_keygen(h->eng, h->rng(), *(const ac17::Ac17MasterKey*)msk, strs(attributes, n)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_ac17_cp_encrypt(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* pt, size_t len, void** ct) {
  GUARD_BEGIN
  if (h && h->coalesce) {
    rabe_ticket t; t.op = rabe_ticket::AC17_ENC; t.a = pk; t.policy = policy; t.language = language; t.pt.assign(pt, pt + len);
    return queue_call(h, &t, ct, nullptr, nullptr);
  }
  *ct = new ac17::Ac17CpCiphertext(ac17::cp_encrypt(h->eng, h->rng(), *(const ac17::Ac17PublicKey*)pk, policy, Bytes(pt, pt + len), lang_of(language)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_ac17_cp_decrypt(rabe_host* h, const void* sk, const void* ct, uint8_t** out, size_t* len) {
  GUARD_BEGIN
  if (h && h->coalesce) {
    rabe_ticket t; t.op = rabe_ticket::AC17_DEC; t.a = sk; t.ct = ct;
    return queue_call(h, &t, nullptr, out, len);
  }
  return give_bytes(ac17::cp_decrypt(h->eng, *(const ac17::Ac17CpSecretKey*)sk, *(const ac17::Ac17CpCiphertext*)ct), out, len);
  GUARD_END(h)
}
int32_t rabe_ac17_cp_decrypt_gt(rabe_host* h, const void* sk, const void* ct, uint8_t out_gt[384]) {
  GUARD_BEGIN
  Gt g = ac17::cp_decrypt_gt(h->eng, *(const ac17::Ac17CpSecretKey*)sk, *(const ac17::Ac17CpCiphertext*)ct);
  memcpy(out_gt, g.data(), 384);
  return 0;
  GUARD_END(h)
}
int32_t rabe_ac17_cp_encrypt_batch(rabe_host* h, const void* pk, size_t n, const char* const* policies, int32_t language,
                                   const uint8_t* const* plaintexts, const size_t* lens, void** cts) {
  GUARD_BEGIN
  std::vector<rabe_ticket> ts(n);
  for (size_t i = 0; i < n; i++) { ts[i].op = rabe_ticket::AC17_ENC; ts[i].a = pk; ts[i].policy = policies[i]; ts[i].language = language; ts[i].pt.assign(plaintexts[i], plaintexts[i] + lens[i]); }
  run_tickets_now(h, ts);
  return give_objects(h, ts, RABE_AC17_CP_CT, cts);
  GUARD_END(h)
}
int32_t rabe_ac17_cp_encrypt_packed(rabe_host* h, const void* pk, const char* const* policies, size_t n_policies, int32_t language, size_t n_items,
                                    const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* ct_buf, size_t ct_cap,
                                    uint64_t* ct_off) {
  GUARD_BEGIN
  const auto& key = *(const ac17::Ac17PublicKey*)pk;
  const auto pols = strs(policies, n_policies);
  const auto lang = lang_of(language);
  if (!item_policy || !pt_off || !ct_off) throw RabeError("cp_encrypt_packed: null input");
  return pipeline::produce(h->engines(), h->rng(), n_items, CHUNK_AC17, [&](Engine& eng, size_t lo, size_t hi, Rng& r, uint8_t* out, size_t cap, uint64_t* off) {
    return ac17::cp_encrypt_packed(eng, r, key, pols, lang, hi - lo, item_policy + lo, pt_blob, pt_off + lo, out, cap, off);
  }, ct_buf, ct_cap, ct_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_ac17_cp_keygen_packed(rabe_host* h, const void* msk, const char* const* attributes, const size_t* counts, size_t n_sets, size_t n_items,
                                   const uint32_t* item_set, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off) {
  GUARD_BEGIN
  std::vector<std::vector<std::string>> sets(n_sets);
  size_t at = 0;
  for (size_t s = 0; s < n_sets; s++)
    for (size_t k = 0; k < counts[s]; k++) sets[s].push_back(attributes[at++]);
  return ac17::cp_keygen_packed(h->eng, h->rng(), *(const ac17::Ac17MasterKey*)msk, sets, n_items, item_set, sk_buf, sk_cap, sk_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_ac17_cp_decrypt_packed(rabe_host* h, const void* sk, size_t n_items, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off,
                                    uint32_t flags, int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off) {
  GUARD_BEGIN
  std::vector<std::string> errors;
  const auto& key = *(const ac17::Ac17CpSecretKey*)sk;
  const bool trusted = (flags & RABE_PACKED_TRUSTED) != 0;
  if (!pipeline::consume(h->engines(), n_items, CHUNK_AC17, ct_off, ct_len, [&](Engine& eng, size_t lo, size_t hi, int32_t* st, uint8_t* pt, size_t cap, uint64_t* off,
                                                                           std::vector<std::string>* errs) {
        return ac17::cp_decrypt_packed(eng, key, hi - lo, ct_blob, ct_len, ct_off + lo, trusted, st, pt, cap, off, errs);
      }, status, pt_buf, pt_cap, pt_off, &errors))
    return 1;
  for (const auto& e : errors) if (!e.empty()) { set_err(h, e); break; }
  return 0;
  GUARD_END(h)
}
int32_t rabe_ac17_kp_encrypt_packed(rabe_host* h, const void* pk, const char* const* attributes, const size_t* counts, size_t n_sets, size_t n_items,
                                    const uint32_t* item_set, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* ct_buf, size_t ct_cap, uint64_t* ct_off) {
  GUARD_BEGIN
  if (!item_set || !pt_off || !ct_off) throw RabeError("kp_encrypt_packed: null input");
  std::vector<std::vector<std::string>> sets(n_sets);
  size_t at = 0;
  for (size_t s = 0; s < n_sets; s++)
    for (size_t k = 0; k < counts[s]; k++) sets[s].push_back(attributes[at++]);
  return ac17::kp_encrypt_packed(h->eng, h->rng(), *(const ac17::Ac17PublicKey*)pk, sets, n_items, item_set, pt_blob, pt_off, ct_buf, ct_cap, ct_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_ac17_kp_decrypt_packed(rabe_host* h, const void* sk, size_t n_items, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off,
                                    uint32_t flags, int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off) {
  GUARD_BEGIN
  std::vector<std::string> errors;
  if (!ac17::kp_decrypt_packed(h->eng, *(const ac17::Ac17KpSecretKey*)sk, n_items, ct_blob, ct_len, ct_off, (flags & RABE_PACKED_TRUSTED) != 0, status,
                               pt_buf, pt_cap, pt_off, &errors))
    return 1;
  for (const auto& e : errors) if (!e.empty()) { set_err(h, e); break; }
  return 0;
  GUARD_END(h)
}
int32_t rabe_bsw_keygen_packed(rabe_host* h, const void* pk, const void* msk, const char* const* attributes, const size_t* counts, size_t n_sets,
                               size_t n_items, const uint32_t* item_set, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off) {
  GUARD_BEGIN
  std::vector<std::vector<std::string>> sets(n_sets);
  size_t at = 0;
  for (size_t s = 0; s < n_sets; s++)
    for (size_t k = 0; k < counts[s]; k++) sets[s].push_back(attributes[at++]);
  return bsw::keygen_packed(h->eng, h->rng(), *(const bsw::CpAbePublicKey*)pk, *(const bsw::CpAbeMasterKey*)msk, sets, n_items, item_set, sk_buf, sk_cap,
                            sk_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_bsw_delegate_packed(rabe_host* h, const void* pk, const void* sk, const char* const* attributes, const size_t* counts, size_t n_subsets,
                                 size_t n_items, const uint32_t* item_subset, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off) {
  GUARD_BEGIN
  std::vector<std::vector<std::string>> sets(n_subsets);
  size_t at = 0;
  for (size_t s = 0; s < n_subsets; s++)
    for (size_t k = 0; k < counts[s]; k++) sets[s].push_back(attributes[at++]);
  return bsw::delegate_packed(h->eng, h->rng(), *(const bsw::CpAbePublicKey*)pk, *(const bsw::CpAbeSecretKey*)sk, sets, n_items, item_subset, sk_buf, sk_cap,
                              sk_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_bsw_encrypt_packed(rabe_host* h, const void* pk, const char* const* policies, size_t n_policies, int32_t language, size_t n_items,
                                const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* ct_buf, size_t ct_cap, uint64_t* ct_off) {
  GUARD_BEGIN
  const auto& key = *(const bsw::CpAbePublicKey*)pk;
  const auto pols = strs(policies, n_policies);
  const auto lang = lang_of(language);
  if (!item_policy || !pt_off || !ct_off) throw RabeError("bsw::encrypt_packed: null input");
  return pipeline::produce(h->engines(), h->rng(), n_items, CHUNK_BSW, [&](Engine& eng, size_t lo, size_t hi, Rng& r, uint8_t* out, size_t cap, uint64_t* off) {
    return bsw::encrypt_packed(eng, r, key, pols, lang, hi - lo, item_policy + lo, pt_blob, pt_off + lo, out, cap, off);
  }, ct_buf, ct_cap, ct_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_bsw_decrypt_packed(rabe_host* h, const void* sk, size_t n_items, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, uint32_t flags,
                                int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off) {
  GUARD_BEGIN
  std::vector<std::string> errors;
  const auto& key = *(const bsw::CpAbeSecretKey*)sk;
  const bool trusted = (flags & RABE_PACKED_TRUSTED) != 0;
  if (!pipeline::consume(h->engines(), n_items, CHUNK_BSW, ct_off, ct_len, [&](Engine& eng, size_t lo, size_t hi, int32_t* st, uint8_t* pt, size_t cap, uint64_t* off,
                                                                          std::vector<std::string>* errs) {
        return bsw::decrypt_packed(eng, key, hi - lo, ct_blob, ct_len, ct_off + lo, trusted, st, pt, cap, off, errs);
      }, status, pt_buf, pt_cap, pt_off, &errors))
    return 1;
  for (const auto& e : errors) if (!e.empty()) { set_err(h, e); break; }
  return 0;
  GUARD_END(h)
}
int32_t rabe_lsw_encrypt_packed(rabe_host* h, const void* pk, const char* const* attributes, const size_t* counts, size_t n_sets, size_t n_items,
                                const uint32_t* item_set, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* ct_buf, size_t ct_cap, uint64_t* ct_off) {
  GUARD_BEGIN
  std::vector<std::vector<std::string>> sets(n_sets);
  size_t at = 0;
  for (size_t s = 0; s < n_sets; s++)
    for (size_t k = 0; k < counts[s]; k++) sets[s].push_back(attributes[at++]);
  return lsw::encrypt_packed(h->eng, h->rng(), *(const lsw::KpAbePublicKey*)pk, sets, n_items, item_set, pt_blob, pt_off, ct_buf, ct_cap, ct_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_lsw_keygen_packed(rabe_host* h, const void* pk, const void* msk, const char* const* policies, size_t n_policies, int32_t language, size_t n_items,
                               const uint32_t* item_policy, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off) {
  GUARD_BEGIN
  const auto& key = *(const lsw::KpAbePublicKey*)pk;
  const auto& master = *(const lsw::KpAbeMasterKey*)msk;
  const auto pols = strs(policies, n_policies);
  const auto lang = lang_of(language);
  if (!item_policy || !sk_off) throw RabeError("lsw::keygen_packed: null input");
  return pipeline::produce(h->engines(), h->rng(), n_items, CHUNK_LSW, [&](Engine& eng, size_t lo, size_t hi, Rng& r, uint8_t* out, size_t cap, uint64_t* off) {
    return lsw::keygen_packed(eng, r, key, master, pols, lang, hi - lo, item_policy + lo, out, cap, off);
  }, sk_buf, sk_cap, sk_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_lsw_decrypt_packed(rabe_host* h, const void* ct, size_t n_items, const uint8_t* sk_blob, size_t sk_len, const uint64_t* sk_off, uint32_t flags,
                                int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off) {
  GUARD_BEGIN
  std::vector<std::string> errors;
  const auto& c = *(const lsw::KpAbeCiphertext*)ct;
  const bool trusted = (flags & RABE_PACKED_TRUSTED) != 0;
  if (!pipeline::consume(h->engines(), n_items, CHUNK_LSW, sk_off, sk_len, [&](Engine& eng, size_t lo, size_t hi, int32_t* st, uint8_t* pt, size_t cap, uint64_t* off,
                                                                          std::vector<std::string>* errs) {
        return lsw::decrypt_packed(eng, c, hi - lo, sk_blob, sk_len, sk_off + lo, trusted, st, pt, cap, off, errs);
      }, status, pt_buf, pt_cap, pt_off, &errors))
    return 1;
  for (const auto& e : errors) if (!e.empty()) { set_err(h, e); break; }
  return 0;
  GUARD_END(h)
}
int32_t rabe_aw11_keygen_packed(rabe_host* h, const void* gk, const void* msk, const char* const* gids, const char* const* attributes, const size_t* counts,
                                size_t n_sets, size_t n_items, const uint32_t* item_set, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off) {
  GUARD_BEGIN
  std::vector<std::vector<std::string>> sets(n_sets);
  size_t at = 0;
  for (size_t s = 0; s < n_sets; s++)
    for (size_t k = 0; k < counts[s]; k++) sets[s].push_back(attributes[at++]);
  return aw11::keygen_packed(h->eng, *(const aw11::Aw11GlobalKey*)gk, *(const aw11::Aw11MasterKey*)msk, strs(gids, n_items), sets, n_items, item_set, sk_buf,
                             sk_cap, sk_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_aw11_encrypt_packed(rabe_host* h, const void* gk, const void* const* pks, size_t n_pks, const char* const* policies, size_t n_policies,
                                 int32_t language, size_t n_items, const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off,
                                 uint8_t* ct_buf, size_t ct_cap, uint64_t* ct_off) {
  GUARD_BEGIN
  std::vector<const aw11::Aw11PublicKey*> p;
  for (size_t i = 0; i < n_pks; i++) p.push_back((const aw11::Aw11PublicKey*)pks[i]);
  const auto& g = *(const aw11::Aw11GlobalKey*)gk;
  const auto pols = strs(policies, n_policies);
  const auto lang = lang_of(language);
  if (!item_policy || !pt_off || !ct_off) throw RabeError("aw11::encrypt_packed: null input");
  return pipeline::produce(h->engines(), h->rng(), n_items, CHUNK_AW11, [&](Engine& eng, size_t lo, size_t hi, Rng& r, uint8_t* out, size_t cap, uint64_t* off) {
    return aw11::encrypt_packed(eng, r, g, p, pols, lang, hi - lo, item_policy + lo, pt_blob, pt_off + lo, out, cap, off);
  }, ct_buf, ct_cap, ct_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_aw11_decrypt_packed(rabe_host* h, const void* gk, const void* sk, size_t n_items, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off,
                                 uint32_t flags, int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off) {
  GUARD_BEGIN
  std::vector<std::string> errors;
  const auto& g = *(const aw11::Aw11GlobalKey*)gk;
  const auto& key = *(const aw11::Aw11SecretKey*)sk;
  const bool trusted = (flags & RABE_PACKED_TRUSTED) != 0;
  if (!pipeline::consume(h->engines(), n_items, CHUNK_AW11, ct_off, ct_len, [&](Engine& eng, size_t lo, size_t hi, int32_t* st, uint8_t* pt, size_t cap, uint64_t* off,
                                                                           std::vector<std::string>* errs) {
        return aw11::decrypt_packed(eng, g, key, hi - lo, ct_blob, ct_len, ct_off + lo, trusted, st, pt, cap, off, errs);
      }, status, pt_buf, pt_cap, pt_off, &errors))
    return 1;
  for (const auto& e : errors) if (!e.empty()) { set_err(h, e); break; }
  return 0;
  GUARD_END(h)
}
int32_t rabe_ac17_cp_decrypt_batch(rabe_host* h, size_t n, const void* const* sks, const void* const* cts, int32_t* status,
                                   uint8_t** plaintexts, size_t* lens) {
  GUARD_BEGIN
  std::vector<rabe_ticket> ts(n);
  for (size_t i = 0; i < n; i++) { ts[i].op = rabe_ticket::AC17_DEC; ts[i].a = sks[i]; ts[i].ct = cts[i]; }
  run_tickets_now(h, ts);
  give_plaintexts(h, ts, status, plaintexts, lens);
  return 0;
  GUARD_END(h)
}

int32_t rabe_ac17_kp_keygen(rabe_host* h, const void* msk, const char* policy, int32_t language, void** sk) {
  GUARD_BEGIN
  *sk = new ac17::Ac17KpSecretKey(ac17::kp_keygen(h->eng, h->rng(), *(const ac17::Ac17MasterKey*)msk, policy, lang_of(language)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_ac17_kp_keygen_packed(rabe_host* h, const void* msk, const char* const* policies, size_t n_policies, int32_t language, size_t n_items,
                                   const uint32_t* item_policy, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off) {
  GUARD_BEGIN
  return ac17::kp_keygen_packed(h->eng, h->rng(), *(const ac17::Ac17MasterKey*)msk, strs(policies, n_policies), lang_of(language), n_items, item_policy, sk_buf,
                                sk_cap, sk_off) ? 0 : 1;
  GUARD_END(h)
}
int32_t rabe_ac17_kp_encrypt(rabe_host* h, const void* pk, const char* const* attributes, size_t n, const uint8_t* data, size_t len, void** ct) {
  GUARD_BEGIN
  *ct = new ac17::Ac17KpCiphertext(ac17::kp_encrypt(h->eng, h->rng(), *(const ac17::Ac17PublicKey*)pk, strs(attributes, n), Bytes(data, data + len)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_ac17_kp_decrypt(rabe_host* h, const void* sk, const void* ct, uint8_t** out, size_t* len) {
  GUARD_BEGIN
  return give_bytes(ac17::kp_decrypt(h->eng, *(const ac17::Ac17KpSecretKey*)sk, *(const ac17::Ac17KpCiphertext*)ct), out, len);
  GUARD_END(h)
}
int32_t rabe_ac17_kp_decrypt_gt(rabe_host* h, const void* sk, const void* ct, uint8_t out_gt[384]) {
  GUARD_BEGIN
  Gt g = ac17::kp_decrypt_gt(h->eng, *(const ac17::Ac17KpSecretKey*)sk, *(const ac17::Ac17KpCiphertext*)ct);
  memcpy(out_gt, g.data(), 384);
  return 0;
  GUARD_END(h)
}

int32_t rabe_ac17_kp_encrypt_batch(rabe_host* h, const void* pk, size_t n, const char* const* attributes, const size_t* counts,
                                   const uint8_t* const* datas, const size_t* lens, void** cts) {
  GUARD_BEGIN
  std::vector<std::vector<std::string>> sets;
  size_t pos = 0;
  for (size_t i = 0; i < n; i++) { sets.push_back(strs(attributes + pos, counts[i])); pos += counts[i]; }
  auto r = ac17::kp_encrypt_batch(h->eng, h->rng(), *(const ac17::Ac17PublicKey*)pk, sets, byte_items(datas, lens, n));
  for (size_t i = 0; i < n; i++) cts[i] = new ac17::Ac17KpCiphertext(r[i]);
  return 0;
  GUARD_END(h)
}
int32_t rabe_ac17_kp_decrypt_batch(rabe_host* h, size_t n, const void* const* sks, const void* const* cts, int32_t* status, uint8_t** plaintexts,
                                   size_t* lens) {
  GUARD_BEGIN
  std::vector<const ac17::Ac17KpSecretKey*> s;
  std::vector<const ac17::Ac17KpCiphertext*> c;
  for (size_t i = 0; i < n; i++) { s.push_back((const ac17::Ac17KpSecretKey*)sks[i]); c.push_back((const ac17::Ac17KpCiphertext*)cts[i]); }
  give_results(h, ac17::kp_decrypt_batch(h->eng, s, c), status, plaintexts, lens);
  return 0;
  GUARD_END(h)
}

// ---------------------------------------------------------------- bsw
int32_t rabe_bsw_delegate(rabe_host* h, const void* pk, const void* sk, const char* const* subset, size_t n, void** out_sk) {
  GUARD_BEGIN
  bsw::CpAbeSecretKey out;
  if (!bsw::delegate(h->eng, h->rng(), *(const bsw::CpAbePublicKey*)pk, *(const bsw::CpAbeSecretKey*)sk, strs(subset, n), &out)) return 1;
  *out_sk = new bsw::CpAbeSecretKey(out);
  return 0;
  GUARD_END(h)
}
int32_t rabe_bsw_setup(rabe_host* h, void** pk, void** msk) {
  GUARD_BEGIN
  auto r = bsw::setup(h->eng, h->rng());
  *pk = new bsw::CpAbePublicKey(r.first);
  *msk = new bsw::CpAbeMasterKey(r.second);
  return 0;
  GUARD_END(h)
}
int32_t rabe_bsw_keygen(rabe_host* h, const void* pk, const void* msk, const char* const* attributes, size_t n, void** sk) {
  GUARD_BEGIN
  bsw::CpAbeSecretKey out;
  if (!bsw::keygen(h->eng, h->rng(), *(const bsw::CpAbePublicKey*)pk, *(const bsw::CpAbeMasterKey*)msk, strs(attributes, n), &out)) return 1;
  *sk = new bsw::CpAbeSecretKey(out);
  return 0;
  GUARD_END(h)
}
int32_t rabe_bsw_encrypt(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* pt, size_t len, void** ct) {
  GUARD_BEGIN
  if (h && h->coalesce) {
    rabe_ticket t; t.op = rabe_ticket::BSW_ENC; t.a = pk; t.policy = policy; t.language = language; t.pt.assign(pt, pt + len);
    return queue_call(h, &t, ct, nullptr, nullptr);
  }
  *ct = new bsw::CpAbeCiphertext(bsw::encrypt(h->eng, h->rng(), *(const bsw::CpAbePublicKey*)pk, policy, lang_of(language), Bytes(pt, pt + len)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_bsw_decrypt(rabe_host* h, const void* sk, const void* ct, uint8_t** out, size_t* len) {
  GUARD_BEGIN
  if (h && h->coalesce) {
    rabe_ticket t; t.op = rabe_ticket::BSW_DEC; t.a = sk; t.ct = ct;
    return queue_call(h, &t, nullptr, out, len);
  }
  return give_bytes(bsw::decrypt(h->eng, *(const bsw::CpAbeSecretKey*)sk, *(const bsw::CpAbeCiphertext*)ct), out, len);
  GUARD_END(h)
}
int32_t rabe_bsw_decrypt_gt(rabe_host* h, const void* sk, const void* ct, uint8_t out_gt[384]) {
  GUARD_BEGIN
  Gt g = bsw::decrypt_gt(h->eng, *(const bsw::CpAbeSecretKey*)sk, *(const bsw::CpAbeCiphertext*)ct);
  memcpy(out_gt, g.data(), 384);
  return 0;
  GUARD_END(h)
}

int32_t rabe_bsw_encrypt_batch(rabe_host* h, const void* pk, size_t n, const char* const* policies, int32_t language,
                               const uint8_t* const* plaintexts, const size_t* lens, void** cts) {
  GUARD_BEGIN
  std::vector<rabe_ticket> ts(n);
  for (size_t i = 0; i < n; i++) { ts[i].op = rabe_ticket::BSW_ENC; ts[i].a = pk; ts[i].policy = policies[i]; ts[i].language = language; ts[i].pt.assign(plaintexts[i], plaintexts[i] + lens[i]); }
  run_tickets_now(h, ts);
  return give_objects(h, ts, RABE_BSW_CT, cts);
  GUARD_END(h)
}
int32_t rabe_bsw_decrypt_batch(rabe_host* h, size_t n, const void* const* sks, const void* const* cts, int32_t* status, uint8_t** plaintexts,
                               size_t* lens) {
  GUARD_BEGIN
  std::vector<rabe_ticket> ts(n);
  for (size_t i = 0; i < n; i++) { ts[i].op = rabe_ticket::BSW_DEC; ts[i].a = sks[i]; ts[i].ct = cts[i]; }
  run_tickets_now(h, ts);
  give_plaintexts(h, ts, status, plaintexts, lens);
  return 0;
  GUARD_END(h)
}

// ---------------------------------------------------------------- lsw
int32_t rabe_lsw_setup(rabe_host* h, void** pk, void** msk) {
  GUARD_BEGIN
  auto r = lsw::setup(h->eng, h->rng());
  *pk = new lsw::KpAbePublicKey(r.first);
  *msk = new lsw::KpAbeMasterKey(r.second);
  return 0;
  GUARD_END(h)
}
int32_t rabe_lsw_keygen(rabe_host* h, const void* pk, const void* msk, const char* policy, int32_t language, void** sk) {
  GUARD_BEGIN
  *sk = new lsw::KpAbeSecretKey(lsw::keygen(h->eng, h->rng(), *(const lsw::KpAbePublicKey*)pk, *(const lsw::KpAbeMasterKey*)msk, policy, lang_of(language)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_lsw_encrypt(rabe_host* h, const void* pk, const char* const* attributes, size_t n, const uint8_t* pt, size_t len, void** ct) {
  GUARD_BEGIN
  if (h && h->coalesce) {
    rabe_ticket t; t.op = rabe_ticket::LSW_ENC; t.a = pk; t.attrs = strs(attributes, n); t.pt.assign(pt, pt + len);
    return queue_call(h, &t, ct, nullptr, nullptr);
  }
  *ct = new lsw::KpAbeCiphertext(lsw::encrypt(h->eng, h->rng(), *(const lsw::KpAbePublicKey*)pk, strs(attributes, n), Bytes(pt, pt + len)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_lsw_decrypt(rabe_host* h, const void* sk, const void* ct, uint8_t** out, size_t* len) {
  GUARD_BEGIN
  if (h && h->coalesce) {
    rabe_ticket t; t.op = rabe_ticket::LSW_DEC; t.a = sk; t.ct = ct;
    return queue_call(h, &t, nullptr, out, len);
  }
  return give_bytes(lsw::decrypt(h->eng, *(const lsw::KpAbeSecretKey*)sk, *(const lsw::KpAbeCiphertext*)ct), out, len);
  GUARD_END(h)
}
int32_t rabe_lsw_decrypt_gt(rabe_host* h, const void* sk, const void* ct, uint8_t out_gt[384]) {
  GUARD_BEGIN
  Gt g = lsw::decrypt_gt(h->eng, *(const lsw::KpAbeSecretKey*)sk, *(const lsw::KpAbeCiphertext*)ct);
  memcpy(out_gt, g.data(), 384);
  return 0;
  GUARD_END(h)
}

int32_t rabe_lsw_keygen_batch(rabe_host* h, const void* pk, const void* msk, size_t n, const char* const* policies, int32_t language, void** sks) {
  GUARD_BEGIN
  auto r = lsw::keygen_batch(h->eng, h->rng(), *(const lsw::KpAbePublicKey*)pk, *(const lsw::KpAbeMasterKey*)msk, strs(policies, n), lang_of(language));
  for (size_t i = 0; i < n; i++) sks[i] = new lsw::KpAbeSecretKey(r[i]);
  return 0;
  GUARD_END(h)
}
int32_t rabe_lsw_decrypt_batch(rabe_host* h, size_t n, const void* const* sks, const void* const* cts, int32_t* status, uint8_t** plaintexts,
                               size_t* lens) {
  GUARD_BEGIN
  std::vector<const lsw::KpAbeSecretKey*> s;
  std::vector<const lsw::KpAbeCiphertext*> c;
  for (size_t i = 0; i < n; i++) { s.push_back((const lsw::KpAbeSecretKey*)sks[i]); c.push_back((const lsw::KpAbeCiphertext*)cts[i]); }
  give_results(h, lsw::decrypt_batch(h->eng, s, c), status, plaintexts, lens);
  return 0;
  GUARD_END(h)
}

// ---------------------------------------------------------------- aw11
int32_t rabe_aw11_setup(rabe_host* h, void** gk) {
  GUARD_BEGIN
  *gk = new aw11::Aw11GlobalKey(aw11::setup(h->eng, h->rng()));
  return 0;
  GUARD_END(h)
}
int32_t rabe_aw11_authgen(rabe_host* h, const void* gk, const char* const* attributes, size_t n, void** pk, void** msk) {
  GUARD_BEGIN
  aw11::Aw11PublicKey p;
  aw11::Aw11MasterKey m;
  if (!aw11::authgen(h->eng, h->rng(), *(const aw11::Aw11GlobalKey*)gk, strs(attributes, n), &p, &m)) return 1;
  *pk = new aw11::Aw11PublicKey(p);
  *msk = new aw11::Aw11MasterKey(m);
  return 0;
  GUARD_END(h)
}
int32_t rabe_aw11_keygen(rabe_host* h, const void* gk, const void* msk, const char* name, const char* const* attributes, size_t n, void** sk) {
  GUARD_BEGIN
  *sk = new aw11::Aw11SecretKey(aw11::keygen(h->eng, *(const aw11::Aw11GlobalKey*)gk, *(const aw11::Aw11MasterKey*)msk, name, strs(attributes, n)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_aw11_add_to_attribute(rabe_host* h, const void* gk, const void* msk, const char* attribute, void* sk) {
  GUARD_BEGIN
  aw11::add_to_attribute(h->eng, *(const aw11::Aw11GlobalKey*)gk, *(const aw11::Aw11MasterKey*)msk, attribute, (aw11::Aw11SecretKey*)sk);
  return 0;
  GUARD_END(h)
}
int32_t rabe_aw11_encrypt(rabe_host* h, const void* gk, const void* const* pks, size_t n_pks, const char* policy, int32_t language,
                          const uint8_t* data, size_t len, void** ct) {
  GUARD_BEGIN
  if (h && h->coalesce) {
    rabe_ticket t; t.op = rabe_ticket::AW11_ENC; t.a = gk; t.pks.assign(pks, pks + n_pks); t.policy = policy; t.language = language; t.pt.assign(data, data + len);
    return queue_call(h, &t, ct, nullptr, nullptr);
  }
  std::vector<const aw11::Aw11PublicKey*> v;
  for (size_t i = 0; i < n_pks; i++) v.push_back((const aw11::Aw11PublicKey*)pks[i]);
  *ct = new aw11::Aw11Ciphertext(aw11::encrypt(h->eng, h->rng(), *(const aw11::Aw11GlobalKey*)gk, v, policy, lang_of(language), Bytes(data, data + len)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_aw11_decrypt(rabe_host* h, const void* gk, const void* sk, const void* ct, uint8_t** out, size_t* len) {
  GUARD_BEGIN
  if (h && h->coalesce) {
    rabe_ticket t; t.op = rabe_ticket::AW11_DEC; t.a = gk; t.b = sk; t.ct = ct;
    return queue_call(h, &t, nullptr, out, len);
  }
  return give_bytes(aw11::decrypt(h->eng, *(const aw11::Aw11GlobalKey*)gk, *(const aw11::Aw11SecretKey*)sk, *(const aw11::Aw11Ciphertext*)ct), out, len);
  GUARD_END(h)
}
int32_t rabe_aw11_decrypt_gt(rabe_host* h, const void* gk, const void* sk, const void* ct, uint8_t out_gt[384]) {
  GUARD_BEGIN
  Gt g = aw11::decrypt_gt(h->eng, *(const aw11::Aw11GlobalKey*)gk, *(const aw11::Aw11SecretKey*)sk, *(const aw11::Aw11Ciphertext*)ct);
  memcpy(out_gt, g.data(), 384);
  return 0;
  GUARD_END(h)
}

int32_t rabe_aw11_encrypt_batch(rabe_host* h, const void* gk, const void* const* pks, size_t n_pks, size_t n, const char* const* policies,
                                int32_t language, const uint8_t* const* datas, const size_t* lens, void** cts) {
  GUARD_BEGIN
  std::vector<rabe_ticket> ts(n);
  for (size_t i = 0; i < n; i++) {
    ts[i].op = rabe_ticket::AW11_ENC; ts[i].a = gk; ts[i].pks.assign(pks, pks + n_pks); ts[i].policy = policies[i]; ts[i].language = language;
    ts[i].pt.assign(datas[i], datas[i] + lens[i]);
  }
  run_tickets_now(h, ts);
  return give_objects(h, ts, RABE_AW11_CT, cts);
  GUARD_END(h)
}
int32_t rabe_aw11_decrypt_batch(rabe_host* h, const void* gk, size_t n, const void* const* sks, const void* const* cts, int32_t* status,
                                uint8_t** plaintexts, size_t* lens) {
  GUARD_BEGIN
  std::vector<rabe_ticket> ts(n);
  for (size_t i = 0; i < n; i++) { ts[i].op = rabe_ticket::AW11_DEC; ts[i].a = gk; ts[i].b = sks[i]; ts[i].ct = cts[i]; }
  run_tickets_now(h, ts);
  give_plaintexts(h, ts, status, plaintexts, lens);
  return 0;
  GUARD_END(h)
}

// ---------------------------------------------------------------- bdabe / mke08
int32_t rabe_bdabe_setup(rabe_host* h, void** pk, void** msk) {
  GUARD_BEGIN
  auto r = bdabe::setup(h->eng, h->rng());
  *pk = new bdabe::BdabePublicKey(r.first);
  *msk = new bdabe::BdabeMasterKey(r.second);
  return 0;
  GUARD_END(h)
}
int32_t rabe_bdabe_authgen(rabe_host* h, const void* pk, const void* msk, const char* name, void** ska) {
  GUARD_BEGIN
  *ska = new bdabe::BdabeSecretAuthorityKey(bdabe::authgen(h->eng, h->rng(), *(const bdabe::BdabePublicKey*)pk, *(const bdabe::BdabeMasterKey*)msk, name));
  return 0;
  GUARD_END(h)
}
int32_t rabe_bdabe_keygen(rabe_host* h, const void* pk, const void* ska, const char* name, void** uk) {
  GUARD_BEGIN
  *uk = new bdabe::BdabeUserKey(bdabe::keygen(h->eng, h->rng(), *(const bdabe::BdabePublicKey*)pk, *(const bdabe::BdabeSecretAuthorityKey*)ska, name));
  return 0;
  GUARD_END(h)
}
int32_t rabe_bdabe_request_attribute_pk(rabe_host* h, const void* pk, const void* ska, const char* attribute, void** pka) {
  GUARD_BEGIN
  *pka = new bdabe::BdabePublicAttributeKey(bdabe::request_attribute_pk(h->eng, *(const bdabe::BdabePublicKey*)pk, *(const bdabe::BdabeSecretAuthorityKey*)ska, attribute));
  return 0;
  GUARD_END(h)
}
int32_t rabe_bdabe_request_attribute_sk(rabe_host* h, void* uk, const void* ska, const char* attribute) {
  GUARD_BEGIN
  auto* k = (bdabe::BdabeUserKey*)uk;
  k->sk_a.push_back(bdabe::request_attribute_sk(h->eng, k->pk, *(const bdabe::BdabeSecretAuthorityKey*)ska, attribute));
  return 0;
  GUARD_END(h)
}
int32_t rabe_bdabe_encrypt(rabe_host* h, const void* pk, const void* const* attr_pks, size_t n_pks, const char* policy, int32_t language,
                           const uint8_t* plaintext, size_t len, void** ct) {
  GUARD_BEGIN
  std::vector<const bdabe::BdabePublicAttributeKey*> v;
  for (size_t i = 0; i < n_pks; i++) v.push_back((const bdabe::BdabePublicAttributeKey*)attr_pks[i]);
  *ct = new bdabe::BdabeCiphertext(bdabe::encrypt(h->eng, h->rng(), *(const bdabe::BdabePublicKey*)pk, v, policy, lang_of(language), Bytes(plaintext, plaintext + len)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_bdabe_decrypt(rabe_host* h, const void* uk, const void* ct, uint8_t** out, size_t* len) {
  GUARD_BEGIN
  return give_bytes(bdabe::decrypt(h->eng, *(const bdabe::BdabeUserKey*)uk, *(const bdabe::BdabeCiphertext*)ct), out, len);
  GUARD_END(h)
}
int32_t rabe_bdabe_decrypt_gt(rabe_host* h, const void* uk, const void* ct, uint8_t out_gt[384]) {
  GUARD_BEGIN
  Gt g = bdabe::decrypt_gt(h->eng, *(const bdabe::BdabeUserKey*)uk, *(const bdabe::BdabeCiphertext*)ct);
  memcpy(out_gt, g.data(), 384);
  return 0;
  GUARD_END(h)
}
int32_t rabe_bdabe_decrypt_batch(rabe_host* h, size_t n, const void* const* uks, const void* const* cts, int32_t* status, uint8_t** plaintexts,
                                 size_t* lens) {
  GUARD_BEGIN
  std::vector<const bdabe::BdabeUserKey*> s;
  std::vector<const bdabe::BdabeCiphertext*> c;
  for (size_t i = 0; i < n; i++) { s.push_back((const bdabe::BdabeUserKey*)uks[i]); c.push_back((const bdabe::BdabeCiphertext*)cts[i]); }
  give_results(h, bdabe::decrypt_batch(h->eng, s, c), status, plaintexts, lens);
  return 0;
  GUARD_END(h)
}
int32_t rabe_bdabe_decrypt_packed(rabe_host* h, const void* uk, size_t n_items, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, uint32_t flags,
                                  int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off) {
  GUARD_BEGIN
  return dnf_decrypt_packed<bdabe::BdabeCiphertext, bdabe::BdabeUserKey>(
      h, RABE_BDABE_CT, uk, n_items, ct_blob, ct_len, ct_off, flags, status, pt_buf, pt_cap, pt_off,
      [&](const std::vector<const bdabe::BdabeUserKey*>& s, const std::vector<const bdabe::BdabeCiphertext*>& c) { return bdabe::decrypt_batch(h->eng, s, c); });
  GUARD_END(h)
}
int32_t rabe_mke08_decrypt_packed(rabe_host* h, const void* uk, size_t n_items, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, uint32_t flags,
                                  int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off) {
  GUARD_BEGIN
  return dnf_decrypt_packed<mke08::Mke08Ciphertext, mke08::Mke08UserKey>(
      h, RABE_MKE08_CT, uk, n_items, ct_blob, ct_len, ct_off, flags, status, pt_buf, pt_cap, pt_off,
      [&](const std::vector<const mke08::Mke08UserKey*>& s, const std::vector<const mke08::Mke08Ciphertext*>& c) { return mke08::decrypt_batch(h->eng, s, c); });
  GUARD_END(h)
}
int32_t rabe_mke08_setup(rabe_host* h, void** pk, void** msk) {
  GUARD_BEGIN
  auto r = mke08::setup(h->eng, h->rng());
  *pk = new mke08::Mke08PublicKey(r.first);
  *msk = new mke08::Mke08MasterKey(r.second);
  return 0;
  GUARD_END(h)
}
int32_t rabe_mke08_keygen(rabe_host* h, const void* pk, const void* msk, const char* name, void** uk) {
  GUARD_BEGIN
  *uk = new mke08::Mke08UserKey(mke08::keygen(h->eng, h->rng(), *(const mke08::Mke08PublicKey*)pk, *(const mke08::Mke08MasterKey*)msk, name));
  return 0;
  GUARD_END(h)
}
int32_t rabe_mke08_authgen(rabe_host* h, const char* name, void** ska) {
  GUARD_BEGIN
  *ska = new mke08::Mke08SecretAuthorityKey(mke08::authgen(h->rng(), name));
  return 0;
  GUARD_END(h)
}
int32_t rabe_mke08_request_authority_pk(rabe_host* h, const void* pk, const char* attribute, const void* ska, void** pka) {
  GUARD_BEGIN
  *pka = new mke08::Mke08PublicAttributeKey(mke08::request_authority_pk(h->eng, *(const mke08::Mke08PublicKey*)pk, attribute, *(const mke08::Mke08SecretAuthorityKey*)ska));
  return 0;
  GUARD_END(h)
}
int32_t rabe_mke08_request_authority_sk(rabe_host* h, void* uk, const char* attribute, const void* ska) {
  GUARD_BEGIN
  auto* k = (mke08::Mke08UserKey*)uk;
  k->sk_a.push_back(mke08::request_authority_sk(h->eng, k->pk, attribute, *(const mke08::Mke08SecretAuthorityKey*)ska));
  return 0;
  GUARD_END(h)
}
int32_t rabe_mke08_encrypt(rabe_host* h, const void* pk, const void* const* attr_pks, size_t n_pks, const char* policy, int32_t language,
                           const uint8_t* plaintext, size_t len, void** ct) {
  GUARD_BEGIN
  std::vector<const mke08::Mke08PublicAttributeKey*> v;
  for (size_t i = 0; i < n_pks; i++) v.push_back((const mke08::Mke08PublicAttributeKey*)attr_pks[i]);
  *ct = new mke08::Mke08Ciphertext(mke08::encrypt(h->eng, h->rng(), *(const mke08::Mke08PublicKey*)pk, v, policy, lang_of(language), Bytes(plaintext, plaintext + len)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_mke08_decrypt(rabe_host* h, const void* uk, const void* ct, uint8_t** out, size_t* len) {
  GUARD_BEGIN
  return give_bytes(mke08::decrypt(h->eng, *(const mke08::Mke08UserKey*)uk, *(const mke08::Mke08Ciphertext*)ct), out, len);
  GUARD_END(h)
}
int32_t rabe_mke08_decrypt_gt(rabe_host* h, const void* uk, const void* ct, uint8_t out_gt[384]) {
  GUARD_BEGIN
  Gt g = mke08::decrypt_gt(h->eng, *(const mke08::Mke08UserKey*)uk, *(const mke08::Mke08Ciphertext*)ct);
  memcpy(out_gt, g.data(), 384);
  return 0;
  GUARD_END(h)
}
int32_t rabe_mke08_decrypt_batch(rabe_host* h, size_t n, const void* const* uks, const void* const* cts, int32_t* status, uint8_t** plaintexts,
                                 size_t* lens) {
  GUARD_BEGIN
  std::vector<const mke08::Mke08UserKey*> s;
  std::vector<const mke08::Mke08Ciphertext*> c;
  for (size_t i = 0; i < n; i++) { s.push_back((const mke08::Mke08UserKey*)uks[i]); c.push_back((const mke08::Mke08Ciphertext*)cts[i]); }
  give_results(h, mke08::decrypt_batch(h->eng, s, c), status, plaintexts, lens);
  return 0;
  GUARD_END(h)
}

// ---------------------------------------------------------------- ghw11
int32_t rabe_ghw11_setup(rabe_host* h, void** pk, void** msk) {
  GUARD_BEGIN
  auto r = ghw11::setup(h->eng, h->rng());
  *pk = new ghw11::Ghw11PublicKey(r.first);
  *msk = new ghw11::Ghw11MasterKey(r.second);
  return 0;
  GUARD_END(h)
}
int32_t rabe_ghw11_keygen(rabe_host* h, const void* pk, const void* msk, const char* const* attributes, size_t n, void** sk) {
  GUARD_BEGIN
  ghw11::Ghw11SecretKey out;
  if (!ghw11::keygen(h->eng, h->rng(), *(const ghw11::Ghw11PublicKey*)pk, *(const ghw11::Ghw11MasterKey*)msk, strs(attributes, n), &out)) return 1;
  *sk = new ghw11::Ghw11SecretKey(out);
  return 0;
  GUARD_END(h)
}
int32_t rabe_ghw11_tkgen(rabe_host* h, const void* sk, void** tk, void** rk) {
  GUARD_BEGIN
  auto r = ghw11::tkgen(h->eng, h->rng(), *(const ghw11::Ghw11SecretKey*)sk);
  *tk = new ghw11::Ghw11TransformKey(r.first);
  *rk = new ghw11::Ghw11RetrieveKey(r.second);
  return 0;
  GUARD_END(h)
}
int32_t rabe_ghw11_encrypt(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* plaintext, size_t len, void** ct) {
  GUARD_BEGIN
  *ct = new ghw11::Ghw11Ciphertext(ghw11::encrypt(h->eng, h->rng(), *(const ghw11::Ghw11PublicKey*)pk, policy, lang_of(language), Bytes(plaintext, plaintext + len)));
  return 0;
  GUARD_END(h)
}
int32_t rabe_ghw11_transform(rabe_host* h, const void* ct, const void* tk, void** tct) {
  GUARD_BEGIN
  *tct = new ghw11::Ghw11TransformCiphertext(ghw11::transform(h->eng, *(const ghw11::Ghw11Ciphertext*)ct, *(const ghw11::Ghw11TransformKey*)tk));
  return 0;
  GUARD_END(h)
}
int32_t rabe_ghw11_transform_batch(rabe_host* h, size_t n, const void* const* cts, const void* const* tks, int32_t* status, void** tcts) {
  GUARD_BEGIN
  std::vector<const ghw11::Ghw11Ciphertext*> c;
  std::vector<const ghw11::Ghw11TransformKey*> t;
  for (size_t i = 0; i < n; i++) { c.push_back((const ghw11::Ghw11Ciphertext*)cts[i]); t.push_back((const ghw11::Ghw11TransformKey*)tks[i]); }
  std::vector<std::string> errors;
  auto r = ghw11::transform_batch(h->eng, c, t, &errors);
  for (size_t i = 0; i < n; i++) {
    status[i] = errors[i].empty() ? 0 : -1;
    tcts[i] = errors[i].empty() ? new ghw11::Ghw11TransformCiphertext(r[i]) : nullptr;
    if (!errors[i].empty()) set_err(h, errors[i]);
  }
  return 0;
  GUARD_END(h)
}
int32_t rabe_ghw11_transform_packed(rabe_host* h, const void* tk, size_t n_items, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, uint32_t flags,
                                    int32_t* status, uint8_t* tct_buf, size_t tct_cap) {
  GUARD_BEGIN
  std::vector<std::string> errors;
  if (!ghw11::transform_packed(h->eng, *(const ghw11::Ghw11TransformKey*)tk, n_items, ct_blob, ct_len, ct_off, (flags & RABE_PACKED_TRUSTED) != 0, status,
                               tct_buf, tct_cap, &errors))
    return 1;
  for (const auto& e : errors) if (!e.empty()) { set_err(h, e); break; }
  return 0;
  GUARD_END(h)
}
int32_t rabe_ghw11_decrypt_out(rabe_host* h, const void* tct, const void* rk, const void* ct, uint8_t** plaintext, size_t* len) {
  GUARD_BEGIN
  return give_bytes(ghw11::decrypt_out(h->eng, *(const ghw11::Ghw11TransformCiphertext*)tct, *(const ghw11::Ghw11RetrieveKey*)rk,
                                       ((const ghw11::Ghw11Ciphertext*)ct)->data), plaintext, len);
  GUARD_END(h)
}
int32_t rabe_ghw11_decrypt_out_gt(rabe_host* h, const void* tct, const void* rk, uint8_t out_gt[384]) {
  GUARD_BEGIN
  Gt g = ghw11::decrypt_out_gt(h->eng, *(const ghw11::Ghw11TransformCiphertext*)tct, *(const ghw11::Ghw11RetrieveKey*)rk);
  memcpy(out_gt, g.data(), 384);
  return 0;
  GUARD_END(h)
}

// sha3_hash_fr (src/utils/hash/mod.rs:23-31) and, for the tests, the two 512-bit reductions side by side
int32_t rabe_hash_fr(const char* label, uint8_t out_le32[32]) {
  Fr f = sha3_hash_fr(label);
  memcpy(out_le32, f.l, 32);
  return 0;
}
int32_t rabe_fr_reduce512(const uint8_t in_le64[64], uint8_t out_fast[32], uint8_t out_division[32]) {
  uint64_t t[8], a[4], b[4];
  memcpy(t, in_le64, 64);
  frdetail::reduce512_fast(a, t);
  frdetail::reduce512(b, t);
  memcpy(out_fast, a, 32);
  memcpy(out_division, b, 32);
  return 0;
}

// ---------------------------------------------------------------- policy utilities (host only)
static std::string jstr(const std::string& s) {
  std::string o = "\"";
  for (char c : s) { if (c == '"' || c == '\\') o += '\\'; o += c; }
  return o + "\"";
}
static std::string hex32(const Fr& f) {
  static const char* d = "0123456789abcdef";
  std::string o;
  const uint8_t* p = (const uint8_t*)f.l;
  for (int i = 0; i < 32; i++) { o += d[p[i] >> 4]; o += d[p[i] & 15]; }
  return o;
}
int32_t rabe_policy_parse(const char* policy, int32_t language, int32_t out_language, char** out) {
  GUARD_BEGIN
  return give_text(serialize_policy(parse_policy(policy, lang_of(language)), lang_of(out_language)), out);
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_policy_msp(const char* policy, int32_t language, char** out) {
  GUARD_BEGIN
  AbePolicy m = calculate_msp(parse_policy(policy, lang_of(language)));
  std::string s = "{\"m\": [";
  for (size_t i = 0; i < m.m.size(); i++) {
    s += i ? ", [" : "[";
    for (size_t j = 0; j < m.m[i].size(); j++) s += (j ? ", " : "") + std::to_string((int)m.m[i][j]);
    s += "]";
  }
  s += "], \"pi\": [";
  for (size_t i = 0; i < m.pi.size(); i++) s += (i ? ", " : "") + jstr(m.pi[i]);
  s += "], \"c\": " + std::to_string(m.c) + "}";
  return give_text(s, out);
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_policy_pruned(const char* policy, int32_t language, const char* const* attributes, size_t n, char** out) {
  GUARD_BEGIN
  PrunedList l;
  bool ok = calc_pruned(strs(attributes, n), parse_policy(policy, lang_of(language)), &l);
  std::string s = std::string("{\"match\": ") + (ok ? "true" : "false") + ", \"list\": [";
  if (ok) for (size_t i = 0; i < l.size(); i++) s += std::string(i ? ", " : "") + "[" + jstr(l[i].first) + ", " + jstr(l[i].second) + "]";
  s += "]}";
  return give_text(s, out);
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_policy_traverse(const char* policy, int32_t language, const char* const* attributes, size_t n, int32_t* result) {
  GUARD_BEGIN
  *result = traverse_policy(strs(attributes, n), parse_policy(policy, lang_of(language))) ? 1 : 0;
  return 0;
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_policy_in_dnf(const char* policy, int32_t language, int32_t* result) {
  GUARD_BEGIN
  *result = policy_in_dnf(parse_policy(policy, lang_of(language))) ? 1 : 0;
  return 0;
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_policy_dnf_terms(const char* policy, int32_t language, const char* const* key_attrs, size_t n, char** out) {
  GUARD_BEGIN
  std::vector<DnfTerm> terms;
  if (!json_to_dnf(parse_policy(policy, lang_of(language)), strs(key_attrs, n), &terms)) throw RabeError("Error in json_to_dnf: could not parse policy as DNF");
  std::string s = "[";
  for (size_t t = 0; t < terms.size(); t++) {
    s += std::string(t ? ", " : "") + "[";
    for (size_t i = 0; i < terms[t].attrs.size(); i++) s += std::string(i ? ", " : "") + jstr(terms[t].attrs[i]);
    s += "]";
  }
  return give_text(s + "]", out);
  GUARD_END((rabe_host*)nullptr)
}
static std::string named_json(const NamedFr& v) {
  std::string s = "[";
  for (size_t i = 0; i < v.size(); i++) s += std::string(i ? ", " : "") + "[" + jstr(v[i].first) + ", \"" + hex32(v[i].second) + "\"]";
  return s + "]";
}
int32_t rabe_policy_shares(const char* policy, int32_t language, const uint8_t secret[32], const uint8_t* tape, size_t n_tape, char** out) {
  GUARD_BEGIN
  std::vector<Fr> t(n_tape);
  for (size_t i = 0; i < n_tape; i++) memcpy(t[i].l, tape + 32 * i, 32);
  TapeRng rng(t);
  NamedFr sh;
  gen_shares_policy(fr_from_bytes(secret), parse_policy(policy, lang_of(language)), rng, &sh);
  return give_text(named_json(sh), out);
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_policy_coeffs(const char* policy, int32_t language, char** out) {
  GUARD_BEGIN
  NamedFr c;
  calc_coefficients(parse_policy(policy, lang_of(language)), fr_one(), &c);
  return give_text(named_json(c), out);
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_encrypt_symmetric(const uint8_t gt[384], const uint8_t* data, size_t len, const uint8_t nonce[12], uint8_t** out, size_t* out_len) {
  GUARD_BEGIN
  return give_bytes(encrypt_symmetric(gt, data, len, nonce), out, out_len);
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_decrypt_symmetric(const uint8_t gt[384], const uint8_t* data, size_t len, uint8_t** out, size_t* out_len) {
  GUARD_BEGIN
  Bytes pt;
  if (!decrypt_symmetric(gt, data, len, &pt)) { set_err(nullptr, "decryption error: aead::Error"); return -1; }
  return give_bytes(pt, out, out_len);
  GUARD_END((rabe_host*)nullptr)
}
// raw primitives behind the KEM/DEM step, exported so that public known-answer vectors can pin them (FIPS-202, the GCM
// specification's 256-bit-key test cases): tests/test_host_policy.py
int32_t rabe_sha3_256(const uint8_t* data, size_t len, uint8_t out[32]) {
  GUARD_BEGIN
  sha3_256(data, len, out);
  return 0;
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_aes256_gcm_encrypt(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* data, size_t len, uint8_t** out, size_t* out_len) {
  GUARD_BEGIN
  return give_bytes(aes256_gcm_encrypt(key, nonce, data, len), out, out_len);       // ciphertext || tag
  GUARD_END((rabe_host*)nullptr)
}
int32_t rabe_aes256_gcm_decrypt(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* data, size_t len, uint8_t** out, size_t* out_len) {
  GUARD_BEGIN
  Bytes pt;
  if (!aes256_gcm_decrypt(key, nonce, data, len, &pt)) { set_err(nullptr, "decryption error: aead::Error"); return -1; }
  return give_bytes(pt, out, out_len);
  GUARD_END((rabe_host*)nullptr)
}

}  // extern "C"
