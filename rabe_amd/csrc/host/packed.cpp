// Packed batch entry points of bsw / lsw / aw11 (include/rabe_host.h: rabe_{bsw,lsw,aw11}_*_packed): n independent calls of the
// reference's scheme functions -- bsw::encrypt / decrypt (src/schemes/bsw/mod.rs:217-318), lsw::keygen / decrypt
// (lsw/mod.rs:121-290), aw11::encrypt / decrypt (aw11/mod.rs:241-366) -- with ONE blob of canonical records + offsets on each side
// of the boundary, fed to the device-resident Level B paths (rhip_{bsw,lsw,aw11}_*_batch, include/rabe_hip.h) instead of the
// per-object pairing jobs.  What stays on the host is what the reference does with strings and bytes: policy parsing, flattening
// the tree into the index tables the share kernels walk, traverse / calc_pruned / calc_coefficients per distinct policy, record
// assembly, KDF + AES-GCM.  Records are the byte form rabe_obj_serialize gives the corresponding struct (host_abi.cpp), so packed
// and object APIs interoperate.
#include "schemes.h"
#include "records.h"

#include <chrono>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <unordered_map>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace rabe {
using namespace host;
PolicyNode parse_or_error(const std::string& policy, PolicyLanguage lang);          // schemes.cpp
void parallel_for(size_t n, const std::function<void(size_t)>& fn);                // schemes.cpp

namespace schemes {
namespace {

inline void put_u32(uint8_t* p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i)); }
inline uint32_t get_u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct Timer {
  bool on;
  const char* what;
  std::chrono::steady_clock::time_point t0;
  explicit Timer(const char* w) : on(getenv("RABE_HOST_TIMING") != nullptr), what(w), t0(std::chrono::steady_clock::now()) {}
  void lap(const char* stage) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[host-timing] %s: %s %.1f ms\n", what, stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// A policy text as the device-level paths want it (include/rabe_hip.h, "Flattened policy trees"): leaves in DFS order -- the order
// gen_shares_policy emits shares (src/utils/secretsharing/mod.rs:82-122) -- each with its root-to-leaf path of (gate, 1-based child
// number); gates in DFS pre-order with their threshold and the offset of their k - 1 coefficient draws (:128-134); per leaf the
// share's name `name_col` (node_index, :74-76), Fr(SHA3(remove_index(name_col))) (what the schemes hash, e.g. bsw/mod.rs:239) and
// the reconstruction coefficient (calc_coefficients, :9-57).  Cached across calls by (language, text).
struct FlatPolicy {
  PolicyNode tree;
  std::vector<std::string> leaf_name, leaf_name_col;
  std::vector<uint32_t> path_off{0}, path_gate, path_x, gate_k, gate_coef_off;
  uint32_t n_coef = 0;
  std::vector<Fr> leaf_hash, leaf_coeff;
  bool has_negative = false;
  size_t names_bytes = 0;          // sum of name_col lengths (record sizes)
};
void flatten_walk(FlatPolicy& f, const PolicyNode& n, std::vector<std::pair<uint32_t, uint32_t>>& path) {
  if (n.type == PolicyType::Leaf) {
    f.leaf_name.push_back(n.name);
    f.leaf_name_col.push_back(node_index(n));
    for (auto& pe : path) { f.path_gate.push_back(pe.first); f.path_x.push_back(pe.second); }
    f.path_off.push_back((uint32_t)f.path_gate.size());
    return;
  }
  if (n.children.size() < 2)
    throw PolicyPanic(n.type == PolicyType::And ? "Error: Invalid policy (AND with just a single child)." : "Error: Invalid policy (OR with just a single child).");
  const uint32_t g = (uint32_t)f.gate_k.size();
  const uint32_t k = n.type == PolicyType::And ? (uint32_t)n.children.size() : 1u;
  f.gate_k.push_back(k);
  f.gate_coef_off.push_back(f.n_coef);
  f.n_coef += k - 1;
  for (size_t i = 0; i < n.children.size(); i++) {
    path.push_back({g, (uint32_t)i + 1});
    flatten_walk(f, n.children[i], path);
    path.pop_back();
  }
}
std::shared_ptr<const FlatPolicy> flat_policy(const std::string& pol, PolicyLanguage language) {
  static std::mutex mu;
  static std::map<std::pair<int, std::string>, std::shared_ptr<const FlatPolicy>> cache;
  const auto key = std::make_pair((int)language, pol);
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
  }
  auto f = std::make_shared<FlatPolicy>();
  f->tree = parse_or_error(pol, language);
  std::vector<std::pair<uint32_t, uint32_t>> path;
  flatten_walk(*f, f->tree, path);
  NamedFr coeff;
  calc_coefficients(f->tree, fr_one(), &coeff);
  for (size_t i = 0; i < f->leaf_name_col.size(); i++) {
    f->leaf_hash.push_back(sha3_hash_fr(remove_index(f->leaf_name_col[i])));
    f->leaf_coeff.push_back(coeff[i].second);
    f->has_negative = f->has_negative || is_negative(f->leaf_name[i]);
    f->names_bytes += f->leaf_name_col[i].size();
  }
  std::lock_guard<std::mutex> g(mu);
  if (cache.size() >= 1024) cache.clear();
  cache[key] = f;
  return f;
}

// uploads that tolerate empty vectors (the device entry points are never handed a null array)
DBuf up32(Engine& eng, const std::vector<uint32_t>& v) {
  static const uint32_t zero = 0;
  return v.empty() ? DBuf(&eng, &zero, 4) : DBuf(&eng, v.data(), v.size() * 4);
}
DBuf up_bytes(Engine& eng, const std::vector<uint8_t>& v) {
  static const uint8_t zero[32] = {0};
  return v.empty() ? DBuf(&eng, zero, 32) : DBuf(&eng, v.data(), v.size());
}

// the concatenated tables of a call's distinct policies, on the device
struct DevTrees {
  std::vector<uint32_t> first_leaf, first_gate;
  DBuf path_off, path_gate, path_x, gate_k, gate_coef_off, leaf_hash;
  DevTrees(Engine& eng, const std::vector<std::shared_ptr<const FlatPolicy>>& pols) {
    std::vector<uint32_t> po{0}, pg, px, gk, gc;
    std::vector<Fr> lh;
    for (const auto& f : pols) {
      first_leaf.push_back((uint32_t)lh.size());
      first_gate.push_back((uint32_t)gk.size());
      const uint32_t base = (uint32_t)pg.size();
      for (size_t i = 1; i < f->path_off.size(); i++) po.push_back(base + f->path_off[i]);
      pg.insert(pg.end(), f->path_gate.begin(), f->path_gate.end());
      px.insert(px.end(), f->path_x.begin(), f->path_x.end());
      gk.insert(gk.end(), f->gate_k.begin(), f->gate_k.end());
      gc.insert(gc.end(), f->gate_coef_off.begin(), f->gate_coef_off.end());
      lh.insert(lh.end(), f->leaf_hash.begin(), f->leaf_hash.end());
    }
    path_off = up32(eng, po); path_gate = up32(eng, pg); path_x = up32(eng, px); gate_k = up32(eng, gk); gate_coef_off = up32(eng, gc);
    leaf_hash = up_bytes(eng, flatten_fr(lh));
  }
};

// randomness of a batch in the reference's per-call draw order, item after item; OS randomness has no order, so blocks of items
// may then draw on their own sources in parallel
template <class DRAW>
void draw_items(Rng& rng, size_t n, DRAW draw) {
  struct Turn { Rng& r; explicit Turn(Rng& x) : r(x) { r.begin_draws(); } ~Turn() { r.end_draws(); } } turn(rng);
  if (rng.unordered() && n >= 256) {
    // an item of these schemes draws hundreds of values (one per gate coefficient and leaf): small blocks, so that every core draws
    const size_t per = 16, blocks = (n + per - 1) / per;
    parallel_for(blocks, [&](size_t b) {
      BatchRng local;
      for (size_t i = b * per; i < n && i < (b + 1) * per; i++) draw(local, i);
    });
  } else {
    for (size_t i = 0; i < n; i++) draw(rng, i);
  }
}

// bounds of the records of an untrusted blob: monotone, inside [0, len); span = total size of the well-formed ones
uint64_t check_offsets(size_t n, const uint64_t* off, size_t len, std::vector<std::string>* errors) {
  uint64_t span = 0;
  for (size_t i = 0; i < n; i++) {
    if (off[i] > off[i + 1] || off[i + 1] > len) (*errors)[i] = "deserialize: record offsets are not monotone inside the blob";
    else span += off[i + 1] - off[i];
  }
  return span;
}
struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  void need(size_t k) const { if ((size_t)(end - p) < k) throw RabeError("deserialize: truncated input"); }
  uint32_t u32() { need(4); uint32_t v = get_u32(p); p += 4; return v; }
  const uint8_t* raw(size_t k) { need(k); const uint8_t* q = p; p += k; return q; }
  std::pair<const char*, uint32_t> str() { uint32_t l = u32(); return {(const char*)raw(l), l}; }
};
inline bool same(const std::pair<const char*, uint32_t>& a, const std::string& b) { return a.second == b.size() && memcmp(a.first, b.data(), b.size()) == 0; }

// The plans of one call by (language, policy text), looked up WITHOUT copying the text and without serialising the readers: every item
// of a batch asks for its policy's plan, the texts are kilobytes (a 200-leaf policy: ~10 KB), and a mutex around a std::map of strings
// made the parse stage of a 4096-item call 8 ms long.  Misses go through the caller's (locking) maker and are remembered here.
template <class Plan>
class FastPlans {
 public:
  std::shared_ptr<Plan> find(const char* txt, size_t len, PolicyLanguage lang) {
    const uint64_t h = hash(txt, len, lang);
    std::shared_lock<std::shared_mutex> g(mu_);
    auto it = tab_.find(h);
    if (it != tab_.end())
      for (const auto& e : it->second) if (e.lang == lang && e.text.size() == len && memcmp(e.text.data(), txt, len) == 0) return e.plan;
    return nullptr;
  }
  void put(const char* txt, size_t len, PolicyLanguage lang, const std::shared_ptr<Plan>& plan) {
    const uint64_t h = hash(txt, len, lang);
    std::unique_lock<std::shared_mutex> g(mu_);
    auto& bucket = tab_[h];
    for (const auto& e : bucket) if (e.lang == lang && e.text.size() == len && memcmp(e.text.data(), txt, len) == 0) return;
    bucket.push_back(Entry{std::string(txt, len), lang, plan});
  }
 private:
  struct Entry { std::string text; PolicyLanguage lang; std::shared_ptr<Plan> plan; };
  static uint64_t hash(const char* txt, size_t len, PolicyLanguage lang) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)lang;
    for (size_t i = 0; i + 8 <= len; i += 8) { uint64_t w; memcpy(&w, txt + i, 8); h = (h ^ w) * 1099511628211ull; }
    for (size_t i = len & ~(size_t)7; i < len; i++) h = (h ^ (uint8_t)txt[i]) * 1099511628211ull;
    return h;
  }
  std::shared_mutex mu_;
  std::unordered_map<uint64_t, std::vector<Entry>> tab_;
};

// where an item's sealed plaintext sits in the caller's blob (opened on the device: records.h, open_sealed_records)
struct Sealed { const uint8_t* p = nullptr; uint32_t len = 0; };

}  // namespace

// ================================================================================================================= AC17 keygen
namespace ac17 {
namespace {
struct MskTables { rhip_g1_table* g = nullptr; rhip_g2_table* h = nullptr; };
void* make_msk_tables(Engine& eng, const void* arg) {
  const Ac17MasterKey& msk = *(const Ac17MasterKey*)arg;
  std::unique_ptr<MskTables> t(new MskTables());
  eng.check(rhip_g1_table_create(eng.ctx(), (const rhip_g1*)msk.g.data(), &t->g), "rhip_g1_table_create");
  int32_t rc = rhip_g1_table_add_w16(eng.ctx(), t->g);
  if (!rc) rc = rhip_g2_table_create(eng.ctx(), (const rhip_g2*)msk.h.data(), &t->h);
  if (!rc) rc = rhip_g2_table_add_w16(eng.ctx(), t->h);
  if (rc) { rhip_g1_table_destroy(t->g); if (t->h) rhip_g2_table_destroy(t->h); eng.check(rc, "master-key tables"); }
  return t.release();
}
void destroy_msk_tables(void* h) {
  MskTables* t = (MskTables*)h;
  rhip_g1_table_destroy(t->g);
  rhip_g2_table_destroy(t->h);
  delete t;
}
}  // namespace
// window tables of a master key's g and h, built on first use and kept (cp_keygen, cp_keygen_packed)
void msk_tables(Engine& eng, const Ac17MasterKey& msk, rhip_g1_table** g, rhip_g2_table** h) {
  std::string key((const char*)msk.g.data(), 64);
  key.append((const char*)msk.h.data(), 128);
  const MskTables* tb = (const MskTables*)eng.aux("ac17_msk_tables", key, make_msk_tables, &msk, destroy_msk_tables, 4);
  *g = tb->g;
  *h = tb->h;
}

// n calls of ac17::cp_keygen (ac17/mod.rs:191-264) under one master key -- a key authority issuing keys in bulk.  Item i gets the
// attribute list sets[item_set[i]]; draw order per item as in the reference: r0, r1, sigma per attribute (list order), sigma'.
// Record = Ac17CpSecretKey: attribute strings, k_0[3], rows (name, k[3]), k_p[3].  Items that share a list run as one launch of the
// Level B kernels (the label hashes of a list are computed once); the window tables of the master key's g and h are kept across calls.
bool cp_keygen_packed(Engine& eng, Rng& rng, const Ac17MasterKey& msk, const std::vector<std::vector<std::string>>& sets, size_t n,
                      const uint32_t* item_set, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  Timer tm("ac17::cp_keygen_packed");
  Engine::ArenaScope arena(eng);
  eng.scrub_when_done();          // master-key-derived scalars pass through the staging buffers
  if (msk.a.size() != 2 || msk.b.size() != 2 || msk.g_k.size() != 3) throw RabeError("malformed Ac17MasterKey: a, b must have 2 and g_k 3 elements");
  if (n && (!item_set || !out_off)) throw RabeError("cp_keygen_packed: null input");
  std::vector<size_t> fixed(sets.size());
  for (size_t s = 0; s < sets.size(); s++) {
    if (sets[s].empty()) throw RabeError("empty attributes!");
    fixed[s] = 4 + 4 + 384 + 4 + 4 + 192;
    for (const auto& a : sets[s]) fixed[s] += (4 + a.size()) + (4 + a.size() + 4 + 192);
  }
  for (size_t i = 0; i < n; i++) if (item_set[i] >= sets.size()) throw RabeError("cp_keygen_packed: item_set out of range");
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + fixed[item_set[i]];
  if (!out_buf || out_cap < out_off[n]) return false;
  if (!n) return true;
  // items grouped by attribute list (stable: the order inside a group is the item order)
  std::vector<std::vector<size_t>> members(sets.size());
  for (size_t i = 0; i < n; i++) members[item_set[i]].push_back(i);
  std::vector<size_t> slot(n), sig_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) sig_off[i + 1] = sig_off[i] + sets[item_set[i]].size();
  // randomness, item after item: r0, r1, sigma_y ..., sigma'
  uint8_t* h_r = eng.pinned(0, n * 96 + sig_off[n] * 32 + 32);
  uint8_t* h_sp = h_r + n * 64;
  uint8_t* h_sig = h_sp + n * 32;
  draw_items(rng, n, [&](Rng& r, size_t i) {
    Fr r0 = r.next_fr(), r1 = r.next_fr();
    memcpy(h_r + 64 * i, r0.l, 32);
    memcpy(h_r + 64 * i + 32, r1.l, 32);
    for (size_t y = sig_off[i]; y < sig_off[i + 1]; y++) { Fr sg = r.next_fr(); memcpy(h_sig + 32 * y, sg.l, 32); }
    Fr sp = r.next_fr();
    memcpy(h_sp + 32 * i, sp.l, 32);
  });
  tm.lap("draws");
  MskTables tables;
  msk_tables(eng, msk, &tables.g, &tables.h);
  const MskTables* tb = &tables;
  std::vector<Fr> H01;
  for (int l = 0; l < 3; l++)
    for (int t = 0; t < 2; t++) H01.push_back(sha3_hash_fr(std::string("01") + std::to_string(l) + std::to_string(t)));
  std::vector<Fr> a_inv(2);
  for (int t = 0; t < 2; t++)
    if (!fr_inv(msk.a[t], &a_inv[t])) throw std::runtime_error("called `Option::unwrap()` on a `None` value (Fr::inverse of zero)");
  DBuf dgk = up_bytes(eng, flatten(msk.g_k)), da = up_bytes(eng, flatten_fr(a_inv)), db = up_bytes(eng, flatten_fr(msk.b)), dH01 = up_bytes(eng, flatten_fr(H01));
  rhip_ctx* cx = eng.ctx();
  // per group: gather its items' randomness into contiguous staging, launch, fetch
  struct Group { size_t first_out; size_t cnt; size_t n_attr; };
  std::vector<Group> groups(sets.size());
  size_t tot_items = 0, tot_rows = 0;
  for (size_t s = 0; s < sets.size(); s++) { groups[s] = {tot_items, members[s].size(), sets[s].size()}; tot_items += members[s].size(); tot_rows += members[s].size() * sets[s].size(); }
  uint8_t* g_r = eng.pinned(1, tot_items * 96 + tot_rows * 32 + 32);              // r | sigma' | sigma, group-major
  uint8_t* g_sp = g_r + tot_items * 64;
  uint8_t* g_sig = g_sp + tot_items * 32;
  std::vector<size_t> grp_row0(sets.size() + 1, 0);
  for (size_t s = 0; s < sets.size(); s++) grp_row0[s + 1] = grp_row0[s] + members[s].size() * sets[s].size();
  for (size_t s = 0; s < sets.size(); s++)
    parallel_for(members[s].size(), [&](size_t q) {
      const size_t i = members[s][q], o = groups[s].first_out + q;
      slot[i] = o;
      memcpy(g_r + 64 * o, h_r + 64 * i, 64);
      memcpy(g_sp + 32 * o, h_sp + 32 * i, 32);
      memcpy(g_sig + 32 * (grp_row0[s] + q * sets[s].size()), h_sig + 32 * sig_off[i], 32 * sets[s].size());
    });
  DBuf d_r(&eng, tot_items * 64), d_sp(&eng, tot_items * 32), d_sig(&eng, tot_rows * 32 + 4), d_k0(&eng, tot_items * 384), d_k(&eng, tot_rows * 192 + 4),
      d_kp(&eng, tot_items * 192);
  eng.check(rhip_upload_async(cx, d_r.ptr(), g_r, tot_items * 64), "upload");
  eng.check(rhip_upload_async(cx, d_sp.ptr(), g_sp, tot_items * 32), "upload");
  eng.check(rhip_upload_async(cx, d_sig.ptr(), g_sig, tot_rows * 32), "upload");
  std::vector<DBuf> dH;
  for (size_t s = 0; s < sets.size(); s++) {
    if (members[s].empty()) { dH.emplace_back(); continue; }
    std::vector<Fr> H;
    for (const auto& attr : sets[s])
      for (int l = 0; l < 3; l++)
        for (int t = 0; t < 2; t++) H.push_back(sha3_hash_fr(attr + std::to_string(l) + std::to_string(t)));
    dH.push_back(up_bytes(eng, flatten_fr(H)));
    const Group& g = groups[s];
    eng.check(rhip_ac17_cp_keygen_batch(cx, tb->g, tb->h, dgk.as<rhip_g1>(), da.as<rhip_fr>(), db.as<rhip_fr>(), g.cnt, g.n_attr, dH.back().as<rhip_fr>(),
                                        dH01.as<rhip_fr>(), (const rhip_fr*)(d_r.as<uint8_t>() + 64 * g.first_out),
                                        (const rhip_fr*)(d_sig.as<uint8_t>() + 32 * grp_row0[s]), (const rhip_fr*)(d_sp.as<uint8_t>() + 32 * g.first_out),
                                        (rhip_g2*)(d_k0.as<uint8_t>() + 384 * g.first_out), (rhip_g1*)(d_k.as<uint8_t>() + 192 * grp_row0[s]),
                                        (rhip_g1*)(d_kp.as<uint8_t>() + 192 * g.first_out)), "rhip_ac17_cp_keygen_batch");
  }
  uint8_t* h_k = eng.pinned(2, tot_rows * 192 + tot_items * (384 + 192) + 4);
  uint8_t* h_k0 = h_k + tot_rows * 192;
  uint8_t* h_kp = h_k0 + tot_items * 384;
  eng.check(rhip_download_async(cx, h_k, d_k.ptr(), tot_rows * 192), "download");
  eng.check(rhip_download_async(cx, h_k0, d_k0.ptr(), tot_items * 384), "download");
  eng.check(rhip_download_async(cx, h_kp, d_kp.ptr(), tot_items * 192), "download");
  eng.check(rhip_sync(cx), "rhip_sync");
  tm.lap("device + copies");
  parallel_for(n, [&](size_t i) {
    const size_t s = item_set[i], o = slot[i], q = o - groups[s].first_out;
    const auto& attrs = sets[s];
    uint8_t* w = out_buf + out_off[i];
    put_u32(w, (uint32_t)attrs.size()); w += 4;
    for (const auto& a : attrs) { put_u32(w, (uint32_t)a.size()); w += 4; memcpy(w, a.data(), a.size()); w += a.size(); }
    put_u32(w, 3); w += 4;
    memcpy(w, h_k0 + 384 * o, 384); w += 384;
    put_u32(w, (uint32_t)attrs.size()); w += 4;
    const uint8_t* rows = h_k + 192 * (grp_row0[s] + q * attrs.size());
    for (size_t y = 0; y < attrs.size(); y++) {
      put_u32(w, (uint32_t)attrs[y].size()); w += 4;
      memcpy(w, attrs[y].data(), attrs[y].size()); w += attrs[y].size();
      put_u32(w, 3); w += 4;
      memcpy(w, rows + 192 * y, 192); w += 192;
    }
    put_u32(w, 3); w += 4;
    memcpy(w, h_kp + 192 * o, 192);
  });
  tm.lap("assembly");
  return true;
}
}  // namespace ac17

// ================================================================================================================= BSW
namespace bsw {
namespace {
void* make_pk(Engine& eng, const void* arg) {
  const CpAbePublicKey& pk = *(const CpAbePublicKey*)arg;
  rhip_bsw_pk* d = nullptr;
  eng.check(rhip_bsw_pk_create(eng.ctx(), (const rhip_g1*)pk.g1.data(), (const rhip_g2*)pk.g2.data(), (const rhip_g1*)pk.h.data(),
                               (const rhip_gt*)pk.e_gg_alpha.data(), &d), "rhip_bsw_pk_create");
  return d;
}
void destroy_pk(void* h) { rhip_bsw_pk_destroy((rhip_bsw_pk*)h); }
void* make_sk_lines(Engine& eng, const void* arg) {          // arg: d | d_j[0].g2 | d_j[1].g2 ... (128 B each)
  const std::string& pts = *(const std::string*)arg;
  DBuf d(&eng, pts.data(), pts.size());
  rhip_bsw_sk_lines* lines = nullptr;
  eng.check(rhip_bsw_sk_prepare(eng.ctx(), 1, pts.size() / 128 - 1, d.as<rhip_g2>(), (const rhip_g2*)(d.as<uint8_t>() + 128), &lines), "rhip_bsw_sk_prepare");
  return lines;
}
void destroy_sk_lines(void* h) { rhip_bsw_sk_lines_destroy((rhip_bsw_sk_lines*)h); }
struct GenTables { rhip_g1_table* g1 = nullptr; rhip_g2_table* g2 = nullptr; };
void* make_gen_tables(Engine& eng, const void* arg) {
  const CpAbePublicKey& pk = *(const CpAbePublicKey*)arg;
  std::unique_ptr<GenTables> t(new GenTables());
  eng.check(rhip_g1_table_create(eng.ctx(), (const rhip_g1*)pk.g1.data(), &t->g1), "rhip_g1_table_create");
  int32_t rc = rhip_g1_table_add_w16(eng.ctx(), t->g1);
  if (!rc) rc = rhip_g2_table_create(eng.ctx(), (const rhip_g2*)pk.g2.data(), &t->g2);
  if (!rc) rc = rhip_g2_table_add_w16(eng.ctx(), t->g2);
  if (rc) { rhip_g1_table_destroy(t->g1); if (t->g2) rhip_g2_table_destroy(t->g2); eng.check(rc, "generator tables"); }
  return t.release();
}
void destroy_gen_tables(void* h) {
  GenTables* t = (GenTables*)h;
  rhip_g1_table_destroy(t->g1);
  rhip_g2_table_destroy(t->g2);
  delete t;
}
}  // namespace

// n calls of bsw::keygen (bsw/mod.rs:125-152) under one master key -- a key authority issuing keys in bulk.  Item i gets the attribute list
// sets[item_set[i]]; draw order per item: r (:130), then r_j per attribute (:142).  Record = CpAbeSecretKey: d, rows (name, g1*r_j,
// g2*r + (g2*h(j))*r_j).  Every element is a fixed-base multiple of a generator: d = g2_alpha/beta + g2*(r/beta) (the first term is
// computed once per call), D_j.g1 = g1*r_j, D_j.g2 = g2*(r + h(j) r_j) -- three window-table launches for the whole batch.
bool keygen_packed(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const CpAbeMasterKey& msk, const std::vector<std::vector<std::string>>& sets, size_t n,
                   const uint32_t* item_set, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  Timer tm("bsw::keygen_packed");
  Engine::ArenaScope arena(eng);
  eng.scrub_when_done();          // master-key-derived scalars pass through the staging buffers
  if (n && (!item_set || !out_off)) throw RabeError("bsw::keygen_packed: null input");
  std::vector<size_t> fixed(sets.size());
  std::vector<std::vector<Fr>> hashes(sets.size());
  for (size_t s = 0; s < sets.size(); s++) {
    if (sets[s].empty()) throw RabeError("bsw::keygen_packed: an empty attribute list (bsw::keygen returns None for it)");
    fixed[s] = 128 + 4;
    for (const auto& a : sets[s]) { fixed[s] += 4 + a.size() + 64 + 128; hashes[s].push_back(sha3_hash_fr(a)); }
  }
  for (size_t i = 0; i < n; i++) if (item_set[i] >= sets.size()) throw RabeError("bsw::keygen_packed: item_set out of range");
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + fixed[item_set[i]];
  if (!out_buf || out_cap < out_off[n]) return false;
  if (!n) return true;
  Fr beta_inv;
  if (!fr_inv(msk.beta, &beta_inv)) throw std::runtime_error("called `Option::unwrap()` on a `None` value (Fr::inverse of zero)");
  std::vector<size_t> row_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) row_off[i + 1] = row_off[i] + sets[item_set[i]].size();
  const size_t total = row_off[n];
  uint8_t* h_k = eng.pinned(0, (n + 2 * total) * 32 + 32);          // r/beta per key | r_j per row | r + h(j) r_j per row
  uint8_t* h_kd = h_k;
  uint8_t* h_k1 = h_k + n * 32;
  uint8_t* h_k2 = h_k1 + total * 32;
  draw_items(rng, n, [&](Rng& r, size_t i) {
    const Fr ri = r.next_fr();
    const Fr kd = fr_mul(ri, beta_inv);
    memcpy(h_kd + 32 * i, kd.l, 32);
    const auto& hs = hashes[item_set[i]];
    for (size_t y = 0; y < hs.size(); y++) {
      const Fr rj = r.next_fr();
      const Fr k2 = fr_add(ri, fr_mul(hs[y], rj));
      memcpy(h_k1 + 32 * (row_off[i] + y), rj.l, 32);
      memcpy(h_k2 + 32 * (row_off[i] + y), k2.l, 32);
    }
  });
  tm.lap("draws + scalars");
  const GenTables* tb;
  {
    std::string key((const char*)pk.g1.data(), 64);
    key.append((const char*)pk.g2.data(), 128);
    tb = (const GenTables*)eng.aux("bsw_gen_tables", key, make_gen_tables, &pk, destroy_gen_tables, 4);
  }
  const G2 a_const = eng.g2_mul({msk.g2_alpha}, {beta_inv})[0];          // g2_alpha / beta, once per call
  std::vector<uint8_t> a_rep(n * 128);
  for (size_t i = 0; i < n; i++) memcpy(a_rep.data() + 128 * i, a_const.data(), 128);
  rhip_ctx* cx = eng.ctx();
  DBuf d_kd(&eng, n * 32), d_k1(&eng, total * 32 + 4), d_k2(&eng, total * 32 + 4), d_a = up_bytes(eng, a_rep), d_dp(&eng, n * 128), d_d(&eng, n * 128),
      d_g1(&eng, total * 64 + 4), d_g2(&eng, total * 128 + 4);
  eng.check(rhip_upload_async(cx, d_kd.ptr(), h_kd, n * 32), "upload");
  eng.check(rhip_upload_async(cx, d_k1.ptr(), h_k1, total * 32), "upload");
  eng.check(rhip_upload_async(cx, d_k2.ptr(), h_k2, total * 32), "upload");
  eng.check(rhip_g2_table_mul(cx, tb->g2, n, d_kd.as<rhip_fr>(), d_dp.as<rhip_g2>()), "rhip_g2_table_mul");
  eng.check(rhip_g2_add(cx, n, d_a.as<rhip_g2>(), d_dp.as<rhip_g2>(), d_d.as<rhip_g2>()), "rhip_g2_add");
  eng.check(rhip_g1_table_mul(cx, tb->g1, total, d_k1.as<rhip_fr>(), d_g1.as<rhip_g1>()), "rhip_g1_table_mul");
  eng.check(rhip_g2_table_mul(cx, tb->g2, total, d_k2.as<rhip_fr>(), d_g2.as<rhip_g2>()), "rhip_g2_table_mul");
  uint8_t* h_o = eng.pinned(1, n * 128 + total * 192 + 4);
  uint8_t* h_g1 = h_o + n * 128;
  uint8_t* h_g2 = h_g1 + total * 64;
  eng.check(rhip_download_async(cx, h_o, d_d.ptr(), n * 128), "download");
  eng.check(rhip_download_async(cx, h_g1, d_g1.ptr(), total * 64), "download");
  eng.check(rhip_download_async(cx, h_g2, d_g2.ptr(), total * 128), "download");
  eng.check(rhip_sync(cx), "rhip_sync");
  tm.lap("device + copies");
  parallel_for(n, [&](size_t i) {
    const auto& attrs = sets[item_set[i]];
    uint8_t* w = out_buf + out_off[i];
    memcpy(w, h_o + 128 * i, 128); w += 128;
    put_u32(w, (uint32_t)attrs.size()); w += 4;
    for (size_t y = 0; y < attrs.size(); y++) {
      put_u32(w, (uint32_t)attrs[y].size()); w += 4;
      memcpy(w, attrs[y].data(), attrs[y].size()); w += attrs[y].size();
      memcpy(w, h_g1 + 64 * (row_off[i] + y), 64); w += 64;
      memcpy(w, h_g2 + 128 * (row_off[i] + y), 128); w += 128;
    }
  });
  tm.lap("assembly");
  return true;
}

// n calls of bsw::delegate (bsw/mod.rs:162-206) on ONE key: item i delegates `sk` to subsets[item_subset[i]].  Draw order per item: r
// (:178), then r_j per attribute of the subset in its order (:183).  d' = d + f * r; per attribute (D_j.g1 + g1 * r_j,
// D_j.g2 + g2 * (h(j) r_j + r)): three window-table launches (f's table is built on first use and kept) and three batched additions;
// records = CpAbeSecretKey, written on the device.  A subset that is empty or not contained in the key's attributes (is_subset,
// tools/mod.rs:24-28 -- delegate returns None) fails the call.
namespace {
void* make_f_table(Engine& eng, const void* arg) {
  rhip_g2_table* t = nullptr;
  eng.check(rhip_g2_table_create(eng.ctx(), (const rhip_g2*)((const G2*)arg)->data(), &t), "rhip_g2_table_create");
  int32_t rc = rhip_g2_table_add_w16(eng.ctx(), t);
  if (rc) { rhip_g2_table_destroy(t); eng.check(rc, "rhip_g2_table_add_w16"); }
  return t;
}
void destroy_f_table(void* h) { rhip_g2_table_destroy((rhip_g2_table*)h); }
}  // namespace
bool delegate_packed(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const CpAbeSecretKey& sk, const std::vector<std::vector<std::string>>& subsets, size_t n,
                     const uint32_t* item_subset, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  Timer tm("bsw::delegate_packed");
  Engine::ArenaScope arena(eng);
  eng.scrub_when_done();          // the key's elements pass through the staging buffers
  if (n && (!item_subset || !out_off)) throw RabeError("bsw::delegate_packed: null input");
  std::vector<std::vector<uint32_t>> row_of(subsets.size());          // subset attribute -> the key's row
  std::vector<std::vector<Fr>> hashes(subsets.size());
  std::vector<RecordLayout> layouts(subsets.size());
  for (size_t s_ = 0; s_ < subsets.size(); s_++) {
    if (subsets[s_].empty()) throw RabeError("bsw::delegate_packed: an empty subset (bsw::delegate returns None for it)");
    RecordLayout& L = layouts[s_];
    L.src(0, 0, 128);
    L.u32((uint32_t)subsets[s_].size());
    for (size_t y = 0; y < subsets[s_].size(); y++) {
      const std::string& a = subsets[s_][y];
      uint32_t r = 0;
      while (r < sk.d_j.size() && sk.d_j[r].string != a) r++;
      if (r == sk.d_j.size()) throw RabeError("bsw::delegate_packed: the subset is not contained in the key's attributes (bsw::delegate returns None)");
      row_of[s_].push_back(r);
      hashes[s_].push_back(sha3_hash_fr(a));
      L.str(a);
      L.src(1, (uint32_t)(64 * y), 64);
      L.src(2, (uint32_t)(128 * y), 128);
    }
  }
  for (size_t i = 0; i < n; i++) if (item_subset[i] >= subsets.size()) throw RabeError("bsw::delegate_packed: item_subset out of range");
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + layouts[item_subset[i]].bytes();
  if (!out_buf || out_cap < out_off[n]) return false;
  if (!n) return true;
  std::vector<size_t> row_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) row_off[i + 1] = row_off[i] + subsets[item_subset[i]].size();
  const size_t total = row_off[n];
  uint8_t* h_k = eng.pinned(0, (n + 2 * total) * 32 + 32);          // r per item | r_j per row | h(j) r_j + r per row
  uint8_t* h_kr = h_k;
  uint8_t* h_k1 = h_k + n * 32;
  uint8_t* h_k2 = h_k1 + total * 32;
  uint8_t* h_old = eng.pinned(1, n * 128 + total * 192 + 4);         // d per item | D_j.g1 per row | D_j.g2 per row
  uint8_t* h_o1 = h_old + n * 128;
  uint8_t* h_o2 = h_o1 + total * 64;
  draw_items(rng, n, [&](Rng& r, size_t i) {
    const Fr ri = r.next_fr();
    memcpy(h_kr + 32 * i, ri.l, 32);
    memcpy(h_old + 128 * i, sk.d.data(), 128);
    const size_t s_ = item_subset[i];
    for (size_t y = 0; y < hashes[s_].size(); y++) {
      const Fr rj = r.next_fr();
      const Fr k2 = fr_add(fr_mul(hashes[s_][y], rj), ri);
      memcpy(h_k1 + 32 * (row_off[i] + y), rj.l, 32);
      memcpy(h_k2 + 32 * (row_off[i] + y), k2.l, 32);
      memcpy(h_o1 + 64 * (row_off[i] + y), sk.d_j[row_of[s_][y]].g1.data(), 64);
      memcpy(h_o2 + 128 * (row_off[i] + y), sk.d_j[row_of[s_][y]].g2.data(), 128);
    }
  });
  tm.lap("draws + scalars");
  const GenTables* tb;
  {
    std::string key((const char*)pk.g1.data(), 64);
    key.append((const char*)pk.g2.data(), 128);
    tb = (const GenTables*)eng.aux("bsw_gen_tables", key, make_gen_tables, &pk, destroy_gen_tables, 4);
  }
  rhip_g2_table* ft = (rhip_g2_table*)eng.aux("bsw_f_table", std::string((const char*)pk.f.data(), 128), make_f_table, &pk.f, destroy_f_table, 4);
  rhip_ctx* cx = eng.ctx();
  DBuf d_k(&eng, (n + 2 * total) * 32 + 32), d_old(&eng, n * 128 + total * 192 + 4), d_fr(&eng, n * 128), d_m1(&eng, total * 64 + 4), d_m2(&eng, total * 128 + 4),
      d_d(&eng, n * 128), d_g1(&eng, total * 64 + 4), d_g2(&eng, total * 128 + 4);
  eng.check(rhip_upload_async(cx, d_k.ptr(), h_k, (n + 2 * total) * 32), "upload");
  eng.check(rhip_upload_async(cx, d_old.ptr(), h_old, n * 128 + total * 192), "upload");
  const rhip_fr* kr = d_k.as<rhip_fr>();
  const uint8_t* old = d_old.as<uint8_t>();
  eng.check(rhip_g2_table_mul(cx, ft, n, kr, d_fr.as<rhip_g2>()), "rhip_g2_table_mul");
  eng.check(rhip_g2_add(cx, n, (const rhip_g2*)old, d_fr.as<rhip_g2>(), d_d.as<rhip_g2>()), "rhip_g2_add");
  eng.check(rhip_g1_table_mul(cx, tb->g1, total, kr + n, d_m1.as<rhip_g1>()), "rhip_g1_table_mul");
  eng.check(rhip_g1_add(cx, total, (const rhip_g1*)(old + n * 128), d_m1.as<rhip_g1>(), d_g1.as<rhip_g1>()), "rhip_g1_add");
  eng.check(rhip_g2_table_mul(cx, tb->g2, total, kr + n + total, d_m2.as<rhip_g2>()), "rhip_g2_table_mul");
  eng.check(rhip_g2_add(cx, total, (const rhip_g2*)(old + n * 128 + total * 64), d_m2.as<rhip_g2>(), d_g2.as<rhip_g2>()), "rhip_g2_add");
  std::vector<uint64_t> src_off(3 * n);
  for (size_t i = 0; i < n; i++) { src_off[i] = 128ull * i; src_off[n + i] = 64ull * row_off[i]; src_off[2 * n + i] = 128ull * row_off[i]; }
  emit_plain_records(eng, layouts, n, item_subset, {d_d.ptr(), d_g1.ptr(), d_g2.ptr()}, src_off, out_off, out_buf);
  tm.lap("device: fixed-base multiplications, additions, records; one copy out");
  return true;
}

// n calls of bsw::encrypt (bsw/mod.rs:217-251).  Draw order per item: secret (:228), msg (:229), the gate coefficients of
// gen_shares_policy (secretsharing/mod.rs:128-134), the AES nonce (aes/mod.rs:17).  Record = CpAbeCiphertext:
//   policy text, language, c, c_p, leaf count, per leaf (name_col, g1 * q_y, (g2 * h(name)) * q_y), sealed plaintext.
bool encrypt_packed(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const std::vector<std::string>& policies, PolicyLanguage language, size_t n,
                    const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  Timer tm("bsw::encrypt_packed");
  Engine::ArenaScope arena(eng);
  std::vector<std::shared_ptr<const FlatPolicy>> pols;
  for (const auto& p : policies) pols.push_back(flat_policy(p, language));
  for (size_t i = 0; i < n; i++) if (item_policy[i] >= policies.size()) throw RabeError("bsw::encrypt_packed: item_policy out of range");
  std::vector<size_t> fixed(policies.size());
  for (size_t p = 0; p < policies.size(); p++)
    fixed[p] = 4 + policies[p].size() + 1 + 64 + 384 + 4 + pols[p]->leaf_name.size() * (4 + 64 + 128) + pols[p]->names_bytes + 4;
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + fixed[item_policy[i]] + (pt_off[i + 1] - pt_off[i]) + 28;
  if (!out_buf || out_cap < out_off[n]) return false;
  std::vector<uint32_t> leaf_off(n + 1, 0), coef_off(n + 1, 0), tree_leaf(n), tree_gate(n);
  for (size_t i = 0; i < n; i++) {
    leaf_off[i + 1] = leaf_off[i] + (uint32_t)pols[item_policy[i]]->leaf_name.size();
    coef_off[i + 1] = coef_off[i] + pols[item_policy[i]]->n_coef;
  }
  const size_t total = leaf_off[n], total_coef = coef_off[n];
  tm.lap("policies");
  uint8_t* h_in = eng.pinned(0, n * 64 + (total_coef + 1) * 32);          // secret | msg exponent | coefficients
  uint8_t* h_sec = h_in;
  uint8_t* h_rho = h_in + n * 32;
  uint8_t* h_coef = h_in + n * 64;
  std::vector<std::array<uint8_t, 12>> nonces(n);
  draw_items(rng, n, [&](Rng& r, size_t i) {
    Fr s = r.next_fr(), rho = r.next_fr();
    memcpy(h_sec + 32 * i, s.l, 32);
    memcpy(h_rho + 32 * i, rho.l, 32);
    for (uint32_t c = coef_off[i]; c < coef_off[i + 1]; c++) { Fr a = r.next_fr(); memcpy(h_coef + 32 * (size_t)c, a.l, 32); }
    r.fill(nonces[i].data(), 12);
  });
  tm.lap("draws");
  rhip_ctx* cx = eng.ctx();
  std::string key((const char*)pk.g1.data(), 64);
  key.append((const char*)pk.g2.data(), 128).append((const char*)pk.h.data(), 64).append((const char*)pk.e_gg_alpha.data(), 384);
  rhip_bsw_pk* dpk = (rhip_bsw_pk*)eng.aux("bsw_pk", key, make_pk, &pk, destroy_pk);
  DevTrees dt(eng, pols);
  for (size_t i = 0; i < n; i++) { tree_leaf[i] = dt.first_leaf[item_policy[i]]; tree_gate[i] = dt.first_gate[item_policy[i]]; }
  DBuf d_leaf_off(&eng, leaf_off.data(), (n + 1) * 4), d_tl(&eng, tree_leaf.data(), n * 4), d_tg(&eng, tree_gate.data(), n * 4),
      d_coef_off(&eng, coef_off.data(), n * 4), d_in(&eng, n * 64 + (total_coef + 1) * 32), d_msg(&eng, n * 384), d_c(&eng, n * 64), d_cp(&eng, n * 384),
      d_g1(&eng, total * 64), d_g2(&eng, total * 128);
  eng.check(rhip_upload_async(cx, d_in.ptr(), h_in, n * 64 + total_coef * 32), "upload");
  const rhip_fr* dsec = d_in.as<rhip_fr>();
  eng.check(rhip_gt_table_pow(cx, eng.gt_generator_table(), n, dsec + n, d_msg.as<rhip_gt>()), "rhip_gt_table_pow");
  eng.check(rhip_bsw_encrypt_batch(cx, dpk, n, total, d_leaf_off.as<uint32_t>(), d_tl.as<uint32_t>(), d_tg.as<uint32_t>(), dt.path_off.as<uint32_t>(),
                                   dt.path_gate.as<uint32_t>(), dt.path_x.as<uint32_t>(), dt.gate_k.as<uint32_t>(), dt.gate_coef_off.as<uint32_t>(),
                                   dt.leaf_hash.as<rhip_fr>(), dsec, dsec + 2 * n, d_coef_off.as<uint32_t>(), d_msg.as<rhip_gt>(), d_c.as<rhip_g1>(),
                                   d_cp.as<rhip_gt>(), d_g1.as<rhip_g1>(), d_g2.as<rhip_g2>()), "rhip_bsw_encrypt_batch");
  // records and sealing on the device (records.h)
  std::vector<RecordLayout> layouts(policies.size());
  for (size_t p_ = 0; p_ < policies.size(); p_++) {
    RecordLayout& L = layouts[p_];
    const FlatPolicy& f = *pols[p_];
    L.str(policies[p_]);
    L.u8((language == PolicyLanguage::HumanPolicy) ? 1 : 0);
    L.src(0, 0, 64);
    L.src(1, 0, 384);
    L.u32((uint32_t)f.leaf_name.size());
    for (size_t y = 0; y < f.leaf_name.size(); y++) {
      L.str(f.leaf_name_col[y]);
      L.src(2, (uint32_t)(64 * y), 64);
      L.src(3, (uint32_t)(128 * y), 128);
    }
    if (L.bytes() + 4 != fixed[p_]) throw RabeError("bsw::encrypt_packed: record layout and size disagree");
  }
  std::vector<uint64_t> src_off(4 * n);
  for (size_t i = 0; i < n; i++) { src_off[i] = 64ull * i; src_off[n + i] = 384ull * i; src_off[2 * n + i] = 64ull * leaf_off[i]; src_off[3 * n + i] = 128ull * leaf_off[i]; }
  emit_sealed_records(eng, layouts, n, item_policy, {d_c.ptr(), d_cp.ptr(), d_g1.ptr(), d_g2.ptr()}, src_off, d_msg.ptr(), (const uint8_t*)nonces.data(),
                      pt_blob, pt_off, out_off, out_buf);
  tm.lap("device: group arithmetic, records, sealing; one copy out");
  return true;
}

// n calls of bsw::decrypt (bsw/mod.rs:260-318) with one key.  Per distinct policy text: traverse_policy, calc_pruned and the
// coefficients, turned into the selection entries the device path takes -- entry = (ciphertext leaf row, key attribute row, z):
// for every pruned (name, name_col) the FIRST ciphertext row named name_col, the FIRST key row named name, and one entry per
// coefficient named name_col (:283-299 as index lists).  status / errors / buffers as ac17::cp_decrypt_packed.
bool decrypt_packed(Engine& eng, const CpAbeSecretKey& sk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                    int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors) {
  Timer tm("bsw::decrypt_packed");
  Engine::ArenaScope arena(eng);
  errors->assign(n, "");
  if (!ct_off || (n && !ct_blob)) throw RabeError("bsw::decrypt_packed: null input");
  const uint64_t span = check_offsets(n, ct_off, ct_len, errors);
  if (!pt_buf || pt_cap < span) return false;
  BlobGather gather(eng, ct_blob, ct_len);          // the blob starts for the device now, beside the parsing below (records.h)
  std::vector<std::string> attr;
  for (const auto& v : sk.d_j) attr.push_back(v.string);
  struct Plan {
    std::shared_ptr<const FlatPolicy> flat; std::string err;
    struct E { std::string name_col; uint32_t sk_row; Fr z; uint32_t std_ct_row; };
    std::vector<E> ent;
  };
  std::map<std::pair<int, std::string>, std::shared_ptr<Plan>> plans;
  std::mutex plans_mu;
  FastPlans<Plan> fast_plans;
  auto plan_of = [&](const std::string& text, PolicyLanguage lang) -> std::shared_ptr<Plan> {
    std::lock_guard<std::mutex> g(plans_mu);
    auto key = std::make_pair((int)lang, text);
    auto it = plans.find(key);
    if (it != plans.end()) return it->second;
    auto pl = std::make_shared<Plan>();
    try {
      pl->flat = flat_policy(text, lang);
      const PolicyNode& tree = pl->flat->tree;
      if (!traverse_policy(attr, tree)) throw RabeError("Error in bsw/encrypt: attributes do not match policy.");
      PrunedList pruned;
      if (!calc_pruned(attr, tree, &pruned)) throw RabeError("Error in bsw/encrypt: attributes do not match policy.");
      for (const auto& pr : pruned) {
        size_t dj = 0;
        while (dj < sk.d_j.size() && sk.d_j[dj].string != pr.first) dj++;
        if (dj == sk.d_j.size()) continue;
        size_t std_row = 0;
        while (std_row < pl->flat->leaf_name_col.size() && pl->flat->leaf_name_col[std_row] != pr.second) std_row++;
        for (size_t y = 0; y < pl->flat->leaf_name_col.size(); y++)
          if (pl->flat->leaf_name_col[y] == pr.second) pl->ent.push_back({pr.second, (uint32_t)dj, pl->flat->leaf_coeff[y], (uint32_t)std_row});
      }
    } catch (const std::exception& ex) {
      pl->err = ex.what();
      if (pl->err.empty()) pl->err = "policy error";
    }
    plans[key] = pl;
    return pl;
  };
  struct View { const uint8_t* c; const uint8_t* cp; uint32_t rows; std::vector<const uint8_t*> g1, g2; std::shared_ptr<Plan> plan;
                std::vector<uint32_t> ct_row; bool standard; };
  std::vector<View> v(n);
  std::vector<Sealed> sealed(n);
  parallel_for(n, [&](size_t i) {
    if (!(*errors)[i].empty()) return;
    try {
      Cursor r{ct_blob + ct_off[i], ct_blob + ct_off[i + 1]};
      auto pol = r.str();
      const PolicyLanguage lang = *r.raw(1) ? PolicyLanguage::HumanPolicy : PolicyLanguage::JsonPolicy;
      v[i].c = r.raw(64);
      v[i].cp = r.raw(384);
      const uint32_t rows = r.u32();
      if ((size_t)rows * 196 > (size_t)(r.end - r.p)) throw RabeError("deserialize: truncated input");
      v[i].rows = rows;
      v[i].g1.resize(rows);
      v[i].g2.resize(rows);
      std::vector<std::pair<const char*, uint32_t>> names(rows);
      for (uint32_t y = 0; y < rows; y++) { names[y] = r.str(); v[i].g1[y] = r.raw(64); v[i].g2[y] = r.raw(128); }
      sealed[i].len = r.u32();
      sealed[i].p = r.raw(sealed[i].len);
      auto pl = fast_plans.find(pol.first, pol.second, lang);
      if (!pl) { pl = plan_of(std::string(pol.first, pol.second), lang); fast_plans.put(pol.first, pol.second, lang, pl); }
      if (!pl->err.empty()) throw RabeError(pl->err);
      v[i].plan = pl;
      const auto& std_names = pl->flat->leaf_name_col;
      bool standard = rows == std_names.size();
      for (uint32_t y = 0; y < rows && standard; y++) standard = same(names[y], std_names[y]);
      v[i].standard = standard;
      if (!standard) {                              // rows in another order / other names: the name-matching loop itself (:283-287)
        for (const auto& e : pl->ent) {
          uint32_t y = 0;
          while (y < rows && !same(names[y], e.name_col)) y++;
          v[i].ct_row.push_back(y);               // y == rows: no such row -> the entry is skipped below
        }
      }
    } catch (const std::exception& ex) {
      (*errors)[i] = ex.what();
      if ((*errors)[i].empty()) (*errors)[i] = "malformed record";
    }
  });
  tm.lap("parse + plan");
  std::vector<size_t> live;
  std::vector<uint32_t> leaf_off{0}, pair_off{0}, sel_start, sel_ct, sel_sk;
  std::vector<Fr> sel_z;
  std::map<const Plan*, uint32_t> shared_start;          // standard-layout items of one policy share their selection entries
  size_t max_pairs = 1;
  for (size_t i = 0; i < n; i++) {
    if (!(*errors)[i].empty()) continue;
    live.push_back(i);
    leaf_off.push_back(leaf_off.back() + v[i].rows);
    const Plan& pl = *v[i].plan;
    uint32_t m = 0;
    if (v[i].standard) {
      auto it = shared_start.find(&pl);
      if (it == shared_start.end()) {
        it = shared_start.insert({&pl, (uint32_t)sel_ct.size()}).first;
        for (const auto& e : pl.ent) { sel_ct.push_back(e.std_ct_row); sel_sk.push_back(e.sk_row); sel_z.push_back(e.z); }
      }
      sel_start.push_back(it->second);
      m = (uint32_t)pl.ent.size();
    } else {
      sel_start.push_back((uint32_t)sel_ct.size());
      for (size_t e = 0; e < pl.ent.size(); e++)
        if (v[i].ct_row[e] < v[i].rows) { sel_ct.push_back(v[i].ct_row[e]); sel_sk.push_back(pl.ent[e].sk_row); sel_z.push_back(pl.ent[e].z); m++; }
    }
    pair_off.push_back(pair_off.back() + 2 * m + 1);
    if ((size_t)2 * m + 1 > max_pairs) max_pairs = 2 * m + 1;
  }
  const size_t m_items = live.size();
  std::vector<uint64_t> sealed_off(m_items);
  std::vector<uint32_t> sealed_len(m_items);
  DBuf d_out(&eng, m_items * 384 + 4);
  // what the walk's verdicts refer to outlives the block below: they are read after the open (records.h: retract_item)
  std::unique_ptr<MemberChecks> mc;          // the decoding checks run on the side context, beside the decrypt kernels (common.h)
  std::unique_ptr<WalkedG2> walked;
  std::vector<uint32_t> walked_idx, walked_off;
  DBuf d_g2;
  if (m_items) {
    const size_t total = leaf_off[m_items];
    DBuf d_c(&eng, m_items * 64), d_cp(&eng, m_items * 384), d_g1(&eng, total * 64 + 4);
    d_g2 = DBuf(&eng, total * 128 + 4);
    std::vector<uint64_t> dst_off(4 * m_items);
    for (size_t j = 0; j < m_items; j++) {
      const View& w = v[live[j]];
      const uint8_t* rec = ct_blob + ct_off[live[j]];
      sealed_off[j] = (uint64_t)(sealed[live[j]].p - ct_blob);
      sealed_len[j] = sealed[live[j]].len;
      dst_off[j] = 64ull * j; dst_off[m_items + j] = 384ull * j; dst_off[2 * m_items + j] = 64ull * leaf_off[j]; dst_off[3 * m_items + j] = 128ull * leaf_off[j];
      int shape = w.standard ? gather.find(w.plan.get()) : -1;          // standard layout: policy text and names as encrypt writes them -> one skeleton
      if (shape < 0) {
        std::vector<RecordLayout::Part> parts;
        parts.push_back({(uint32_t)(w.c - rec), 64, 0, 0});
        parts.push_back({(uint32_t)(w.cp - rec), 384, 1, 0});
        for (uint32_t y = 0; y < w.rows; y++) {
          parts.push_back({(uint32_t)(w.g1[y] - rec), 64, 2, 64 * y});
          parts.push_back({(uint32_t)(w.g2[y] - rec), 128, 3, 128 * y});
        }
        shape = (int)gather.add_shape(w.standard ? (const void*)w.plan.get() : nullptr, std::move(parts));
      }
      gather.item(ct_off[live[j]], (uint32_t)shape);
    }
    tm.lap("shapes");
    rhip_ctx* cx = eng.ctx();
    std::vector<uint8_t> kg1, kg2;
    for (const auto& a : sk.d_j) { kg1.insert(kg1.end(), a.g1.begin(), a.g1.end()); kg2.insert(kg2.end(), a.g2.begin(), a.g2.end()); }
    std::vector<uint32_t> sk_attr_off{0, (uint32_t)sk.d_j.size()};
    auto fz = flatten_fr(sel_z);
    DBuf d_leaf_off = up32(eng, leaf_off),
        d_pair_off = up32(eng, pair_off), d_sel_start = up32(eng, sel_start), d_sel_ct = up32(eng, sel_ct), d_sel_sk = up32(eng, sel_sk),
        d_sel_z = up_bytes(eng, fz), d_skd(&eng, sk.d.data(), 128), d_kg1 = up_bytes(eng, kg1), d_kg2 = up_bytes(eng, kg2),
        d_sk_attr_off = up32(eng, sk_attr_off);
    gather.run({d_c.ptr(), d_cp.ptr(), d_g1.ptr(), d_g2.ptr()}, dst_off);
    // the key's prepared lines (d and every d_j.g2: 17 KB per point) are a function of the key alone: kept across calls
    rhip_bsw_sk_lines* lines = nullptr;
    if (!sk.d_j.empty()) {
      std::string key((const char*)sk.d.data(), 128);
      for (const auto& a : sk.d_j) key.append((const char*)a.g2.data(), 128);
      lines = (rhip_bsw_sk_lines*)eng.aux("bsw_sk_lines", key, make_sk_lines, &key, destroy_sk_lines, 4);
    }
    if (!trusted) {
      mc.reset(new MemberChecks(eng));
      mc->add(1, d_c.ptr(), m_items); mc->add(1, d_g1.ptr(), total, d_leaf_off.as<uint32_t>(), m_items);
      mc->add(3, d_cp.ptr(), m_items);
      // Cy.g2 of every selected leaf is the walking argument of a pairing (the key's side replays prepared lines): the decrypt's own
      // Miller loops say whether it is a member of G2; leaves the policy did not select get the stand-alone test (common.h: WalkedG2)
      if (walk_checks() && lines) {
        // "every leaf is walked" may only be concluded from the counts for records in the standard layout (their selection is the plan's:
        // distinct rows); rows matched by NAME can coincide in a crafted record, and then some other row is walked by nobody
        // ... and the count argument needs the selected rows of an item to be DISTINCT: it is verified, not assumed (two plan entries
        // that named one row would leave another one walked by nobody)
        bool all = true;
        std::vector<uint8_t> seen;
        for (size_t j = 0; j < m_items && all; j++) {
          const uint32_t mj = (pair_off[j + 1] - pair_off[j] - 1) / 2, leaves = leaf_off[j + 1] - leaf_off[j];
          all = v[live[j]].standard && mj == leaves;
          if (!all) break;
          seen.assign(leaves, 0);
          for (uint32_t e = 0; e < mj && all; e++) {
            const uint32_t row = sel_ct[sel_start[j] + e];
            all = row < leaves && !seen[row];
            if (all) seen[row] = 1;
          }
        }
        if (!all) {
          walked_off.push_back(0);
          for (size_t j = 0; j < m_items; j++) {
            const uint32_t mj = (pair_off[j + 1] - pair_off[j] - 1) / 2;
            for (uint32_t e = 0; e < mj; e++) walked_idx.push_back(leaf_off[j] + sel_ct[sel_start[j] + e]);
            walked_off.push_back((uint32_t)walked_idx.size());
          }
        }
        walked.reset(new WalkedG2(eng, *mc, d_g2.ptr(), total, d_leaf_off.as<uint32_t>(), leaf_off, 1, all ? nullptr : &walked_idx, all ? nullptr : &walked_off));
      } else {
        mc->add(2, d_g2.ptr(), total, d_leaf_off.as<uint32_t>(), m_items);
      }
    }
    // one key for all ciphertexts: its scaled Dj.g1 are computed once per selection entry (ciphertexts that share a policy share them)
    if (walked) walked->arm();
    int32_t rc = rhip_bsw_decrypt_batch_one_sk(cx, m_items, max_pairs, pair_off[m_items], sel_ct.size(), d_pair_off.as<uint32_t>(), d_sel_start.as<uint32_t>(),
                                               d_sel_ct.as<uint32_t>(), d_sel_sk.as<uint32_t>(), d_sel_z.as<rhip_fr>(), d_c.as<rhip_g1>(), d_cp.as<rhip_gt>(),
                                               d_g1.as<rhip_g1>(), d_g2.as<rhip_g2>(), d_leaf_off.as<uint32_t>(), d_skd.as<rhip_g2>(), d_kg1.as<rhip_g1>(),
                                               d_kg2.as<rhip_g2>(), d_sk_attr_off.as<uint32_t>(), lines, d_out.as<rhip_gt>());
    eng.check(rc, "rhip_bsw_decrypt_batch");
    if (mc) {
      mc->collect();
      const auto &ok_c = mc->ok(0), &ok_g1 = mc->ok(1), &ok_cp = mc->ok(2);
      std::vector<uint8_t> ok_g2(m_items, 1);
      if (!walked) { const auto& e = mc->ok(3); ok_g2.assign(e.begin(), e.end()); }
      for (size_t j = 0; j < m_items; j++) {
        const char* bad = !ok_c[j] ? "deserialize: c is not a point of G1 (FieldError::NotMember)" : !ok_cp[j] ? "deserialize: c_p is not a member of Gt (FieldError::NotMember)" : nullptr;
        if (!bad && (!ok_g1[j] || !ok_g2[j])) bad = "deserialize: a leaf element is not a group member (FieldError::NotMember)";
        if (bad) (*errors)[live[j]] = bad;
      }
    }
  }
  // KDF + AES-GCM open on the device: the decrypted Gt never leaves HBM; plaintext bytes come back in one copy
  open_sealed_records(eng, n, live, d_out.ptr(), gather.dev_blob(), sealed_off, sealed_len, status, pt_buf, pt_off, errors);
  if (walked) {
    std::vector<uint8_t> ok_g2;
    walked->finish(&ok_g2);
    for (size_t j = 0; j < m_items; j++)
      if (!ok_g2[j]) retract_item(live[j], "deserialize: a leaf element is not a group member (FieldError::NotMember)", status, pt_buf, pt_off, errors);
  }
  tm.lap(trusted ? "device: gather, pairings, open" : "device: gather, pairings, open; membership beside");
  return true;
}
}  // namespace bsw


// ================================================================================================================= LSW
namespace lsw {
namespace {
void* make_pk(Engine& eng, const void* arg) {
  const KpAbePublicKey& pk = *(const KpAbePublicKey*)arg;
  rhip_lsw_pk* d = nullptr;
  eng.check(rhip_lsw_pk_create(eng.ctx(), (const rhip_g1*)pk.g1.data(), (const rhip_g2*)pk.g2.data(), &d), "rhip_lsw_pk_create");
  return d;
}
void destroy_pk(void* h) { rhip_lsw_pk_destroy((rhip_lsw_pk*)h); }
void* make_e2_lines(Engine& eng, const void* arg) {
  const std::string& pt = *(const std::string*)arg;
  DBuf d(&eng, pt.data(), pt.size());
  rhip_g2_lines* lines = nullptr;
  eng.check(rhip_g2_lines_prepare(eng.ctx(), 1, d.as<rhip_g2>(), &lines), "rhip_g2_lines_prepare");
  return lines;
}
void destroy_e2_lines(void* h) { rhip_g2_lines_destroy((rhip_g2_lines*)h); }
}  // namespace

namespace {
struct EncTables { rhip_g1_table* g1 = nullptr; rhip_g1_table* g1_b = nullptr; rhip_g1_table* g1_b2 = nullptr; rhip_g1_table* h_b = nullptr;
                   rhip_g2_table* g2 = nullptr; rhip_gt_table* egg = nullptr; };
void destroy_enc_tables(void* h) {
  EncTables* t = (EncTables*)h;
  if (t->g1) rhip_g1_table_destroy(t->g1);
  if (t->g1_b) rhip_g1_table_destroy(t->g1_b);
  if (t->g1_b2) rhip_g1_table_destroy(t->g1_b2);
  if (t->h_b) rhip_g1_table_destroy(t->h_b);
  if (t->g2) rhip_g2_table_destroy(t->g2);
  if (t->egg) rhip_gt_table_destroy(t->egg);
  delete t;
}
void* make_enc_tables(Engine& eng, const void* arg) {
  const KpAbePublicKey& pk = *(const KpAbePublicKey*)arg;
  EncTables* t = new EncTables();
  rhip_ctx* cx = eng.ctx();
  int32_t rc = rhip_g1_table_create(cx, (const rhip_g1*)pk.g1.data(), &t->g1);
  if (!rc) rc = rhip_g1_table_add_w16(cx, t->g1);
  if (!rc) rc = rhip_g1_table_create(cx, (const rhip_g1*)pk.g1_b.data(), &t->g1_b);
  if (!rc) rc = rhip_g1_table_add_w16(cx, t->g1_b);
  if (!rc) rc = rhip_g1_table_create(cx, (const rhip_g1*)pk.g1_b2.data(), &t->g1_b2);
  if (!rc) rc = rhip_g1_table_add_w16(cx, t->g1_b2);
  if (!rc) rc = rhip_g1_table_create(cx, (const rhip_g1*)pk.h_b.data(), &t->h_b);
  if (!rc) rc = rhip_g1_table_add_w16(cx, t->h_b);
  if (!rc) rc = rhip_g2_table_create(cx, (const rhip_g2*)pk.g2.data(), &t->g2);
  if (!rc) rc = rhip_g2_table_add_w16(cx, t->g2);
  if (!rc) rc = rhip_gt_table_create(cx, (const rhip_gt*)pk.e_gg_alpha.data(), &t->egg);
  if (!rc) rc = rhip_gt_table_add_w16(cx, t->egg);
  if (rc) { destroy_enc_tables(t); eng.check(rc, "lsw public-key tables"); }
  return t;
}
}  // namespace
// n calls of lsw::encrypt (lsw/mod.rs:180-219): item i under the attribute list sets[item_set[i]].  Draw order per item: secret (:188), one
// sx per attribute (:196-200, with the reference's index quirk: sx[0] loses sx[i], not the new element), the message exponent, the nonce.
// Record = KpAbeCiphertext: e1 = e_gg_alpha^secret * msg, e2 = g2*secret, rows (name, g1*(h(a) secret), g1_b*sx_i, g1_b2*(sx_i h(a)) + h_b*sx_i),
// sealed plaintext.  Every element is a fixed-base multiple of a public-key element: window-table launches for the whole batch.
bool encrypt_packed(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const std::vector<std::vector<std::string>>& sets, size_t n, const uint32_t* item_set,
                    const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  Timer tm("lsw::encrypt_packed");
  Engine::ArenaScope arena(eng);
  if (n && (!item_set || !pt_off || !out_off)) throw RabeError("lsw::encrypt_packed: null input");
  std::vector<size_t> fixed(sets.size());
  std::vector<std::vector<Fr>> hashes(sets.size());
  for (size_t s = 0; s < sets.size(); s++) {
    if (sets[s].empty()) throw RabeError("attributes or data empty");
    fixed[s] = 384 + 128 + 4 + 4;
    for (const auto& a : sets[s]) { fixed[s] += 4 + a.size() + 3 * 64; hashes[s].push_back(sha3_hash_fr(a)); }
  }
  for (size_t i = 0; i < n; i++) {
    if (item_set[i] >= sets.size()) throw RabeError("lsw::encrypt_packed: item_set out of range");
    if (pt_off[i + 1] <= pt_off[i]) throw RabeError("attributes or data empty");
  }
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + fixed[item_set[i]] + (pt_off[i + 1] - pt_off[i]) + 28;
  if (!out_buf || out_cap < out_off[n]) return false;
  if (!n) return true;
  std::vector<size_t> row_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) row_off[i + 1] = row_off[i] + sets[item_set[i]].size();
  const size_t total = row_off[n];
  // scalars: per item secret | msg exponent; per row h*secret | sx_i | sx_i*h
  uint8_t* h_k = eng.pinned(0, (2 * n + 3 * total) * 32 + 32);
  uint8_t* h_sec = h_k;
  uint8_t* h_rho = h_sec + n * 32;
  uint8_t* h_a = h_rho + n * 32;
  uint8_t* h_b = h_a + total * 32;
  uint8_t* h_c = h_b + total * 32;
  std::vector<std::array<uint8_t, 12>> nonces(n);
  draw_items(rng, n, [&](Rng& r, size_t i) {
    const Fr secret = r.next_fr();
    const auto& hs = hashes[item_set[i]];
    std::vector<Fr> sx{secret};
    for (size_t y = 0; y < hs.size(); y++) {
      sx.push_back(r.next_fr());
      sx[0] = fr_sub(sx[0], sx[y]);                 // :197-200 as it is written
    }
    for (size_t y = 0; y < hs.size(); y++) {
      const Fr a = fr_mul(hs[y], secret), c = fr_mul(sx[y], hs[y]);
      memcpy(h_a + 32 * (row_off[i] + y), a.l, 32);
      memcpy(h_b + 32 * (row_off[i] + y), sx[y].l, 32);
      memcpy(h_c + 32 * (row_off[i] + y), c.l, 32);
    }
    const Fr rho = r.next_fr();
    memcpy(h_sec + 32 * i, secret.l, 32);
    memcpy(h_rho + 32 * i, rho.l, 32);
    r.fill(nonces[i].data(), 12);
  });
  tm.lap("draws + scalars");
  const EncTables* tb;
  {
    std::string key((const char*)pk.g1.data(), 64);
    key.append((const char*)pk.g2.data(), 128).append((const char*)pk.g1_b.data(), 64).append((const char*)pk.g1_b2.data(), 64).append((const char*)pk.h_b.data(), 64);
    key.append((const char*)pk.e_gg_alpha.data(), 384);
    tb = (const EncTables*)eng.aux("lsw_enc_tables", key, make_enc_tables, &pk, destroy_enc_tables, 2);
  }
  rhip_gt_table* gen = eng.gt_generator_table();
  rhip_ctx* cx = eng.ctx();
  DBuf d_k(&eng, (2 * n + 3 * total) * 32 + 4), d_msg(&eng, n * 384), d_pw(&eng, n * 384), d_e1(&eng, n * 384), d_e2(&eng, n * 128), d_r1(&eng, total * 64 + 4),
      d_r2(&eng, total * 64 + 4), d_t1(&eng, total * 64 + 4), d_t2(&eng, total * 64 + 4), d_r3(&eng, total * 64 + 4);
  eng.check(rhip_upload_async(cx, d_k.ptr(), h_k, (2 * n + 3 * total) * 32), "upload");
  const rhip_fr* k_sec = d_k.as<rhip_fr>();
  const rhip_fr* k_rho = k_sec + n;
  const rhip_fr* k_a = k_rho + n;
  const rhip_fr* k_b = k_a + total;
  const rhip_fr* k_c = k_b + total;
  eng.check(rhip_gt_table_pow(cx, gen, n, k_rho, d_msg.as<rhip_gt>()), "rhip_gt_table_pow");                 // rng.gen::<Gt>() = e(g1, g2)^rho
  eng.check(rhip_gt_table_pow(cx, tb->egg, n, k_sec, d_pw.as<rhip_gt>()), "rhip_gt_table_pow");
  eng.check(rhip_gt_mul(cx, n, d_pw.as<rhip_gt>(), d_msg.as<rhip_gt>(), d_e1.as<rhip_gt>()), "rhip_gt_mul");
  eng.check(rhip_g2_table_mul(cx, tb->g2, n, k_sec, d_e2.as<rhip_g2>()), "rhip_g2_table_mul");
  eng.check(rhip_g1_table_mul(cx, tb->g1, total, k_a, d_r1.as<rhip_g1>()), "rhip_g1_table_mul");
  eng.check(rhip_g1_table_mul(cx, tb->g1_b, total, k_b, d_r2.as<rhip_g1>()), "rhip_g1_table_mul");
  eng.check(rhip_g1_table_mul(cx, tb->g1_b2, total, k_c, d_t1.as<rhip_g1>()), "rhip_g1_table_mul");
  eng.check(rhip_g1_table_mul(cx, tb->h_b, total, k_b, d_t2.as<rhip_g1>()), "rhip_g1_table_mul");
  eng.check(rhip_g1_add(cx, total, d_t1.as<rhip_g1>(), d_t2.as<rhip_g1>(), d_r3.as<rhip_g1>()), "rhip_g1_add");
  // records and sealing on the device (records.h); layout = attribute set
  std::vector<RecordLayout> layouts(sets.size());
  for (size_t p_ = 0; p_ < sets.size(); p_++) {
    RecordLayout& L = layouts[p_];
    L.src(0, 0, 384);
    L.src(1, 0, 128);
    L.u32((uint32_t)sets[p_].size());
    for (size_t y = 0; y < sets[p_].size(); y++) {
      L.str(sets[p_][y]);
      L.src(2, (uint32_t)(64 * y), 64);
      L.src(3, (uint32_t)(64 * y), 64);
      L.src(4, (uint32_t)(64 * y), 64);
    }
  }
  std::vector<uint64_t> src_off(5 * n);
  for (size_t i = 0; i < n; i++) {
    src_off[i] = 384ull * i; src_off[n + i] = 128ull * i;
    src_off[2 * n + i] = src_off[3 * n + i] = src_off[4 * n + i] = 64ull * row_off[i];
  }
  emit_sealed_records(eng, layouts, n, item_set, {d_e1.ptr(), d_e2.ptr(), d_r1.ptr(), d_r2.ptr(), d_r3.ptr()}, src_off, d_msg.ptr(),
                      (const uint8_t*)nonces.data(), pt_blob, pt_off, out_off, out_buf);
  tm.lap("device: group arithmetic, records, sealing; one copy out");
  return true;
}

// n calls of lsw::keygen (lsw/mod.rs:121-170).  Draw order per item: the gate coefficients of gen_shares_policy(alpha1), then one
// `random` per share (:136).  Record = KpAbeSecretKey: policy text, language, leaf count, per leaf (name, d1, d2, d3, d4, d5) with
// d3..d5 the identity for positive leaves and d1, d2 the identity for negative ones ("!x", :137-146: rhip_lsw_keygen_batch_signed).
bool keygen_packed(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const KpAbeMasterKey& msk, const std::vector<std::string>& policies,
                   PolicyLanguage language, size_t n, const uint32_t* item_policy, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  Timer tm("lsw::keygen_packed");
  Engine::ArenaScope arena(eng);
  eng.scrub_when_done();          // master-key-derived scalars pass through the staging buffers
  std::vector<std::shared_ptr<const FlatPolicy>> pols;
  std::vector<std::vector<std::string>> striped(policies.size());
  std::vector<size_t> fixed(policies.size());
  bool any_negative = false;
  for (size_t p = 0; p < policies.size(); p++) {
    pols.push_back(flat_policy(policies[p], language));
    any_negative = any_negative || pols[p]->has_negative;
    fixed[p] = 4 + policies[p].size() + 1 + 4;
    for (const auto& nc : pols[p]->leaf_name_col) { striped[p].push_back(remove_index(nc)); fixed[p] += 4 + striped[p].back().size() + 64 + 128 + 3 * 64; }
  }
  for (size_t i = 0; i < n; i++) if (item_policy[i] >= policies.size()) throw RabeError("lsw::keygen_packed: item_policy out of range");
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + fixed[item_policy[i]];
  if (!out_buf || out_cap < out_off[n]) return false;
  std::vector<uint32_t> leaf_off(n + 1, 0), coef_off(n + 1, 0), tree_leaf(n), tree_gate(n);
  for (size_t i = 0; i < n; i++) {
    leaf_off[i + 1] = leaf_off[i] + (uint32_t)pols[item_policy[i]]->leaf_name.size();
    coef_off[i + 1] = coef_off[i] + pols[item_policy[i]]->n_coef;
  }
  const size_t total = leaf_off[n], total_coef = coef_off[n];
  uint8_t* h_in = eng.pinned(0, 64 + (total_coef + total + 1) * 32);        // alpha1 | alpha2 | coefficients | randoms
  memcpy(h_in, msk.alpha1.l, 32);
  memcpy(h_in + 32, msk.alpha2.l, 32);
  uint8_t* h_coef = h_in + 64;
  uint8_t* h_rand = h_coef + total_coef * 32;
  draw_items(rng, n, [&](Rng& r, size_t i) {
    for (uint32_t c = coef_off[i]; c < coef_off[i + 1]; c++) { Fr a = r.next_fr(); memcpy(h_coef + 32 * (size_t)c, a.l, 32); }
    for (uint32_t y = leaf_off[i]; y < leaf_off[i + 1]; y++) { Fr a = r.next_fr(); memcpy(h_rand + 32 * (size_t)y, a.l, 32); }
  });
  tm.lap("policies + draws");
  rhip_ctx* cx = eng.ctx();
  std::string key((const char*)pk.g1.data(), 64);
  key.append((const char*)pk.g2.data(), 128);
  rhip_lsw_pk* dpk = (rhip_lsw_pk*)eng.aux("lsw_pk", key, make_pk, &pk, destroy_pk);
  DevTrees dt(eng, pols);
  for (size_t i = 0; i < n; i++) { tree_leaf[i] = dt.first_leaf[item_policy[i]]; tree_gate[i] = dt.first_gate[item_policy[i]]; }
  DBuf d_leaf_off = up32(eng, leaf_off), d_tl = up32(eng, tree_leaf), d_tg = up32(eng, tree_gate), d_coef_off = up32(eng, coef_off),
       d_in(&eng, 64 + (total_coef + total + 1) * 32), d_d1(&eng, total * 64 + 4), d_d2(&eng, total * 128 + 4);
  eng.check(rhip_upload_async(cx, d_in.ptr(), h_in, 64 + (total_coef + total) * 32), "upload");
  const rhip_fr* din = d_in.as<rhip_fr>();
  DBuf d_d345;
  if (!any_negative) {
    eng.check(rhip_lsw_keygen_batch(cx, dpk, n, total, d_leaf_off.as<uint32_t>(), d_tl.as<uint32_t>(), d_tg.as<uint32_t>(), dt.path_off.as<uint32_t>(),
                                    dt.path_gate.as<uint32_t>(), dt.path_x.as<uint32_t>(), dt.gate_k.as<uint32_t>(), dt.gate_coef_off.as<uint32_t>(),
                                    dt.leaf_hash.as<rhip_fr>(), din, din + 2, d_coef_off.as<uint32_t>(), din + 2 + total_coef, d_d1.as<rhip_g1>(),
                                    d_d2.as<rhip_g2>()), "rhip_lsw_keygen_batch");
  } else {                                   // negative leaves (lsw/mod.rs:137-146): d3, d4, d5 from the share, b and the master key's h_g1
    std::vector<uint32_t> leaf_neg;
    for (const auto& f : pols) for (const auto& nm : f->leaf_name) leaf_neg.push_back(is_negative(nm) ? 1u : 0u);
    DBuf d_neg = up32(eng, leaf_neg), d_b(&eng, msk.b.l, 32);
    d_d345 = DBuf(&eng, total * 192 + 4);
    rhip_g1* d3 = d_d345.as<rhip_g1>();
    eng.check(rhip_lsw_keygen_batch_signed(cx, dpk, n, total, d_leaf_off.as<uint32_t>(), d_tl.as<uint32_t>(), d_tg.as<uint32_t>(), dt.path_off.as<uint32_t>(),
                                           dt.path_gate.as<uint32_t>(), dt.path_x.as<uint32_t>(), dt.gate_k.as<uint32_t>(), dt.gate_coef_off.as<uint32_t>(),
                                           dt.leaf_hash.as<rhip_fr>(), d_neg.as<uint32_t>(), din, d_b.as<rhip_fr>(), (const rhip_g1*)msk.h_g1.data(),
                                           din + 2, d_coef_off.as<uint32_t>(), din + 2 + total_coef, d_d1.as<rhip_g1>(), d_d2.as<rhip_g2>(), d3,
                                           d3 + total, d3 + 2 * total), "rhip_lsw_keygen_batch_signed");
  }
  // the records are written on the device (records.h): policy text, leaf names and counts from the policy's template, the elements dropped
  // in (d3 .. d5 are the point at infinity -- zeros -- for a policy without negative leaves)
  std::vector<RecordLayout> layouts(policies.size());
  for (size_t p_ = 0; p_ < policies.size(); p_++) {
    RecordLayout& L = layouts[p_];
    L.str(policies[p_]);
    L.u8((language == PolicyLanguage::HumanPolicy) ? 1 : 0);
    L.u32((uint32_t)striped[p_].size());
    for (size_t y = 0; y < striped[p_].size(); y++) {
      L.str(striped[p_][y]);
      L.src(0, (uint32_t)(64 * y), 64);
      L.src(1, (uint32_t)(128 * y), 128);
      if (any_negative) { L.src(2, (uint32_t)(64 * y), 64); L.src(3, (uint32_t)(64 * y), 64); L.src(4, (uint32_t)(64 * y), 64); }
      else { const uint8_t zeros[192] = {0}; L.lit(zeros, 192); }
    }
  }
  std::vector<const void*> srcs{d_d1.ptr(), d_d2.ptr()};
  std::vector<uint64_t> src_off((any_negative ? 5 : 2) * n);
  for (size_t i = 0; i < n; i++) { src_off[i] = 64ull * leaf_off[i]; src_off[n + i] = 128ull * leaf_off[i]; }
  if (any_negative) {
    const uint8_t* d3 = d_d345.as<uint8_t>();
    srcs.push_back(d3); srcs.push_back(d3 + total * 64); srcs.push_back(d3 + 2 * total * 64);
    for (size_t i = 0; i < n; i++) src_off[2 * n + i] = src_off[3 * n + i] = src_off[4 * n + i] = 64ull * leaf_off[i];
  }
  emit_plain_records(eng, layouts, n, item_policy, srcs, src_off, out_off, out_buf);
  tm.lap("device: shares, fixed-base multiplications, records; one copy out");
  return true;
}

// n calls of lsw::decrypt (lsw/mod.rs:228-290): n keys (a blob of KpAbeSecretKey records) against ONE ciphertext (BASELINE config 4:
// a pre-made ciphertext, a fresh key per item).  Per distinct key policy: calc_pruned over the ciphertext's attributes and, for every
// pruned (name, name_col), the FIRST key row and the FIRST ciphertext row named `name` and the FIRST coefficient named name_col
// (:249-263).  A pruned negative attribute (the reference's TODO branch, :265-278) fails the item here: the object API reproduces it.
bool decrypt_packed(Engine& eng, const KpAbeCiphertext& ct, size_t n, const uint8_t* sk_blob, size_t sk_len, const uint64_t* sk_off, bool trusted,
                    int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors) {
  Timer tm("lsw::decrypt_packed");
  Engine::ArenaScope arena(eng);
  errors->assign(n, "");
  if (!sk_off || (n && !sk_blob)) throw RabeError("lsw::decrypt_packed: null input");
  (void)check_offsets(n, sk_off, sk_len, errors);
  const size_t pt_each = ct.ct.size() >= 28 ? ct.ct.size() - 28 : 0;
  if (!pt_buf || pt_cap < n * pt_each) return false;
  BlobGather gather(eng, sk_blob, sk_len);          // the blob starts for the device now, beside the parsing below (records.h)
  std::vector<std::string> attr;
  for (const auto& a : ct.ej) attr.push_back(a.name);
  struct Plan {
    std::shared_ptr<const FlatPolicy> flat; std::string err; std::vector<std::string> std_names;
    struct E { std::string name; uint32_t ct_row; Fr c; uint32_t std_sk_row; };
    std::vector<E> ent;
  };
  std::map<std::pair<int, std::string>, std::shared_ptr<Plan>> plans;
  std::mutex plans_mu;
  FastPlans<Plan> fast_plans;
  auto plan_of = [&](const std::string& text, PolicyLanguage lang) -> std::shared_ptr<Plan> {
    std::lock_guard<std::mutex> g(plans_mu);
    auto key = std::make_pair((int)lang, text);
    auto it = plans.find(key);
    if (it != plans.end()) return it->second;
    auto pl = std::make_shared<Plan>();
    try {
      pl->flat = flat_policy(text, lang);
      for (const auto& nc : pl->flat->leaf_name_col) pl->std_names.push_back(remove_index(nc));
      PrunedList list;
      if (!calc_pruned(attr, pl->flat->tree, &list)) throw RabeError("Error in lsw/decrypt: attributes do not match policy.");
      for (const auto& a : list) {
        if (is_negative(a.first)) throw RabeError("lsw::decrypt_packed: a negative attribute is selected; rabe_lsw_decrypt reproduces the reference's branch");
        size_t cr = 0, sr = 0, co = 0;
        while (cr < ct.ej.size() && ct.ej[cr].name != a.first) cr++;
        while (sr < pl->std_names.size() && pl->std_names[sr] != a.first) sr++;
        while (co < pl->flat->leaf_name_col.size() && pl->flat->leaf_name_col[co] != a.second) co++;
        if (cr == ct.ej.size() || co == pl->flat->leaf_name_col.size()) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
        pl->ent.push_back({a.first, (uint32_t)cr, pl->flat->leaf_coeff[co], (uint32_t)sr});
      }
    } catch (const std::exception& ex) {
      pl->err = ex.what();
      if (pl->err.empty()) pl->err = "policy error";
    }
    plans[key] = pl;
    return pl;
  };
  struct View { uint32_t rows; std::vector<const uint8_t*> d1, d2; std::shared_ptr<Plan> plan; std::vector<uint32_t> sk_row; bool standard; };
  std::vector<View> v(n);
  parallel_for(n, [&](size_t i) {
    if (!(*errors)[i].empty()) return;
    try {
      Cursor r{sk_blob + sk_off[i], sk_blob + sk_off[i + 1]};
      auto pol = r.str();
      const PolicyLanguage lang = *r.raw(1) ? PolicyLanguage::HumanPolicy : PolicyLanguage::JsonPolicy;
      const uint32_t rows = r.u32();
      if ((size_t)rows * 388 > (size_t)(r.end - r.p)) throw RabeError("deserialize: truncated input");
      v[i].rows = rows;
      v[i].d1.resize(rows);
      v[i].d2.resize(rows);
      std::vector<std::pair<const char*, uint32_t>> names(rows);
      for (uint32_t y = 0; y < rows; y++) { names[y] = r.str(); v[i].d1[y] = r.raw(64); v[i].d2[y] = r.raw(128); (void)r.raw(192); }
      auto pl = fast_plans.find(pol.first, pol.second, lang);
      if (!pl) { pl = plan_of(std::string(pol.first, pol.second), lang); fast_plans.put(pol.first, pol.second, lang, pl); }
      if (!pl->err.empty()) throw RabeError(pl->err);
      v[i].plan = pl;
      bool standard = rows == pl->std_names.size();
      for (uint32_t y = 0; y < rows && standard; y++) standard = same(names[y], pl->std_names[y]);
      v[i].standard = standard;
      if (!standard) {
        for (const auto& e : pl->ent) {
          uint32_t y = 0;
          while (y < rows && !same(names[y], e.name)) y++;
          if (y == rows) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
          v[i].sk_row.push_back(y);
        }
      } else {
        for (const auto& e : pl->ent) if (e.std_sk_row >= rows) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
      }
    } catch (const std::exception& ex) {
      (*errors)[i] = ex.what();
      if ((*errors)[i].empty()) (*errors)[i] = "malformed record";
    }
  });
  tm.lap("parse + plan");
  std::vector<size_t> live;
  std::vector<uint32_t> leaf_off{0}, pair_off{0}, sel_start, sel_sk, sel_ct;
  std::vector<Fr> sel_z;
  std::map<const Plan*, uint32_t> shared_start;
  size_t max_pairs = 1;
  for (size_t i = 0; i < n; i++) {
    if (!(*errors)[i].empty()) continue;
    live.push_back(i);
    leaf_off.push_back(leaf_off.back() + v[i].rows);
    const Plan& pl = *v[i].plan;
    if (v[i].standard) {
      auto it = shared_start.find(&pl);
      if (it == shared_start.end()) {
        it = shared_start.insert({&pl, (uint32_t)sel_sk.size()}).first;
        for (const auto& e : pl.ent) { sel_sk.push_back(e.std_sk_row); sel_ct.push_back(e.ct_row); sel_z.push_back(e.c); }
      }
      sel_start.push_back(it->second);
    } else {
      sel_start.push_back((uint32_t)sel_sk.size());
      for (size_t e = 0; e < pl.ent.size(); e++) { sel_sk.push_back(v[i].sk_row[e]); sel_ct.push_back(pl.ent[e].ct_row); sel_z.push_back(pl.ent[e].c); }
    }
    const uint32_t m = (uint32_t)pl.ent.size();
    pair_off.push_back(pair_off.back() + m + 1);
    if ((size_t)m + 1 > max_pairs) max_pairs = m + 1;
  }
  const size_t m_items = live.size();
  DBuf d_out(&eng, m_items * 384 + 4);
  std::unique_ptr<MemberChecks> mc;
  std::unique_ptr<WalkedG2> walked;          // read after the open (records.h: retract_item): it and what it refers to outlive the block
  std::vector<uint32_t> walked_idx, walked_off;
  DBuf d_d2;
  if (m_items) {
    const size_t total = leaf_off[m_items];
    DBuf d_d1(&eng, total * 64 + 4);
    d_d2 = DBuf(&eng, total * 128 + 4);
    std::vector<uint64_t> dst_off(2 * m_items);
    for (size_t j = 0; j < m_items; j++) {
      const View& w = v[live[j]];
      const uint8_t* rec = sk_blob + sk_off[live[j]];
      dst_off[j] = 64ull * leaf_off[j]; dst_off[m_items + j] = 128ull * leaf_off[j];
      int shape = w.standard ? gather.find(w.plan.get()) : -1;
      if (shape < 0) {
        std::vector<RecordLayout::Part> parts;
        for (uint32_t y = 0; y < w.rows; y++) {
          parts.push_back({(uint32_t)(w.d1[y] - rec), 64, 0, 64 * y});
          parts.push_back({(uint32_t)(w.d2[y] - rec), 128, 1, 128 * y});
        }
        shape = (int)gather.add_shape(w.standard ? (const void*)w.plan.get() : nullptr, std::move(parts));
      }
      gather.item(sk_off[live[j]], (uint32_t)shape);
    }
    tm.lap("shapes");
    rhip_ctx* cx = eng.ctx();
    std::vector<uint8_t> e1j, e1rep(m_items * 384);
    for (const auto& a : ct.ej) e1j.insert(e1j.end(), a.e1.begin(), a.e1.end());
    for (size_t j = 0; j < m_items; j++) memcpy(e1rep.data() + 384 * j, ct.e1.data(), 384);

    DBuf d_leaf_off = up32(eng, leaf_off), d_pair_off = up32(eng, pair_off),
        d_sel_start = up32(eng, sel_start), d_sel_sk = up32(eng, sel_sk), d_sel_ct = up32(eng, sel_ct), d_sel_z = up_bytes(eng, flatten_fr(sel_z)),
        d_e1 = up_bytes(eng, e1rep), d_e2(&eng, ct.e2.data(), 128), d_e1j = up_bytes(eng, e1j);
    gather.run({d_d1.ptr(), d_d2.ptr()}, dst_off);
    std::string e2_key((const char*)ct.e2.data(), 128);         // the ciphertext's prepared e2 lines: kept across calls
    rhip_g2_lines* lines = (rhip_g2_lines*)eng.aux("lsw_e2_lines", e2_key, make_e2_lines, &e2_key, destroy_e2_lines, 4);
    if (!trusted) {
      mc.reset(new MemberChecks(eng));
      mc->add(1, d_d1.ptr(), total, d_leaf_off.as<uint32_t>(), m_items);
      // D2 of every selected key leaf is the walking argument of a pairing (e2 replays prepared lines): membership out of the decrypt's
      // own Miller loops, the stand-alone test for the leaves the selection left out (common.h: WalkedG2)
      if (walk_checks() && lines) {
        bool all = true;          // concluded from the counts for standard-layout records only (see bsw::decrypt_packed)
        for (size_t j = 0; j < m_items && all; j++) all = v[live[j]].standard && pair_off[j + 1] - pair_off[j] - 1 == leaf_off[j + 1] - leaf_off[j];
        if (!all) {
          walked_off.push_back(0);
          for (size_t j = 0; j < m_items; j++) {
            const uint32_t mj = pair_off[j + 1] - pair_off[j] - 1;
            for (uint32_t e = 0; e < mj; e++) walked_idx.push_back(leaf_off[j] + sel_sk[sel_start[j] + e]);
            walked_off.push_back((uint32_t)walked_idx.size());
          }
        }
        walked.reset(new WalkedG2(eng, *mc, d_d2.ptr(), total, d_leaf_off.as<uint32_t>(), leaf_off, 1, all ? nullptr : &walked_idx, all ? nullptr : &walked_off));
      } else {
        mc->add(2, d_d2.ptr(), total, d_leaf_off.as<uint32_t>(), m_items);
      }
    }
    if (walked) walked->arm();
    // one ciphertext for all keys: the scaled ciphertext rows are computed once per selection entry (keys that share a policy share them)
    int32_t rc = rhip_lsw_decrypt_batch_one_ct(cx, m_items, max_pairs, pair_off[m_items], sel_sk.size(), d_pair_off.as<uint32_t>(), d_sel_start.as<uint32_t>(),
                                               d_sel_sk.as<uint32_t>(), d_sel_ct.as<uint32_t>(), d_sel_z.as<rhip_fr>(), d_e1.as<rhip_gt>(), d_e2.as<rhip_g2>(),
                                               d_e1j.as<rhip_g1>(), d_d1.as<rhip_g1>(), d_d2.as<rhip_g2>(), d_leaf_off.as<uint32_t>(), (const uint32_t*)nullptr, lines,
                                               d_out.as<rhip_gt>());
    eng.check(rc, "rhip_lsw_decrypt_batch");
    if (mc) {
      mc->collect();
      const auto& ok1 = mc->ok(0);
      std::vector<uint8_t> ok2(m_items, 1);
      if (!walked) { const auto& e = mc->ok(1); ok2.assign(e.begin(), e.end()); }
      for (size_t j = 0; j < m_items; j++)
        if (!ok1[j] || !ok2[j]) (*errors)[live[j]] = "deserialize: a key element is not a group member (FieldError::NotMember)";
    }
  }
  // the ONE ciphertext's sealed data, opened under every key's Gt on the device (KDF + AES-GCM; the Gt never leaves HBM)
  DBuf d_sealed(&eng, ct.ct.data(), ct.ct.size());
  std::vector<uint64_t> sealed_off(m_items, 0);
  std::vector<uint32_t> sealed_len(m_items, (uint32_t)ct.ct.size());
  open_sealed_records(eng, n, live, d_out.ptr(), d_sealed.as<uint8_t>(), sealed_off, sealed_len, status, pt_buf, pt_off, errors);
  if (walked) {
    std::vector<uint8_t> ok2;
    walked->finish(&ok2);
    for (size_t j = 0; j < m_items; j++)
      if (!ok2[j]) retract_item(live[j], "deserialize: a key element is not a group member (FieldError::NotMember)", status, pt_buf, pt_off, errors);
  }
  tm.lap(trusted ? "device: gather, pairings, open" : "device: gather, pairings, open; membership beside");
  return true;
}
}  // namespace lsw

// ================================================================================================================= AW11
namespace aw11 {
namespace {
std::string upper(const std::string& s) {
  std::string o = s;
  for (auto& c : o) if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');
  return o;
}
struct PkArg { const Aw11GlobalKey* gk; std::vector<const Aw11PkAttr*> attrs; };
void* make_pk(Engine& eng, const void* arg) {
  const PkArg& a = *(const PkArg*)arg;
  std::vector<uint8_t> egg, g2y;
  for (const auto* t : a.attrs) { egg.insert(egg.end(), t->egg_alpha.begin(), t->egg_alpha.end()); g2y.insert(g2y.end(), t->g2_y.begin(), t->g2_y.end()); }
  rhip_aw11_pk* d = nullptr;
  eng.check(rhip_aw11_pk_create(eng.ctx(), (const rhip_g1*)a.gk->g1.data(), (const rhip_g2*)a.gk->g2.data(), a.attrs.size(), (const rhip_gt*)egg.data(),
                                (const rhip_g2*)g2y.data(), &d), "rhip_aw11_pk_create");
  return d;
}
void destroy_pk(void* h) { rhip_aw11_pk_destroy((rhip_aw11_pk*)h); }
}  // namespace

// n calls of aw11::encrypt (aw11/mod.rs:241-289) under the same authority keys.  Draw order per item: s (:257), the gate
// coefficients of the s-shares then of the 0-shares (:259-260), msg (:262), one r_x per share (:267), the nonce.  Record =
// Aw11Ciphertext: policy, c_0, row count, per row (NAME_COL upper-cased, c1, c2, c3), sealed data.  A leaf whose attribute no
// authority key lists is silently dropped by the reference (:269-271); here it is an error (use rabe_aw11_encrypt for that case).
bool encrypt_packed(Engine& eng, Rng& rng, const Aw11GlobalKey& gk, const std::vector<const Aw11PublicKey*>& pks, const std::vector<std::string>& policies,
                    PolicyLanguage language, size_t n, const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf,
                    size_t out_cap, uint64_t* out_off) {
  Timer tm("aw11::encrypt_packed");
  Engine::ArenaScope arena(eng);
  PkArg arg{&gk, {}};
  std::string key((const char*)gk.g1.data(), 64);
  key.append((const char*)gk.g2.data(), 128);
  for (const auto* pk : pks) for (const auto& t : pk->attr) { arg.attrs.push_back(&t); key.append((const char*)t.egg_alpha.data(), 384).append((const char*)t.g2_y.data(), 128); }
  if (arg.attrs.empty()) throw RabeError("aw11::encrypt_packed: no authority attributes");
  std::vector<std::shared_ptr<const FlatPolicy>> pols;
  std::vector<std::vector<std::string>> row_name(policies.size());
  std::vector<uint32_t> leaf_attr;
  std::vector<size_t> fixed(policies.size());
  for (size_t p = 0; p < policies.size(); p++) {
    pols.push_back(flat_policy(policies[p], language));
    (void)calculate_msp(pols[p]->tree);                    // built and unused in the reference (:253-255) -- but it must not panic
    fixed[p] = 4 + policies[p].size() + 1 + 384 + 4 + 4;
    for (const auto& nc : pols[p]->leaf_name_col) {
      const std::string up = upper(nc), want = remove_index(up);
      size_t a = 0;
      while (a < arg.attrs.size() && arg.attrs[a]->name != want) a++;
      if (a == arg.attrs.size()) throw RabeError("aw11::encrypt_packed: attribute " + want + " is in no authority key (rabe_aw11_encrypt drops such rows like the reference)");
      leaf_attr.push_back((uint32_t)a);
      row_name[p].push_back(up);
      fixed[p] += 4 + up.size() + 384 + 128 + 128;
    }
  }
  for (size_t i = 0; i < n; i++) if (item_policy[i] >= policies.size()) throw RabeError("aw11::encrypt_packed: item_policy out of range");
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + fixed[item_policy[i]] + (pt_off[i + 1] - pt_off[i]) + 28;
  if (!out_buf || out_cap < out_off[n]) return false;
  std::vector<uint32_t> row_off(n + 1, 0), coef_off(n + 1, 0), tree_leaf(n), tree_gate(n), n_coef(n);
  for (size_t i = 0; i < n; i++) {
    const FlatPolicy& f = *pols[item_policy[i]];
    row_off[i + 1] = row_off[i] + (uint32_t)f.leaf_name.size();
    coef_off[i + 1] = coef_off[i] + 2 * f.n_coef;
    n_coef[i] = f.n_coef;
  }
  const size_t total = row_off[n], total_coef = coef_off[n];
  uint8_t* h_in = eng.pinned(0, (2 * n + total_coef + total + 1) * 32);      // s | msg exponent | coefficients | r_x
  uint8_t* h_s = h_in;
  uint8_t* h_rho = h_in + n * 32;
  uint8_t* h_coef = h_in + 2 * n * 32;
  uint8_t* h_rand = h_coef + total_coef * 32;
  std::vector<std::array<uint8_t, 12>> nonces(n);
  draw_items(rng, n, [&](Rng& r, size_t i) {
    Fr s = r.next_fr();
    memcpy(h_s + 32 * i, s.l, 32);
    for (uint32_t c = coef_off[i]; c < coef_off[i + 1]; c++) { Fr a = r.next_fr(); memcpy(h_coef + 32 * (size_t)c, a.l, 32); }
    Fr rho = r.next_fr();
    memcpy(h_rho + 32 * i, rho.l, 32);
    for (uint32_t y = row_off[i]; y < row_off[i + 1]; y++) { Fr a = r.next_fr(); memcpy(h_rand + 32 * (size_t)y, a.l, 32); }
    r.fill(nonces[i].data(), 12);
  });
  tm.lap("policies + draws");
  rhip_ctx* cx = eng.ctx();
  rhip_aw11_pk* dpk = (rhip_aw11_pk*)eng.aux("aw11_pk", key, make_pk, &arg, destroy_pk, 2);
  DevTrees dt(eng, pols);
  for (size_t i = 0; i < n; i++) { tree_leaf[i] = dt.first_leaf[item_policy[i]]; tree_gate[i] = dt.first_gate[item_policy[i]]; }
  DBuf d_row_off = up32(eng, row_off), d_tl = up32(eng, tree_leaf), d_tg = up32(eng, tree_gate), d_nc = up32(eng, n_coef), d_coef_off = up32(eng, coef_off),
       d_leaf_attr = up32(eng, leaf_attr), d_in(&eng, (2 * n + total_coef + total + 1) * 32), d_msg(&eng, n * 384), d_c0(&eng, n * 384),
       d_c1(&eng, total * 384 + 4), d_c2(&eng, total * 128 + 4), d_c3(&eng, total * 128 + 4);
  eng.check(rhip_upload_async(cx, d_in.ptr(), h_in, (2 * n + total_coef + total) * 32), "upload");
  const rhip_fr* din = d_in.as<rhip_fr>();
  eng.check(rhip_gt_table_pow(cx, eng.gt_generator_table(), n, din + n, d_msg.as<rhip_gt>()), "rhip_gt_table_pow");
  eng.check(rhip_aw11_encrypt_batch(cx, dpk, n, total, d_row_off.as<uint32_t>(), d_tl.as<uint32_t>(), d_tg.as<uint32_t>(), d_nc.as<uint32_t>(),
                                    dt.path_off.as<uint32_t>(), dt.path_gate.as<uint32_t>(), dt.path_x.as<uint32_t>(), dt.gate_k.as<uint32_t>(),
                                    dt.gate_coef_off.as<uint32_t>(), d_leaf_attr.as<uint32_t>(), din, din + 2 * n, d_coef_off.as<uint32_t>(),
                                    din + 2 * n + total_coef, d_msg.as<rhip_gt>(), d_c0.as<rhip_gt>(), d_c1.as<rhip_gt>(), d_c2.as<rhip_g2>(),
                                    d_c3.as<rhip_g2>()), "rhip_aw11_encrypt_batch");
  // records and sealing on the device (records.h)
  std::vector<RecordLayout> layouts(policies.size());
  for (size_t p_ = 0; p_ < policies.size(); p_++) {
    RecordLayout& L = layouts[p_];
    L.str(policies[p_]);
    L.u8((language == PolicyLanguage::HumanPolicy) ? 1 : 0);
    L.src(0, 0, 384);
    L.u32((uint32_t)row_name[p_].size());
    for (size_t y = 0; y < row_name[p_].size(); y++) {
      L.str(row_name[p_][y]);
      L.src(1, (uint32_t)(384 * y), 384);
      L.src(2, (uint32_t)(128 * y), 128);
      L.src(3, (uint32_t)(128 * y), 128);
    }
  }
  std::vector<uint64_t> src_off(4 * n);
  for (size_t i = 0; i < n; i++) { src_off[i] = 384ull * i; src_off[n + i] = 384ull * row_off[i]; src_off[2 * n + i] = src_off[3 * n + i] = 128ull * row_off[i]; }
  emit_sealed_records(eng, layouts, n, item_policy, {d_c0.ptr(), d_c1.ptr(), d_c2.ptr(), d_c3.ptr()}, src_off, d_msg.ptr(), (const uint8_t*)nonces.data(),
                      pt_blob, pt_off, out_off, out_buf);
  tm.lap("device: group arithmetic, records, sealing; one copy out");
  return true;
}

namespace {
void* make_g1_table(Engine& eng, const void* arg) {
  const G1& g1 = *(const G1*)arg;
  rhip_g1_table* t = nullptr;
  eng.check(rhip_g1_table_create(eng.ctx(), (const rhip_g1*)g1.data(), &t), "rhip_g1_table_create");
  const int32_t rc = rhip_g1_table_add_w16(eng.ctx(), t);
  if (rc) { rhip_g1_table_destroy(t); eng.check(rc, "rhip_g1_table_add_w16"); }
  return t;
}
void destroy_g1_table(void* h) { rhip_g1_table_destroy((rhip_g1_table*)h); }
}  // namespace
// n calls of aw11::keygen (aw11/mod.rs:165-231) by ONE authority: user i (gids[i]) gets the attribute list sets[item_set[i]].  No randomness:
// K_x = g1*alpha_x + H(gid)*y_x with H(gid) = g1*h(gid) is g1*(alpha_x + h(gid) y_x) -- one window-table launch for the whole batch.
// Record = Aw11SecretKey: gid, rows (upper-cased attribute name, K_x).  An attribute the authority does not own is the reference's
// unwrap panic (:213); an empty gid or list its RabeError.
bool keygen_packed(Engine& eng, const Aw11GlobalKey& gk, const Aw11MasterKey& msk, const std::vector<std::string>& gids,
                   const std::vector<std::vector<std::string>>& sets, size_t n, const uint32_t* item_set, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  Timer tm("aw11::keygen_packed");
  Engine::ArenaScope arena(eng);
  eng.scrub_when_done();          // master-key-derived scalars pass through the staging buffers
  if (n && (!item_set || !out_off)) throw RabeError("aw11::keygen_packed: null input");
  if (gids.size() != n) throw RabeError("aw11::keygen_packed: one gid per item");
  std::vector<std::vector<const Aw11MkAttr*>> auth(sets.size());
  std::vector<std::vector<std::string>> names(sets.size());
  std::vector<size_t> fixed(sets.size());
  for (size_t s = 0; s < sets.size(); s++) {
    if (sets[s].empty()) throw RabeError("empty _attributes");
    fixed[s] = 4 + 4;
    for (const auto& a : sets[s]) {
      if (a.empty()) throw RabeError("empty _attributes");
      const Aw11MkAttr* hit = nullptr;
      for (const auto& m : msk.attr) if (m.name == a) { hit = &m; break; }
      if (!hit) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
      auth[s].push_back(hit);
      names[s].push_back(upper(hit->name));
      fixed[s] += 4 + names[s].back().size() + 64;
    }
  }
  for (size_t i = 0; i < n; i++) {
    if (item_set[i] >= sets.size()) throw RabeError("aw11::keygen_packed: item_set out of range");
    if (gids[i].empty()) throw RabeError("empty _name");
  }
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + fixed[item_set[i]] + gids[i].size();
  if (!out_buf || out_cap < out_off[n]) return false;
  if (!n) return true;
  std::vector<size_t> row_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) row_off[i + 1] = row_off[i] + sets[item_set[i]].size();
  const size_t total = row_off[n];
  uint8_t* h_k = eng.pinned(0, total * 32 + 32);
  parallel_for(n, [&](size_t i) {
    const Fr hg = sha3_hash_fr(gids[i]);
    const auto& au = auth[item_set[i]];
    for (size_t y = 0; y < au.size(); y++) {
      const Fr k = fr_add(au[y]->alpha, fr_mul(hg, au[y]->y));
      memcpy(h_k + 32 * (row_off[i] + y), k.l, 32);
    }
  });
  tm.lap("scalars");
  const rhip_g1_table* tb = (const rhip_g1_table*)eng.aux("aw11_g1_table", std::string((const char*)gk.g1.data(), 64), make_g1_table, &gk.g1, destroy_g1_table, 4);
  rhip_ctx* cx = eng.ctx();
  DBuf d_k(&eng, total * 32 + 4), d_out(&eng, total * 64 + 4);
  eng.check(rhip_upload_async(cx, d_k.ptr(), h_k, total * 32), "upload");
  eng.check(rhip_g1_table_mul(cx, tb, total, d_k.as<rhip_fr>(), d_out.as<rhip_g1>()), "rhip_g1_table_mul");
  uint8_t* h_o = eng.pinned(1, total * 64 + 4);
  eng.check(rhip_download_async(cx, h_o, d_out.ptr(), total * 64), "download");
  eng.check(rhip_sync(cx), "rhip_sync");
  tm.lap("device + copies");
  parallel_for(n, [&](size_t i) {
    const auto& nm = names[item_set[i]];
    uint8_t* w = out_buf + out_off[i];
    put_u32(w, (uint32_t)gids[i].size()); w += 4;
    memcpy(w, gids[i].data(), gids[i].size()); w += gids[i].size();
    put_u32(w, (uint32_t)nm.size()); w += 4;
    for (size_t y = 0; y < nm.size(); y++) {
      put_u32(w, (uint32_t)nm[y].size()); w += 4;
      memcpy(w, nm[y].data(), nm[y].size()); w += nm[y].size();
      memcpy(w, h_o + 64 * (row_off[i] + y), 64); w += 64;
    }
  });
  tm.lap("assembly");
  return true;
}

// n calls of aw11::decrypt (aw11/mod.rs:298-366) with one key.  Per distinct policy: traverse_policy, calc_pruned, and per pruned
// (name, name_col) the FIRST key attribute named `name`, the FIRST ciphertext row named name_col (literal spelling, :325-333) and the
// FIRST coefficient named name_col.
bool decrypt_packed(Engine& eng, const Aw11GlobalKey& gk, const Aw11SecretKey& sk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off,
                    bool trusted, int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors) {
  Timer tm("aw11::decrypt_packed");
  Engine::ArenaScope arena(eng);
  errors->assign(n, "");
  if (!ct_off || (n && !ct_blob)) throw RabeError("aw11::decrypt_packed: null input");
  const uint64_t span = check_offsets(n, ct_off, ct_len, errors);
  if (!pt_buf || pt_cap < span) return false;
  BlobGather gather(eng, ct_blob, ct_len);          // the blob starts for the device now, beside the parsing below (records.h)
  std::vector<std::string> str_attr;
  for (const auto& a : sk.attr) str_attr.push_back(a.first);
  struct Plan {
    std::shared_ptr<const FlatPolicy> flat; std::string err; std::vector<std::string> std_names;
    struct E { std::string name_col; uint32_t sk_row; Fr c; uint32_t std_ct_row; };
    std::vector<E> ent;
  };
  std::map<std::pair<int, std::string>, std::shared_ptr<Plan>> plans;
  std::mutex plans_mu;
  FastPlans<Plan> fast_plans;
  auto plan_of = [&](const std::string& text, PolicyLanguage lang) -> std::shared_ptr<Plan> {
    std::lock_guard<std::mutex> g(plans_mu);
    auto key = std::make_pair((int)lang, text);
    auto it = plans.find(key);
    if (it != plans.end()) return it->second;
    auto pl = std::make_shared<Plan>();
    try {
      pl->flat = flat_policy(text, lang);
      for (const auto& nc : pl->flat->leaf_name_col) pl->std_names.push_back(upper(nc));
      if (!traverse_policy(str_attr, pl->flat->tree)) throw RabeError("Error: attributes in sk do not match policy in ct.");
      PrunedList list;
      if (!calc_pruned(str_attr, pl->flat->tree, &list)) throw RabeError("Error in aw11/decrypt: attributes in sk do not match policy in ct.");
      for (const auto& cur : list) {
        size_t sr = 0, co = 0, cr = 0;
        while (sr < sk.attr.size() && sk.attr[sr].first != cur.first) sr++;
        while (co < pl->flat->leaf_name_col.size() && pl->flat->leaf_name_col[co] != cur.second) co++;
        while (cr < pl->std_names.size() && pl->std_names[cr] != cur.second) cr++;
        if (sr == sk.attr.size() || co == pl->flat->leaf_name_col.size()) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
        pl->ent.push_back({cur.second, (uint32_t)sr, pl->flat->leaf_coeff[co], (uint32_t)cr});
      }
    } catch (const std::exception& ex) {
      pl->err = ex.what();
      if (pl->err.empty()) pl->err = "policy error";
    }
    plans[key] = pl;
    return pl;
  };
  struct View { const uint8_t* c0; uint32_t rows; std::vector<const uint8_t*> c1, c2, c3; std::shared_ptr<Plan> plan; std::vector<uint32_t> ct_row; bool standard; };
  std::vector<View> v(n);
  std::vector<Sealed> sealed(n);
  parallel_for(n, [&](size_t i) {
    if (!(*errors)[i].empty()) return;
    try {
      Cursor r{ct_blob + ct_off[i], ct_blob + ct_off[i + 1]};
      auto pol = r.str();
      const PolicyLanguage lang = *r.raw(1) ? PolicyLanguage::HumanPolicy : PolicyLanguage::JsonPolicy;
      v[i].c0 = r.raw(384);
      const uint32_t rows = r.u32();
      if ((size_t)rows * 644 > (size_t)(r.end - r.p)) throw RabeError("deserialize: truncated input");
      v[i].rows = rows;
      v[i].c1.resize(rows); v[i].c2.resize(rows); v[i].c3.resize(rows);
      std::vector<std::pair<const char*, uint32_t>> names(rows);
      for (uint32_t y = 0; y < rows; y++) { names[y] = r.str(); v[i].c1[y] = r.raw(384); v[i].c2[y] = r.raw(128); v[i].c3[y] = r.raw(128); }
      sealed[i].len = r.u32();
      sealed[i].p = r.raw(sealed[i].len);
      auto pl = fast_plans.find(pol.first, pol.second, lang);
      if (!pl) { pl = plan_of(std::string(pol.first, pol.second), lang); fast_plans.put(pol.first, pol.second, lang, pl); }
      if (!pl->err.empty()) throw RabeError(pl->err);
      v[i].plan = pl;
      bool standard = rows == pl->std_names.size();
      for (uint32_t y = 0; y < rows && standard; y++) standard = same(names[y], pl->std_names[y]);
      v[i].standard = standard;
      for (size_t e = 0; e < pl->ent.size(); e++) {
        uint32_t y = standard ? pl->ent[e].std_ct_row : 0;
        if (!standard) while (y < rows && !same(names[y], pl->ent[e].name_col)) y++;
        if (y >= rows) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
        if (!standard) v[i].ct_row.push_back(y);
      }
    } catch (const std::exception& ex) {
      (*errors)[i] = ex.what();
      if ((*errors)[i].empty()) (*errors)[i] = "malformed record";
    }
  });
  tm.lap("parse + plan");
  std::vector<size_t> live;
  std::vector<uint32_t> row_off{0}, pair_off{0}, sel_start, sel_ct, sel_sk;
  std::vector<Fr> sel_z;
  std::map<const Plan*, uint32_t> shared_start;
  size_t max_pairs = 1;
  for (size_t i = 0; i < n; i++) {
    if (!(*errors)[i].empty()) continue;
    live.push_back(i);
    row_off.push_back(row_off.back() + v[i].rows);
    const Plan& pl = *v[i].plan;
    if (v[i].standard) {
      auto it = shared_start.find(&pl);
      if (it == shared_start.end()) {
        it = shared_start.insert({&pl, (uint32_t)sel_ct.size()}).first;
        for (const auto& e : pl.ent) { sel_ct.push_back(e.std_ct_row); sel_sk.push_back(e.sk_row); sel_z.push_back(e.c); }
      }
      sel_start.push_back(it->second);
    } else {
      sel_start.push_back((uint32_t)sel_ct.size());
      for (size_t e = 0; e < pl.ent.size(); e++) { sel_ct.push_back(v[i].ct_row[e]); sel_sk.push_back(pl.ent[e].sk_row); sel_z.push_back(pl.ent[e].c); }
    }
    const uint32_t m = (uint32_t)pl.ent.size();
    pair_off.push_back(pair_off.back() + m + 1);
    if ((size_t)m + 1 > max_pairs) max_pairs = m + 1;
  }
  const size_t m_items = live.size();
  std::vector<uint64_t> sealed_off(m_items);
  std::vector<uint32_t> sealed_len(m_items);
  DBuf d_out(&eng, m_items * 384 + 4);
  std::unique_ptr<MemberChecks> mc;
  std::unique_ptr<WalkedG2> walked;          // read after the open (records.h: retract_item): it and what it refers to outlive the block
  std::vector<uint32_t> walked_idx, walked_off;
  DBuf d_c2;
  if (m_items) {
    const size_t total = row_off[m_items];
    DBuf d_c0(&eng, m_items * 384), d_c1(&eng, total * 384 + 4), d_c3(&eng, total * 128 + 4);
    d_c2 = DBuf(&eng, total * 128 + 4);
    std::vector<uint64_t> dst_off(4 * m_items);
    for (size_t j = 0; j < m_items; j++) {
      const View& w = v[live[j]];
      const uint8_t* rec = ct_blob + ct_off[live[j]];
      sealed_off[j] = (uint64_t)(sealed[live[j]].p - ct_blob);
      sealed_len[j] = sealed[live[j]].len;
      dst_off[j] = 384ull * j; dst_off[m_items + j] = 384ull * row_off[j]; dst_off[2 * m_items + j] = dst_off[3 * m_items + j] = 128ull * row_off[j];
      int shape = w.standard ? gather.find(w.plan.get()) : -1;
      if (shape < 0) {
        std::vector<RecordLayout::Part> parts;
        parts.push_back({(uint32_t)(w.c0 - rec), 384, 0, 0});
        for (uint32_t y = 0; y < w.rows; y++) {
          parts.push_back({(uint32_t)(w.c1[y] - rec), 384, 1, 384 * y});
          parts.push_back({(uint32_t)(w.c2[y] - rec), 128, 2, 128 * y});
          parts.push_back({(uint32_t)(w.c3[y] - rec), 128, 3, 128 * y});
        }
        shape = (int)gather.add_shape(w.standard ? (const void*)w.plan.get() : nullptr, std::move(parts));
      }
      gather.item(ct_off[live[j]], (uint32_t)shape);
    }
    tm.lap("shapes");
    rhip_ctx* cx = eng.ctx();
    G1 hash = eng.g1_mul({gk.g1}, {sha3_hash_fr(sk.gid)})[0];            // H(gid) = g1 * h(gid), hashed inside decrypt (:318)
    std::vector<uint8_t> kk;
    for (const auto& a : sk.attr) kk.insert(kk.end(), a.second.begin(), a.second.end());
    std::vector<uint32_t> sk_attr_off{0, (uint32_t)sk.attr.size()}, sk_idx(m_items, 0);
    DBuf d_row_off = up32(eng, row_off),
        d_pair_off = up32(eng, pair_off), d_sel_start = up32(eng, sel_start), d_sel_ct = up32(eng, sel_ct), d_sel_sk = up32(eng, sel_sk),
        d_sel_z = up_bytes(eng, flatten_fr(sel_z)), d_hash(&eng, hash.data(), 64), d_kk = up_bytes(eng, kk), d_sk_attr_off = up32(eng, sk_attr_off),
        d_sk_idx = up32(eng, sk_idx);
    gather.run({d_c0.ptr(), d_c1.ptr(), d_c2.ptr(), d_c3.ptr()}, dst_off);
    if (!trusted) {
      mc.reset(new MemberChecks(eng));
      mc->add(3, d_c0.ptr(), m_items); mc->add(3, d_c1.ptr(), total, d_row_off.as<uint32_t>(), m_items);
      mc->add(2, d_c3.ptr(), total, d_row_off.as<uint32_t>(), m_items);          // C3 enters its pairing as a SUM: every term keeps the stand-alone test
      // C2 of every selected row is the walking argument of a pairing; one more argument walks per item (the sum of the C3 terms)
      if (walk_checks()) {
        bool all = true;          // concluded from the counts for standard-layout records only (see bsw::decrypt_packed)
        for (size_t j = 0; j < m_items && all; j++) all = v[live[j]].standard && pair_off[j + 1] - pair_off[j] - 1 == row_off[j + 1] - row_off[j];
        if (!all) {
          walked_off.push_back(0);
          for (size_t j = 0; j < m_items; j++) {
            const uint32_t mj = pair_off[j + 1] - pair_off[j] - 1;
            for (uint32_t e = 0; e < mj; e++) walked_idx.push_back(row_off[j] + sel_ct[sel_start[j] + e]);
            walked_off.push_back((uint32_t)walked_idx.size());
          }
        }
        walked.reset(new WalkedG2(eng, *mc, d_c2.ptr(), total, d_row_off.as<uint32_t>(), row_off, 1, all ? nullptr : &walked_idx, all ? nullptr : &walked_off, 1));
      } else {
        mc->add(2, d_c2.ptr(), total, d_row_off.as<uint32_t>(), m_items);
      }
    }
    if (walked) walked->arm();
    int32_t rc = rhip_aw11_decrypt_batch(cx, m_items, max_pairs, pair_off[m_items], sel_ct.size(), d_pair_off.as<uint32_t>(), d_sel_start.as<uint32_t>(),
                                         d_sel_ct.as<uint32_t>(), d_sel_sk.as<uint32_t>(), d_sel_z.as<rhip_fr>(), d_c0.as<rhip_gt>(), d_c1.as<rhip_gt>(),
                                         d_c2.as<rhip_g2>(), d_c3.as<rhip_g2>(), d_row_off.as<uint32_t>(), d_hash.as<rhip_g1>(), d_kk.as<rhip_g1>(),
                                         d_sk_attr_off.as<uint32_t>(), d_sk_idx.as<uint32_t>(), d_out.as<rhip_gt>());
    eng.check(rc, "rhip_aw11_decrypt_batch");
    if (mc) {
      mc->collect();
      const auto &ok0 = mc->ok(0), &ok1 = mc->ok(1), &ok3 = mc->ok(2);
      std::vector<uint8_t> ok2(m_items, 1);
      if (!walked) { const auto& e = mc->ok(3); ok2.assign(e.begin(), e.end()); }
      for (size_t j = 0; j < m_items; j++) {
        const bool bad = !ok0[j] || !ok1[j] || !ok2[j] || !ok3[j];
        if (bad) (*errors)[live[j]] = "deserialize: a ciphertext element is not a group member (FieldError::NotMember)";
      }
    }
  }
  // KDF + AES-GCM open on the device: the decrypted Gt never leaves HBM; plaintext bytes come back in one copy
  open_sealed_records(eng, n, live, d_out.ptr(), gather.dev_blob(), sealed_off, sealed_len, status, pt_buf, pt_off, errors);
  if (walked) {
    std::vector<uint8_t> ok2;
    walked->finish(&ok2);
    for (size_t j = 0; j < m_items; j++)
      if (!ok2[j]) retract_item(live[j], "deserialize: a ciphertext element is not a group member (FieldError::NotMember)", status, pt_buf, pt_off, errors);
  }
  tm.lap(trusted ? "device: gather, pairings, open" : "device: gather, pairings, open; membership beside");
  return true;
}
}  // namespace aw11

// ================================================================================================================= GHW11
namespace ghw11 {
namespace {
void* make_tk_lines(Engine& eng, const void* arg) {          // arg: k_z | l_z | k_x[0] | k_x[1] ... (128 B each)
  const std::string& pts = *(const std::string*)arg;
  DBuf d(&eng, pts.data(), pts.size());
  rhip_g2_lines* lines = nullptr;
  eng.check(rhip_g2_lines_prepare(eng.ctx(), pts.size() / 128, d.as<rhip_g2>(), &lines), "rhip_g2_lines_prepare");
  return lines;
}
void destroy_tk_lines(void* h) { rhip_g2_lines_destroy((rhip_g2_lines*)h); }
}  // namespace

// n calls of ghw11::transform (ghw11/mod.rs:227-295) under ONE transform key -- the outsourced half of a decryption, what a server
// holding users' transform keys runs.  Records in: Ghw11Ciphertext (policy, c, c1, rows (name, c_i, d_i), sealed
// data); records out: Ghw11TransformCiphertext = c | t, 768 bytes per item at out_buf + 768 i (zeros where status[i] = -1).
// Per distinct policy: traverse_policy, calc_pruned, and per pruned (name, name_col) the FIRST coefficient named name_col, the FIRST
// key attribute named `name`, the FIRST ciphertext row named name_col (:259-281); a missing one is the reference's unwrap panic.
// Every G2 argument is the key's: the batch replays prepared lines (kept across calls) and does no G2 arithmetic.
bool transform_packed(Engine& eng, const Ghw11TransformKey& tk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                      int32_t* status, uint8_t* out_buf, size_t out_cap, std::vector<std::string>* errors) {
  Timer tm("ghw11::transform_packed");
  Engine::ArenaScope arena(eng);
  errors->assign(n, "");
  if (!ct_off || (n && !ct_blob) || !status) throw RabeError("ghw11::transform_packed: null input");
  if (!out_buf || out_cap < n * 768) return false;
  (void)check_offsets(n, ct_off, ct_len, errors);
  std::vector<std::string> attr;
  for (const auto& a : tk.attr_key_z) attr.push_back(a.string);
  struct Plan {
    std::shared_ptr<const FlatPolicy> flat; std::string err;
    struct E { std::string name_col; uint32_t tk_attr; Fr w; uint32_t std_ct_row; };
    std::vector<E> ent;
  };
  std::map<std::pair<int, std::string>, std::shared_ptr<Plan>> plans;
  std::mutex plans_mu;
  FastPlans<Plan> fast_plans;
  auto plan_of = [&](const std::string& text, PolicyLanguage lang) -> std::shared_ptr<Plan> {
    std::lock_guard<std::mutex> g(plans_mu);
    auto key = std::make_pair((int)lang, text);
    auto it = plans.find(key);
    if (it != plans.end()) return it->second;
    auto pl = std::make_shared<Plan>();
    try {
      pl->flat = flat_policy(text, lang);
      const auto& names = pl->flat->leaf_name_col;
      if (!traverse_policy(attr, pl->flat->tree)) throw RabeError("Error: attributes in tk do not match policy in ct.");
      PrunedList list;
      if (!calc_pruned(attr, pl->flat->tree, &list)) throw RabeError("Error in Ghw11/decrypt: attributes in sk do not match policy in ct.");
      for (const auto& cur : list) {
        size_t a = 0, co = 0;
        while (a < tk.attr_key_z.size() && tk.attr_key_z[a].string != cur.first) a++;
        while (co < names.size() && names[co] != cur.second) co++;
        if (a == tk.attr_key_z.size() || co == names.size()) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
        pl->ent.push_back({cur.second, (uint32_t)a, pl->flat->leaf_coeff[co], (uint32_t)co});
      }
    } catch (const std::exception& ex) {
      pl->err = ex.what();
      if (pl->err.empty()) pl->err = "policy error";
    }
    plans[key] = pl;
    return pl;
  };
  struct View { const uint8_t* c; const uint8_t* c1; uint32_t rows; std::vector<const uint8_t*> ci, di; std::shared_ptr<Plan> plan;
                std::vector<uint32_t> ct_row; bool standard; };
  std::vector<View> v(n);
  parallel_for(n, [&](size_t i) {
    if (!(*errors)[i].empty()) return;
    try {
      Cursor r{ct_blob + ct_off[i], ct_blob + ct_off[i + 1]};
      auto pol = r.str();
      const PolicyLanguage lang = *r.raw(1) ? PolicyLanguage::HumanPolicy : PolicyLanguage::JsonPolicy;
      v[i].c = r.raw(384);
      v[i].c1 = r.raw(64);
      const uint32_t rows = r.u32();
      if ((size_t)rows * 132 > (size_t)(r.end - r.p)) throw RabeError("deserialize: truncated input");
      v[i].rows = rows;
      v[i].ci.resize(rows);
      v[i].di.resize(rows);
      std::vector<std::pair<const char*, uint32_t>> names(rows);
      for (uint32_t y = 0; y < rows; y++) { names[y] = r.str(); v[i].ci[y] = r.raw(64); v[i].di[y] = r.raw(64); }
      const uint32_t dl = r.u32();
      (void)r.raw(dl);                                   // the sealed data stays with the client (decrypt_out)
      auto pl = fast_plans.find(pol.first, pol.second, lang);
      if (!pl) { pl = plan_of(std::string(pol.first, pol.second), lang); fast_plans.put(pol.first, pol.second, lang, pl); }
      if (!pl->err.empty()) throw RabeError(pl->err);
      v[i].plan = pl;
      const auto& std_names = pl->flat->leaf_name_col;
      bool standard = rows == std_names.size();
      for (uint32_t y = 0; y < rows && standard; y++) standard = same(names[y], std_names[y]);
      v[i].standard = standard;
      if (!standard) {
        for (const auto& e : pl->ent) {
          uint32_t y = 0;
          while (y < rows && !same(names[y], e.name_col)) y++;
          if (y == rows) throw RabeError("called `Option::unwrap()` on a `None` value");
          v[i].ct_row.push_back(y);
        }
      }
    } catch (const std::exception& ex) {
      (*errors)[i] = ex.what();
      if ((*errors)[i].empty()) (*errors)[i] = "malformed record";
    }
  });
  tm.lap("parse + plan");
  std::vector<size_t> live;
  std::vector<uint32_t> row_off{0}, pair_off{0}, sel_start, sel_ct, sel_tk;
  std::vector<Fr> sel_w;
  std::map<const Plan*, uint32_t> shared_start;
  size_t max_pairs = 2;
  for (size_t i = 0; i < n; i++) {
    if (!(*errors)[i].empty()) continue;
    live.push_back(i);
    row_off.push_back(row_off.back() + v[i].rows);
    const Plan& pl = *v[i].plan;
    if (v[i].standard) {
      auto it = shared_start.find(&pl);
      if (it == shared_start.end()) {
        it = shared_start.insert({&pl, (uint32_t)sel_ct.size()}).first;
        for (const auto& e : pl.ent) { sel_ct.push_back(e.std_ct_row); sel_tk.push_back(e.tk_attr); sel_w.push_back(e.w); }
      }
      sel_start.push_back(it->second);
    } else {
      sel_start.push_back((uint32_t)sel_ct.size());
      for (size_t e = 0; e < pl.ent.size(); e++) { sel_ct.push_back(v[i].ct_row[e]); sel_tk.push_back(pl.ent[e].tk_attr); sel_w.push_back(pl.ent[e].w); }
    }
    const uint32_t m = (uint32_t)pl.ent.size();
    pair_off.push_back(pair_off.back() + m + 2);
    if ((size_t)m + 2 > max_pairs) max_pairs = m + 2;
  }
  const size_t m_items = live.size();
  uint8_t* h_out = nullptr;
  if (m_items) {
    const size_t total = row_off[m_items];
    uint8_t* h_l = eng.pinned(1, total * 128 + 4);
    uint8_t* h_x = eng.pinned(2, m_items * (64 + 384 + 384));
    parallel_for(m_items, [&](size_t j) {
      const View& w = v[live[j]];
      memcpy(h_x + 64 * j, w.c1, 64);
      memcpy(h_x + m_items * 64 + 384 * j, w.c, 384);
      for (uint32_t y = 0; y < w.rows; y++) {
        memcpy(h_l + (size_t)(row_off[j] + y) * 64, w.ci[y], 64);
        memcpy(h_l + total * 64 + (size_t)(row_off[j] + y) * 64, w.di[y], 64);
      }
    });
    tm.lap("pack");
    rhip_ctx* cx = eng.ctx();
    std::string key((const char*)tk.k_z.data(), 128);
    key.append((const char*)tk.l_z.data(), 128);
    for (const auto& a : tk.attr_key_z) key.append((const char*)a.k_x.data(), 128);
    rhip_g2_lines* lines = (rhip_g2_lines*)eng.aux("ghw11_tk_lines", key, make_tk_lines, &key, destroy_tk_lines, 4);
    DBuf d_c1(&eng, m_items * 64), d_c(&eng, m_items * 384), d_ci(&eng, total * 64 + 4), d_di(&eng, total * 64 + 4), d_row_off = up32(eng, row_off),
        d_pair_off = up32(eng, pair_off), d_sel_start = up32(eng, sel_start), d_sel_ct = up32(eng, sel_ct), d_sel_tk = up32(eng, sel_tk),
        d_sel_w = up_bytes(eng, flatten_fr(sel_w)), d_out(&eng, m_items * 384);
    eng.check(rhip_upload_async(cx, d_c1.ptr(), h_x, m_items * 64), "upload");
    eng.check(rhip_upload_async(cx, d_ci.ptr(), h_l, total * 64), "upload");
    eng.check(rhip_upload_async(cx, d_di.ptr(), h_l + total * 64, total * 64), "upload");
    std::unique_ptr<MemberChecks> mc;
    if (!trusted) {
      eng.check(rhip_upload_async(cx, d_c.ptr(), h_x + m_items * 64, m_items * 384), "upload");
      mc.reset(new MemberChecks(eng));
      mc->add(1, d_c1.ptr(), m_items); mc->add(1, d_ci.ptr(), total, d_row_off.as<uint32_t>(), m_items);
      mc->add(1, d_di.ptr(), total, d_row_off.as<uint32_t>(), m_items); mc->add(3, d_c.ptr(), m_items);
    }
    int32_t rc = rhip_ghw11_transform_batch(cx, m_items, max_pairs, pair_off[m_items], sel_ct.size(), d_pair_off.as<uint32_t>(), d_sel_start.as<uint32_t>(),
                                            d_sel_ct.as<uint32_t>(), d_sel_tk.as<uint32_t>(), d_sel_w.as<rhip_fr>(), d_c1.as<rhip_g1>(), d_ci.as<rhip_g1>(),
                                            d_di.as<rhip_g1>(), d_row_off.as<uint32_t>(), lines, d_out.as<rhip_gt>());
    h_out = h_x + m_items * 448;
    if (rc == RHIP_OK) rc = rhip_download_async(cx, h_out, d_out.ptr(), m_items * 384);
    if (rc == RHIP_OK) rc = rhip_sync(cx);
    eng.check(rc, "rhip_ghw11_transform_batch");
    if (mc) {
      mc->collect();
      const auto &ok_c1 = mc->ok(0), &ok_ci = mc->ok(1), &ok_di = mc->ok(2), &ok_c = mc->ok(3);
      for (size_t j = 0; j < m_items; j++)
        if (!ok_c1[j] || !ok_ci[j] || !ok_di[j] || !ok_c[j]) (*errors)[live[j]] = "deserialize: a ciphertext element is not a group member (FieldError::NotMember)";
    }
  }
  tm.lap(trusted ? "device + copies" : "device + copies, membership beside");
  std::vector<size_t> slot(n, (size_t)-1);
  for (size_t j = 0; j < m_items; j++) slot[live[j]] = j;
  parallel_for(n, [&](size_t i) {
    uint8_t* o = out_buf + 768 * i;
    if (!(*errors)[i].empty() || slot[i] == (size_t)-1) { memset(o, 0, 768); status[i] = -1; return; }
    memcpy(o, v[i].c, 384);
    memcpy(o + 384, h_out + 384 * slot[i], 384);
    status[i] = 0;
  });
  tm.lap("assembly");
  return true;
}
}  // namespace ghw11

}  // namespace schemes
}  // namespace rabe
