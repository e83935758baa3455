// Host layer implementation: Engine wrapper + rabe::schemes::{ac17,bsw,lsw,aw11} over the C ABI.
// String / Fr work happens here exactly where the reference does it on the CPU; every group operation is a
// batched launch of the HIP engine (no CPU group arithmetic exists in this layer).
#include "schemes.h"
#include "records.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <exception>
#include <functional>
#include <future>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <unordered_map>

#include <errno.h>
#include <stdio.h>
#include <sys/random.h>

namespace rabe {

using namespace host;

// ------------------------------------------------------------------------------------------------ Rng / Engine
void OsRng::fill(uint8_t* out, size_t n) {
  while (n) {
    if (pool_pos == sizeof(pool)) {
      size_t off = 0;
      while (off < sizeof(pool)) {
        ssize_t r = getrandom(pool + off, sizeof(pool) - off, 0);
        if (r < 0 && errno == EINTR) continue;            // a signal during the read: try again
        if (r <= 0) throw RabeError("getrandom failed");
        off += (size_t)r;
      }
      pool_pos = 0;
    }
    size_t k = std::min(n, sizeof(pool) - pool_pos);
    memcpy(out, pool + pool_pos, k);
    memset(pool + pool_pos, 0, k);            // consumed randomness does not linger
    pool_pos += k;
    out += k;
    n -= k;
  }
}

BatchRng::BatchRng() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (host::aeshw::available()) {
    uint8_t seed[40];
    os.fill(seed, sizeof seed);
    auto* k = new host::aeshw::Key;
    host::aeshw::expand(seed, k);
    memcpy(&iv_hi, seed + 32, 8);
    explicit_bzero(seed, sizeof seed);
    key = k;
  }
#endif
}
BatchRng::~BatchRng() {
  memset(pool, 0, sizeof pool);
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (key) {
    explicit_bzero(key, sizeof(host::aeshw::Key));
    delete (host::aeshw::Key*)key;
  }
#endif
}
void BatchRng::fill(uint8_t* out, size_t n) {
  if (!key) { os.fill(out, n); return; }
  while (n) {
    if (pool_pos == sizeof(pool)) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
      host::aeshw::keystream(*(const host::aeshw::Key*)key, iv_hi, &counter, pool, sizeof(pool) / 16);
#endif
      pool_pos = 0;
    }
    const size_t k = std::min(n, sizeof(pool) - pool_pos);
    memcpy(out, pool + pool_pos, k);
    memset(pool + pool_pos, 0, k);            // consumed randomness does not linger
    pool_pos += k;
    out += k;
    n -= k;
  }
}

int& Engine::tl_lane() {
  static thread_local int lane = 0;
  return lane;
}
Engine::LaneScope::LaneScope(int lane) : prev(Engine::tl_lane()) { Engine::tl_lane() = lane; }
Engine::LaneScope::~LaneScope() { Engine::tl_lane() = prev; }
Engine::Engine(int device) : device_(device) {
  lanes_.emplace_back(new Lane());
  int32_t rc = rhip_ctx_create(device, &lanes_[0]->ctx);
  if (rc != RHIP_OK) throw RabeError(std::string("no usable HIP device: ") + rhip_last_error(nullptr));
}
void Engine::ensure_lanes(size_t count) {
  std::lock_guard<std::recursive_mutex> g(mu_);
  while (lanes_.size() < count) {
    std::unique_ptr<Lane> l(new Lane());
    check(rhip_ctx_create(device_, &l->ctx), "rhip_ctx_create (lane)");
    lanes_.push_back(std::move(l));
  }
}
size_t Engine::lane_count() {
  std::lock_guard<std::recursive_mutex> g(mu_);
  return lanes_.size();
}
rhip_ac17_pk* Engine::ac17_pk(const G1& g, const std::vector<G2>& h_a, const std::vector<Gt>& e_gh_ka) {
  if (h_a.size() != 3 || e_gh_ka.size() != 2) throw RabeError("malformed Ac17PublicKey: h_a must have 3 and e_gh_ka 2 elements");
  auto fha = flatten(h_a), fe = flatten(e_gh_ka);
  std::string key((const char*)g.data(), g.size());
  key.append((const char*)fha.data(), fha.size());
  key.append((const char*)fe.data(), fe.size());
  std::lock_guard<std::recursive_mutex> lk(mu_);
  auto it = pk17_.find(key);
  if (it != pk17_.end()) { it->second.used = ++use_clock_; return it->second.h; }
  if (pk17_.size() >= 4) {                       // bounded: each entry holds ~1.7 GB of tables; the least recently used one goes
    auto lru = pk17_.begin();
    for (auto c = pk17_.begin(); c != pk17_.end(); ++c) if (c->second.used < lru->second.used) lru = c;
    retire(lru->second.h, [](void* h) { rhip_ac17_pk_destroy((rhip_ac17_pk*)h); });
    pk17_.erase(lru);
  }
  rhip_ac17_pk* dpk = nullptr;
  check(rhip_ac17_pk_create(ctx(), (const rhip_g1*)g.data(), (const rhip_g2*)fha.data(), (const rhip_gt*)fe.data(), &dpk), "rhip_ac17_pk_create");
  // signed 20-bit windows for g (13 instead of 16 additions per row element, +0.44 GB): the device-level default of bench.py;
  // RABE_G_WINDOW=16 keeps the plain 16-bit table, larger values trade more memory (include/rabe_hip.h)
  int w = 20;
  if (const char* env = getenv("RABE_G_WINDOW")) w = atoi(env);
  if (w > 16) check(rhip_ac17_pk_set_g_window(ctx(), dpk, w), "rhip_ac17_pk_set_g_window");
  pk17_[key] = Pk17{dpk, ++use_clock_};
  return dpk;
}
uint8_t* Engine::pinned(int slot, size_t bytes) {
  Lane& l = *lanes_[cur_lane()];
  if (l.pin_bytes[slot] < bytes) {
    if (l.pin[slot]) rhip_host_free(l.ctx, l.pin[slot]);
    l.pin[slot] = nullptr;
    l.pin_bytes[slot] = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    check(rhip_host_alloc(l.ctx, want, &l.pin[slot]), "rhip_host_alloc");
    l.pin_bytes[slot] = want;
  }
  if (bytes > l.pin_touched[slot]) l.pin_touched[slot] = bytes;
  return (uint8_t*)l.pin[slot];
}
void Engine::scrub_when_done() { lanes_[cur_lane()]->scrub = 31; }
// device arena, pinned slot 0 and the ParamPack block (pinned slot 3: emit_sealed_records stages plaintexts of up to 8 MB through it)
void Engine::scrub_session_when_done() { lanes_[cur_lane()]->scrub |= 3 | 16; }
void Engine::pinned_reserve(size_t bytes) {
  Lane& l = *lanes_[cur_lane()];
  if (l.pin3_used + bytes <= l.pin_bytes[3]) return;
  // grow NOW, at a point where nothing of this call reads the block on a helper thread: what is already in it moves along
  check(rhip_sync(l.ctx), "rhip_sync");
  if (l.side) check(rhip_sync(l.side), "rhip_sync");
  void* fresh = nullptr;
  const size_t want = l.pin3_used + bytes + bytes / 8 + (1u << 20);
  check(rhip_host_alloc(l.ctx, want, &fresh), "rhip_host_alloc");
  if (l.pin[3] && l.pin3_used) memcpy(fresh, l.pin[3], l.pin3_used);
  if (l.pin[3]) rhip_host_free(l.ctx, l.pin[3]);
  l.pin[3] = fresh;
  l.pin_bytes[3] = want;
}
uint8_t* Engine::pinned_bump(size_t bytes) {
  Lane& l = *lanes_[cur_lane()];
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (l.pin3_used + need > l.pin_bytes[3]) {
    if (l.pin3_hold > 0) throw RabeError("pinned_bump: the staging block would move while a deferred copy still reads it (pinned_reserve first)");
    check(rhip_sync(l.ctx), "rhip_sync");            // copies out of the old block have finished: it can go
    if (l.side) check(rhip_sync(l.side), "rhip_sync");
    if (l.pin[3]) rhip_host_free(l.ctx, l.pin[3]);
    l.pin[3] = nullptr;
    l.pin_bytes[3] = 0;
    const size_t want = 2 * (l.pin3_used + need) + (1u << 20);
    check(rhip_host_alloc(l.ctx, want, &l.pin[3]), "rhip_host_alloc");
    l.pin_bytes[3] = want;
    l.pin3_used = 0;
  }
  uint8_t* p = (uint8_t*)l.pin[3] + l.pin3_used;
  l.pin3_used += need;
  return p;
}
void Engine::pinned_hold(int delta) { lanes_[cur_lane()]->pin3_hold += delta; }
Engine::ArenaScope::ArenaScope(Engine& eng) : e(eng) {
  Lane& l = *e.lanes_[e.cur_lane()];
  if (l.arena_depth++ == 0) { l.arena_used = 0; l.arena_want = 0; l.pin3_used = 0; l.scrub = 0; for (auto& t : l.pin_touched) t = 0; }
}
rhip_ctx* Engine::side_ctx() {
  Lane& l = *lanes_[cur_lane()];
  if (!l.side) check(rhip_ctx_create(device_, &l.side), "rhip_ctx_create (side)");
  return l.side;
}
// RABE_NO_WALK_CHECKS=1: every G2 element gets the stand-alone subgroup test again (A/B runs)
bool walk_checks() { static const bool off = getenv("RABE_NO_WALK_CHECKS") != nullptr; return !off; }
MemberChecks::MemberChecks(Engine& eng) : eng_(eng), cx_(getenv("RABE_MEMBER_INLINE") ? eng.ctx() : eng.side_ctx()) {
  if (cx_ != eng.ctx()) eng.check(rhip_ctx_wait_for(cx_, eng.ctx()), "rhip_ctx_wait_for");
}
void MemberChecks::add(int which, const void* dev, size_t count, const uint32_t* dev_seg_off, size_t n_seg, uint32_t scale) {
  const bool fold = dev_seg_off != nullptr;
  flags_.emplace_back(fold ? n_seg : count, 1u);
  dev_.emplace_back(&eng_, (fold ? n_seg : count) * 4);
  if (!count) return;
  if (which == 3 && cx_ != eng_.ctx() && count >= 4096) {          // Gt: beside the decrypt's final exponentiation (see collect()); a small batch's Miller kernel leaves the chip empty anyway
    if (!requested_) { eng_.check(rhip_ctx_release_after_miller(eng_.ctx(), cx_), "rhip_ctx_release_after_miller"); requested_ = true; }
    later_.push_back({flags_.size() - 1, dev, count, dev_seg_off, n_seg, scale});
    return;
  }
  launch(which, flags_.size() - 1, dev, count, dev_seg_off, n_seg, scale);
}
void MemberChecks::launch(int which, size_t k, const void* dev, size_t count, const uint32_t* dev_seg_off, size_t n_seg, uint32_t scale) {
  const bool fold = dev_seg_off != nullptr;
  DBuf per_element;
  if (fold) per_element = DBuf(&eng_, count * 4);
  uint32_t* ok = fold ? per_element.as<uint32_t>() : dev_[k].as<uint32_t>();
  int32_t rc = which == 1 ? rhip_g1_on_curve(cx_, count, (const rhip_g1*)dev, ok)
             : which == 2 ? rhip_g2_in_subgroup(cx_, count, (const rhip_g2*)dev, ok)
             : which == 4 ? rhip_g2_on_curve(cx_, count, (const rhip_g2*)dev, ok)
                          : rhip_gt_is_member(cx_, count, (const rhip_gt*)dev, ok);
  eng_.check(rc, "membership pass");
  if (fold) {
    eng_.check(rhip_flags_all(cx_, n_seg, dev_seg_off, scale, ok, dev_[k].as<uint32_t>()), "rhip_flags_all");
    scratch_.push_back(std::move(per_element));            // lives until collect(): the side stream still reads it
  }
}
MemberChecks::~MemberChecks() {
  // a request no decrypt consumed (an exception on the way) must not hold the side stream at some later launch
  if (requested_) (void)rhip_ctx_release_after_miller(eng_.ctx(), nullptr);
}
void MemberChecks::add_g2_at(const void* dev, const std::vector<uint32_t>& idx) {
  flags_.emplace_back(idx.size(), 1u);
  dev_.emplace_back(&eng_, idx.size() * 4);
  if (idx.empty()) return;
  // the index list goes up on the main context (asynchronously, from the lane's pinned staging); the side context waits for it
  DBuf d_idx(&eng_, idx.size() * 4);
  uint8_t* const h = eng_.pinned_bump(idx.size() * 4);
  memcpy(h, idx.data(), idx.size() * 4);
  eng_.check(rhip_upload_async(eng_.ctx(), d_idx.ptr(), h, idx.size() * 4), "upload");
  if (cx_ != eng_.ctx()) eng_.check(rhip_ctx_wait_for(cx_, eng_.ctx()), "rhip_ctx_wait_for");
  eng_.check(rhip_g2_in_subgroup_at(cx_, idx.size(), d_idx.as<uint32_t>(), (const rhip_g2*)dev, dev_.back().as<uint32_t>()), "rhip_g2_in_subgroup_at");
  scratch_.push_back(std::move(d_idx));
}
WalkedG2::WalkedG2(Engine& eng, MemberChecks& mc, const void* dev_g2, size_t count, const uint32_t* dev_seg_off, const std::vector<uint32_t>& seg_off,
                   uint32_t scale, const std::vector<uint32_t>* walked_idx, const std::vector<uint32_t>* walked_off, uint32_t extra_walks)
    : eng_(eng), mc_(mc), dev_(dev_g2), count_(count), n_seg_(seg_off.size() - 1), seg_off_(seg_off), scale_(scale), extra_(extra_walks), widx_(walked_idx),
      woff_(walked_off) {
  k_curve_ = mc_.add_count();
  mc_.add(4, dev_g2, count, dev_seg_off, n_seg_, scale);
  if (woff_) {                                   // the complement of the walked elements: stand-alone, beside the decrypt
    std::vector<uint8_t> walked(count, 0);
    for (uint32_t e : *widx_) if (e < count) walked[e] = 1;
    std::vector<uint32_t> rest;
    for (size_t j = 0; j < n_seg_; j++)
      for (size_t e = (size_t)seg_off_[j] * scale; e < (size_t)seg_off_[j + 1] * scale; e++)
        if (!walked[e]) { rest.push_back((uint32_t)e); rest_item_.push_back((uint32_t)j); }
    if (!rest.empty()) {
      k_rest_ = mc_.add_count();
      mc_.add_g2_at(dev_g2, rest);
      has_rest_ = true;
    }
  }
  d_verdicts_ = DBuf(&eng_, 2 * n_seg_ * 4);
  eng_.check(rhip_memset_async(eng_.ctx(), d_verdicts_.ptr(), 0, 2 * n_seg_ * 4), "memset");
}
void WalkedG2::arm() {
  eng_.check(rhip_ctx_collect_walk_verdicts(eng_.ctx(), d_verdicts_.as<uint32_t>(), d_verdicts_.as<uint32_t>() + n_seg_), "rhip_ctx_collect_walk_verdicts");
  armed_ = true;
}
WalkedG2::~WalkedG2() {
  if (armed_) (void)rhip_ctx_collect_walk_verdicts(eng_.ctx(), nullptr, nullptr);
}
void WalkedG2::finish(std::vector<uint8_t>* ok) {
  eng_.check(rhip_ctx_collect_walk_verdicts(eng_.ctx(), nullptr, nullptr), "rhip_ctx_collect_walk_verdicts");          // a path that did not consume the request
  armed_ = false;
  std::vector<uint32_t> v(2 * n_seg_);
  eng_.check(rhip_download(eng_.ctx(), v.data(), d_verdicts_.ptr(), v.size() * 4), "download");
  ok->assign(n_seg_, 1);
  const auto& curve = mc_.ok(k_curve_);
  for (size_t j = 0; j < n_seg_; j++) if (!curve[j] || v[j]) (*ok)[j] = 0;
  if (has_rest_) {
    const auto& r = mc_.ok(k_rest_);
    for (size_t t = 0; t < r.size(); t++) if (!r[t]) (*ok)[rest_item_[t]] = 0;
  }
  // items whose walk examined fewer arguments than expected: their walked elements get the stand-alone test now
  std::vector<uint32_t> again, again_item;
  for (size_t j = 0; j < n_seg_; j++) {
    if (!(*ok)[j]) continue;
    const size_t lo = (size_t)seg_off_[j] * scale_, hi = (size_t)seg_off_[j + 1] * scale_;
    const size_t expected = woff_ ? (size_t)((*woff_)[j + 1] - (*woff_)[j]) : hi - lo;
    if (v[n_seg_ + j] == expected + extra_) continue;
    if (woff_) for (uint32_t t = (*woff_)[j]; t < (*woff_)[j + 1]; t++) { again.push_back((*widx_)[t]); again_item.push_back((uint32_t)j); }
    else for (size_t e = lo; e < hi; e++) { again.push_back((uint32_t)e); again_item.push_back((uint32_t)j); }
  }
  if (!again.empty()) {
    DBuf d_idx(&eng_, again.size() * 4), d_ok(&eng_, again.size() * 4);
    eng_.check(rhip_upload(eng_.ctx(), d_idx.ptr(), again.data(), again.size() * 4), "upload");
    eng_.check(rhip_g2_in_subgroup_at(eng_.ctx(), again.size(), d_idx.as<uint32_t>(), (const rhip_g2*)dev_, d_ok.as<uint32_t>()), "rhip_g2_in_subgroup_at");
    std::vector<uint32_t> r(again.size());
    eng_.check(rhip_download(eng_.ctx(), r.data(), d_ok.ptr(), r.size() * 4), "download");
    for (size_t t = 0; t < r.size(); t++) if (!r[t]) (*ok)[again_item[t]] = 0;
  }
}
void MemberChecks::collect() {
  for (const auto& d : later_) launch(3, d.k, d.dev, d.count, d.seg, d.n_seg, d.scale);
  later_.clear();
  if (requested_) {            // consumed by the decrypt the caller queued -- or by nothing (a path without that launch): withdrawn, the checks ran at once
    (void)rhip_ctx_release_after_miller(eng_.ctx(), nullptr);
    requested_ = false;
  }
  for (size_t k = 0; k < flags_.size(); k++)
    if (!flags_[k].empty()) eng_.check(rhip_download_async(cx_, flags_[k].data(), dev_[k].ptr(), flags_[k].size() * 4), "download");
  eng_.check(rhip_sync(cx_), "rhip_sync (membership pass)");
  scratch_.clear();
}
Engine::ArenaScope::~ArenaScope() {
  Lane& l = *e.lanes_[e.cur_lane()];
  if (--l.arena_depth) return;
  rhip_sync(l.ctx);                               // nothing of this call may still read the block when the next call reuses it
  if (l.side) rhip_sync(l.side);
  if (l.scrub) {                                  // scrub_when_done(): secrets do not outlive the call in staging memory
    if ((l.scrub & 1) && l.arena && l.arena_used) { rhip_memset_async(l.ctx, l.arena, 0, l.arena_used); rhip_sync(l.ctx); }
    for (int s = 0; s < 3; s++)
      if ((l.scrub & (2 << s)) && l.pin[s] && l.pin_touched[s]) explicit_bzero(l.pin[s], l.pin_touched[s] < l.pin_bytes[s] ? l.pin_touched[s] : l.pin_bytes[s]);
    if ((l.scrub & 16) && l.pin[3] && l.pin3_used) explicit_bzero(l.pin[3], l.pin3_used < l.pin_bytes[3] ? l.pin3_used : l.pin_bytes[3]);
    l.scrub = 0;
  }
  // grow for the next call -- up to a cap (RABE_ARENA_MAX_GB, default 16): the block is never given back, so one huge batch must not
  // pin a large part of the device for the life of the engine; beyond the cap the buffers that do not fit are plain allocations again
  static const size_t cap = [] { const char* e = getenv("RABE_ARENA_MAX_GB"); const long g = e ? atol(e) : 16; return (size_t)(g > 0 ? g : 0) << 30; }();
  size_t want = l.arena_want + l.arena_want / 4 + (1u << 20);
  if (want > cap) want = cap;
  if (want > l.arena_bytes) {
    if (l.arena) rhip_free(l.ctx, l.arena);
    l.arena = nullptr;
    l.arena_bytes = 0;
    void* p = nullptr;
    if (rhip_malloc(l.ctx, want, &p) == RHIP_OK) { l.arena = (uint8_t*)p; l.arena_bytes = want; }
  }
}
void* Engine::arena_take(size_t bytes) {
  Lane& l = *lanes_[cur_lane()];
  static const bool off = getenv("RABE_NO_ARENA") != nullptr;          // diagnostics: every buffer its own hipMalloc again
  if (!l.arena_depth || off) return nullptr;
  const size_t need = (bytes + 255) & ~(size_t)255;
  l.arena_want += need;
  if (l.arena_used + need > l.arena_bytes) return nullptr;
  void* p = l.arena + l.arena_used;
  l.arena_used += need;
  return p;
}
bool Engine::arena_owns(const void* p) const {
  for (const auto& l : lanes_)
    if (l->arena && (const uint8_t*)p >= l->arena && (const uint8_t*)p < l->arena + l->arena_bytes) return true;
  return false;
}
rhip_gt_table* Engine::gt_generator_table() {
  std::lock_guard<std::recursive_mutex> lk(mu_);
  if (!e_gen_tbl_) {
    const Gt& g = gt_generator();
    check(rhip_gt_table_create(ctx(), (const rhip_gt*)g.data(), &e_gen_tbl_), "rhip_gt_table_create");
    check(rhip_gt_table_add_w16(ctx(), e_gen_tbl_), "rhip_gt_table_add_w16");
  }
  return e_gen_tbl_;
}
void* Engine::aux(const std::string& kind, const std::string& key, void* (*make)(Engine&, const void*), const void* arg, void (*destroy)(void*), size_t cap) {
  std::lock_guard<std::recursive_mutex> lk(mu_);
  auto& m = aux_[kind];
  auto it = m.find(key);
  if (it != m.end()) { it->second.used = ++use_clock_; return it->second.h; }
  if (m.size() >= cap && !m.empty()) {            // the least recently used entry goes; never one this call was just handed (cap >= 2)
    auto lru = m.begin();
    for (auto c = m.begin(); c != m.end(); ++c) if (c->second.used < lru->second.used) lru = c;
    retire(lru->second.h, lru->second.destroy);
    m.erase(lru);
  }
  void* h = make(*this, arg);
  m[key] = Aux{h, destroy, ++use_clock_};
  return h;
}
void Engine::retire(void* h, void (*destroy)(void*)) {        // mu_ held
  if (busy_ > 0) parked_.push_back(Parked{++busy_clock_, h, destroy});
  else destroy(h);
}
Engine::Busy::Busy(Engine& eng) : e(eng), start(0) {
  std::lock_guard<std::recursive_mutex> lk(e.mu_);
  e.busy_++;
  start = ++e.busy_clock_;
  e.busy_starts_.insert(start);
}
Engine::Busy::~Busy() {
  std::vector<Parked> dead;
  {
    std::lock_guard<std::recursive_mutex> lk(e.mu_);
    --e.busy_;
    auto it = e.busy_starts_.find(start);
    if (it != e.busy_starts_.end()) e.busy_starts_.erase(it);
    const uint64_t oldest = e.busy_starts_.empty() ? ~(uint64_t)0 : *e.busy_starts_.begin();
    size_t keep = 0;
    for (auto& p : e.parked_) {
      if (p.at < oldest) dead.push_back(p);       // every scope that began before the eviction has ended
      else e.parked_[keep++] = p;
    }
    e.parked_.resize(keep);
  }
  for (auto& d : dead) d.destroy(d.h);            // every operation that could have seen these handles has waited for its stream
}
Engine::~Engine() {
  for (auto& d : parked_) d.destroy(d.h);
  for (auto& k : aux_) for (auto& c : k.second) c.second.destroy(c.second.h);
  if (e_gen_tbl_) rhip_gt_table_destroy(e_gen_tbl_);
  for (auto& c : pk17_) rhip_ac17_pk_destroy(c.second.h);
  for (auto& c : t1_) rhip_g1_table_destroy(c.second);
  for (auto& c : t2_) rhip_g2_table_destroy(c.second);
  for (auto& c : tt_) rhip_gt_table_destroy(c.second);
  for (auto& l : lanes_) {
    for (int i = 0; i < 4; i++) if (l->pin[i]) rhip_host_free(l->ctx, l->pin[i]);
    if (l->arena) rhip_free(l->ctx, l->arena);
  }
  for (size_t i = lanes_.size(); i-- > 0;) {
    if (lanes_[i]->side) rhip_ctx_destroy(lanes_[i]->side);
    rhip_ctx_destroy(lanes_[i]->ctx);
  }
}
void Engine::check(int32_t rc, const char* what) const {
  if (rc != RHIP_OK) throw RabeError(std::string(what) + " failed: " + rhip_last_error(ctx()));
}

DBuf::DBuf(Engine* e, size_t bytes) : eng_(e), n_(bytes) {
  p_ = e->arena_take(bytes ? bytes : 4);
  if (!p_) e->check(rhip_malloc(e->ctx(), bytes ? bytes : 4, &p_), "rhip_malloc");
}
DBuf::DBuf(Engine* e, const void* host_data, size_t bytes) : DBuf(e, bytes) {
  if (bytes) e->check(rhip_upload(e->ctx(), p_, host_data, bytes), "rhip_upload");
}
static void dbuf_release(Engine* e, void* p) { if (p && !e->arena_owns(p)) rhip_free(e->ctx(), p); }
DBuf::~DBuf() { dbuf_release(eng_, p_); }
DBuf& DBuf::operator=(DBuf&& o) noexcept {
  if (this != &o) { dbuf_release(eng_, p_); eng_ = o.eng_; p_ = o.p_; n_ = o.n_; o.p_ = nullptr; }
  return *this;
}
void DBuf::download(void* host, size_t bytes) const { if (bytes) eng_->check(rhip_download(eng_->ctx(), host, p_, bytes), "rhip_download"); }

template <class OUT, size_t N>
static std::vector<std::array<uint8_t, N>> unflatten(const std::vector<uint8_t>& raw) {
  std::vector<std::array<uint8_t, N>> v(raw.size() / N);
  for (size_t i = 0; i < v.size(); i++) memcpy(v[i].data(), raw.data() + i * N, N);
  return v;
}
template <size_t N>
static std::vector<std::array<uint8_t, N>> fetch(const DBuf& d, size_t n) {
  std::vector<uint8_t> raw(n * N);
  d.download(raw.data(), raw.size());
  return unflatten<void, N>(raw);
}

// ---- fixed-base fast path (common.h): partition the operands by base; groups of >= fixed_base_min elements go
// through a cached window table, the rest through the generic variable-base kernel
template <size_t N, class TBL, class CREATE, class MUL, class GENERIC>
std::vector<std::array<uint8_t, N>> Engine::mul_grouped(const std::vector<std::array<uint8_t, N>>& p, const std::vector<Fr>& k,
                                                         std::map<std::string, TBL*>& cache, CREATE create, MUL mul, GENERIC generic) {
  const size_t n = p.size();
  std::vector<std::array<uint8_t, N>> out(n);
  if (!n) return out;
  // group the operands by base value without copying them: keys point into `p`
  struct Key { const uint8_t* b; };
  struct KeyHash {
    size_t operator()(const Key& k) const {
      uint64_t h = 1469598103934665603ull;
      for (size_t i = 0; i + 8 <= N; i += 8) { uint64_t w; memcpy(&w, k.b + i, 8); h = (h ^ w) * 1099511628211ull; }
      return (size_t)h;
    }
  };
  struct KeyEq { bool operator()(const Key& a, const Key& b) const { return memcmp(a.b, b.b, N) == 0; } };
  // pass 1: how often does each base occur (no per-base allocation: most calls carry all-distinct bases)
  std::unordered_map<Key, uint32_t, KeyHash, KeyEq> count;
  count.reserve(n);
  for (size_t i = 0; i < n; i++) count[Key{p[i].data()}]++;
  // bases that repeat often enough -- in this call or over the calls so far (a server meets the same public-key elements again
  // and again) -- or already have a table, leave the generic path
  const size_t need = fixed_base_min;
  std::unordered_map<Key, size_t, KeyHash, KeyEq> slot;       // base -> index into `lists`
  std::vector<std::vector<size_t>> lists;
  std::map<std::string, size_t>& seen = seen_[N == 64 ? 0 : N == 128 ? 1 : 2];
  if (seen.size() > 65536) seen.clear();
  for (auto& g : count) {
    size_t total = g.second;
    if (g.second >= 16 && need > 1 && need != (size_t)-1) {   // only bases with some weight are remembered
      size_t& acc = seen[std::string((const char*)g.first.b, N)];
      acc += g.second;
      total = acc;
    }
    if (total >= need) { slot[g.first] = lists.size(); lists.emplace_back(); }
  }
  // every table group is a launch of its own: a call with hundreds of medium-sized groups (aw11: 2 x 200 attribute bases, 1024 uses
  // each) is better served by ONE generic launch -- then only groups that are big in this very call keep their table
  if (slot.size() > 16 && need > 1) {
    std::unordered_map<Key, size_t, KeyHash, KeyEq> keep;
    lists.clear();
    for (auto& g : count)
      if (slot.count(g.first) && g.second >= 4096) { keep[g.first] = lists.size(); lists.emplace_back(); }
    slot.swap(keep);
  }
  std::vector<std::string> cached_keys;                        // keeps the Key pointers below alive
  cached_keys.reserve(cache.size());
  for (auto& c : cache) cached_keys.push_back(c.first);
  for (auto& ck : cached_keys) {
    Key k2{(const uint8_t*)ck.data()};
    auto it = count.find(k2);
    if (it != count.end() && !slot.count(k2) && (slot.size() < 16 || it->second >= 4096)) { slot[k2] = lists.size(); lists.emplace_back(); }
  }
  // pass 2: distribute
  std::vector<size_t> rest;
  if (slot.empty()) {
    rest.resize(n);
    for (size_t i = 0; i < n; i++) rest[i] = i;
  } else {
    for (size_t i = 0; i < n; i++) {
      auto it = slot.find(Key{p[i].data()});
      if (it == slot.end()) rest.push_back(i); else lists[it->second].push_back(i);
    }
  }
  // groups in order of first appearance: a deterministic launch sequence
  std::vector<std::pair<size_t, const std::vector<size_t>*>> big;
  for (auto& l : lists) if (!l.empty()) big.push_back({l[0], &l});
  std::sort(big.begin(), big.end());
  for (auto& bg : big) {
    const std::vector<size_t>& idx = *bg.second;
    const std::string key((const char*)p[idx[0]].data(), N);
    const bool cached = cache.count(key) != 0;
    if (!cached) {
      if (cache.size() >= 1024) {                     // bounded: drop everything rather than track recency
        for (auto& c : cache) this->destroy_table(c.second);
        cache.clear();
      }
      TBL* t = nullptr;
      check(create(ctx(), p[idx[0]].data(), &t), "fixed-base table build");
      cache[key] = t;
    }
    std::vector<Fr> kk;
    for (size_t i : idx) kk.push_back(k[i]);
    auto fk = flatten_fr(kk);
    DBuf dk(this, fk.data(), fk.size()), dout(this, idx.size() * N);
    check(mul(ctx(), cache[key], idx.size(), dk.as<rhip_fr>(), dout.ptr()), "fixed-base table multiplication");
    auto r = fetch<N>(dout, idx.size());
    for (size_t j = 0; j < idx.size(); j++) out[idx[j]] = r[j];
  }
  if (rest.size() == n) {                       // nothing repeated: hand the call through untouched
    return generic(p, k);
  }
  if (!rest.empty()) {
    std::vector<std::array<uint8_t, N>> rp;
    std::vector<Fr> rk;
    for (size_t i : rest) { rp.push_back(p[i]); rk.push_back(k[i]); }
    auto r = generic(rp, rk);
    for (size_t j = 0; j < rest.size(); j++) out[rest[j]] = r[j];
  }
  return out;
}
void Engine::destroy_table(rhip_g1_table* t) { rhip_g1_table_destroy(t); }
void Engine::destroy_table(rhip_g2_table* t) { rhip_g2_table_destroy(t); }
void Engine::destroy_table(rhip_gt_table* t) { rhip_gt_table_destroy(t); }

std::vector<G1> Engine::g1_mul(const std::vector<G1>& p, const std::vector<Fr>& k) {
  std::lock_guard<std::recursive_mutex> lk(mu_);
  return mul_grouped<64>(p, k, t1_,
      [](rhip_ctx* c, const uint8_t* b, rhip_g1_table** t) { return rhip_g1_table_create(c, (const rhip_g1*)b, t); },
      [](rhip_ctx* c, const rhip_g1_table* t, size_t n, const rhip_fr* dk, void* o) { return rhip_g1_table_mul(c, t, n, dk, (rhip_g1*)o); },
      [this](const std::vector<G1>& rp, const std::vector<Fr>& rk) {
        size_t n = rp.size();
        auto fp = flatten(rp); auto fk = flatten_fr(rk);
        DBuf dp(this, fp.data(), fp.size()), dk(this, fk.data(), fk.size()), out(this, n * 64);
        check(rhip_g1_mul(ctx(), n, dp.as<rhip_g1>(), dk.as<rhip_fr>(), out.as<rhip_g1>()), "rhip_g1_mul");
        return fetch<64>(out, n);
      });
}
std::vector<G2> Engine::g2_mul(const std::vector<G2>& p, const std::vector<Fr>& k) {
  std::lock_guard<std::recursive_mutex> lk(mu_);
  return mul_grouped<128>(p, k, t2_,
      [](rhip_ctx* c, const uint8_t* b, rhip_g2_table** t) { return rhip_g2_table_create(c, (const rhip_g2*)b, t); },
      [](rhip_ctx* c, const rhip_g2_table* t, size_t n, const rhip_fr* dk, void* o) { return rhip_g2_table_mul(c, t, n, dk, (rhip_g2*)o); },
      [this](const std::vector<G2>& rp, const std::vector<Fr>& rk) {
        size_t n = rp.size();
        auto fp = flatten(rp); auto fk = flatten_fr(rk);
        DBuf dp(this, fp.data(), fp.size()), dk(this, fk.data(), fk.size()), out(this, n * 128);
        check(rhip_g2_mul(ctx(), n, dp.as<rhip_g2>(), dk.as<rhip_fr>(), out.as<rhip_g2>()), "rhip_g2_mul");
        return fetch<128>(out, n);
      });
}
std::vector<Gt> Engine::gt_pow(const std::vector<Gt>& a, const std::vector<Fr>& k) {
  std::lock_guard<std::recursive_mutex> lk(mu_);
  return mul_grouped<384>(a, k, tt_,
      [](rhip_ctx* c, const uint8_t* b, rhip_gt_table** t) { return rhip_gt_table_create(c, (const rhip_gt*)b, t); },
      [](rhip_ctx* c, const rhip_gt_table* t, size_t n, const rhip_fr* dk, void* o) { return rhip_gt_table_pow(c, t, n, dk, (rhip_gt*)o); },
      [this](const std::vector<Gt>& ra, const std::vector<Fr>& rk) {
        size_t n = ra.size();
        auto fa = flatten(ra); auto fk = flatten_fr(rk);
        DBuf da(this, fa.data(), fa.size()), dk(this, fk.data(), fk.size()), out(this, n * 384);
        check(rhip_gt_pow(ctx(), n, da.as<rhip_gt>(), dk.as<rhip_fr>(), out.as<rhip_gt>()), "rhip_gt_pow");
        return fetch<384>(out, n);
      });
}
std::vector<Gt> Engine::gt_mul(const std::vector<Gt>& a, const std::vector<Gt>& b) {
  size_t n = a.size();
  auto fa = flatten(a); auto fb = flatten(b);
  DBuf da(this, fa.data(), fa.size()), db(this, fb.data(), fb.size()), out(this, n * 384);
  check(rhip_gt_mul(ctx(), n, da.as<rhip_gt>(), db.as<rhip_gt>(), out.as<rhip_gt>()), "rhip_gt_mul");
  return fetch<384>(out, n);
}
std::vector<Gt> Engine::pairing(const std::vector<G1>& p, const std::vector<G2>& q) {
  size_t n = p.size();
  auto fp = flatten(p); auto fq = flatten(q);
  DBuf dp(this, fp.data(), fp.size()), dq(this, fq.data(), fq.size()), out(this, n * 384);
  check(rhip_pairing(ctx(), n, dp.as<rhip_g1>(), dq.as<rhip_g2>(), out.as<rhip_gt>()), "rhip_pairing");
  return fetch<384>(out, n);
}
const Gt& Engine::gt_generator() {
  std::lock_guard<std::recursive_mutex> lk(mu_);
  if (!have_e_) { e_gen_ = pairing({g1_generator()}, {g2_generator()})[0]; have_e_ = true; }
  return e_gen_;
}
Gt Engine::random_gt(Rng& rng) {
  Gt g = gt_generator();
  return gt_pow({g}, {rng.next_fr()})[0];
}

// small helpers over Level E used by the schemes below
static std::vector<G2> g2_add(Engine& e, const std::vector<G2>& a, const std::vector<G2>& b) {
  size_t n = a.size();
  auto fa = flatten(a); auto fb = flatten(b);
  DBuf da(&e, fa.data(), fa.size()), db(&e, fb.data(), fb.size()), out(&e, n * 128);
  e.check(rhip_g2_add(e.ctx(), n, da.as<rhip_g2>(), db.as<rhip_g2>(), out.as<rhip_g2>()), "rhip_g2_add");
  return fetch<128>(out, n);
}
static std::vector<G1> g1_add(Engine& e, const std::vector<G1>& a, const std::vector<G1>& b) {
  size_t n = a.size();
  auto fa = flatten(a); auto fb = flatten(b);
  DBuf da(&e, fa.data(), fa.size()), db(&e, fb.data(), fb.size()), out(&e, n * 64);
  e.check(rhip_g1_add(e.ctx(), n, da.as<rhip_g1>(), db.as<rhip_g1>(), out.as<rhip_g1>()), "rhip_g1_add");
  return fetch<64>(out, n);
}
static Fr must_inv(const Fr& a) {
  Fr o;
  if (!fr_inv(a, &o)) throw std::runtime_error("called `Option::unwrap()` on a `None` value (Fr::inverse of zero)");
  return o;
}
PolicyNode parse_or_error(const std::string& policy, PolicyLanguage lang) {
  try {
    return parse_policy(policy, lang);
  } catch (const PolicyError& e) {
    throw RabeError(std::string("Json Policy Error / parse: ") + e.what());
  }
}
static Bytes seal(Rng& rng, const Gt& msg, const Bytes& plaintext) {       // encrypt_symmetric, aes/mod.rs:10-26
  uint8_t nonce[12];
  rng.fill(nonce, 12);
  return encrypt_symmetric(msg.data(), plaintext.data(), plaintext.size(), nonce);
}
static Bytes open_or_error(const Gt& msg, const Bytes& ct) {               // decrypt_symmetric, aes/mod.rs:29-44
  Bytes out;
  if (!decrypt_symmetric(msg.data(), ct.data(), ct.size(), &out)) throw RabeError("decryption error: aead::Error");
  return out;
}

// RABE_HOST_TIMING=1: stage timings of the batch entry points on stderr (development aid)
struct StageTimer {
  bool on;
  const char* what;
  std::chrono::steady_clock::time_point t0;
  explicit StageTimer(const char* w) : on(getenv("RABE_HOST_TIMING") != nullptr), what(w), t0(std::chrono::steady_clock::now()) {}
  void lap(const char* stage) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[host-timing] %s: %s %.1f ms\n", what, stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
// ---- batched decryption tail shared by bsw / lsw / aw11: item i yields
//   lead_i * prod_j gbase_ij^gexp_ij * FE( prod_j ML(scal_ij * base_ij, q_ij) )
// with ONE g1_mul, ONE pairing-product launch (one final exponentiation per item) and ONE gt_pow for the whole batch.
struct PairingJob {
  std::vector<G1> base;
  std::vector<Fr> scal;
  std::vector<G2> q;
  std::vector<Gt> gbase;
  std::vector<Fr> gexp;
  // one more pair whose G1 argument is a sum: ML(sum_j sscal_j * sbase_j, sq)   (ghw11: e(sum_i w_i C_i, L))
  std::vector<G1> sbase;
  std::vector<Fr> sscal;
  G2 sq;
  Gt lead;
  bool lead_one = false;      // lead = 1 (no leading factor)
  bool failed = false;
  std::string error;
};
// With `d_keep`: when the values are complete on the device (no Gt power factors: every scheme but ghw11's retrieval), the live items'
// values STAY there -- *d_keep receives the array (one rhip_gt per live job, in order), *live_out the job indices, and the returned
// host vector is not filled: the caller opens the sealed plaintexts on the device (open_jobs) and no Gt crosses PCIe.
static std::vector<Gt> run_pairing_jobs(Engine& e, const std::vector<PairingJob>& jobs, DBuf* d_keep = nullptr, std::vector<size_t>* live_out = nullptr) {
  StageTimer tm("run_pairing_jobs");
  std::vector<size_t> live;
  for (size_t i = 0; i < jobs.size(); i++) if (!jobs[i].failed) live.push_back(i);
  std::vector<Gt> out(jobs.size());
  if (live_out) *live_out = live;
  if (live.empty()) return out;
  std::vector<Gt> gb;
  std::vector<Fr> gk;
  bool any_sum = false, any_pair = false;
  for (size_t i : live) {
    const PairingJob& j = jobs[i];
    gb.insert(gb.end(), j.gbase.begin(), j.gbase.end());
    gk.insert(gk.end(), j.gexp.begin(), j.gexp.end());
    any_sum = any_sum || !j.sbase.empty();
    any_pair = any_pair || !j.base.empty() || !j.sbase.empty();
  }
  Gt gt_one{};
  gt_one[0] = 1;
  std::vector<Gt> acc(live.size());
  for (size_t t = 0; t < live.size(); t++) acc[t] = jobs[live[t]].lead_one ? gt_one : jobs[live[t]].lead;
  if (any_pair) {
    tm.lap("concat");
    // the whole batch in ONE device call (rhip_pairing_jobs): NAF scaling of the G1 arguments, the summed argument with shared
    // doublings, every item's pairs on a few Fq12 accumulators, one final exponentiation per item, times the leading factor
    std::vector<uint8_t> fb, fs, fq, fsb, fss, fsq;
    std::vector<uint32_t> pair_off{0}, sum_off{0};
    size_t max_pairs = 0, max_terms = 0;
    for (size_t i : live) {
      const PairingJob& j = jobs[i];
      for (const auto& x : j.base) fb.insert(fb.end(), x.begin(), x.end());
      for (const auto& x : j.scal) fs.insert(fs.end(), (const uint8_t*)x.l, (const uint8_t*)x.l + 32);
      for (const auto& x : j.q) fq.insert(fq.end(), x.begin(), x.end());
      for (const auto& x : j.sbase) fsb.insert(fsb.end(), x.begin(), x.end());
      for (const auto& x : j.sscal) fss.insert(fss.end(), (const uint8_t*)x.l, (const uint8_t*)x.l + 32);
      G2 sq{};
      if (!j.sbase.empty()) sq = j.sq;
      fsq.insert(fsq.end(), sq.begin(), sq.end());
      pair_off.push_back(pair_off.back() + (uint32_t)j.base.size());
      sum_off.push_back(sum_off.back() + (uint32_t)j.sbase.size());
      max_pairs = std::max(max_pairs, j.base.size());
      max_terms = std::max(max_terms, j.sbase.size());
    }
    auto flead = flatten(acc);
    const size_t m = live.size(), np = pair_off.back(), nt = sum_off.back();
    DBuf dbase(&e, fb.data(), fb.size()), dscal(&e, fs.data(), fs.size()), dq(&e, fq.data(), fq.size()), dpo(&e, pair_off.data(), pair_off.size() * 4),
        dsb(&e, fsb.data(), fsb.size()), dss(&e, fss.data(), fss.size()), dsq(&e, fsq.data(), fsq.size()), dso(&e, sum_off.data(), sum_off.size() * 4),
        dlead(&e, flead.data(), flead.size()), dout(&e, m * 384);
    tm.lap("flatten + upload");
    e.check(rhip_pairing_jobs(e.ctx(), m, max_pairs, np, dpo.as<uint32_t>(), dbase.as<rhip_g1>(), dscal.as<rhip_fr>(), dq.as<rhip_g2>(), max_terms, nt,
                              any_sum ? dso.as<uint32_t>() : (const uint32_t*)nullptr, dsb.as<rhip_g1>(), dss.as<rhip_fr>(), dsq.as<rhip_g2>(),
                              dlead.as<rhip_gt>(), dout.as<rhip_gt>()), "rhip_pairing_jobs");
    if (d_keep && gb.empty()) {
      *d_keep = std::move(dout);
      tm.lap("pairing jobs (values kept on the device)");
      return out;
    }
    acc = fetch<384>(dout, m);
    tm.lap("pairing jobs");
  }
  if (!gb.empty()) {
    std::vector<Gt> pw = e.gt_pow(gb, gk);
    // each item's factors folded by one lane: a single launch instead of one gt_mul round per factor position
    std::vector<uint32_t> goff{0};
    for (size_t t = 0; t < live.size(); t++) goff.push_back(goff.back() + (uint32_t)jobs[live[t]].gbase.size());
    auto fpw = flatten(pw);
    DBuf dpw(&e, fpw.data(), fpw.size()), dgoff(&e, goff.data(), goff.size() * 4), dprod(&e, live.size() * 384);
    e.check(rhip_gt_product(e.ctx(), live.size(), dgoff.as<uint32_t>(), dpw.as<rhip_gt>(), dprod.as<rhip_gt>()), "rhip_gt_product");
    acc = e.gt_mul(acc, fetch<384>(dprod, live.size()));
  }
  for (size_t t = 0; t < live.size(); t++) out[live[t]] = acc[t];
  tm.lap("gt powers + fold");
  return out;
}
static std::vector<schemes::DecryptResult> open_jobs(Engine& e, const std::vector<PairingJob>& jobs, const std::vector<const Bytes*>& sealed) {
  Engine::ArenaScope scope(e);
  DBuf d_gt;
  std::vector<size_t> live;
  std::vector<Gt> gts = run_pairing_jobs(e, jobs, &d_gt, &live);
  std::vector<schemes::DecryptResult> out(jobs.size());
  if (d_gt.ptr()) {
    // KDF + AES-GCM open on the device (round 4, Level S): the sealed parts go up as one blob, plaintext bytes come back
    const size_t n = jobs.size();
    std::vector<uint64_t> s_off(live.size());
    std::vector<uint32_t> s_len(live.size());
    size_t total = 0;
    for (size_t t = 0; t < live.size(); t++) { s_off[t] = total; s_len[t] = (uint32_t)sealed[live[t]]->size(); total += s_len[t]; }
    Bytes blob(total ? total : 1);
    for (size_t t = 0; t < live.size(); t++) if (s_len[t]) memcpy(blob.data() + s_off[t], sealed[live[t]]->data(), s_len[t]);
    DBuf d_blob(&e, blob.data(), blob.size());
    std::vector<int32_t> status(n, -1);
    std::vector<uint64_t> pt_off(n + 1, 0);
    Bytes pt(total ? total : 1);
    std::vector<std::string> errors(n);
    for (size_t i = 0; i < n; i++) if (jobs[i].failed) errors[i] = jobs[i].error.empty() ? std::string("failed") : jobs[i].error;
    schemes::open_sealed_records(e, n, live, d_gt.ptr(), d_blob.as<uint8_t>(), s_off, s_len, status.data(), pt.data(), pt_off.data(), &errors);
    for (size_t i = 0; i < n; i++) {
      if (jobs[i].failed) out[i] = {false, {}, jobs[i].error};
      else if (status[i] == 0) out[i] = {true, Bytes(pt.begin() + (size_t)pt_off[i], pt.begin() + (size_t)pt_off[i + 1]), ""};
      else out[i] = {false, {}, errors[i].empty() ? std::string("decryption error: aead::Error") : errors[i]};
    }
    return out;
  }
  for (size_t i = 0; i < jobs.size(); i++) {
    if (jobs[i].failed) { out[i] = {false, {}, jobs[i].error}; continue; }
    Bytes pt;
    if (decrypt_symmetric(gts[i].data(), sealed[i]->data(), sealed[i]->size(), &pt)) out[i] = {true, pt, ""};
    else out[i] = {false, {}, "decryption error: aead::Error"};
  }
  return out;
}
// Host-side planning of a batch (share generation, hashing, pruning: string and Fr work) runs on all cores; the
// randomness is pulled from the generator beforehand, item after item, so results do not depend on the thread count.
// A persistent pool (round 3 spawned and joined up to 64 threads per call: ~2 ms of a packed call's ~25 ms, several times per call).
// A call hands out `tickets` for its job; pool threads that take one run the job's index loop beside the caller, who always takes
// part itself -- so a parallel_for from inside a pool thread (or from several caller threads at once: the pipelined entry points)
// cannot deadlock, it only finds fewer helpers.
namespace {
struct PfJob {
  size_t n;
  const std::function<void(size_t)>* fn;
  std::atomic<size_t> next{0};
  std::atomic<int> helpers{0};          // pool threads still inside this job
  std::exception_ptr first;
  std::mutex mu;
  std::condition_variable done;
  void run() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= n) return;
      try {
        (*fn)(i);
      } catch (...) {
        std::lock_guard<std::mutex> g(mu);
        if (!first) first = std::current_exception();
      }
    }
  }
};
class PfPool {
 public:
  static PfPool& get() { static PfPool* p = new PfPool(); return *p; }          // leaked on purpose: no join at process exit
  unsigned size() const { return (unsigned)threads_.size(); }
  void offer(const std::shared_ptr<PfJob>& job, unsigned tickets) {
    {
      std::lock_guard<std::mutex> g(mu_);
      for (unsigned t = 0; t < tickets; t++) q_.push_back(job);
    }
    if (tickets == 1) cv_.notify_one(); else cv_.notify_all();
  }
  // tickets nobody took yet are withdrawn (the caller finished the indices itself)
  void withdraw(const std::shared_ptr<PfJob>& job) {
    std::lock_guard<std::mutex> g(mu_);
    for (auto it = q_.begin(); it != q_.end();) it = (*it == job) ? q_.erase(it) : it + 1;
  }
 private:
  PfPool() {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 64) nt = 64;
    if (nt < 2) nt = 2;
    for (unsigned t = 0; t + 1 < nt; t++) threads_.emplace_back([this] { loop(); });
    for (auto& t : threads_) t.detach();
  }
  void loop() {
    for (;;) {
      std::shared_ptr<PfJob> job;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this] { return !q_.empty(); });
        job = q_.front();
        q_.pop_front();
        job->helpers.fetch_add(1);
      }
      job->run();
      {
        std::lock_guard<std::mutex> g(job->mu);
        job->helpers.fetch_sub(1);
      }
      job->done.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<PfJob>> q_;
  std::vector<std::thread> threads_;
};
}  // namespace
void parallel_for(size_t n, const std::function<void(size_t)>& fn) {
  PfPool& pool = PfPool::get();
  if (n < 16 || pool.size() == 0) { for (size_t i = 0; i < n; i++) fn(i); return; }
  auto job = std::make_shared<PfJob>();
  job->n = n;
  job->fn = &fn;
  pool.offer(job, (unsigned)std::min<size_t>(pool.size(), n - 1));
  job->run();
  pool.withdraw(job);                        // from here on no new helper can enter (taking a ticket and `helpers++` happen under the pool's lock)
  {
    std::unique_lock<std::mutex> g(job->mu);
    job->done.wait(g, [&] { return job->helpers.load() == 0; });
  }
  if (job->first) std::rethrow_exception(job->first);
}
// Items of one batch usually repeat a few policies: the parsed tree and its Lagrange coefficients (pure functions of
// the policy text) are computed once per distinct (text, language) within a call.
struct PolicyMemo {
  struct Entry { PolicyNode tree; NamedFr coeff; bool msp_checked = false; };
  std::map<std::pair<std::string, int>, std::shared_ptr<Entry>> m;
  std::mutex mu;
  const Entry& get(const std::string& policy, PolicyLanguage lang) {
    std::lock_guard<std::mutex> g(mu);
    auto key = std::make_pair(policy, (int)lang);
    auto it = m.find(key);
    if (it != m.end()) return *it->second;
    auto e = std::make_shared<Entry>();
    e->tree = parse_or_error(policy, lang);
    calc_coefficients(e->tree, fr_one(), &e->coeff);
    m[key] = e;
    return *e;
  }
};
// plan(i, &job) fills job i or throws RabeError (-> that item fails); panics (std::runtime_error) propagate like the reference's
template <class PLAN>
static std::vector<PairingJob> plan_jobs(size_t n, PLAN plan) {
  StageTimer tm("plan_jobs");
  std::vector<PairingJob> jobs(n);
  parallel_for(n, [&](size_t i) {
    try {
      plan(i, &jobs[i]);
    } catch (const RabeError& ex) {
      jobs[i] = PairingJob();
      jobs[i].failed = true;
      jobs[i].error = ex.what();
    }
  });
  tm.lap("plan (parallel)");
  return jobs;
}

namespace schemes {

// ================================================================================================ AC17
namespace ac17 {
static const size_t ASSUMPTION_SIZE = 2;     // ac17/mod.rs:138

std::pair<Ac17PublicKey, Ac17MasterKey> setup(Engine& eng, Rng& rng) {      // :141-182
  G1 g = eng.random_g1(rng);
  G2 h = eng.random_g2(rng);
  Gt e_gh = eng.pairing({g}, {h})[0];
  std::vector<Fr> a, b, k;
  for (size_t i = 0; i < ASSUMPTION_SIZE; i++) { a.push_back(rng.next_fr()); b.push_back(rng.next_fr()); }
  for (size_t i = 0; i < ASSUMPTION_SIZE + 1; i++) k.push_back(rng.next_fr());
  std::vector<G2> h_a = eng.g2_mul({h, h}, {a[0], a[1]});
  h_a.push_back(h);
  std::vector<G1> g_k = eng.g1_mul({g, g, g}, k);
  std::vector<Gt> e_gh_ka = eng.gt_pow({e_gh, e_gh}, {fr_add(fr_mul(k[0], a[0]), k[2]), fr_add(fr_mul(k[1], a[1]), k[2])});
  return {Ac17PublicKey{g, h_a, e_gh_ka}, Ac17MasterKey{g, h, g_k, a, b}};
}

Ac17CpSecretKey cp_keygen(Engine& eng, Rng& rng, const Ac17MasterKey& msk, const std::vector<std::string>& attributes) {   // :191-264
  if (attributes.empty()) throw RabeError("empty attributes!");
  if (msk.a.size() != 2 || msk.b.size() != 2 || msk.g_k.size() != 3) throw RabeError("malformed Ac17MasterKey: a, b must have 2 and g_k 3 elements");
  const size_t n = attributes.size();
  // draw order: r0, r1, sigma per attribute (loop order), sigma'
  std::vector<Fr> r{rng.next_fr(), rng.next_fr()};
  std::vector<Fr> sigma;
  for (size_t i = 0; i < n; i++) sigma.push_back(rng.next_fr());
  Fr sigma_p = rng.next_fr();
  std::vector<Fr> H, H01;
  for (const auto& attr : attributes)
    for (int l = 0; l < 3; l++)
      for (int t = 0; t < 2; t++) H.push_back(sha3_hash_fr(attr + std::to_string(l) + std::to_string(t)));
  for (int l = 0; l < 3; l++)
    for (int t = 0; t < 2; t++) H01.push_back(sha3_hash_fr(std::string("01") + std::to_string(l) + std::to_string(t)));
  std::vector<Fr> a_inv{must_inv(msk.a[0]), must_inv(msk.a[1])};
  rhip_g1_table* gt = nullptr;
  rhip_g2_table* ht = nullptr;
  msk_tables(eng, msk, &gt, &ht);                // kept across calls (a key authority issues many keys under one master key)
  auto fgk = flatten(msk.g_k), fa = flatten_fr(a_inv), fb = flatten_fr(msk.b), fH = flatten_fr(H), fH01 = flatten_fr(H01), fr_ = flatten_fr(r),
       fs = flatten_fr(sigma), fsp = flatten_fr({sigma_p});
  DBuf dgk(&eng, fgk.data(), fgk.size()), da(&eng, fa.data(), fa.size()), db(&eng, fb.data(), fb.size()), dH(&eng, fH.data(), fH.size()),
      dH01(&eng, fH01.data(), fH01.size()), dr(&eng, fr_.data(), fr_.size()), ds(&eng, fs.data(), fs.size()), dsp(&eng, fsp.data(), fsp.size());
  DBuf dk0(&eng, 3 * 128), dk(&eng, n * 3 * 64), dkp(&eng, 3 * 64);
  int32_t rc = rhip_ac17_cp_keygen_batch(eng.ctx(), gt, ht, dgk.as<rhip_g1>(), da.as<rhip_fr>(), db.as<rhip_fr>(), 1, n, dH.as<rhip_fr>(),
                                         dH01.as<rhip_fr>(), dr.as<rhip_fr>(), ds.as<rhip_fr>(), dsp.as<rhip_fr>(), dk0.as<rhip_g2>(),
                                         dk.as<rhip_g1>(), dkp.as<rhip_g1>());
  std::vector<G2> k0;
  std::vector<G1> k, kp;
  if (rc == RHIP_OK) { k0 = fetch<128>(dk0, 3); k = fetch<64>(dk, n * 3); kp = fetch<64>(dkp, 3); }
  eng.check(rc, "rhip_ac17_cp_keygen_batch");
  Ac17CpSecretKey out;
  out.attr = attributes;
  out.sk.k_0 = k0;
  for (size_t i = 0; i < n; i++) out.sk.k.push_back({attributes[i], {k[3 * i], k[3 * i + 1], k[3 * i + 2]}});
  out.sk.k_p = kp;
  return out;
}

// per-policy Fr table A[row][l][t] (SURVEY.md Appendix B.3 = ac17/mod.rs:305-348 with the hashes pre-combined)
static std::vector<Fr> policy_table(const AbePolicy& msp) {
  const size_t cols = msp.m.empty() ? 0 : msp.m[0].size();
  std::vector<Fr> colh(cols * 6);
  for (size_t j = 0; j < cols; j++)
    for (int l = 0; l < 3; l++)
      for (int t = 0; t < 2; t++)
        colh[(j * 3 + l) * 2 + t] = sha3_hash_fr(std::string("0") + std::to_string(j + 1) + std::to_string(l) + std::to_string(t));
  std::vector<Fr> A;
  for (size_t i = 0; i < msp.m.size(); i++)
    for (int l = 0; l < 3; l++)
      for (int t = 0; t < 2; t++) {
        Fr v = sha3_hash_fr(msp.pi[i] + std::to_string(l) + std::to_string(t));
        for (size_t j = 0; j < cols; j++) {
          if (msp.m[i][j] == 1) v = fr_add(v, colh[(j * 3 + l) * 2 + t]);
          else if (msp.m[i][j] == -1) v = fr_sub(v, colh[(j * 3 + l) * 2 + t]);
        }
        A.push_back(v);
      }
  return A;
}

std::vector<Ac17CpCiphertext> cp_encrypt_batch(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::string>& policies,
                                               const std::vector<Bytes>& plaintexts, PolicyLanguage language) {
  const size_t n = policies.size();
  if (plaintexts.size() != n) throw RabeError("cp_encrypt_batch: policies / plaintexts length mismatch");
  // host: parse + MSP once per distinct policy string
  std::map<std::string, size_t> index;
  std::vector<AbePolicy> msps;
  std::vector<uint32_t> a_off{0};
  std::vector<Fr> A;
  std::vector<size_t> item_pol(n);
  for (size_t i = 0; i < n; i++) {
    auto it = index.find(policies[i]);
    if (it == index.end()) {
      PolicyNode tree = parse_or_error(policies[i], language);
      AbePolicy msp = calculate_msp(tree);
      std::vector<Fr> tab = policy_table(msp);
      A.insert(A.end(), tab.begin(), tab.end());
      a_off.push_back(a_off.back() + (uint32_t)msp.m.size());
      msps.push_back(std::move(msp));
      it = index.insert({policies[i], msps.size() - 1}).first;
    }
    item_pol[i] = it->second;
  }
  // randomness in the reference's per-call draw order: s0, s1, msg, nonce -- item after item
  std::vector<Fr> s, msg_k;
  std::vector<std::array<uint8_t, 12>> nonces(n);
  for (size_t i = 0; i < n; i++) {
    s.push_back(rng.next_fr());
    s.push_back(rng.next_fr());
    msg_k.push_back(rng.next_fr());                    // rng.gen::<Gt>() = e(G1::one(), G2::one())^Fr::random
    rng.fill(nonces[i].data(), 12);
  }
  std::vector<Gt> msgs = n ? eng.gt_pow(std::vector<Gt>(n, eng.gt_generator()), msg_k) : std::vector<Gt>();   // one launch for the batch
  std::vector<uint32_t> item_a_off(n), row_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) {
    item_a_off[i] = a_off[item_pol[i]];
    row_off[i + 1] = row_off[i] + (uint32_t)msps[item_pol[i]].m.size();
  }
  const size_t total_rows = row_off[n];
  rhip_ac17_pk* dpk = eng.ac17_pk(pk.g, pk.h_a, pk.e_gh_ka);
  auto fA = flatten_fr(A), fs = flatten_fr(s), fm = flatten(msgs);
  DBuf dA(&eng, fA.data(), fA.size()), dio(&eng, item_a_off.data(), n * 4), dro(&eng, row_off.data(), (n + 1) * 4), ds(&eng, fs.data(), fs.size()),
      dm(&eng, fm.data(), fm.size()), dc0(&eng, n * 3 * 128), dc(&eng, total_rows * 3 * 64), dcp(&eng, n * 384);
  int32_t rc = rhip_ac17_cp_encrypt_batch(eng.ctx(), dpk, n, dA.as<rhip_fr>(), dio.as<uint32_t>(), dro.as<uint32_t>(), total_rows,
                                          ds.as<rhip_fr>(), dm.as<rhip_gt>(), dc0.as<rhip_g2>(), dc.as<rhip_g1>(), dcp.as<rhip_gt>());
  std::vector<G2> c0;
  std::vector<G1> c;
  std::vector<Gt> cp;
  if (rc == RHIP_OK) { c0 = fetch<128>(dc0, n * 3); c = fetch<64>(dc, total_rows * 3); cp = fetch<384>(dcp, n); }
  eng.check(rc, "rhip_ac17_cp_encrypt_batch");
  std::vector<Ac17CpCiphertext> out(n);
  parallel_for(n, [&](size_t i) {               // struct assembly + KDF + AES-GCM per item, on all cores
    const AbePolicy& msp = msps[item_pol[i]];
    out[i].policy = {policies[i], language};
    out[i].ct.c_0 = {c0[3 * i], c0[3 * i + 1], c0[3 * i + 2]};
    out[i].ct.c.reserve(msp.m.size());
    for (size_t r = 0; r < msp.m.size(); r++) {
      size_t g = (size_t)row_off[i] + r;
      out[i].ct.c.push_back({msp.pi[r], {c[3 * g], c[3 * g + 1], c[3 * g + 2]}});
    }
    out[i].ct.c_p = cp[i];
    out[i].ct.ct = encrypt_symmetric(msgs[i].data(), plaintexts[i].data(), plaintexts[i].size(), nonces[i].data());
  });
  return out;
}

Ac17CpCiphertext cp_encrypt(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::string& policy, const Bytes& plaintext,
                            PolicyLanguage language) {
  return cp_encrypt_batch(eng, rng, pk, {policy}, {plaintext}, language)[0];
}

// One decrypt call, abstracted over CP (:385-430) and KP (:625-675): the attribute list that must satisfy the
// policy, the policy, the ciphertext body and the key body (KP keys have no k_p: three points at infinity).
struct DecItem {
  const std::vector<std::string>* attrs;
  const PolicyRef* policy;
  const Ac17Ciphertext* ct;
  const Ac17SecretKey* sk;
  const char* err_traverse;
  const char* err_pruned;
};
// Gt of n decrypt calls; errors[i] non-empty when item i does not decrypt (no group work is done for it)
static std::vector<Gt> decrypt_items(Engine& eng, const std::vector<DecItem>& items, std::vector<std::string>* errors) {
  const size_t n = items.size();
  errors->assign(n, "");
  std::vector<uint8_t> ct_c0, ct_c, ct_cp, sk_k0, sk_k, sk_kp;
  std::vector<uint32_t> ct_row_off{0}, sk_row_off{0}, sk_idx, ct_sel, sk_sel, ct_sel_off{0}, sk_sel_off{0};
  std::vector<size_t> live;
  std::map<const Ac17SecretKey*, uint32_t> key_slot;       // items that share a key object share its device copy (and its prepared lines)
  // per item, on all cores: parse (once per distinct policy), traverse, prune, and the name-matching loops of
  // :403-414 / :643-654 as index lists
  struct Plan { std::vector<uint32_t> ct_sel, sk_sel; };
  std::vector<Plan> plans(n);
  PolicyMemo memo;
  parallel_for(n, [&](size_t i) {
    const DecItem& it = items[i];
    try {
      // fixed-size vectors (ASSUMPTION_SIZE + 1 = 3): the kernels index [item * 3 + j]; a malformed object fails its own item only
      if (it.ct->c_0.size() != 3 || it.sk->k_0.size() != 3 || !(it.sk->k_p.empty() || it.sk->k_p.size() == 3))
        throw RabeError("malformed AC17 object: c_0 / k_0 / k_p must have 3 elements");
      for (const auto& row : it.ct->c) if (row.second.size() != 3) throw RabeError("malformed AC17 ciphertext: a row does not have 3 elements");
      for (const auto& row : it.sk->k) if (row.second.size() != 3) throw RabeError("malformed AC17 key: a row does not have 3 elements");
      const PolicyNode& tree = memo.get(it.policy->first, it.policy->second).tree;       // a policy that does not parse fails its item
      if (!traverse_policy(*it.attrs, tree)) { (*errors)[i] = it.err_traverse; return; }
      PrunedList lst;
      if (!calc_pruned(*it.attrs, tree, &lst)) { (*errors)[i] = it.err_pruned; return; }
      for (const auto& cur : lst) {
        for (size_t r = 0; r < it.ct->c.size(); r++) if (it.ct->c[r].first == cur.first) plans[i].ct_sel.push_back((uint32_t)r);
        for (size_t r = 0; r < it.sk->k.size(); r++) if (it.sk->k[r].first == cur.first) plans[i].sk_sel.push_back((uint32_t)r);
      }
    } catch (const RabeError& ex) {
      (*errors)[i] = ex.what();            // like plan_jobs of the other schemes: one bad item does not abort the batch
    }
  });
  for (size_t i = 0; i < n; i++) {
    if (!(*errors)[i].empty()) continue;
    const DecItem& it = items[i];
    ct_sel.insert(ct_sel.end(), plans[i].ct_sel.begin(), plans[i].ct_sel.end());
    sk_sel.insert(sk_sel.end(), plans[i].sk_sel.begin(), plans[i].sk_sel.end());
    ct_sel_off.push_back((uint32_t)ct_sel.size());
    sk_sel_off.push_back((uint32_t)sk_sel.size());
    for (const auto& x : it.ct->c_0) ct_c0.insert(ct_c0.end(), x.begin(), x.end());
    for (const auto& row : it.ct->c) for (const auto& x : row.second) ct_c.insert(ct_c.end(), x.begin(), x.end());
    ct_cp.insert(ct_cp.end(), it.ct->c_p.begin(), it.ct->c_p.end());
    ct_row_off.push_back(ct_row_off.back() + (uint32_t)it.ct->c.size());
    auto slot = key_slot.find(it.sk);
    if (slot == key_slot.end()) {
      slot = key_slot.emplace(it.sk, (uint32_t)key_slot.size()).first;
      for (const auto& x : it.sk->k_0) sk_k0.insert(sk_k0.end(), x.begin(), x.end());
      for (const auto& row : it.sk->k) for (const auto& x : row.second) sk_k.insert(sk_k.end(), x.begin(), x.end());
      if (it.sk->k_p.size() == 3) { for (const auto& x : it.sk->k_p) sk_kp.insert(sk_kp.end(), x.begin(), x.end()); }
      else sk_kp.insert(sk_kp.end(), 3 * 64, 0);        // KP: prod_h starts from G1::zero()
      sk_row_off.push_back(sk_row_off.back() + (uint32_t)it.sk->k.size());
    }
    sk_idx.push_back(slot->second);
    live.push_back(i);
  }
  std::vector<Gt> out(n);
  const size_t m = live.size();
  if (!m) return out;
  DBuf d1(&eng, ct_c0.data(), ct_c0.size()), d2(&eng, ct_c.data(), ct_c.size()), d3(&eng, ct_row_off.data(), ct_row_off.size() * 4),
      d4(&eng, ct_cp.data(), ct_cp.size()), d5(&eng, sk_k0.data(), sk_k0.size()), d6(&eng, sk_k.data(), sk_k.size()),
      d7(&eng, sk_row_off.data(), sk_row_off.size() * 4), d8(&eng, sk_kp.data(), sk_kp.size()), d9(&eng, sk_idx.data(), sk_idx.size() * 4),
      d10(&eng, ct_sel.data(), ct_sel.size() * 4), d11(&eng, ct_sel_off.data(), ct_sel_off.size() * 4), d12(&eng, sk_sel.data(), sk_sel.size() * 4),
      d13(&eng, sk_sel_off.data(), sk_sel_off.size() * 4), dout(&eng, m * 384);
  if (m >= 64) {
    // throughput path: the keys' k_0 lines are prepared once for the call, each (item, j) lane runs both pairings
    rhip_ac17_sk_lines* lines = nullptr;
    eng.check(rhip_ac17_sk_prepare(eng.ctx(), key_slot.size(), d5.as<rhip_g2>(), &lines), "rhip_ac17_sk_prepare");
    int32_t rc = rhip_ac17_cp_decrypt_batch_prepared(eng.ctx(), m, d1.as<rhip_g2>(), d2.as<rhip_g1>(), d3.as<uint32_t>(), d4.as<rhip_gt>(), lines,
                                                     d6.as<rhip_g1>(), d7.as<uint32_t>(), d8.as<rhip_g1>(), d9.as<uint32_t>(), d10.as<uint32_t>(),
                                                     d11.as<uint32_t>(), d12.as<uint32_t>(), d13.as<uint32_t>(), dout.as<rhip_gt>());
    if (rc == RHIP_OK) rc = rhip_sync(eng.ctx());
    rhip_ac17_sk_lines_destroy(lines);
    eng.check(rc, "rhip_ac17_cp_decrypt_batch_prepared");
  } else {
    eng.check(rhip_ac17_cp_decrypt_batch(eng.ctx(), m, d1.as<rhip_g2>(), d2.as<rhip_g1>(), d3.as<uint32_t>(), d4.as<rhip_gt>(), d5.as<rhip_g2>(),
                                         d6.as<rhip_g1>(), d7.as<uint32_t>(), d8.as<rhip_g1>(), d9.as<uint32_t>(), d10.as<uint32_t>(),
                                         d11.as<uint32_t>(), d12.as<uint32_t>(), d13.as<uint32_t>(), dout.as<rhip_gt>()),
              "rhip_ac17_cp_decrypt_batch");
  }
  std::vector<Gt> got = fetch<384>(dout, m);
  for (size_t j = 0; j < m; j++) out[live[j]] = got[j];
  return out;
}
static std::vector<Gt> decrypt_gts(Engine& eng, const std::vector<const Ac17CpSecretKey*>& sks, const std::vector<const Ac17CpCiphertext*>& cts,
                                   std::vector<std::string>* errors) {
  std::vector<DecItem> items;
  for (size_t i = 0; i < cts.size(); i++)
    items.push_back({&sks[i]->attr, &cts[i]->policy, &cts[i]->ct, &sks[i]->sk, "Error in cp_decrypt: attributes in SK do not match policy in CT.",
                     "Error: attributes in sk do not match policy in ct."});
  return decrypt_items(eng, items, errors);
}

std::vector<DecryptResult> cp_decrypt_batch(Engine& eng, const std::vector<const Ac17CpSecretKey*>& sks,
                                            const std::vector<const Ac17CpCiphertext*>& cts) {
  std::vector<std::string> errors;
  std::vector<Gt> gts = decrypt_gts(eng, sks, cts, &errors);
  std::vector<DecryptResult> out(cts.size());
  for (size_t i = 0; i < cts.size(); i++) {
    if (!errors[i].empty()) { out[i] = {false, {}, errors[i]}; continue; }
    Bytes pt;
    if (decrypt_symmetric(gts[i].data(), cts[i]->ct.ct.data(), cts[i]->ct.ct.size(), &pt)) out[i] = {true, pt, ""};
    else out[i] = {false, {}, "decryption error: aead::Error"};
  }
  return out;
}
Gt cp_decrypt_gt(Engine& eng, const Ac17CpSecretKey& sk, const Ac17CpCiphertext& ct) {
  std::vector<std::string> errors;
  std::vector<Gt> g = decrypt_gts(eng, {&sk}, {&ct}, &errors);
  if (!errors[0].empty()) throw RabeError(errors[0]);
  return g[0];
}
Bytes cp_decrypt(Engine& eng, const Ac17CpSecretKey& sk, const Ac17CpCiphertext& ct) {     // :385-430
  return open_or_error(cp_decrypt_gt(eng, sk, ct), ct.ct.ct);
}

// ---------------------------------------------------------------------------------------------- packed batches
// The same two functions with packed input and output (no per-item objects on either side of the C ABI): n ciphertexts are
// one blob of canonical records (the byte form of rabe_obj_serialize for Ac17CpCiphertext) with an offset array.  Everything
// per item that is not group arithmetic -- record assembly, KDF, AES-GCM, parsing, pruning -- runs on all host cores; the
// group arithmetic is one launch set; PCIe copies go through pinned staging buffers.
static inline void put_u32(uint8_t* p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static inline uint32_t get_u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
// parse + MSP + Fr table of a policy text, cached across calls (a server encrypts under the same few policies again and again)
struct EncPolicy { AbePolicy msp; std::vector<Fr> tab; size_t fixed_bytes; };
static std::shared_ptr<const EncPolicy> enc_policy(const std::string& pol, PolicyLanguage language) {
  static std::mutex mu;
  static std::map<std::pair<int, std::string>, std::shared_ptr<const EncPolicy>> cache;
  const auto key = std::make_pair((int)language, pol);
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
  }
  auto e = std::make_shared<EncPolicy>();
  e->msp = calculate_msp(parse_or_error(pol, language));
  e->tab = policy_table(e->msp);
  e->fixed_bytes = 4 + pol.size() + 1 + 4 + 3 * 128 + 4 + 384 + 4;      // record size without the sealed plaintext
  for (const auto& name : e->msp.pi) e->fixed_bytes += 4 + name.size() + 4 + 3 * 64;
  std::lock_guard<std::mutex> g(mu);
  if (cache.size() >= 1024) cache.clear();
  cache[key] = e;
  return e;
}
// out_buf / out_cap: caller-allocated (reusable) output; out_off: n + 1 caller-allocated entries, always filled.  Returns false
// -- before any randomness is drawn or work is done -- when out_cap < out_off[n], the size the records need.
// KP-ABE's encrypt (:556-616) is the same arithmetic with a table of plain label hashes: rows = the attribute list, no MSP columns.  The
// record differs only in its head (the attribute strings instead of policy text + language).
static std::shared_ptr<const EncPolicy> enc_attr_set(const std::vector<std::string>& attrs) {
  auto e = std::make_shared<EncPolicy>();
  e->msp.pi = attrs;
  e->msp.m.assign(attrs.size(), std::vector<int8_t>{0});
  for (const auto& a : attrs)
    for (int l = 0; l < 3; l++)
      for (int t = 0; t < 2; t++) e->tab.push_back(sha3_hash_fr(a + std::to_string(l) + std::to_string(t)));
  e->fixed_bytes = 4 + 4 + 3 * 128 + 4 + 384 + 4;
  for (const auto& a : attrs) e->fixed_bytes += (4 + a.size()) + (4 + a.size() + 4 + 3 * 64);
  return e;
}
static bool encrypt_packed_core(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::shared_ptr<const EncPolicy>>& pols,
                                const std::vector<std::string>* policies /* CP: the texts; KP: nullptr */, PolicyLanguage language, size_t n,
                                const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool cp_encrypt_packed(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::string>& policies, PolicyLanguage language, size_t n,
                       const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  std::vector<std::shared_ptr<const EncPolicy>> pols;
  for (const auto& pol : policies) pols.push_back(enc_policy(pol, language));
  return encrypt_packed_core(eng, rng, pk, pols, &policies, language, n, item_policy, pt_blob, pt_off, out_buf, out_cap, out_off);
}
// n calls of ac17::kp_encrypt: item i is encrypted under the attribute list sets[item_set[i]]; records = Ac17KpCiphertext
bool kp_encrypt_packed(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::vector<std::string>>& sets, size_t n, const uint32_t* item_set,
                       const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  std::vector<std::shared_ptr<const EncPolicy>> pols;
  for (const auto& a : sets) pols.push_back(enc_attr_set(a));
  return encrypt_packed_core(eng, rng, pk, pols, nullptr, PolicyLanguage::JsonPolicy, n, item_set, pt_blob, pt_off, out_buf, out_cap, out_off);
}
static bool encrypt_packed_core(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::shared_ptr<const EncPolicy>>& pols,
                                const std::vector<std::string>* policies, PolicyLanguage language, size_t n,
                                const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  StageTimer tm(policies ? "ac17::cp_encrypt_packed" : "ac17::kp_encrypt_packed");
  Engine::ArenaScope arena(eng);
  if (pk.h_a.size() != 3 || pk.e_gh_ka.size() != 2) throw RabeError("malformed Ac17PublicKey");
  std::vector<uint32_t> a_off{0};
  std::vector<Fr> A;
  for (const auto& e : pols) {
    A.insert(A.end(), e->tab.begin(), e->tab.end());
    a_off.push_back(a_off.back() + (uint32_t)e->msp.m.size());
  }
  for (size_t i = 0; i < n; i++) if (item_policy[i] >= pols.size()) throw RabeError("encrypt_packed: item_policy out of range");
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + pols[item_policy[i]]->fixed_bytes + (pt_off[i + 1] - pt_off[i]) + 28;
  if (!out_buf || out_cap < out_off[n]) return false;
  tm.lap("policies");
  // randomness in the reference's per-call draw order: s0, s1, msg, nonce -- item after item
  uint8_t* h_s = eng.pinned(0, n * (64 + 32));
  uint8_t* h_rho = h_s + n * 64;
  std::vector<std::array<uint8_t, 12>> nonces(n);
  auto draw_item = [&](Rng& r, size_t i) {
    Fr s0 = r.next_fr(), s1 = r.next_fr(), rho = r.next_fr();
    memcpy(h_s + 64 * i, s0.l, 32);
    memcpy(h_s + 64 * i + 32, s1.l, 32);
    memcpy(h_rho + 32 * i, rho.l, 32);
    r.fill(nonces[i].data(), 12);
  };
  {
    struct Turn { Rng& r; explicit Turn(Rng& x) : r(x) { r.begin_draws(); } ~Turn() { r.end_draws(); } } turn(rng);
    if (rng.unordered() && n >= 1024) {          // OS randomness has no order: blocks of items draw on their own sources
      const size_t blocks = (n + 255) / 256;
      parallel_for(blocks, [&](size_t b) {
        BatchRng local;
        for (size_t i = b * 256; i < n && i < (b + 1) * 256; i++) draw_item(local, i);
      });
    } else {
      for (size_t i = 0; i < n; i++) draw_item(rng, i);
    }
  }
  std::vector<uint32_t> item_a_off(n), row_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) {
    item_a_off[i] = a_off[item_policy[i]];
    row_off[i + 1] = row_off[i] + (uint32_t)pols[item_policy[i]]->msp.m.size();
  }
  const size_t total_rows = row_off[n];
  tm.lap("draws");
  rhip_ac17_pk* dpk = eng.ac17_pk(pk.g, pk.h_a, pk.e_gh_ka);
  rhip_gt_table* egt = eng.gt_generator_table();
  auto fA = flatten_fr(A);
  DBuf dA(&eng, fA.data(), fA.size()), dio(&eng, item_a_off.data(), n * 4), ds(&eng, n * 64), drho(&eng, n * 32),
      dm(&eng, n * 384), dc0(&eng, n * 3 * 128), dc(&eng, total_rows * 3 * 64 + 4), dcp(&eng, n * 384);
  rhip_ctx* cx = eng.ctx();
  eng.check(rhip_upload_async(cx, ds.ptr(), h_s, n * 64), "upload");
  eng.check(rhip_upload_async(cx, drho.ptr(), h_rho, n * 32), "upload");
  // records and sealing on the device (records.h): per policy a template of the literal bytes with the elements dropped in
  std::vector<RecordLayout> layouts(pols.size());
  for (size_t p_ = 0; p_ < pols.size(); p_++) {
    RecordLayout& L = layouts[p_];
    const AbePolicy& msp = pols[p_]->msp;
    if (policies) {                               // Ac17CpCiphertext: policy text + language
      L.str((*policies)[p_]);
      L.u8((language == PolicyLanguage::HumanPolicy) ? 1 : 0);
    } else {                                      // Ac17KpCiphertext: the attribute strings
      L.u32((uint32_t)msp.pi.size());
      for (const auto& a : msp.pi) L.str(a);
    }
    L.u32(3);
    L.src(0, 0, 384);
    L.u32((uint32_t)msp.m.size());
    for (size_t r = 0; r < msp.m.size(); r++) {
      L.str(msp.pi[r]);
      L.u32(3);
      L.src(1, (uint32_t)(192 * r), 192);
    }
    L.src(2, 0, 384);
    if (L.bytes() + 4 != pols[p_]->fixed_bytes) throw RabeError("ac17 encrypt_packed: record layout and size disagree");
  }
  // The batch can go through the device in PARTS of >= 16 384 items, a part's records copied out (10.4 KB per item at 50 rows, 15 ms per
  // 65 536 items) on the side stream while the next part's group arithmetic runs; randomness was drawn above for the whole batch, item
  // after item, so the bytes do not depend on the cut (tests/test_gpu_packed.py).  OFF by default (RABE_AC17_ENC_PARTS=4 turns it on):
  // measured no faster -- 37.2 against 36.9 ms at 65 536 items -- because the runtime's D2H copy of such a block is a blit kernel that
  // covers the chip: k_table_pow_gt of the next part takes 5.3 instead of 1.3 ms under it.
  static const size_t max_parts = [] { const char* e = getenv("RABE_AC17_ENC_PARTS"); const long v = e ? atol(e) : 1; return (size_t)(v > 0 ? v : 1); }();
  const size_t parts = std::max<size_t>(1, std::min<size_t>(max_parts, n / 16384));
  std::vector<PendingCopy> pending(parts);
  if (parts > 1) eng.pinned_reserve((size_t)out_off[n] + parts * ((size_t)8 << 20) + 300 * n);          // staging of every part + their parameter packs: no growth inside the loop
  // every part's row offsets (relative to the part's first row), uploaded once and asynchronously: nothing inside the loop may wait for
  // the stream, or the host would queue part k + 1 only after part k has finished
  std::vector<uint32_t> ro_all(n + parts);
  std::vector<size_t> ro_at(parts);
  for (size_t part = 0, at = 0; part < parts; part++) {
    const size_t lo = n * part / parts, hi = n * (part + 1) / parts;
    ro_at[part] = at;
    for (size_t i = lo; i <= hi; i++) ro_all[at++] = row_off[i] - row_off[lo];
  }
  uint8_t* const h_ro = eng.pinned_bump(ro_all.size() * 4);
  memcpy(h_ro, ro_all.data(), ro_all.size() * 4);
  DBuf d_ro_all(&eng, ro_all.size() * 4);
  eng.check(rhip_upload_async(cx, d_ro_all.ptr(), h_ro, ro_all.size() * 4), "upload");
  for (size_t part = 0; part < parts; part++) {
    const size_t lo = n * part / parts, hi = n * (part + 1) / parts, np = hi - lo;
    const uint32_t* const ro = ro_all.data() + ro_at[part];
    const uint32_t* const dro_p = d_ro_all.as<uint32_t>() + ro_at[part];
    const size_t rows_p = ro[np];
    rhip_g1* const dc_p = (rhip_g1*)(dc.as<uint8_t>() + 192ull * row_off[lo]);
    eng.check(rhip_gt_table_pow(cx, egt, np, drho.as<rhip_fr>() + lo, dm.as<rhip_gt>() + lo), "rhip_gt_table_pow");
    eng.check(rhip_ac17_cp_encrypt_batch(cx, dpk, np, dA.as<rhip_fr>(), dio.as<uint32_t>() + lo, dro_p, rows_p, ds.as<rhip_fr>() + 2 * lo,
                                         dm.as<rhip_gt>() + lo, dc0.as<rhip_g2>() + 3 * lo, dc_p, dcp.as<rhip_gt>() + lo), "rhip_ac17_cp_encrypt_batch");
    std::vector<uint64_t> src_off(3 * np);
    for (size_t i = 0; i < np; i++) { src_off[i] = 384ull * i; src_off[np + i] = 192ull * ro[i]; src_off[2 * np + i] = 384ull * i; }
    emit_sealed_records(eng, layouts, np, item_policy + lo, {dc0.as<uint8_t>() + 384ull * lo, (const void*)dc_p, dcp.as<uint8_t>() + 384ull * lo}, src_off,
                        dm.as<uint8_t>() + 384ull * lo, (const uint8_t*)nonces.data() + 12 * lo, pt_blob, pt_off + lo, out_off + lo, out_buf,
                        parts > 1 ? &pending[part] : nullptr);
  }
  for (auto& pc : pending) pc.wait(eng);
  tm.lap("device: group arithmetic, records, sealing; copies out beside the next part");
  return true;
}

static void* make_ac17_sk_lines(Engine& eng, const void* arg) {
  const std::string& k0 = *(const std::string*)arg;
  DBuf d(&eng, k0.data(), k0.size());
  rhip_ac17_sk_lines* lines = nullptr;
  eng.check(rhip_ac17_sk_prepare(eng.ctx(), 1, d.as<rhip_g2>(), &lines), "rhip_ac17_sk_prepare");
  return lines;
}
static void destroy_ac17_sk_lines(void* h) { rhip_ac17_sk_lines_destroy((rhip_ac17_sk_lines*)h); }
// status[i]: 0 ok, -1 the key does not satisfy the policy / malformed record / authentication failure (errors[i] says which).
// pt_buf / pt_cap: caller-allocated; a capacity of ct_off[n] bytes always suffices (a plaintext is 28 bytes shorter than its sealed
// form).  Returns false, before any work, when pt_cap is smaller than that.
//
// The records come from outside: the offsets are validated against ct_len before anything is read (monotone, inside the blob -- an
// item whose own bounds are bad fails alone), and unless `trusted` is set every decoded element goes through one batched membership
// pass on the GPU (coordinates < p, rows on the G1 curve, c_0 in the r-torsion of the twist, c_p in the order-r subgroup of Fq12:
// what rabe-bn's decoding establishes, FieldError::NotMember) -- the Gt arithmetic downstream (cyclotomic squarings, conjugate as
// inverse) is only valid inside the subgroup.  `trusted` is for ciphertexts this process produced itself.
// KP-ABE's decrypt (:625-675) is the same product of pairings with the roles of the two sides' names swapped: the policy is the KEY's, the
// attribute list the ciphertext's (its record starts with the attribute strings instead of policy text + language); k_p is absent.
// plans of the packed decrypt, kept across calls: one bucket per key fingerprint (attributes or policy + row names), inside it one plan
// per policy text.  A bucket that has grown past its cap is REPLACED (calls that still hold the old one finish on it).
struct Ac17PolPlan {
  std::string text; PolicyLanguage lang; std::string err; PrunedList lst; std::vector<uint32_t> sk_sel;
  std::mutex mu; std::atomic<bool> have_rows{false}; std::vector<std::string> row_names; std::vector<uint32_t> ct_sel;
};
struct Ac17PlanBucket {
  std::unordered_map<uint64_t, std::vector<std::shared_ptr<Ac17PolPlan>>> plans;
  std::shared_mutex mu;
  std::atomic<size_t> n{0};
};
struct Ac17PlanCache {
  static std::shared_ptr<Ac17PlanBucket> get(const std::string& fingerprint) {
    static std::mutex mu;
    static std::unordered_map<std::string, std::shared_ptr<Ac17PlanBucket>> m;
    std::lock_guard<std::mutex> g(mu);
    if (m.size() > 64) m.clear();                                     // keys come and go: bounded
    auto& b = m[fingerprint];
    if (!b || b->n.load() > 4096) b = std::make_shared<Ac17PlanBucket>();       // policies come and go as well
    return b;
  }
};
struct DecKey {
  const Ac17SecretKey& sk;
  const std::vector<std::string>* cp_attrs;      // CP: the key's attributes; the policy comes with every ciphertext
  const PolicyRef* kp_policy;                    // KP: the key's policy; the attributes come with every ciphertext
};
static bool decrypt_packed_core(Engine& eng, const DecKey& sk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                                int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors);
bool cp_decrypt_packed(Engine& eng, const Ac17CpSecretKey& sk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                       int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors) {
  if (sk.sk.k_p.size() != 3) throw RabeError("malformed Ac17CpSecretKey");
  return decrypt_packed_core(eng, DecKey{sk.sk, &sk.attr, nullptr}, n, ct_blob, ct_len, ct_off, trusted, status, pt_buf, pt_cap, pt_off, errors);
}
bool kp_decrypt_packed(Engine& eng, const Ac17KpSecretKey& sk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                       int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors) {
  if (!(sk.sk.k_p.empty() || sk.sk.k_p.size() == 3)) throw RabeError("malformed Ac17KpSecretKey");
  return decrypt_packed_core(eng, DecKey{sk.sk, nullptr, &sk.policy}, n, ct_blob, ct_len, ct_off, trusted, status, pt_buf, pt_cap, pt_off, errors);
}
static bool decrypt_packed_core(Engine& eng, const DecKey& key, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                                int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors) {
  const bool kp = key.kp_policy != nullptr;
  const Ac17SecretKey& core = key.sk;
  StageTimer tm(kp ? "ac17::kp_decrypt_packed" : "ac17::cp_decrypt_packed");
  Engine::ArenaScope arena(eng);
  errors->assign(n, "");
  if (!ct_off || (n && !ct_blob)) throw RabeError("cp_decrypt_packed: null input");
  // bounds first: everything below trusts [ct_off[i], ct_off[i+1]) to lie inside the blob
  uint64_t span = 0;
  for (size_t i = 0; i < n; i++) {
    if (ct_off[i] > ct_off[i + 1] || ct_off[i + 1] > ct_len) (*errors)[i] = "deserialize: record offsets are not monotone inside the blob";
    else span += ct_off[i + 1] - ct_off[i];
  }
  if (!pt_buf || pt_cap < span) return false;
  if (core.k_0.size() != 3) throw RabeError("malformed AC17 secret key");
  for (const auto& row : core.k) if (row.second.size() != 3) throw RabeError("malformed AC17 secret key: a row does not have 3 elements");
  PolicyNode kp_tree;                            // KP: the key's policy, parsed once (a policy that does not parse fails the call like kp_decrypt)
  if (kp) kp_tree = parse_or_error(key.kp_policy->first, key.kp_policy->second);
  // per distinct policy text (items of a batch repeat a few): the tree, the verdict for this key, the key-side selection and --
  // for the row layout the first item with that policy shows -- the ciphertext-side selection.  The plans are a pure function of
  // (key attributes / key policy, key row names, policy text): they are kept ACROSS calls (Ac17PlanCache above) -- a queue batch of a few
  // dozen requests spent 1.5 of its 6 ms re-parsing and re-pruning the same sixteen policies.
  typedef Ac17PolPlan PolPlan;
  std::string fp(kp ? "K" : "C");
  if (kp) { fp += key.kp_policy->first; fp.push_back((char)('0' + (int)key.kp_policy->second)); }
  else for (const auto& a : *key.cp_attrs) { fp += a; fp.push_back('\x1f'); }
  fp.push_back('\x1e');
  for (const auto& row : core.k) { fp += row.first; fp.push_back('\x1f'); }
  const std::shared_ptr<Ac17PlanBucket> bucket_owner = Ac17PlanCache::get(fp);
  auto& plans = bucket_owner->plans;
  std::shared_mutex& plans_mu = bucket_owner->mu;
  // what a plan's shared row layout looks like in THIS call's gather (filled after the parse, serially)
  struct PlanShape { int shape = -1; uint32_t e_c0 = 0, e_row0 = 0, e_cp = 0, e_rows = 0; };
  std::unordered_map<const PolPlan*, PlanShape> shapes;
  auto plan_of = [&](const uint8_t* txt, size_t len, PolicyLanguage lang) -> PolPlan* {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)lang;
    for (size_t i = 0; i + 8 <= len; i += 8) { uint64_t w; memcpy(&w, txt + i, 8); h = (h ^ w) * 1099511628211ull; }
    for (size_t i = len & ~(size_t)7; i < len; i++) h = (h ^ txt[i]) * 1099511628211ull;
    {
      std::shared_lock<std::shared_mutex> g(plans_mu);          // the common case: the policy has been seen, readers do not serialise
      auto it = plans.find(h);
      if (it != plans.end())
        for (auto& e : it->second) if (e->lang == lang && e->text.size() == len && memcmp(e->text.data(), txt, len) == 0) return e.get();
    }
    std::unique_lock<std::shared_mutex> g(plans_mu);
    auto& bucket = plans[h];
    for (auto& e : bucket) if (e->lang == lang && e->text.size() == len && memcmp(e->text.data(), txt, len) == 0) return e.get();
    auto e = std::make_shared<PolPlan>();
    e->text.assign((const char*)txt, len);
    e->lang = lang;
    try {
      if (kp) {                                  // e->text = the record's head: u32 count, then (u32 length, bytes) per attribute
        std::vector<std::string> attrs;
        const uint8_t* q = (const uint8_t*)e->text.data();
        const uint32_t cnt = get_u32(q); q += 4;
        for (uint32_t k = 0; k < cnt; k++) { const uint32_t l = get_u32(q); q += 4; attrs.emplace_back((const char*)q, l); q += l; }
        if (!traverse_policy(attrs, kp_tree)) throw RabeError("Error in kp_decrypt: attributes in ct do not match policy in sk.");
        if (!calc_pruned(attrs, kp_tree, &e->lst)) throw RabeError("Error in kp_decrypt: pruned attributes in sk do not match policy in ct.");
      } else {
        PolicyNode tree = parse_or_error(e->text, lang);
        if (!traverse_policy(*key.cp_attrs, tree)) throw RabeError("Error in cp_decrypt: attributes in SK do not match policy in CT.");
        if (!calc_pruned(*key.cp_attrs, tree, &e->lst)) throw RabeError("Error: attributes in sk do not match policy in ct.");
      }
      for (const auto& cur : e->lst)
        for (size_t r = 0; r < core.k.size(); r++) if (core.k[r].first == cur.first) e->sk_sel.push_back((uint32_t)r);
    } catch (const std::exception& ex) {
      e->err = ex.what();
      if (e->err.empty()) e->err = "policy error";
    }
    bucket.push_back(e);
    bucket_owner->n++;
    return e.get();
  };
  BlobGather gather(eng, ct_blob, ct_len);          // the blob starts for the device now, beside the parsing below (records.h)
  struct View { const uint8_t* c0; const uint8_t* cp; const uint8_t* sealed; uint32_t sealed_len; uint32_t rows; const uint8_t* first_row;
                std::vector<const uint8_t*> row_ptr; const std::vector<uint32_t>* ct_sel; const std::vector<uint32_t>* sk_sel;
                std::vector<uint32_t> own_sel; PolPlan* plan = nullptr; };
  std::vector<View> v(n);
  parallel_for(n, [&](size_t i) {
    if (!(*errors)[i].empty()) return;
    try {
      const uint8_t* p = ct_blob + ct_off[i];
      const uint8_t* end = ct_blob + ct_off[i + 1];
      auto need = [&](size_t k) { if ((size_t)(end - p) < k) throw RabeError("deserialize: truncated input"); };
      const uint8_t* pol;                         // what the plan is keyed by: CP the policy text, KP the whole attribute head
      uint32_t pl;
      PolicyLanguage lang = PolicyLanguage::JsonPolicy;
      if (kp) {
        pol = p;
        need(4); const uint32_t cnt = get_u32(p); p += 4;
        if ((size_t)cnt * 4 > (size_t)(end - p)) throw RabeError("deserialize: truncated input");
        for (uint32_t k = 0; k < cnt; k++) { need(4); const uint32_t l = get_u32(p); p += 4; need(l); p += l; }
        pl = (uint32_t)(p - pol);
      } else {
        need(4); pl = get_u32(p); p += 4;
        need((size_t)pl + 1);
        pol = p; p += pl;
        lang = *p++ ? PolicyLanguage::HumanPolicy : PolicyLanguage::JsonPolicy;
      }
      need(4 + 384); if (get_u32(p) != 3) throw RabeError("deserialize: c_0 does not have 3 elements"); p += 4;
      v[i].c0 = p; p += 384;
      need(4); const uint32_t rows = get_u32(p); p += 4;
      if ((size_t)rows * 200 > (size_t)(end - p)) throw RabeError("deserialize: truncated input");
      v[i].rows = rows;
      v[i].row_ptr.resize(rows);
      static thread_local std::vector<std::pair<const char*, uint32_t>> names;      // per-thread scratch: no allocation per item
      names.resize(rows);
      for (uint32_t r = 0; r < rows; r++) {
        need(4); const uint32_t nl = get_u32(p); p += 4;
        need((size_t)nl + 4 + 192);
        names[r] = {(const char*)p, nl}; p += nl;
        if (get_u32(p) != 3) throw RabeError("deserialize: an AC17 row does not have 3 elements"); p += 4;
        v[i].row_ptr[r] = p; p += 192;
      }
      need(384 + 4); v[i].cp = p; p += 384;
      v[i].sealed_len = get_u32(p); p += 4;
      need(v[i].sealed_len);
      v[i].sealed = p;
      PolPlan* pp = plan_of(pol, pl, lang);
      if (!pp->err.empty()) throw RabeError(pp->err);
      v[i].sk_sel = &pp->sk_sel;
      v[i].plan = pp;
      auto select = [&](std::vector<uint32_t>* out) {          // the name-matching loop of ac17/mod.rs:403-408 as an index list
        for (const auto& cur : pp->lst)
          for (uint32_t r = 0; r < rows; r++)
            if (names[r].second == cur.first.size() && memcmp(names[r].first, cur.first.data(), cur.first.size()) == 0) out->push_back(r);
      };
      bool shared = false;
      if (!pp->have_rows.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> g(pp->mu);
        if (!pp->have_rows.load(std::memory_order_relaxed)) {
          for (uint32_t r = 0; r < rows; r++) pp->row_names.emplace_back(names[r].first, names[r].second);
          select(&pp->ct_sel);
          pp->have_rows.store(true, std::memory_order_release);
        }
      }
      if (pp->row_names.size() == rows) {          // read-only from here on
        shared = true;
        for (uint32_t r = 0; r < rows && shared; r++)
          shared = pp->row_names[r].size() == names[r].second && memcmp(pp->row_names[r].data(), names[r].first, names[r].second) == 0;
      }
      if (shared) v[i].ct_sel = &pp->ct_sel;
      else { select(&v[i].own_sel); v[i].ct_sel = &v[i].own_sel; }
    } catch (const std::exception& ex) {
      (*errors)[i] = ex.what();
      if ((*errors)[i].empty()) (*errors)[i] = "malformed record";
    }
  });
  tm.lap("parse + plan");
  std::vector<size_t> live;
  for (size_t i = 0; i < n; i++) if ((*errors)[i].empty()) live.push_back(i);
  const size_t m = live.size();
  // offsets first (a scan), then the selection lists copied into place on all cores (65 536 items x ~30 entries x 2: 6 ms when serial)
  std::vector<uint32_t> ct_row_off(m + 1, 0), ct_sel_off(m + 1, 0), sk_sel_off(m + 1, 0);
  for (size_t j = 0; j < m; j++) {
    const View& w = v[live[j]];
    ct_row_off[j + 1] = ct_row_off[j] + w.rows;
    ct_sel_off[j + 1] = ct_sel_off[j] + (uint32_t)w.ct_sel->size();
    sk_sel_off[j + 1] = sk_sel_off[j] + (uint32_t)w.sk_sel->size();
  }
  // the lists themselves are written straight into pinned staging (65 536 items: 2 x 8 MB) and uploaded from there
  const size_t n_ct_sel = ct_sel_off[m], n_sk_sel = sk_sel_off[m];
  uint32_t* const ct_sel = (uint32_t*)eng.pinned_bump((n_ct_sel + n_sk_sel + 2) * 4);
  uint32_t* const sk_sel = ct_sel + n_ct_sel + 1;
  {
    const size_t per = 1024, blocks = (m + per - 1) / per;
    parallel_for(blocks, [&](size_t b) {
      for (size_t j = b * per; j < m && j < (b + 1) * per; j++) {
        const View& w = v[live[j]];
        if (!w.ct_sel->empty()) memcpy(ct_sel + ct_sel_off[j], w.ct_sel->data(), w.ct_sel->size() * 4);
        if (!w.sk_sel->empty()) memcpy(sk_sel + sk_sel_off[j], w.sk_sel->data(), w.sk_sel->size() * 4);
      }
    });
  }
  // uploaded at once: a later take from the same pinned block may move the block (Engine::pinned_bump waits for the stream first)
  DBuf d_sel(&eng, (n_ct_sel + n_sk_sel + 2) * 4);
  eng.check(rhip_upload_async(eng.ctx(), d_sel.ptr(), ct_sel, (n_ct_sel + n_sk_sel + 2) * 4), "upload (selection lists)");
  const uint32_t* const d_ct_sel = d_sel.as<uint32_t>();
  const uint32_t* const d_sk_sel = d_ct_sel + n_ct_sel + 1;
  if (!m) {
    pt_off[0] = 0;
    for (size_t i = 0; i < n; i++) { pt_off[i + 1] = 0; status[i] = -1; }
    return true;
  }
  const size_t total_rows = ct_row_off[m];
  rhip_ctx* cx = eng.ctx();
  // The elements are gathered out of the device copy of the blob.  Items that share a policy and its row names share one part list --
  // their records have the same skeleton, hence the same relative offsets (checked on the ends); any other item brings its own.
  std::vector<uint64_t> sealed_off(m);
  std::vector<uint32_t> sealed_len(m);
  auto parts_of = [&](const View& w, const uint8_t* rec) {
    std::vector<RecordLayout::Part> parts;
    parts.push_back({(uint32_t)(w.c0 - rec), 384, 0, 0});
    for (uint32_t r = 0; r < w.rows; r++) parts.push_back({(uint32_t)(w.row_ptr[r] - rec), 192, 1, 192 * r});
    parts.push_back({(uint32_t)(w.cp - rec), 384, 2, 0});
    return parts;
  };
  for (size_t j = 0; j < m; j++) {
    const View& w = v[live[j]];
    const uint8_t* rec = ct_blob + ct_off[live[j]];
    sealed_off[j] = (uint64_t)(w.sealed - ct_blob);
    sealed_len[j] = w.sealed_len;
    const uint32_t e_c0 = (uint32_t)(w.c0 - rec), e_row0 = w.rows ? (uint32_t)(w.row_ptr[0] - rec) : 0u, e_cp = (uint32_t)(w.cp - rec);
    int shape = -1;
    if (w.ct_sel != &w.own_sel) {                 // the plan's shared row layout: its shape is remembered in the plan itself
      PlanShape& ps = shapes[w.plan];
      if (ps.shape < 0) {
        ps.shape = (int)gather.add_shape(nullptr, parts_of(w, rec));
        ps.e_c0 = e_c0; ps.e_row0 = e_row0; ps.e_cp = e_cp; ps.e_rows = w.rows;
      }
      if (ps.e_c0 == e_c0 && ps.e_row0 == e_row0 && ps.e_cp == e_cp && ps.e_rows == w.rows) shape = ps.shape;
    }
    if (shape < 0) shape = (int)gather.add_shape(nullptr, parts_of(w, rec));
    gather.item(ct_off[live[j]], (uint32_t)shape);
  }
  std::vector<uint8_t> k0 = flatten(core.k_0), kp_bytes = core.k_p.size() == 3 ? flatten(core.k_p) : std::vector<uint8_t>(3 * 64, 0), kk;   // KP: prod_h starts from G1::zero()
  for (const auto& row : core.k) for (const auto& x : row.second) kk.insert(kk.end(), x.begin(), x.end());
  if (kk.empty()) kk.assign(64, 0);
  std::vector<uint32_t> sk_row_off{0, (uint32_t)core.k.size()}, sk_idx(m, 0);
  DBuf d1(&eng, m * 384), d2(&eng, total_rows * 192 + 4), d4(&eng, m * 384), dout(&eng, m * 384);
  std::vector<uint64_t> dst_off(3 * m);
  for (size_t j = 0; j < m; j++) { dst_off[j] = 384ull * j; dst_off[m + j] = 192ull * ct_row_off[j]; dst_off[2 * m + j] = 384ull * j; }
  ParamPack pp(eng);
  const size_t h3 = pp.add(ct_row_off), h6 = pp.add(kk), h7 = pp.add(sk_row_off), h8 = pp.add(kp_bytes), h9 = pp.add(sk_idx),
               h11 = pp.add(ct_sel_off), h13 = pp.add(sk_sel_off);
  pp.upload();

  tm.lap("selection tables");
  gather.run({d1.ptr(), d2.ptr(), d4.ptr()}, dst_off);
  const uint32_t* d3 = pp.dev<uint32_t>(h3);
  // decoding checks over the gathered elements, on the side context beside the decrypt kernels; a non-member fails its item only (the
  // batch still runs: the kernels terminate on any input, the item's result is discarded below)
  std::unique_ptr<MemberChecks> mc;
  std::unique_ptr<WalkedG2> walked;
  DBuf seg_keep;
  if (!trusted) {
    mc.reset(new MemberChecks(eng));
    mc->add(1, d2.ptr(), 3 * total_rows, d3, m, 3);
    mc->add(3, d4.ptr(), m);
    // c_0: the decrypt walks all three elements of every ciphertext -- their subgroup membership comes out of its Miller loops
    if (walk_checks()) {
      std::vector<uint32_t> seg(m + 1);
      for (size_t j = 0; j <= m; j++) seg[j] = (uint32_t)j;
      DBuf d_seg(&eng, (m + 1) * 4);
      uint8_t* const hs = eng.pinned_bump((m + 1) * 4);
      memcpy(hs, seg.data(), (m + 1) * 4);
      eng.check(rhip_upload_async(cx, d_seg.ptr(), hs, (m + 1) * 4), "upload");
      eng.check(rhip_ctx_wait_for(mc->ctx(), cx), "rhip_ctx_wait_for");
      seg_keep = std::move(d_seg);
      walked.reset(new WalkedG2(eng, *mc, d1.ptr(), 3 * m, seg_keep.as<uint32_t>(), seg, 3));
    } else {
      mc->add(2, d1.ptr(), 3 * m);
    }
  }
  // the key's prepared k_0 lines are a function of the key alone: kept across calls (a server decrypts with the same key again and again)
  std::string k0_key((const char*)k0.data(), k0.size());
  rhip_ac17_sk_lines* lines = (rhip_ac17_sk_lines*)eng.aux("ac17_sk_lines", k0_key, make_ac17_sk_lines, &k0_key, destroy_ac17_sk_lines, 4);
  if (walked) walked->arm();
  eng.check(rhip_ac17_cp_decrypt_batch_prepared(cx, m, d1.as<rhip_g2>(), d2.as<rhip_g1>(), d3, d4.as<rhip_gt>(), lines, pp.dev<rhip_g1>(h6),
                                                pp.dev<uint32_t>(h7), pp.dev<rhip_g1>(h8), pp.dev<uint32_t>(h9), d_ct_sel,
                                                pp.dev<uint32_t>(h11), d_sk_sel, pp.dev<uint32_t>(h13), dout.as<rhip_gt>()),
            "rhip_ac17_cp_decrypt_batch_prepared");
  if (mc) {
    mc->collect();
    const auto &ok_rows = mc->ok(0), &ok_cp = mc->ok(1);
    std::vector<uint8_t> ok_c0(m, 1);
    if (!walked) { const auto& e = mc->ok(2); for (size_t j = 0; j < m; j++) for (int t = 0; t < 3; t++) if (!e[3 * j + t]) ok_c0[j] = 0; }
    for (size_t j = 0; j < m; j++) {
      const char* bad = !ok_c0[j] ? "deserialize: c_0 element is not a member of G2 (FieldError::NotMember)" : nullptr;
      if (!bad && !ok_rows[j]) bad = "deserialize: a row element is not a point of G1 (FieldError::NotMember)";
      if (!bad && !ok_cp[j]) bad = "deserialize: c_p is not a member of Gt (FieldError::NotMember)";
      if (bad) (*errors)[live[j]] = bad;
    }
  }
  // KDF + AES-GCM open on the device: the decrypted Gt never leaves HBM; plaintext bytes come back in one copy
  open_sealed_records(eng, n, live, dout.ptr(), gather.dev_blob(), sealed_off, sealed_len, status, pt_buf, pt_off, errors);
  if (walked) {          // the walk's verdicts are read AFTER the open was queued behind the pairings (records.h: retract_item)
    std::vector<uint8_t> ok_c0;
    walked->finish(&ok_c0);
    for (size_t j = 0; j < m; j++)
      if (!ok_c0[j]) retract_item(live[j], "deserialize: c_0 element is not a member of G2 (FieldError::NotMember)", status, pt_buf, pt_off, errors);
  }
  tm.lap(trusted ? "device: gather, pairings, open" : "device: gather, pairings, open; membership beside");
  return true;
}

// ---------------------------------------------------------------------------------------------- KP-ABE
// The Fr half of kp_keygen (:455-538) for one key: 3 scalars per MSP row (multiples of g), in row order, and br = (b0 r0, b1 r1, r0 + r1)
// (multiples of h).  `hashes`: the label hashes of a policy -- rows h(pi_i || l || t) then columns h("0" || j || l || t), l major -- which do
// not depend on the key.
struct KpPolicy { AbePolicy msp; std::vector<Fr> row_h /*[rows][t][l]*/, col_h /*[cols][t][l], j = 0 unused*/; };
static KpPolicy kp_policy(const std::string& policy, PolicyLanguage lang) {
  KpPolicy kp;
  kp.msp = calculate_msp(parse_or_error(policy, lang));
  const size_t cols = kp.msp.m[0].size(), rows = kp.msp.m.size();
  kp.col_h.assign(cols * 6, fr_zero());
  for (size_t j = 1; j < cols; j++)
    for (int t = 0; t < 2; t++)
      for (int l = 0; l < 3; l++) kp.col_h[(j * 2 + t) * 3 + l] = sha3_hash_fr(std::string("0") + std::to_string(j) + std::to_string(l) + std::to_string(t));
  kp.row_h.resize(rows * 6);
  for (size_t i = 0; i < rows; i++)
    for (int t = 0; t < 2; t++)
      for (int l = 0; l < 3; l++) kp.row_h[(i * 2 + t) * 3 + l] = sha3_hash_fr(kp.msp.pi[i] + std::to_string(l) + std::to_string(t));
  return kp;
}
static void kp_keygen_scalars(const Ac17MasterKey& msk, const KpPolicy& kp, const Fr a_inv[2], const Fr& r0, const Fr& r1, const Fr* sigma_prime,
                              const Fr* sigma, Fr* scal /*[3 rows]*/, Fr br[3]) {
  const AbePolicy& msp = kp.msp;
  const size_t cols = msp.m[0].size(), rows = msp.m.size();
  br[0] = fr_mul(msk.b[0], r0); br[1] = fr_mul(msk.b[1], r1); br[2] = fr_add(r0, r1);
  // T[j][t] = sum_{j' <= j} ( sum_l h("0"||j'||l||t) br_l / a_t - sigma'_{j'-1} ): the reference's `_temp` is declared
  // outside the column loop and never reset (:496), so it accumulates over the columns -- restated verbatim.
  std::vector<Fr> T(cols * 2, fr_zero());
  for (int t = 0; t < 2; t++) {
    Fr acc = fr_zero();
    for (size_t j = 1; j < cols; j++) {
      Fr hsum = fr_zero();
      for (int l = 0; l < 3; l++) hsum = fr_add(hsum, fr_mul(kp.col_h[(j * 2 + t) * 3 + l], br[l]));
      acc = fr_add(acc, fr_sub(fr_mul(hsum, a_inv[t]), sigma_prime[j - 1]));
      T[j * 2 + t] = acc;
    }
  }
  for (size_t i = 0; i < rows; i++) {
    for (int t = 0; t < 2; t++) {
      Fr hsum = sigma[i];
      for (int l = 0; l < 3; l++) hsum = fr_add(hsum, fr_mul(kp.row_h[(i * 2 + t) * 3 + l], br[l]));
      Fr k = fr_mul(hsum, a_inv[t]);
      for (size_t j = 1; j < cols; j++) {
        if (msp.m[i][j] == 1) k = fr_add(k, T[j * 2 + t]);
        else if (msp.m[i][j] == -1) k = fr_sub(k, T[j * 2 + t]);
      }
      scal[3 * i + t] = k;
    }
    Fr k3 = fr_neg(sigma[i]);
    for (size_t j = 1; j < cols; j++) {
      if (msp.m[i][j] == 1) k3 = fr_sub(k3, sigma_prime[j - 1]);
      else if (msp.m[i][j] == -1) k3 = fr_add(k3, sigma_prime[j - 1]);
    }
    scal[3 * i + 2] = k3;
  }
}
Ac17KpSecretKey kp_keygen(Engine& eng, Rng& rng, const Ac17MasterKey& msk, const std::string& policy, PolicyLanguage lang) {   // :439-547
  if (msk.a.size() != 2 || msk.b.size() != 2 || msk.g_k.size() != 3) throw RabeError("malformed Ac17MasterKey: a, b must have 2 and g_k 3 elements");
  const KpPolicy kp = kp_policy(policy, lang);
  const AbePolicy& msp = kp.msp;
  const size_t cols = msp.m[0].size(), rows = msp.m.size();
  // draw order: r0, r1 (:455-459); sigma'_1..sigma'_{c-1} (:471-474); sigma_i per row (:481)
  Fr r0 = rng.next_fr(), r1 = rng.next_fr();
  std::vector<Fr> sigma_prime;
  for (size_t j = 0; j + 1 < cols; j++) sigma_prime.push_back(rng.next_fr());
  std::vector<Fr> sigma;
  for (size_t i = 0; i < rows; i++) sigma.push_back(rng.next_fr());
  Fr a_inv[2] = {must_inv(msk.a[0]), must_inv(msk.a[1])};
  std::vector<Fr> scal(3 * rows);
  Fr br[3];
  sigma_prime.push_back(fr_zero());              // never read (cols - 1 entries are), keeps .data() valid for a one-column policy
  kp_keygen_scalars(msk, kp, a_inv, r0, r1, sigma_prime.data(), sigma.data(), scal.data(), br);
  std::vector<G1> pts = eng.g1_mul(std::vector<G1>(scal.size(), msk.g), scal);
  // +/- g_k[t] where the first MSP column is +/-1
  std::vector<G1> add_a, add_b;
  std::vector<size_t> where;
  std::vector<G1> neg_gk = eng.g1_mul(msk.g_k, std::vector<Fr>(3, fr_neg(fr_one())));
  for (size_t i = 0; i < rows; i++)
    for (int t = 0; t < 3; t++)
      if (msp.m[i][0] != 0) { add_a.push_back(pts[3 * i + t]); add_b.push_back(msp.m[i][0] == 1 ? msk.g_k[t] : neg_gk[t]); where.push_back(3 * i + t); }
  if (!where.empty()) {
    std::vector<G1> sum = g1_add(eng, add_a, add_b);
    for (size_t q = 0; q < where.size(); q++) pts[where[q]] = sum[q];
  }
  Ac17KpSecretKey out;
  out.policy = {policy, lang};
  out.sk.k_0 = eng.g2_mul({msk.h, msk.h, msk.h}, {br[0], br[1], br[2]});
  for (size_t i = 0; i < rows; i++) out.sk.k.push_back({msp.pi[i], {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}});
  return out;
}
// n calls of ac17::kp_keygen (:439-547) under one master key: item i's policy = policies[item_policy[i]].  Draw order per item as above.
// The Fr half runs on all host cores; the group half is ONE fixed-base launch per group (window tables of msk.g / msk.h, kept across
// calls), one batched addition for the rows whose first MSP column is +/-1 (+/- g_k), and the records are written on the device.
// Record = Ac17KpSecretKey: policy text, language, k_0 (3 G2), rows (name, 3 G1), an empty k_p.  Buffers as cp_keygen_packed.
bool kp_keygen_packed(Engine& eng, Rng& rng, const Ac17MasterKey& msk, const std::vector<std::string>& policies, PolicyLanguage lang, size_t n,
                      const uint32_t* item_policy, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  StageTimer tm("ac17::kp_keygen_packed");
  Engine::ArenaScope arena(eng);
  eng.scrub_when_done();          // master-key-derived scalars pass through the staging buffers
  if (msk.a.size() != 2 || msk.b.size() != 2 || msk.g_k.size() != 3) throw RabeError("malformed Ac17MasterKey: a, b must have 2 and g_k 3 elements");
  if (n && (!item_policy || !out_off)) throw RabeError("kp_keygen_packed: null input");
  std::vector<KpPolicy> kps;
  std::vector<RecordLayout> layouts(policies.size());
  for (size_t p_ = 0; p_ < policies.size(); p_++) {
    kps.push_back(kp_policy(policies[p_], lang));
    RecordLayout& L = layouts[p_];
    const AbePolicy& msp = kps.back().msp;
    L.str(policies[p_]);
    L.u8(lang == PolicyLanguage::HumanPolicy ? 1 : 0);
    L.u32(3);
    L.src(0, 0, 384);
    L.u32((uint32_t)msp.m.size());
    for (size_t r = 0; r < msp.m.size(); r++) { L.str(msp.pi[r]); L.u32(3); L.src(1, (uint32_t)(192 * r), 192); }
    L.u32(0);                                   // k_p: empty for a KP key
  }
  for (size_t i = 0; i < n; i++) if (item_policy[i] >= policies.size()) throw RabeError("kp_keygen_packed: item_policy out of range");
  out_off[0] = 0;
  for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + layouts[item_policy[i]].bytes();
  if (!out_buf || out_cap < out_off[n]) return false;
  if (!n) return true;
  std::vector<uint32_t> row_off(n + 1, 0), draw_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) {
    const AbePolicy& msp = kps[item_policy[i]].msp;
    row_off[i + 1] = row_off[i] + (uint32_t)msp.m.size();
    draw_off[i + 1] = draw_off[i] + 2 + (uint32_t)(msp.m[0].size() - 1) + (uint32_t)msp.m.size();
  }
  const size_t total_rows = row_off[n];
  std::vector<Fr> draws(draw_off[n] + 1);
  {
    struct Turn { Rng& r; explicit Turn(Rng& x) : r(x) { r.begin_draws(); } ~Turn() { r.end_draws(); } } turn(rng);
    if (rng.unordered() && n >= 1024) {
      const size_t blocks = (n + 255) / 256;
      parallel_for(blocks, [&](size_t b) {
        BatchRng local;
        for (size_t i = b * 256; i < n && i < (b + 1) * 256; i++) for (uint32_t d = draw_off[i]; d < draw_off[i + 1]; d++) draws[d] = local.next_fr();
      });
    } else {
      for (size_t d = 0; d < draw_off[n]; d++) draws[d] = rng.next_fr();
    }
  }
  tm.lap("policies + draws");
  Fr a_inv[2] = {must_inv(msk.a[0]), must_inv(msk.a[1])};
  uint8_t* h_scal = eng.pinned(0, total_rows * 96 + n * 96 + 32);          // row scalars | br
  uint8_t* h_br = h_scal + total_rows * 96;
  uint8_t* h_add = eng.pinned(1, total_rows * 192 + 4);                     // +/- g_k[t] per row element, the point at infinity where nothing is added
  G1 neg_gk[3];
  {
    std::vector<G1> ng = eng.g1_mul(msk.g_k, std::vector<Fr>(3, fr_neg(fr_one())));
    for (int t = 0; t < 3; t++) neg_gk[t] = ng[t];
  }
  parallel_for(n, [&](size_t i) {
    const KpPolicy& kp = kps[item_policy[i]];
    const size_t cols = kp.msp.m[0].size(), rows = kp.msp.m.size();
    const Fr* d = draws.data() + draw_off[i];
    std::vector<Fr> scal(3 * rows);
    Fr br[3];
    kp_keygen_scalars(msk, kp, a_inv, d[0], d[1], d + 2, d + 2 + (cols - 1), scal.data(), br);
    memcpy(h_scal + 96 * (size_t)row_off[i], scal.data(), 96 * rows);
    memcpy(h_br + 96 * i, br, 96);
    uint8_t* ad = h_add + 192 * (size_t)row_off[i];
    for (size_t r = 0; r < rows; r++)
      for (int t = 0; t < 3; t++) {
        const int8_t c = kp.msp.m[r][0];
        if (c == 0) memset(ad + 192 * r + 64 * t, 0, 64);
        else memcpy(ad + 192 * r + 64 * t, (c == 1 ? msk.g_k[t] : neg_gk[t]).data(), 64);
      }
  });
  tm.lap("scalars");
  rhip_g1_table* gt = nullptr;
  rhip_g2_table* ht = nullptr;
  msk_tables(eng, msk, &gt, &ht);
  rhip_ctx* cx = eng.ctx();
  DBuf d_scal(&eng, total_rows * 96 + n * 96), d_add(&eng, total_rows * 192 + 4), d_pts(&eng, total_rows * 192 + 4), d_k(&eng, total_rows * 192 + 4),
      d_k0(&eng, n * 384);
  eng.check(rhip_upload_async(cx, d_scal.ptr(), h_scal, total_rows * 96 + n * 96), "upload");
  eng.check(rhip_upload_async(cx, d_add.ptr(), h_add, total_rows * 192), "upload");
  eng.check(rhip_g1_table_mul(cx, gt, 3 * total_rows, d_scal.as<rhip_fr>(), d_pts.as<rhip_g1>()), "rhip_g1_table_mul");
  eng.check(rhip_g1_add(cx, 3 * total_rows, d_pts.as<rhip_g1>(), d_add.as<rhip_g1>(), d_k.as<rhip_g1>()), "rhip_g1_add");
  eng.check(rhip_g2_table_mul(cx, ht, 3 * n, (const rhip_fr*)(d_scal.as<uint8_t>() + total_rows * 96), d_k0.as<rhip_g2>()), "rhip_g2_table_mul");
  std::vector<uint64_t> src_off(2 * n);
  for (size_t i = 0; i < n; i++) { src_off[i] = 384ull * i; src_off[n + i] = 192ull * row_off[i]; }
  emit_plain_records(eng, layouts, n, item_policy, {d_k0.ptr(), d_k.ptr()}, src_off, out_off, out_buf);
  tm.lap("device: fixed-base multiplications, records; one copy out");
  return true;
}

std::vector<Ac17KpCiphertext> kp_encrypt_batch(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::vector<std::string>>& attribute_sets,
                                               const std::vector<Bytes>& datas) {   // :556-616, n times
  if (attribute_sets.size() != datas.size()) throw RabeError("kp_encrypt_batch: attribute sets / datas length mismatch");
  const size_t n = attribute_sets.size();
  std::vector<Ac17KpCiphertext> out(n);
  if (!n) return out;
  // draw order per call: s0, s1, msg, nonce -- item after item
  std::vector<Fr> s, msg_k;
  std::vector<std::array<uint8_t, 12>> nonces(n);
  for (size_t i = 0; i < n; i++) {
    s.push_back(rng.next_fr());
    s.push_back(rng.next_fr());
    msg_k.push_back(rng.next_fr());
    rng.fill(nonces[i].data(), 12);
  }
  // C[y][l] = g*(s0 h(y||l||0) + s1 h(y||l||1)): the CP row kernel with a table of plain label hashes, one block of rows per item
  std::vector<uint32_t> item_a_off(n), row_off(n + 1, 0);
  for (size_t i = 0; i < n; i++) { item_a_off[i] = row_off[i]; row_off[i + 1] = row_off[i] + (uint32_t)attribute_sets[i].size(); }
  const size_t total_rows = row_off[n];
  std::vector<Fr> A(total_rows * 6);
  parallel_for(n, [&](size_t i) {
    for (size_t y = 0; y < attribute_sets[i].size(); y++)
      for (int l = 0; l < 3; l++)
        for (int t = 0; t < 2; t++)
          A[((size_t)row_off[i] + y) * 6 + l * 2 + t] = sha3_hash_fr(attribute_sets[i][y] + std::to_string(l) + std::to_string(t));
  });
  std::vector<Gt> msgs = eng.gt_pow(std::vector<Gt>(n, eng.gt_generator()), msg_k);
  rhip_ac17_pk* dpk = eng.ac17_pk(pk.g, pk.h_a, pk.e_gh_ka);
  auto fA = flatten_fr(A), fs = flatten_fr(s), fm = flatten(msgs);
  DBuf dA(&eng, fA.data(), fA.size()), dio(&eng, item_a_off.data(), n * 4), dro(&eng, row_off.data(), (n + 1) * 4), ds(&eng, fs.data(), fs.size()),
      dm(&eng, fm.data(), fm.size()), dc0(&eng, n * 3 * 128), dc(&eng, total_rows * 3 * 64), dcp(&eng, n * 384);
  int32_t rc = rhip_ac17_cp_encrypt_batch(eng.ctx(), dpk, n, dA.as<rhip_fr>(), dio.as<uint32_t>(), dro.as<uint32_t>(), total_rows,
                                          ds.as<rhip_fr>(), dm.as<rhip_gt>(), dc0.as<rhip_g2>(), dc.as<rhip_g1>(), dcp.as<rhip_gt>());
  std::vector<G2> c0;
  std::vector<G1> c;
  std::vector<Gt> cp;
  if (rc == RHIP_OK) { c0 = fetch<128>(dc0, n * 3); c = fetch<64>(dc, total_rows * 3); cp = fetch<384>(dcp, n); }
  eng.check(rc, "rhip_ac17_cp_encrypt_batch");
  parallel_for(n, [&](size_t i) {
    out[i].attr = attribute_sets[i];
    out[i].ct.c_0 = {c0[3 * i], c0[3 * i + 1], c0[3 * i + 2]};
    for (size_t y = 0; y < attribute_sets[i].size(); y++) {
      size_t g = (size_t)row_off[i] + y;
      out[i].ct.c.push_back({attribute_sets[i][y], {c[3 * g], c[3 * g + 1], c[3 * g + 2]}});
    }
    out[i].ct.c_p = cp[i];
    out[i].ct.ct = encrypt_symmetric(msgs[i].data(), datas[i].data(), datas[i].size(), nonces[i].data());
  });
  return out;
}
Ac17KpCiphertext kp_encrypt(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::string>& attributes, const Bytes& data) {   // :556-616
  return kp_encrypt_batch(eng, rng, pk, {attributes}, {data})[0];
}
std::vector<DecryptResult> kp_decrypt_batch(Engine& eng, const std::vector<const Ac17KpSecretKey*>& sks, const std::vector<const Ac17KpCiphertext*>& cts) {
  if (sks.size() != cts.size()) throw RabeError("kp_decrypt_batch: sks and cts differ in length");
  std::vector<DecItem> items;
  for (size_t i = 0; i < cts.size(); i++)
    items.push_back({&cts[i]->attr, &sks[i]->policy, &cts[i]->ct, &sks[i]->sk, "Error in kp_decrypt: attributes in ct do not match policy in sk.",
                     "Error in kp_decrypt: pruned attributes in sk do not match policy in ct."});
  std::vector<std::string> errors;
  std::vector<Gt> g = decrypt_items(eng, items, &errors);
  std::vector<DecryptResult> out(cts.size());
  for (size_t i = 0; i < cts.size(); i++) {
    if (!errors[i].empty()) { out[i] = {false, {}, errors[i]}; continue; }
    Bytes pt;
    if (decrypt_symmetric(g[i].data(), cts[i]->ct.ct.data(), cts[i]->ct.ct.size(), &pt)) out[i] = {true, pt, ""};
    else out[i] = {false, {}, "decryption error: aead::Error"};
  }
  return out;
}
Gt kp_decrypt_gt(Engine& eng, const Ac17KpSecretKey& sk, const Ac17KpCiphertext& ct) {      // :625-675
  std::vector<std::string> errors;
  std::vector<DecItem> items{{&ct.attr, &sk.policy, &ct.ct, &sk.sk, "Error in kp_decrypt: attributes in ct do not match policy in sk.",
                              "Error in kp_decrypt: pruned attributes in sk do not match policy in ct."}};
  std::vector<Gt> g = decrypt_items(eng, items, &errors);
  if (!errors[0].empty()) throw RabeError(errors[0]);
  return g[0];
}
Bytes kp_decrypt(Engine& eng, const Ac17KpSecretKey& sk, const Ac17KpCiphertext& ct) { return open_or_error(kp_decrypt_gt(eng, sk, ct), ct.ct.ct); }
}  // namespace ac17

// ================================================================================================ BSW
namespace bsw {
std::pair<CpAbePublicKey, CpAbeMasterKey> setup(Engine& eng, Rng& rng) {       // :92-114
  G1 g1 = eng.random_g1(rng);
  G2 g2 = eng.random_g2(rng);
  Fr beta = rng.next_fr();
  Fr alpha = rng.next_fr();
  G1 h = eng.g1_mul({g1}, {beta})[0];
  std::vector<G2> fa = eng.g2_mul({g2, g2}, {must_inv(beta), alpha});
  Gt e = eng.pairing({g1}, {fa[1]})[0];
  return {CpAbePublicKey{g1, g2, h, fa[0], e}, CpAbeMasterKey{beta, fa[1]}};
}
bool keygen(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const CpAbeMasterKey& msk, const std::vector<std::string>& attributes,
            CpAbeSecretKey* out) {      // :125-152
  if (attributes.empty()) return false;
  Fr r = rng.next_fr();
  G2 g2_r = eng.g2_mul({pk.g2}, {r})[0];
  G2 sum = g2_add(eng, {msk.g2_alpha}, {g2_r})[0];
  out->d = eng.g2_mul({sum}, {must_inv(msk.beta)})[0];
  out->d_j.clear();
  std::vector<Fr> rj, hr;
  for (const auto& j : attributes) {
    Fr x = rng.next_fr();
    rj.push_back(x);
    hr.push_back(fr_mul(sha3_hash_fr(j), x));          // (g2*h(j))*r_j = g2*(h(j) r_j)
  }
  std::vector<G1> a1 = eng.g1_mul(std::vector<G1>(attributes.size(), pk.g1), rj);
  std::vector<G2> a2 = eng.g2_mul(std::vector<G2>(attributes.size(), pk.g2), hr);
  a2 = g2_add(eng, std::vector<G2>(attributes.size(), g2_r), a2);
  for (size_t i = 0; i < attributes.size(); i++) out->d_j.push_back({attributes[i], a1[i], a2[i]});
  return true;
}
bool delegate(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const CpAbeSecretKey& sk, const std::vector<std::string>& subset,
              CpAbeSecretKey* out) {       // :162-206
  for (const auto& a : subset) {            // is_subset (tools/mod.rs:24-28)
    bool found = false;
    for (const auto& d : sk.d_j) if (d.string == a) { found = true; break; }
    if (!found) return false;
  }
  if (subset.empty()) return false;
  Fr r = rng.next_fr();
  std::vector<G1> old1;
  std::vector<G2> old2;
  std::vector<Fr> rj, hr;
  for (const auto& a : subset) {
    Fr x = rng.next_fr();
    rj.push_back(x);
    hr.push_back(fr_add(fr_mul(sha3_hash_fr(a), x), r));        // (g2*h(a))*r_j + g2*r = g2*(h(a) r_j + r)
    for (const auto& d : sk.d_j) if (d.string == a) { old1.push_back(d.g1); old2.push_back(d.g2); break; }
  }
  std::vector<G1> n1 = g1_add(eng, old1, eng.g1_mul(std::vector<G1>(subset.size(), pk.g1), rj));
  std::vector<G2> n2 = g2_add(eng, old2, eng.g2_mul(std::vector<G2>(subset.size(), pk.g2), hr));
  out->d = g2_add(eng, {sk.d}, {eng.g2_mul({pk.f}, {r})[0]})[0];
  out->d_j.clear();
  for (size_t i = 0; i < subset.size(); i++) out->d_j.push_back({subset[i], n1[i], n2[i]});
  return true;
}
CpAbeCiphertext encrypt(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const std::string& policy, PolicyLanguage language,
                        const Bytes& plaintext) {       // :217-251
  Fr secret = rng.next_fr();
  Gt msg = eng.random_gt(rng);
  PolicyNode tree = parse_or_error(policy, language);
  NamedFr shares;
  gen_shares_policy(secret, tree, rng, &shares);
  CpAbeCiphertext ct;
  ct.policy = {policy, language};
  ct.c = eng.g1_mul({pk.h}, {secret})[0];
  ct.c_p = eng.gt_mul({eng.gt_pow({pk.e_gg_alpha}, {secret})[0]}, {msg})[0];
  std::vector<Fr> k1, k2;
  for (const auto& sh : shares) {
    k1.push_back(sh.second);
    k2.push_back(fr_mul(sha3_hash_fr(remove_index(sh.first)), sh.second));
  }
  std::vector<G1> p1 = eng.g1_mul(std::vector<G1>(shares.size(), pk.g1), k1);
  std::vector<G2> p2 = eng.g2_mul(std::vector<G2>(shares.size(), pk.g2), k2);
  for (size_t i = 0; i < shares.size(); i++) ct.c_y.push_back({shares[i].first, p1[i], p2[i]});
  ct.data = seal(rng, msg, plaintext);
  return ct;
}
static void plan_decrypt(PolicyMemo& memo, const CpAbeSecretKey& sk, const CpAbeCiphertext& ct, PairingJob* job) {       // :260-308
  std::vector<std::string> attr;
  for (const auto& v : sk.d_j) attr.push_back(v.string);
  const PolicyMemo::Entry& pe = memo.get(ct.policy.first, ct.policy.second);
  const PolicyNode& tree = pe.tree;
  if (!traverse_policy(attr, tree)) throw RabeError("Error in bsw/encrypt: attributes do not match policy.");
  PrunedList pruned;
  if (!calc_pruned(attr, tree, &pruned)) throw RabeError("Error in bsw/encrypt: attributes do not match policy.");
  const NamedFr& z = pe.coeff;
  // msg = c_p * A / e(c, d),  A = prod ( e(Cy.g1, Dj.g2) / e(Dj.g1, Cy.g2) )^z
  //     = c_p * FE( ML(-c, d) * prod ML(z Cy.g1, Dj.g2) ML(-z Dj.g1, Cy.g2) )        
  std::vector<G1>& base = job->base;
  std::vector<Fr>& scal = job->scal;
  std::vector<G2>& q = job->q;
  job->lead = ct.c_p;
  base.push_back(ct.c); scal.push_back(fr_neg(fr_one())); q.push_back(sk.d);
  for (const auto& pr : pruned) {
    const CpAbeAttribute* cy = nullptr;
    const CpAbeAttribute* dj = nullptr;
    for (const auto& x : ct.c_y) if (x.string == pr.second) { cy = &x; break; }
    if (!cy) continue;
    for (const auto& x : sk.d_j) if (x.string == pr.first) { dj = &x; break; }
    if (!dj) continue;
    for (const auto& zt : z) {
      if (zt.first == pr.second) {
        base.push_back(cy->g1); scal.push_back(zt.second); q.push_back(dj->g2);
        base.push_back(dj->g1); scal.push_back(fr_neg(zt.second)); q.push_back(cy->g2);
      }
    }
  }
}
Gt decrypt_gt(Engine& eng, const CpAbeSecretKey& sk, const CpAbeCiphertext& ct) {
  std::vector<PairingJob> jobs(1);
  PolicyMemo memo;
  plan_decrypt(memo, sk, ct, &jobs[0]);
  return run_pairing_jobs(eng, jobs)[0];
}
Bytes decrypt(Engine& eng, const CpAbeSecretKey& sk, const CpAbeCiphertext& ct) { return open_or_error(decrypt_gt(eng, sk, ct), ct.data); }
std::vector<DecryptResult> decrypt_batch(Engine& eng, const std::vector<const CpAbeSecretKey*>& sks, const std::vector<const CpAbeCiphertext*>& cts) {
  if (sks.size() != cts.size()) throw RabeError("decrypt_batch: sks and cts differ in length");
  PolicyMemo memo;
  std::vector<PairingJob> jobs = plan_jobs(cts.size(), [&](size_t i, PairingJob* j) { plan_decrypt(memo, *sks[i], *cts[i], j); });
  std::vector<const Bytes*> sealed;
  for (const auto* c : cts) sealed.push_back(&c->data);
  return open_jobs(eng, jobs, sealed);
}
std::vector<CpAbeCiphertext> encrypt_batch(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const std::vector<std::string>& policies,
                                           PolicyLanguage language, const std::vector<Bytes>& plaintexts) {
  if (policies.size() != plaintexts.size()) throw RabeError("encrypt_batch: policies and plaintexts differ in length");
  const size_t n = policies.size();
  std::vector<CpAbeCiphertext> cts(n);
  // plan: every draw of item i before any draw of item i+1, in encrypt's order (secret, msg, gate coefficients, nonce)
  struct Item { NamedFr shares; std::array<uint8_t, 12> nonce; size_t o1, o2; Fr secret, msg_k; const PolicyNode* tree; std::vector<Fr> draws;
                std::vector<Fr> k2; };
  std::vector<Item> items(n);
  PolicyMemo memo;
  StageTimer tm("bsw::encrypt_batch");
  for (size_t i = 0; i < n; i++) {              // sequential: the generator is consumed in encrypt's order
    Item& it = items[i];
    it.secret = rng.next_fr();
    it.msg_k = rng.next_fr();
    it.tree = &memo.get(policies[i], language).tree;
    const size_t nd = count_share_draws(*it.tree);
    for (size_t d = 0; d < nd; d++) it.draws.push_back(rng.next_fr());
    rng.fill(it.nonce.data(), 12);
    cts[i].policy = {policies[i], language};
  }
  tm.lap("draws (sequential)");
  parallel_for(n, [&](size_t i) {               // parallel: shares and label hashes
    Item& it = items[i];
    VecFrSource src(it.draws.data(), it.draws.size());
    gen_shares_policy(it.secret, *it.tree, src, &it.shares);
    for (const auto& sh : it.shares) it.k2.push_back(fr_mul(sha3_hash_fr(remove_index(sh.first)), sh.second));
  });
  tm.lap("shares (parallel)");
  std::vector<G1> b1; std::vector<Fr> k1;
  std::vector<G2> b2; std::vector<Fr> k2;
  std::vector<Gt> bt; std::vector<Fr> kt;
  for (size_t i = 0; i < n; i++) {
    Item& it = items[i];
    it.o1 = b1.size();
    it.o2 = b2.size();
    b1.push_back(pk.h); k1.push_back(it.secret);
    bt.push_back(pk.e_gg_alpha); kt.push_back(it.secret);
    bt.push_back(eng.gt_generator()); kt.push_back(it.msg_k);
    for (size_t y = 0; y < it.shares.size(); y++) {
      b1.push_back(pk.g1); k1.push_back(it.shares[y].second);
      b2.push_back(pk.g2); k2.push_back(it.k2[y]);
    }
  }
  if (!n) return cts;
  tm.lap("concat");
  std::vector<G1> r1 = eng.g1_mul(b1, k1);
  tm.lap("g1_mul");
  std::vector<G2> r2 = b2.empty() ? std::vector<G2>() : eng.g2_mul(b2, k2);
  tm.lap("g2_mul");
  std::vector<Gt> rt = eng.gt_pow(bt, kt);
  tm.lap("gt_pow");
  std::vector<Gt> ea, msg;
  for (size_t i = 0; i < n; i++) { ea.push_back(rt[2 * i]); msg.push_back(rt[2 * i + 1]); }
  std::vector<Gt> cp = eng.gt_mul(ea, msg);
  tm.lap("gt_mul");
  for (size_t i = 0; i < n; i++) {
    cts[i].c = r1[items[i].o1];
    cts[i].c_p = cp[i];
    for (size_t y = 0; y < items[i].shares.size(); y++)
      cts[i].c_y.push_back({items[i].shares[y].first, r1[items[i].o1 + 1 + y], r2[items[i].o2 + y]});
    cts[i].data = encrypt_symmetric(msg[i].data(), plaintexts[i].data(), plaintexts[i].size(), items[i].nonce.data());
  }
  tm.lap("assemble");
  return cts;
}
}  // namespace bsw

// ================================================================================================ LSW
namespace lsw {
std::pair<KpAbePublicKey, KpAbeMasterKey> setup(Engine& eng, Rng& rng) {         // :86-110
  Fr alpha1 = rng.next_fr(), alpha2 = rng.next_fr(), b = rng.next_fr();
  Fr alpha = fr_mul(alpha1, alpha2);
  G1 g1 = eng.random_g1(rng);
  G2 g2 = eng.random_g2(rng);
  G1 h_g1 = eng.random_g1(rng);
  G2 h_g2 = eng.random_g2(rng);
  std::vector<G1> m = eng.g1_mul({g1, g1, h_g1}, {b, fr_mul(b, b), b});
  Gt e = eng.gt_pow({eng.pairing({g1}, {g2})[0]}, {alpha})[0];
  return {KpAbePublicKey{g1, g2, m[0], m[1], m[2], e}, KpAbeMasterKey{alpha1, alpha2, b, h_g1, h_g2}};
}
static G1 g1_zero() { G1 z{}; return z; }
static G2 g2_zero() { G2 z{}; return z; }
std::vector<KpAbeSecretKey> keygen_batch(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const KpAbeMasterKey& msk,
                                         const std::vector<std::string>& policies, PolicyLanguage language) {        // :121-170, n times
  const size_t n = policies.size();
  std::vector<KpAbeSecretKey> sks(n);
  // all items' scalar multiplications in one launch per group.  Draw order per item: the gate coefficients of
  // gen_shares_policy, then one `random` per share in share order (drawn inside the loop at :136); every draw of
  // item i before item i+1 (sequential), the arithmetic on all cores.
  struct Slot { size_t item, row; bool neg; size_t i1, i2; };
  struct Item { const PolicyNode* tree; std::vector<Fr> draws; std::vector<G1> b1; std::vector<Fr> k1; std::vector<G2> b2; std::vector<Fr> k2;
                std::vector<Slot> slots; };
  std::vector<Item> items(n);
  PolicyMemo memo;
  for (size_t it = 0; it < n; it++) {
    items[it].tree = &memo.get(policies[it], language).tree;
    const size_t nd = count_share_draws(*items[it].tree) + count_leaves(*items[it].tree);
    for (size_t d = 0; d < nd; d++) items[it].draws.push_back(rng.next_fr());
    sks[it].policy = {policies[it], language};
  }
  parallel_for(n, [&](size_t it) {
    Item& im = items[it];
    VecFrSource src(im.draws.data(), im.draws.size());
    NamedFr shares;
    gen_shares_policy(msk.alpha1, *im.tree, src, &shares);
    KpAbeSecretKey& sk = sks[it];
    std::vector<G1>& b1 = im.b1;
    std::vector<Fr>& k1 = im.k1;
    std::vector<G2>& b2 = im.b2;
    std::vector<Fr>& k2 = im.k2;
    for (const auto& sh : shares) {
      std::string striped = remove_index(sh.first);
      Fr random = src.next_fr();
      Slot s{it, sk.dj.size(), is_negative(striped), b1.size(), b2.size()};
      if (s.neg) {
        Fr share_hash = sha3_hash_fr(striped);
        // d3 = g1*share + g1_b2*random ; d4 = g1_b*(hash*random) + h_g1*random ; d5 = g1*(-random)
        b1.push_back(pk.g1); k1.push_back(sh.second);
        b1.push_back(pk.g1_b2); k1.push_back(random);
        b1.push_back(pk.g1_b); k1.push_back(fr_mul(share_hash, random));
        b1.push_back(msk.h_g1); k1.push_back(random);
        b1.push_back(pk.g1); k1.push_back(fr_neg(random));
      } else {
        // d1 = g1*(alpha2*share) + (g1*h(y))*random = g1*(alpha2*share + h(y)*random) ; d2 = g2*random
        b1.push_back(pk.g1); k1.push_back(fr_add(fr_mul(msk.alpha2, sh.second), fr_mul(sha3_hash_fr(striped), random)));
        b2.push_back(pk.g2); k2.push_back(random);
      }
      im.slots.push_back(s);
      sk.dj.push_back({striped, g1_zero(), g2_zero(), g1_zero(), g1_zero(), g1_zero()});
    }
  });
  std::vector<G1> b1;
  std::vector<Fr> k1;
  std::vector<G2> b2;
  std::vector<Fr> k2;
  std::vector<Slot> slots;
  for (size_t it = 0; it < n; it++) {
    Item& im = items[it];
    for (Slot s : im.slots) { s.i1 += b1.size(); s.i2 += b2.size(); slots.push_back(s); }
    b1.insert(b1.end(), im.b1.begin(), im.b1.end());
    k1.insert(k1.end(), im.k1.begin(), im.k1.end());
    b2.insert(b2.end(), im.b2.begin(), im.b2.end());
    k2.insert(k2.end(), im.k2.begin(), im.k2.end());
  }
  std::vector<G1> r1 = b1.empty() ? std::vector<G1>() : eng.g1_mul(b1, k1);
  std::vector<G2> r2 = b2.empty() ? std::vector<G2>() : eng.g2_mul(b2, k2);
  std::vector<G1> sa, sb;
  for (const auto& sl : slots)
    if (sl.neg) { sa.push_back(r1[sl.i1]); sb.push_back(r1[sl.i1 + 1]); sa.push_back(r1[sl.i1 + 2]); sb.push_back(r1[sl.i1 + 3]); }
  std::vector<G1> sums = sa.empty() ? std::vector<G1>() : g1_add(eng, sa, sb);
  size_t ns = 0;
  for (const auto& sl : slots) {
    KpAbeKeyRow& row = sks[sl.item].dj[sl.row];
    if (sl.neg) {
      row.d3 = sums[ns];
      row.d4 = sums[ns + 1];
      row.d5 = r1[sl.i1 + 4];
      ns += 2;
    } else {
      row.d1 = r1[sl.i1];
      row.d2 = r2[sl.i2];
    }
  }
  return sks;
}
KpAbeSecretKey keygen(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const KpAbeMasterKey& msk, const std::string& policy,
                      PolicyLanguage language) {        // :121-170
  return keygen_batch(eng, rng, pk, msk, {policy}, language)[0];
}
KpAbeCiphertext encrypt(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const std::vector<std::string>& attributes, const Bytes& plaintext) {   // :180-219
  if (attributes.empty() || plaintext.empty()) throw RabeError("attributes or data empty");
  Fr secret = rng.next_fr();
  std::vector<Fr> sx{secret};
  for (size_t i = 0; i < attributes.size(); i++) {
    sx.push_back(rng.next_fr());
    sx[0] = fr_sub(sx[0], sx[i]);       // the reference's index quirk (:197-200): subtracts sx[i], not the new element
  }
  std::vector<G1> base;
  std::vector<Fr> k;
  for (size_t i = 0; i < attributes.size(); i++) {
    Fr h = sha3_hash_fr(attributes[i]);
    base.push_back(pk.g1); k.push_back(fr_mul(h, secret));                 // (g1*h(a))*secret
    base.push_back(pk.g1_b); k.push_back(sx[i]);
    base.push_back(pk.g1_b2); k.push_back(fr_mul(sx[i], h));
    base.push_back(pk.h_b); k.push_back(sx[i]);
  }
  std::vector<G1> r = eng.g1_mul(base, k);
  Gt msg = eng.random_gt(rng);
  KpAbeCiphertext ct;
  ct.e1 = eng.gt_mul({eng.gt_pow({pk.e_gg_alpha}, {secret})[0]}, {msg})[0];
  ct.e2 = eng.g2_mul({pk.g2}, {secret})[0];
  for (size_t i = 0; i < attributes.size(); i++) {
    G1 e3 = g1_add(eng, {r[4 * i + 2]}, {r[4 * i + 3]})[0];
    ct.ej.push_back({attributes[i], r[4 * i], r[4 * i + 1], e3});
  }
  ct.ct = seal(rng, msg, plaintext);
  return ct;
}
static void plan_decrypt(PolicyMemo& memo, const KpAbeSecretKey& sk, const KpAbeCiphertext& ct, PairingJob* job) {      // :228-290
  std::vector<std::string> attr;
  for (const auto& a : ct.ej) attr.push_back(a.name);
  const PolicyMemo::Entry& pe = memo.get(sk.policy.first, sk.policy.second);
  const PolicyNode& tree = pe.tree;
  PrunedList list;
  if (!calc_pruned(attr, tree, &list)) throw RabeError("Error in lsw/decrypt: attributes do not match policy.");
  const NamedFr& coeff_list = pe.coeff;
  // prod_t = prod z_y^coeff with z_y = e(D1, e2) / e(E1, D2) for positive leaves; a negative leaf re-uses the
  // previous z_y (the reference's TODO branch, :265-278).  msg = e1 / prod_t
  //   = e1 * FE( prod ML(-c*D1, e2) * ML(c*E1, D2) ).
  std::vector<G1>& base = job->base;
  std::vector<Fr>& scal = job->scal;
  std::vector<G2>& q = job->q;
  job->lead = ct.e1;
  const KpAbeKeyRow* cur_sk = nullptr;
  const KpAbeCtRow* cur_ct = nullptr;
  for (const auto& a : list) {
    const KpAbeKeyRow* sk_attr = nullptr;
    const KpAbeCtRow* ct_attr = nullptr;
    const Fr* coeff = nullptr;
    for (const auto& x : sk.dj) if (x.name == a.first) { sk_attr = &x; break; }
    for (const auto& x : ct.ej) if (x.name == a.first) { ct_attr = &x; break; }
    for (const auto& x : coeff_list) if (x.first == a.second) { coeff = &x.second; break; }
    if (!sk_attr || !ct_attr || !coeff) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
    if (!is_negative(a.first)) { cur_sk = sk_attr; cur_ct = ct_attr; }
    if (!cur_sk) continue;            // z_y still Gt::one()
    // every e(-c D1, e2) shares its G2 argument: prod_x e(-c_x D1_x, e2) = e(sum_x -c_x D1_x, e2) -- one Miller loop for all
    job->sbase.push_back(cur_sk->d1); job->sscal.push_back(fr_neg(*coeff)); job->sq = ct.e2;
    base.push_back(cur_ct->e1); scal.push_back(*coeff); q.push_back(cur_sk->d2);
  }
}
Gt decrypt_gt(Engine& eng, const KpAbeSecretKey& sk, const KpAbeCiphertext& ct) {
  std::vector<PairingJob> jobs(1);
  PolicyMemo memo;
  plan_decrypt(memo, sk, ct, &jobs[0]);
  if (jobs[0].base.empty() && jobs[0].sbase.empty()) return ct.e1;
  return run_pairing_jobs(eng, jobs)[0];
}
Bytes decrypt(Engine& eng, const KpAbeSecretKey& sk, const KpAbeCiphertext& ct) { return open_or_error(decrypt_gt(eng, sk, ct), ct.ct); }
std::vector<DecryptResult> decrypt_batch(Engine& eng, const std::vector<const KpAbeSecretKey*>& sks, const std::vector<const KpAbeCiphertext*>& cts) {
  if (sks.size() != cts.size()) throw RabeError("decrypt_batch: sks and cts differ in length");
  PolicyMemo memo;
  std::vector<PairingJob> jobs = plan_jobs(cts.size(), [&](size_t i, PairingJob* j) { plan_decrypt(memo, *sks[i], *cts[i], j); });
  std::vector<const Bytes*> sealed;
  for (const auto* c : cts) sealed.push_back(&c->ct);
  return open_jobs(eng, jobs, sealed);
}
}  // namespace lsw

// ================================================================================================ AW11
namespace aw11 {
static std::string upper(const std::string& s) {
  std::string o = s;
  for (auto& c : o) if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');      // ASCII part of str::to_uppercase
  return o;
}
Aw11GlobalKey setup(Engine& eng, Rng& rng) { G1 a = eng.random_g1(rng); G2 b = eng.random_g2(rng); return Aw11GlobalKey{a, b}; }   // :100-108
bool authgen(Engine& eng, Rng& rng, const Aw11GlobalKey& gk, const std::vector<std::string>& attributes, Aw11PublicKey* pk, Aw11MasterKey* msk) {   // :121-151
  if (attributes.empty()) return false;
  pk->attr.clear();
  msk->attr.clear();
  std::vector<Fr> alphas, ys;
  for (const auto& a : attributes) {
    Fr alpha = rng.next_fr(), y = rng.next_fr();
    msk->attr.push_back({upper(a), alpha, y});
    alphas.push_back(alpha);
    ys.push_back(y);
  }
  Gt egg = eng.pairing({gk.g1}, {gk.g2})[0];         // the reference recomputes this constant per attribute (:144)
  std::vector<Gt> e = eng.gt_pow(std::vector<Gt>(attributes.size(), egg), alphas);
  std::vector<G2> g = eng.g2_mul(std::vector<G2>(attributes.size(), gk.g2), ys);
  for (size_t i = 0; i < attributes.size(); i++) pk->attr.push_back({upper(attributes[i]), e[i], g[i]});
  return true;
}
void add_to_attribute(Engine& eng, const Aw11GlobalKey& gk, const Aw11MasterKey& msk, const std::string& attribute, Aw11SecretKey* sk) {   // :200-231
  if (attribute.empty()) throw RabeError("empty _attributes");
  if (sk->gid.empty()) throw RabeError("empty _gid");
  const Aw11MkAttr* auth = nullptr;
  for (const auto& a : msk.attr) if (a.name == attribute) { auth = &a; break; }
  if (!auth) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
  // g1*alpha + (g1*h(gid))*y = g1*(alpha + h(gid) y)
  Fr k = fr_add(auth->alpha, fr_mul(sha3_hash_fr(sk->gid), auth->y));
  sk->attr.push_back({upper(auth->name), eng.g1_mul({gk.g1}, {k})[0]});
}
Aw11SecretKey keygen(Engine& eng, const Aw11GlobalKey& gk, const Aw11MasterKey& msk, const std::string& name,
                     const std::vector<std::string>& attributes) {       // :165-190
  if (attributes.empty()) throw RabeError("empty _attributes");
  if (name.empty()) throw RabeError("empty _name");
  Aw11SecretKey sk{name, {}};
  for (const auto& a : attributes) add_to_attribute(eng, gk, msk, a, &sk);
  return sk;
}
std::vector<Aw11Ciphertext> encrypt_batch(Engine& eng, Rng& rng, const Aw11GlobalKey& gk, const std::vector<const Aw11PublicKey*>& pks,
                                          const std::vector<std::string>& policies, PolicyLanguage language, const std::vector<Bytes>& datas) {   // :241-289, n times
  if (policies.size() != datas.size()) throw RabeError("encrypt_batch: policies and datas differ in length");
  const size_t n = policies.size();
  std::vector<Aw11Ciphertext> cts(n);
  if (!n) return cts;
  StageTimer tm("aw11::encrypt_batch");
  Gt egg = eng.pairing({gk.g1}, {gk.g2})[0];          // the reference recomputes this constant per call (:264)
  tm.lap("pairing");
  struct Item { std::vector<std::string> names; std::array<uint8_t, 12> nonce; size_t ot, o2; const PolicyNode* tree; Fr s, msg_k;
                std::vector<Fr> draws; std::vector<Gt> gb; std::vector<Fr> gk; std::vector<G2> b2; std::vector<Fr> k2; };
  std::vector<Item> items(n);
  PolicyMemo memo;
  // sequential: the generator in encrypt's order -- s, gate coefficients of the s-shares, of the 0-shares, msg, one r_x per
  // share, nonce
  for (size_t it = 0; it < n; it++) {
    Item& im = items[it];
    PolicyMemo::Entry& pe = const_cast<PolicyMemo::Entry&>(memo.get(policies[it], language));
    im.tree = &pe.tree;
    if (!pe.msp_checked) { (void)calculate_msp(pe.tree); pe.msp_checked = true; }   // built and unused in the reference (:253-255) -- but it must not panic
    im.s = rng.next_fr();
    const size_t nd = count_share_draws(*im.tree), nl = count_leaves(*im.tree);
    for (size_t d = 0; d < 2 * nd; d++) im.draws.push_back(rng.next_fr());
    im.msg_k = rng.next_fr();
    for (size_t d = 0; d < nl; d++) im.draws.push_back(rng.next_fr());
    rng.fill(im.nonce.data(), 12);
    cts[it].policy = {policies[it], language};
  }
  tm.lap("draws (sequential)");
  const Gt e_gen = eng.gt_generator();
  parallel_for(n, [&](size_t it) {
    Item& im = items[it];
    VecFrSource src(im.draws.data(), im.draws.size());
    NamedFr s_shares, w_shares;
    gen_shares_policy(im.s, *im.tree, src, &s_shares);
    gen_shares_policy(fr_zero(), *im.tree, src, &w_shares);
    im.gb.push_back(e_gen); im.gk.push_back(im.msg_k);                    // msg
    im.gb.push_back(egg); im.gk.push_back(im.s);                          // egg^s
    for (size_t i = 0; i < s_shares.size(); i++) {
      Fr r_x = src.next_fr();
      std::string up = upper(s_shares[i].first);
      const Aw11PkAttr* pa = nullptr;
      std::string want = remove_index(up);
      for (const auto* pk : pks) {
        for (const auto& t : pk->attr) if (t.name == want) { pa = &t; break; }
        if (pa) break;
      }
      if (!pa) continue;
      im.names.push_back(up);
      im.gb.push_back(egg); im.gk.push_back(s_shares[i].second);
      im.gb.push_back(pa->egg_alpha); im.gk.push_back(r_x);
      im.b2.push_back(gk.g2); im.k2.push_back(r_x);
      im.b2.push_back(pa->g2_y); im.k2.push_back(r_x);
      im.b2.push_back(gk.g2); im.k2.push_back(w_shares[i].second);
    }
  });
  std::vector<Gt> gb;
  std::vector<Fr> gk_;
  std::vector<G2> b2;
  std::vector<Fr> k2;
  for (size_t it = 0; it < n; it++) {
    Item& im = items[it];
    im.ot = gb.size();
    im.o2 = b2.size();
    gb.insert(gb.end(), im.gb.begin(), im.gb.end());
    gk_.insert(gk_.end(), im.gk.begin(), im.gk.end());
    b2.insert(b2.end(), im.b2.begin(), im.b2.end());
    k2.insert(k2.end(), im.k2.begin(), im.k2.end());
  }
  tm.lap("shares + concat");
  std::vector<Gt> ge = eng.gt_pow(gb, gk_);
  tm.lap("gt_pow");
  std::vector<G2> g2r = b2.empty() ? std::vector<G2>() : eng.g2_mul(b2, k2);
  tm.lap("g2_mul");
  // c_0 = msg * egg^s and every row's c1 = egg^share * egg_alpha^r_x in one gt_mul; every row's c3 in one g2_add
  std::vector<Gt> ma, mb;
  std::vector<G2> aa, ab;
  for (size_t it = 0; it < n; it++) {
    const size_t ot = items[it].ot, o2 = items[it].o2;
    ma.push_back(ge[ot]); mb.push_back(ge[ot + 1]);
    for (size_t i = 0; i < items[it].names.size(); i++) {
      ma.push_back(ge[ot + 2 + 2 * i]); mb.push_back(ge[ot + 3 + 2 * i]);
      aa.push_back(g2r[o2 + 3 * i + 1]); ab.push_back(g2r[o2 + 3 * i + 2]);
    }
  }
  tm.lap("gather");
  std::vector<Gt> mm = eng.gt_mul(ma, mb);
  tm.lap("gt_mul");
  std::vector<G2> c3 = aa.empty() ? std::vector<G2>() : g2_add(eng, aa, ab);
  tm.lap("g2_add");
  size_t pm = 0, p3 = 0;
  for (size_t it = 0; it < n; it++) {
    cts[it].c_0 = mm[pm++];
    for (size_t i = 0; i < items[it].names.size(); i++)
      cts[it].c.push_back({items[it].names[i], mm[pm++], g2r[items[it].o2 + 3 * i], c3[p3++]});
    cts[it].ct = encrypt_symmetric(ge[items[it].ot].data(), datas[it].data(), datas[it].size(), items[it].nonce.data());
  }
  tm.lap("assemble");
  return cts;
}
Aw11Ciphertext encrypt(Engine& eng, Rng& rng, const Aw11GlobalKey& gk, const std::vector<const Aw11PublicKey*>& pks, const std::string& policy,
                       PolicyLanguage language, const Bytes& data) {       // :241-289
  return encrypt_batch(eng, rng, gk, pks, {policy}, language, {data})[0];
}
static void plan_decrypt(PolicyMemo& memo, const G1& hash, const Aw11SecretKey& sk, const Aw11Ciphertext& ct, PairingJob* job) {       // :298-366
  std::vector<std::string> str_attr;
  for (const auto& v : sk.attr) str_attr.push_back(v.first);
  const PolicyMemo::Entry& pe = memo.get(ct.policy.first, ct.policy.second);
  const PolicyNode& tree = pe.tree;
  if (!traverse_policy(str_attr, tree)) throw RabeError("Error: attributes in sk do not match policy in ct.");
  PrunedList list;
  bool ok = calc_pruned(str_attr, tree, &list);
  const NamedFr& coeff_list = pe.coeff;
  if (!ok) throw RabeError("Error in aw11/decrypt: attributes in sk do not match policy in ct.");
  // egg_s = prod ( C1 * e(H, C3) / e(K, C2) )^c ; msg = c_0 / egg_s
  //       = c_0 * prod C1^(-c) * FE( prod ML(-c H, C3) ML(c K, C2) ),  H = g1*h(gid)        
  std::vector<G1>& base = job->base;
  std::vector<Fr>& scal = job->scal;
  std::vector<G2>& q = job->q;
  std::vector<Gt>& c1s = job->gbase;
  std::vector<Fr>& c1k = job->gexp;
  job->lead = ct.c_0;
  for (const auto& cur : list) {
    const std::pair<std::string, G1>* sk_attr = nullptr;
    const Aw11CtRow* ct_attr = nullptr;
    const Fr* coeff = nullptr;
    for (const auto& x : sk.attr) if (x.first == cur.first) { sk_attr = &x; break; }
    for (const auto& x : ct.c) if (x.name == cur.second) { ct_attr = &x; break; }
    for (const auto& x : coeff_list) if (x.first == cur.second) { coeff = &x.second; break; }
    if (!sk_attr || !ct_attr || !coeff) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
    base.push_back(hash); scal.push_back(fr_neg(*coeff)); q.push_back(ct_attr->c3);
    base.push_back(sk_attr->second); scal.push_back(*coeff); q.push_back(ct_attr->c2);
    c1s.push_back(ct_attr->c1); c1k.push_back(fr_neg(*coeff));
  }
}
Gt decrypt_gt(Engine& eng, const Aw11GlobalKey& gk, const Aw11SecretKey& sk, const Aw11Ciphertext& ct) {
  G1 hash = eng.g1_mul({gk.g1}, {sha3_hash_fr(sk.gid)})[0];
  std::vector<PairingJob> jobs(1);
  PolicyMemo memo;
  plan_decrypt(memo, hash, sk, ct, &jobs[0]);
  if (jobs[0].base.empty()) return ct.c_0;
  return run_pairing_jobs(eng, jobs)[0];
}
Bytes decrypt(Engine& eng, const Aw11GlobalKey& gk, const Aw11SecretKey& sk, const Aw11Ciphertext& ct) {
  return open_or_error(decrypt_gt(eng, gk, sk, ct), ct.ct);
}
std::vector<DecryptResult> decrypt_batch(Engine& eng, const Aw11GlobalKey& gk, const std::vector<const Aw11SecretKey*>& sks,
                                         const std::vector<const Aw11Ciphertext*>& cts) {
  if (sks.size() != cts.size()) throw RabeError("decrypt_batch: sks and cts differ in length");
  // H(gid) = g1 * h(gid) for every key, one launch
  std::vector<Fr> hk;
  for (const auto* sk : sks) hk.push_back(sha3_hash_fr(sk->gid));
  std::vector<G1> hashes = sks.empty() ? std::vector<G1>() : eng.g1_mul(std::vector<G1>(sks.size(), gk.g1), hk);
  PolicyMemo memo;
  std::vector<PairingJob> jobs = plan_jobs(cts.size(), [&](size_t i, PairingJob* j) { plan_decrypt(memo, hashes[i], *sks[i], *cts[i], j); });
  std::vector<const Bytes*> sealed;
  for (const auto* c : cts) sealed.push_back(&c->ct);
  return open_jobs(eng, jobs, sealed);
}
}  // namespace aw11

// ================================================================================================ GHW11
namespace ghw11 {
std::pair<Ghw11PublicKey, Ghw11MasterKey> setup(Engine& eng, Rng& rng) {        // :92-111
  G1 g1 = eng.random_g1(rng);
  G2 g2 = eng.random_g2(rng);
  Fr a = rng.next_fr();
  G1 g1_a = eng.g1_mul({g1}, {a})[0];
  Fr alpha = rng.next_fr();
  std::vector<G2> m = eng.g2_mul({g2, g2}, {a, alpha});
  Gt e = eng.gt_pow({eng.pairing({g1}, {g2})[0]}, {alpha})[0];
  Ghw11PublicKey pk{g1, g2, g1_a, m[0], e};
  return {pk, Ghw11MasterKey{m[1], pk}};
}
bool keygen(Engine& eng, Rng& rng, const Ghw11PublicKey& pk, const Ghw11MasterKey& msk, const std::vector<std::string>& attributes,
            Ghw11SecretKey* out) {       // :123-152
  if (attributes.empty()) return false;
  Fr r = rng.next_fr();
  // L = g2*r ; K = g2_alpha + g2_a*r ; K_x = (g2*h(x))*r = g2*(h(x) r)
  std::vector<G2> base{pk.g2, pk.g2_a};
  std::vector<Fr> k{r, r};
  for (const auto& j : attributes) { base.push_back(pk.g2); k.push_back(fr_mul(sha3_hash_fr(j), r)); }
  std::vector<G2> m = eng.g2_mul(base, k);
  out->l = m[0];
  out->k = g2_add(eng, {msk.g2_alpha}, {m[1]})[0];
  out->attr_key.clear();
  for (size_t i = 0; i < attributes.size(); i++) out->attr_key.push_back({attributes[i], m[2 + i]});
  return true;
}
std::pair<Ghw11TransformKey, Ghw11RetrieveKey> tkgen(Engine& eng, Rng& rng, const Ghw11SecretKey& sk) {     // :156-180
  Fr z = rng.next_fr();
  Fr z_inv = must_inv(z);
  std::vector<G2> base{sk.k, sk.l};
  for (const auto& a : sk.attr_key) base.push_back(a.k_x);
  std::vector<G2> m = eng.g2_mul(base, std::vector<Fr>(base.size(), z_inv));
  Ghw11TransformKey tk{m[0], m[1], {}};
  for (size_t i = 0; i < sk.attr_key.size(); i++) tk.attr_key_z.push_back({sk.attr_key[i].string, m[2 + i]});
  return {tk, Ghw11RetrieveKey{z}};
}
Ghw11Ciphertext encrypt(Engine& eng, Rng& rng, const Ghw11PublicKey& pk, const std::string& policy, PolicyLanguage language,
                        const Bytes& plaintext) {       // :192-228
  Fr secret = rng.next_fr();
  Fr msg_k = rng.next_fr();                       // rng.gen::<Gt>()
  PolicyNode tree = parse_or_error(policy, language);
  NamedFr shares;
  gen_shares_policy(secret, tree, rng, &shares);
  // C_i = g1_a*share - (g1*h(j))*t_i = g1_a*share + g1*(-h(j) t_i) ; D_i = g1*t_i
  std::vector<G1> base{pk.g1};
  std::vector<Fr> k{secret};
  for (const auto& sh : shares) {
    Fr t_i = rng.next_fr();
    base.push_back(pk.g1_a); k.push_back(sh.second);
    base.push_back(pk.g1); k.push_back(fr_neg(fr_mul(sha3_hash_fr(remove_index(sh.first)), t_i)));
    base.push_back(pk.g1); k.push_back(t_i);
  }
  std::vector<G1> m = eng.g1_mul(base, k);
  std::vector<Gt> pw = eng.gt_pow({pk.e_gg_alpha, eng.gt_generator()}, {secret, msg_k});
  Ghw11Ciphertext ct;
  ct.policy = {policy, language};
  ct.c = eng.gt_mul({pw[0]}, {pw[1]})[0];
  ct.c1 = m[0];
  std::vector<G1> ca, cb;
  for (size_t i = 0; i < shares.size(); i++) { ca.push_back(m[1 + 3 * i]); cb.push_back(m[2 + 3 * i]); }
  std::vector<G1> ci = shares.empty() ? std::vector<G1>() : g1_add(eng, ca, cb);
  for (size_t i = 0; i < shares.size(); i++) ct.ci_di.push_back({shares[i].first, ci[i], m[3 + 3 * i]});
  ct.data = seal(rng, pw[1], plaintext);
  return ct;
}
static void plan_transform(PolicyMemo& memo, const Ghw11Ciphertext& ct, const Ghw11TransformKey& tk, PairingJob* job) {      // :231-295
  std::vector<std::string> attr;
  for (const auto& v : tk.attr_key_z) attr.push_back(v.string);
  const PolicyMemo::Entry& pe = memo.get(ct.policy.first, ct.policy.second);
  if (!traverse_policy(attr, pe.tree)) throw RabeError("Error: attributes in tk do not match policy in ct.");
  PrunedList list;
  bool ok = calc_pruned(attr, pe.tree, &list);
  if (!ok) throw RabeError("Error in Ghw11/decrypt: attributes in sk do not match policy in ct.");
  // t = e(c1, k_z) / ( prod_i e(w_i D_i, K_i) * e(sum_i w_i C_i, l_z) )
  //   = FE( ML(c1, k_z) * prod_i ML(-w_i D_i, K_i) * ML(sum_i (-w_i) C_i, l_z) )
  job->lead_one = true;
  job->base.push_back(ct.c1); job->scal.push_back(fr_one()); job->q.push_back(tk.k_z);
  for (const auto& cur : list) {
    const Fr* coeff = nullptr;
    const Ghw11Attribute* tk_attr = nullptr;
    const Ghw11CtRow* ct_attr = nullptr;
    for (const auto& x : pe.coeff) if (x.first == cur.second) { coeff = &x.second; break; }
    for (const auto& x : tk.attr_key_z) if (x.string == cur.first) { tk_attr = &x; break; }
    for (const auto& x : ct.ci_di) if (x.name == cur.second) { ct_attr = &x; break; }
    if (!coeff || !tk_attr || !ct_attr) throw std::runtime_error("called `Option::unwrap()` on a `None` value");
    const Fr nw = fr_neg(*coeff);
    job->base.push_back(ct_attr->d); job->scal.push_back(nw); job->q.push_back(tk_attr->k_x);
    job->sbase.push_back(ct_attr->c); job->sscal.push_back(nw);
  }
  job->sq = tk.l_z;
}
std::vector<Ghw11TransformCiphertext> transform_batch(Engine& eng, const std::vector<const Ghw11Ciphertext*>& cts,
                                                      const std::vector<const Ghw11TransformKey*>& tks, std::vector<std::string>* errors) {
  if (cts.size() != tks.size()) throw RabeError("transform_batch: cts and tks differ in length");
  PolicyMemo memo;
  std::vector<PairingJob> jobs = plan_jobs(cts.size(), [&](size_t i, PairingJob* j) { plan_transform(memo, *cts[i], *tks[i], j); });
  std::vector<Gt> t = run_pairing_jobs(eng, jobs);
  std::vector<Ghw11TransformCiphertext> out(cts.size());
  if (errors) errors->assign(cts.size(), "");
  for (size_t i = 0; i < cts.size(); i++) {
    if (jobs[i].failed) { if (errors) (*errors)[i] = jobs[i].error; continue; }
    out[i] = Ghw11TransformCiphertext{cts[i]->c, t[i]};
  }
  return out;
}
Ghw11TransformCiphertext transform(Engine& eng, const Ghw11Ciphertext& ct, const Ghw11TransformKey& tk) {
  std::vector<std::string> errors;
  auto r = transform_batch(eng, {&ct}, {&tk}, &errors);
  if (!errors[0].empty()) throw RabeError(errors[0]);
  return r[0];
}
Gt decrypt_out_gt(Engine& eng, const Ghw11TransformCiphertext& pct, const Ghw11RetrieveKey& rk) {       // :298-305
  // msg = c * (t^z)^-1 = c * t^(-z)   (Gt has order r)
  Gt p = eng.gt_pow({pct.t}, {fr_neg(rk.z)})[0];
  return eng.gt_mul({pct.c}, {p})[0];
}
Bytes decrypt_out(Engine& eng, const Ghw11TransformCiphertext& pct, const Ghw11RetrieveKey& rk, const Bytes& data) {
  return open_or_error(decrypt_out_gt(eng, pct, rk), data);
}
}  // namespace ghw11

// ================================================================================================ BDABE / MKE08 (DNF policies)
// What the two schemes share (bdabe/mod.rs:401-475 = mke08/mod.rs:382-470): the authority test on "auth::attribute" names, the
// attribute exponent h(attribute) * h(authority) * secret, the conjunction sums of dnf.rs, and the shape of decrypt:
//   msg = lead * e(c_a, sum S2) * e(sum S1, c_b) / (e(c_c, u2) * e(u1, c_d))
// which is ONE pairing job: m pairs (c_a, S2_i), one pair with the summed G1 argument (sum S1_i, c_b), two pairs with exponent -1.
namespace dnfabe {
static bool from_authority(const std::string& attr, const std::string& authority) {
  size_t count = 0, first = std::string::npos;
  for (size_t pos = attr.find("::"); pos != std::string::npos; pos = attr.find("::", pos + 2)) { if (!count) first = pos; count++; }   // match_indices: non-overlapping
  return count == 1 && attr.substr(0, first) == authority;
}
static Fr attr_exponent(const std::string& attribute, const std::string& authority, const Fr& secret) {
  return fr_mul(fr_mul(sha3_hash_fr(attribute), sha3_hash_fr(authority)), secret);
}
struct AttrKey { const std::string* attr; const G1* g1; const G2* g2; };
static bool is_satisfiable(const std::vector<std::string>& conjunction, const std::vector<AttrKey>& sk) {
  for (const auto& a : conjunction) {
    bool found = false;
    for (const auto& k : sk) if (*k.attr == a) { found = true; break; }
    if (!found) return false;
  }
  return true;
}
// the first satisfiable conjunction of a ciphertext -> the pairing job of decrypt; `lead` is set by the caller
static void plan_term(const std::vector<std::string>& conjunction, const std::vector<AttrKey>& sk, const G1& c_a, const G2& c_b, const G1& c_c,
                      const G2& c_d, const G1& u1, const G2& u2, PairingJob* job) {
  const Fr one = fr_one(), minus_one = fr_neg(fr_one());
  for (const auto& a : conjunction) {
    for (const auto& k : sk) {
      if (*k.attr != a) continue;
      job->base.push_back(c_a); job->scal.push_back(one); job->q.push_back(*k.g2);      // e(c_a, sum S2) factor by factor
      job->sbase.push_back(*k.g1); job->sscal.push_back(one);                            // sum S1
      break;                                                                             // find(): the first match
    }
  }
  job->sq = c_b;
  job->base.push_back(c_c); job->scal.push_back(minus_one); job->q.push_back(u2);
  job->base.push_back(u1); job->scal.push_back(minus_one); job->q.push_back(c_d);
}
// sums / products of the public attribute keys of every DNF term, round by round (one batched launch per round and group)
template <class T, class ADD>
static std::vector<T> fold_terms(const std::vector<DnfTerm>& terms, const std::vector<T>& elems, ADD add) {
  std::vector<T> acc;
  size_t longest = 0;
  for (const auto& t : terms) { acc.push_back(elems[t.keys[0]]); longest = std::max(longest, t.keys.size()); }
  for (size_t r = 1; r < longest; r++) {
    std::vector<T> a, b;
    std::vector<size_t> who;
    for (size_t t = 0; t < terms.size(); t++) if (terms[t].keys.size() > r) { a.push_back(acc[t]); b.push_back(elems[terms[t].keys[r]]); who.push_back(t); }
    std::vector<T> sum = add(a, b);
    for (size_t i = 0; i < who.size(); i++) acc[who[i]] = sum[i];
  }
  return acc;
}
}  // namespace dnfabe

namespace bdabe {
using namespace dnfabe;
std::pair<BdabePublicKey, BdabeMasterKey> setup(Engine& eng, Rng& rng) {          // :149-163
  G1 g1 = eng.random_g1(rng);
  G2 g2 = eng.random_g2(rng);
  G1 p1 = eng.random_g1(rng);
  G2 p2 = eng.random_g2(rng);
  Fr y = rng.next_fr();
  Gt e = eng.gt_pow({eng.pairing({g1}, {g2})[0]}, {y})[0];
  return {BdabePublicKey{g1, g2, p1, p2, e}, BdabeMasterKey{y}};
}
BdabeSecretAuthorityKey authgen(Engine& eng, Rng& rng, const BdabePublicKey& pk, const BdabeMasterKey& msk, const std::string& name) {   // :174-189
  Fr alpha = rng.next_fr();
  Fr beta = fr_sub(msk.y, alpha);
  G1 a1 = eng.g1_mul({pk.g1}, {alpha})[0];
  G2 a2 = eng.g2_mul({pk.g2}, {beta})[0];
  Fr a3 = rng.next_fr();
  return BdabeSecretAuthorityKey{name, a1, a2, a3};
}
BdabeUserKey keygen(Engine& eng, Rng& rng, const BdabePublicKey& pk, const BdabeSecretAuthorityKey& ska, const std::string& name) {   // :202-224
  Fr r_u = rng.next_fr();
  std::vector<G1> a = eng.g1_mul({pk.p1, pk.g1}, {r_u, r_u});
  std::vector<G2> b = eng.g2_mul({pk.p2, pk.g2}, {r_u, r_u});
  BdabeUserKey k;
  k.sk = {g1_add(eng, {ska.a1}, {a[0]})[0], g2_add(eng, {ska.a2}, {b[0]})[0]};
  k.pk = {name, a[1], b[1]};
  return k;
}
BdabePublicAttributeKey request_attribute_pk(Engine& eng, const BdabePublicKey& pk, const BdabeSecretAuthorityKey& ska, const std::string& attribute) {   // :234-265
  if (!from_authority(attribute, ska.name)) throw RabeError("attribute " + attribute + " is not from_authority() or !is_eligible()");
  Fr exp = attr_exponent(attribute, ska.name, ska.a3);
  return BdabePublicAttributeKey{attribute, eng.g1_mul({pk.g1}, {exp})[0], eng.g2_mul({pk.g2}, {exp})[0], eng.gt_pow({pk.e_gg_y}, {exp})[0]};
}
BdabeSecretAttributeKey request_attribute_sk(Engine& eng, const BdabePublicUserKey& pk_u, const BdabeSecretAuthorityKey& ska, const std::string& attribute) {   // :275-305
  if (!from_authority(attribute, ska.name)) throw RabeError("attribute " + attribute + " is not from_authority() or !is_eligible()");
  Fr exp = attr_exponent(attribute, ska.name, ska.a3);
  return BdabeSecretAttributeKey{attribute, eng.g1_mul({pk_u.u1}, {exp})[0], eng.g2_mul({pk_u.u2}, {exp})[0]};
}
BdabeCiphertext encrypt(Engine& eng, Rng& rng, const BdabePublicKey& pk, const std::vector<const BdabePublicAttributeKey*>& attr_pks,
                        const std::string& policy, PolicyLanguage language, const Bytes& plaintext) {        // :317-358
  PolicyNode tree = parse_or_error(policy, language);
  if (!policy_in_dnf(tree)) throw RabeError("Error in bdabe/encrypt: Policy not in DNF.");
  std::vector<std::string> names;
  std::vector<G1> k1;
  std::vector<G2> k2;
  std::vector<Gt> kt;
  for (const auto* k : attr_pks) { names.push_back(k->attr); k1.push_back(k->a1); k2.push_back(k->a2); kt.push_back(k->a3); }
  std::vector<DnfTerm> terms;
  if (!json_to_dnf(tree, names, &terms)) throw std::runtime_error("called `Result::unwrap()` on an `Err` value: Error in json_to_dnf: could not parse policy as DNF");
  // _msg = pairing(rng.gen(), rng.gen()) = e(G1::one(), G2::one())^(a b)
  Fr a = rng.next_fr(), b = rng.next_fr();
  Gt msg = eng.gt_pow({eng.gt_generator()}, {fr_mul(a, b)})[0];
  BdabeCiphertext ct;
  ct.policy = {policy, language};
  if (!terms.empty()) {
    std::vector<Gt> t_gt = fold_terms(terms, kt, [&](const std::vector<Gt>& x, const std::vector<Gt>& y) { return eng.gt_mul(x, y); });
    std::vector<G1> t_g1 = fold_terms(terms, k1, [&](const std::vector<G1>& x, const std::vector<G1>& y) { return g1_add(eng, x, y); });
    std::vector<G2> t_g2 = fold_terms(terms, k2, [&](const std::vector<G2>& x, const std::vector<G2>& y) { return g2_add(eng, x, y); });
    std::vector<Fr> r;
    for (size_t t = 0; t < terms.size(); t++) r.push_back(rng.next_fr());
    const size_t m = terms.size();
    std::vector<G1> b1(m, pk.p1);
    std::vector<G2> b2(m, pk.p2);
    std::vector<Fr> rr = r;
    b1.insert(b1.end(), t_g1.begin(), t_g1.end());
    b2.insert(b2.end(), t_g2.begin(), t_g2.end());
    rr.insert(rr.end(), r.begin(), r.end());
    std::vector<G1> o1 = eng.g1_mul(b1, rr);
    std::vector<G2> o2 = eng.g2_mul(b2, rr);
    std::vector<Gt> e1 = eng.gt_mul(eng.gt_pow(t_gt, r), std::vector<Gt>(m, msg));
    for (size_t t = 0; t < m; t++) ct.j.push_back({terms[t].attrs, e1[t], o1[t], o2[t], o1[m + t], o2[m + t]});
  }
  ct.ct = seal(rng, msg, plaintext);
  return ct;
}
static void plan_decrypt(const BdabeUserKey& sk, const BdabeCiphertext& ct, PairingJob* job) {      // :367-399
  std::vector<std::string> str_attr;
  std::vector<AttrKey> keys;
  for (const auto& k : sk.sk_a) { str_attr.push_back(k.attr); keys.push_back({&k.attr, &k.au1, &k.au2}); }
  PolicyNode tree = parse_or_error(ct.policy.first, ct.policy.second);
  if (!traverse_policy(str_attr, tree)) throw RabeError("Error in bdabe/decrypt: attributes in sk do not match policy in ct.");
  job->lead_one = true;                       // no satisfiable conjunction: msg stays Gt::one() and the AES layer fails
  for (const auto& cj : ct.j) {
    if (!is_satisfiable(cj.attr, keys)) continue;
    job->lead_one = false;
    job->lead = cj.e1;
    plan_term(cj.attr, keys, cj.e2, cj.e3, cj.e4, cj.e5, sk.sk.u1, sk.sk.u2, job);
    break;
  }
}
Gt decrypt_gt(Engine& eng, const BdabeUserKey& sk, const BdabeCiphertext& ct) {
  std::vector<PairingJob> jobs(1);
  plan_decrypt(sk, ct, &jobs[0]);
  return run_pairing_jobs(eng, jobs)[0];
}
Bytes decrypt(Engine& eng, const BdabeUserKey& sk, const BdabeCiphertext& ct) { return open_or_error(decrypt_gt(eng, sk, ct), ct.ct); }
std::vector<DecryptResult> decrypt_batch(Engine& eng, const std::vector<const BdabeUserKey*>& sks, const std::vector<const BdabeCiphertext*>& cts) {
  if (sks.size() != cts.size()) throw RabeError("decrypt_batch: sks and cts differ in length");
  std::vector<PairingJob> jobs = plan_jobs(cts.size(), [&](size_t i, PairingJob* j) { plan_decrypt(*sks[i], *cts[i], j); });
  std::vector<const Bytes*> sealed;
  for (const auto* c : cts) sealed.push_back(&c->ct);
  return open_jobs(eng, jobs, sealed);
}
}  // namespace bdabe

namespace mke08 {
using namespace dnfabe;
std::pair<Mke08PublicKey, Mke08MasterKey> setup(Engine& eng, Rng& rng) {          // :130-149
  G1 g1 = eng.random_g1(rng);
  G2 g2 = eng.random_g2(rng);
  G1 p1 = eng.random_g1(rng);
  G2 p2 = eng.random_g2(rng);
  Fr y1 = rng.next_fr(), y2 = rng.next_fr();
  Gt e = eng.pairing({g1}, {g2})[0];
  std::vector<Gt> ey = eng.gt_pow({e, e}, {y1, y2});
  return {Mke08PublicKey{g1, g2, p1, p2, ey[0], ey[1]}, Mke08MasterKey{eng.g1_mul({g1}, {y1})[0], eng.g2_mul({g2}, {y2})[0]}};
}
Mke08UserKey keygen(Engine& eng, Rng& rng, const Mke08PublicKey& pk, const Mke08MasterKey& msk, const std::string& name) {   // :159-181
  Fr mk_u = rng.next_fr();
  std::vector<G1> a = eng.g1_mul({pk.p1, pk.g1}, {mk_u, mk_u});
  std::vector<G2> b = eng.g2_mul({pk.p2, pk.g2}, {mk_u, mk_u});
  Mke08UserKey k;
  k.sk = {g1_add(eng, {msk.g1}, {a[0]})[0], g2_add(eng, {msk.g2}, {b[0]})[0]};
  k.pk = {name, a[1], b[1]};
  return k;
}
Mke08SecretAuthorityKey authgen(Rng& rng, const std::string& name) { return Mke08SecretAuthorityKey{name, rng.next_fr()}; }   // :189-197
Mke08PublicAttributeKey request_authority_pk(Engine& eng, const Mke08PublicKey& pk, const std::string& attribute, const Mke08SecretAuthorityKey& ska) {   // :207-238
  if (!from_authority(attribute, ska.name)) throw RabeError("attribute " + attribute + " is not from_authority() or !is_eligible()");
  Fr exp = attr_exponent(attribute, ska.name, ska.r);
  std::vector<Gt> g = eng.gt_pow({pk.e_gg_y1, pk.e_gg_y2}, {exp, exp});
  return Mke08PublicAttributeKey{attribute, eng.g1_mul({pk.g1}, {exp})[0], eng.g2_mul({pk.g2}, {exp})[0], g[0], g[1]};
}
Mke08SecretAttributeKey request_authority_sk(Engine& eng, const Mke08PublicUserKey& pk_u, const std::string& attr, const Mke08SecretAuthorityKey& ska) {   // :248-278
  if (!from_authority(attr, ska.name)) throw RabeError("attribute " + attr + " is not from_authority() or !is_eligible()");
  Fr exp = attr_exponent(attr, ska.name, ska.r);
  return Mke08SecretAttributeKey{attr, eng.g1_mul({pk_u.g1}, {exp})[0], eng.g2_mul({pk_u.g2}, {exp})[0]};
}
Mke08Ciphertext encrypt(Engine& eng, Rng& rng, const Mke08PublicKey& pk, const std::vector<const Mke08PublicAttributeKey*>& attr_pks,
                        const std::string& policy, PolicyLanguage language, const Bytes& plaintext) {        // :290-334
  PolicyNode tree = parse_or_error(policy, language);
  if (!policy_in_dnf(tree)) throw RabeError("Error in mke08/encrypt: policy is not in dnf");
  std::vector<std::string> names;
  std::vector<G1> k1;
  std::vector<G2> k2;
  std::vector<Gt> kt1, kt2;
  for (const auto* k : attr_pks) { names.push_back(k->attr); k1.push_back(k->g1); k2.push_back(k->g2); kt1.push_back(k->gt1); kt2.push_back(k->gt2); }
  std::vector<DnfTerm> terms;
  if (!json_to_dnf(tree, names, &terms)) throw std::runtime_error("called `Result::unwrap()` on an `Err` value: Error in json_to_dnf: could not parse policy as DNF");
  // msg1 = pairing(rng.gen(), rng.gen()); msg2 = msg1.pow(rng.gen()); msg = msg1 * msg2
  Fr a = rng.next_fr(), b = rng.next_fr(), c = rng.next_fr();
  Fr ab = fr_mul(a, b);
  std::vector<Gt> ms = eng.gt_pow({eng.gt_generator(), eng.gt_generator(), eng.gt_generator()}, {ab, fr_mul(ab, c), fr_mul(ab, fr_add(fr_one(), c))});
  const Gt msg1 = ms[0], msg2 = ms[1], msg = ms[2];
  Mke08Ciphertext ct;
  ct.policy = {policy, language};
  if (!terms.empty()) {
    auto gmul = [&](const std::vector<Gt>& x, const std::vector<Gt>& y) { return eng.gt_mul(x, y); };
    std::vector<Gt> t1 = fold_terms(terms, kt1, gmul), t2 = fold_terms(terms, kt2, gmul);
    std::vector<G1> t_g1 = fold_terms(terms, k1, [&](const std::vector<G1>& x, const std::vector<G1>& y) { return g1_add(eng, x, y); });
    std::vector<G2> t_g2 = fold_terms(terms, k2, [&](const std::vector<G2>& x, const std::vector<G2>& y) { return g2_add(eng, x, y); });
    std::vector<Fr> r;
    for (size_t t = 0; t < terms.size(); t++) r.push_back(rng.next_fr());
    const size_t m = terms.size();
    std::vector<G1> b1(m, pk.p1);
    std::vector<G2> b2(m, pk.p2);
    std::vector<Fr> rr = r;
    b1.insert(b1.end(), t_g1.begin(), t_g1.end());
    b2.insert(b2.end(), t_g2.begin(), t_g2.end());
    rr.insert(rr.end(), r.begin(), r.end());
    std::vector<G1> o1 = eng.g1_mul(b1, rr);
    std::vector<G2> o2 = eng.g2_mul(b2, rr);
    std::vector<Gt> tt = t1, mm(m, msg1);
    tt.insert(tt.end(), t2.begin(), t2.end());
    mm.insert(mm.end(), m, msg2);
    std::vector<Gt> j = eng.gt_mul(eng.gt_pow(tt, rr), mm);
    for (size_t t = 0; t < m; t++) ct.e.push_back({terms[t].attrs, j[t], j[m + t], o1[t], o2[t], o1[m + t], o2[m + t]});
  }
  ct.ct = seal(rng, msg, plaintext);
  return ct;
}
// leads (j1 * j2) are multiplied in one launch for the whole batch before the pairing jobs run
static bool plan_decrypt(const Mke08UserKey& sk, const Mke08Ciphertext& ct, PairingJob* job, Gt* j1, Gt* j2) {      // :343-380
  std::vector<std::string> str_attr;
  std::vector<AttrKey> keys;
  for (const auto& k : sk.sk_a) { str_attr.push_back(k.attr); keys.push_back({&k.attr, &k.g1, &k.g2}); }
  PolicyNode tree = parse_or_error(ct.policy.first, ct.policy.second);
  if (!traverse_policy(str_attr, tree)) throw RabeError("Error in mke08/decrypt: attributes in sk do not match policy in ct.");
  job->lead_one = true;
  for (const auto& ej : ct.e) {
    if (!is_satisfiable(ej.str, keys)) continue;
    job->lead_one = false;
    *j1 = ej.j1;
    *j2 = ej.j2;
    plan_term(ej.str, keys, ej.j3, ej.j4, ej.j5, ej.j6, sk.sk.g1, sk.sk.g2, job);
    return true;
  }
  return false;
}
static std::vector<PairingJob> plan_batch(Engine& eng, const std::vector<const Mke08UserKey*>& sks, const std::vector<const Mke08Ciphertext*>& cts) {
  const size_t n = cts.size();
  std::vector<Gt> j1(n), j2(n);
  std::vector<char> has(n, 0);
  std::vector<PairingJob> jobs = plan_jobs(n, [&](size_t i, PairingJob* j) { has[i] = plan_decrypt(*sks[i], *cts[i], j, &j1[i], &j2[i]) ? 1 : 0; });
  std::vector<Gt> a, b;
  std::vector<size_t> who;
  for (size_t i = 0; i < n; i++) if (has[i] && !jobs[i].failed) { a.push_back(j1[i]); b.push_back(j2[i]); who.push_back(i); }
  if (!who.empty()) {
    std::vector<Gt> lead = eng.gt_mul(a, b);
    for (size_t t = 0; t < who.size(); t++) jobs[who[t]].lead = lead[t];
  }
  return jobs;
}
Gt decrypt_gt(Engine& eng, const Mke08UserKey& sk, const Mke08Ciphertext& ct) {
  std::vector<PairingJob> jobs(1);
  Gt j1, j2;
  if (plan_decrypt(sk, ct, &jobs[0], &j1, &j2)) jobs[0].lead = eng.gt_mul({j1}, {j2})[0];
  return run_pairing_jobs(eng, jobs)[0];
}
Bytes decrypt(Engine& eng, const Mke08UserKey& sk, const Mke08Ciphertext& ct) { return open_or_error(decrypt_gt(eng, sk, ct), ct.ct); }
std::vector<DecryptResult> decrypt_batch(Engine& eng, const std::vector<const Mke08UserKey*>& sks, const std::vector<const Mke08Ciphertext*>& cts) {
  if (sks.size() != cts.size()) throw RabeError("decrypt_batch: sks and cts differ in length");
  std::vector<PairingJob> jobs = plan_batch(eng, sks, cts);
  std::vector<const Bytes*> sealed;
  for (const auto* c : cts) sealed.push_back(&c->ct);
  return open_jobs(eng, jobs, sealed);
}
}  // namespace mke08

}  // namespace schemes
}  // namespace rabe
