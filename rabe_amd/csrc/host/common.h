// Host layer common pieces: group-element value types (canonical wire format of include/rabe_hip.h),
// RabeError, randomness sources, and a thin RAII wrapper of the engine's C ABI.
//
// This layer mirrors the reference's scheme API (rabe::schemes::{ac17,bsw,lsw,aw11}) in C++ because the
// container has no Rust toolchain; a Rust host would call the same rhip_* entry points.
#pragma once
#include <set>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/rabe_hip.h"
#include "aes_gcm.h"
#include "policy.h"

namespace rabe {

// src/error.rs:19-28
struct RabeError : std::runtime_error {
  explicit RabeError(const std::string& details) : std::runtime_error(details) {}
};

typedef std::vector<uint8_t> Bytes;
typedef std::array<uint8_t, 64> G1;
typedef std::array<uint8_t, 128> G2;
typedef std::array<uint8_t, 384> Gt;
using host::Fr;
using host::PolicyLanguage;

inline G1 g1_generator() {
  G1 g{};
  g[0] = 1;
  g[32] = 2;
  return g;
}
inline G2 g2_generator() {
  static const uint64_t L[16] = {
      0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull,     // x.c0
      0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull,     // x.c1
      0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull,     // y.c0
      0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull};     // y.c1
  G2 g{};
  memcpy(g.data(), L, 128);
  return g;
}

// ---------------------------------------------------------------------------------------------- randomness
// Stands where the reference uses rand::thread_rng() (ac17/mod.rs:143,201,281; bsw/mod.rs:227;
// lsw/mod.rs:129,188; aw11/mod.rs:131,249; aes/mod.rs:11).
struct Rng : host::FrSource {
  virtual void fill(uint8_t* out, size_t n) = 0;
  // true when draws carry no order (OS randomness): a batch may then draw on several threads, each from its own source
  virtual bool unordered() const { return false; }
  // brackets around the draws of one batch call: the chunks of a pipelined batch (pipeline.cpp) take turns here, so an ordered
  // source hands out the same values to the same items as in one unchunked call
  virtual void begin_draws() {}
  virtual void end_draws() {}
};
struct OsRng : Rng {
  // getrandom(2) in 4 KB refills: a batch draws hundreds of thousands of Fr values, one system call each would dominate it
  uint8_t pool[4096];
  size_t pool_pos = sizeof(pool);
  void fill(uint8_t* out, size_t n) override;
  bool unordered() const override { return true; }
  Fr next_fr() override {
    uint8_t b[64];
    fill(b, 64);
    return host::fr_from_le64_reduce(b);     // `Fr::random`: 512 random bits mod r
  }
};
// What the parallel draws of a batch use, one per block of items: an AES-256-CTR stream (AES-NI) keyed with 40 bytes from the OS -- a
// CSPRNG seeded by the OS, which is what rand::thread_rng() is to the reference -- instead of a system call per 4 KB (a 4096-item AW11
// batch draws 158 MB).  Without AES-NI it reads the OS like OsRng.
struct BatchRng : Rng {
  uint8_t pool[4096];
  size_t pool_pos = sizeof(pool);
  void* key = nullptr;            // host::aeshw::Key, owned
  uint64_t iv_hi = 0, counter = 0;
  OsRng os;
  BatchRng();
  ~BatchRng();
  BatchRng(const BatchRng&) = delete;
  BatchRng& operator=(const BatchRng&) = delete;
  void fill(uint8_t* out, size_t n) override;
  bool unordered() const override { return true; }
  Fr next_fr() override {
    uint8_t b[64];
    fill(b, 64);
    return host::fr_from_le64_reduce(b);
  }
};
// Explicit-randomness tape: replays a list of Fr values in draw order; byte draws (the AES
// nonce) take the low bytes of the next tape entry.
struct TapeRng : Rng {
  std::vector<Fr> tape;
  size_t pos = 0;
  explicit TapeRng(std::vector<Fr> t) : tape(std::move(t)) {}
  Fr next_fr() override {
    if (pos >= tape.size()) throw RabeError("randomness tape exhausted");
    return tape[pos++];
  }
  void fill(uint8_t* out, size_t n) override {
    size_t off = 0;
    while (off < n) {
      Fr f = next_fr();
      size_t k = n - off < 32 ? n - off : 32;
      memcpy(out + off, f.l, k);
      off += k;
    }
  }
};

// ---------------------------------------------------------------------------------------------- engine wrapper
class Engine;
class DBuf {
 public:
  DBuf() {}
  DBuf(Engine* e, size_t bytes);
  DBuf(Engine* e, const void* host_data, size_t bytes);
  ~DBuf();
  DBuf(DBuf&& o) noexcept : eng_(o.eng_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
  DBuf& operator=(DBuf&& o) noexcept;
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  void* ptr() const { return p_; }
  template <class T> T* as() const { return (T*)p_; }
  size_t size() const { return n_; }
  void download(void* host, size_t bytes) const;
 private:
  Engine* eng_ = nullptr;
  void* p_ = nullptr;
  size_t n_ = 0;
};

class Engine {
 public:
  explicit Engine(int device = 0);
  ~Engine();
  // A LANE is a context (= HIP stream + workspaces) with its own pinned staging buffers and device arena.  Lane 0 is the engine's
  // own; the pipelined packed entry points (pipeline.cpp) run chunks of a batch on worker threads, one lane each, so that one
  // chunk's record parsing / AES runs beside another's kernels and a third's PCIe copies.  The calling thread picks its lane with a
  // LaneScope; everything below (`ctx`, `pinned`, DBuf) then refers to that lane.  Tables and key handles are shared by all lanes.
  rhip_ctx* ctx() const { return lanes_[cur_lane()]->ctx; }
  int device() const { return device_; }
  void make_current() const { check(rhip_ctx_make_current(lanes_[0]->ctx), "rhip_ctx_make_current"); }
  void ensure_lanes(size_t count);                 // call before handing lanes to threads
  size_t lane_count();                             // lanes that exist: a LaneScope beyond them falls back to lane 0 (cur_lane)
  size_t lanes() const { return lanes_.size(); }
  static int current_lane() { return tl_lane(); }
  struct LaneScope {
    int prev;
    explicit LaneScope(int lane);
    ~LaneScope();
  };
  // Threads that work on the engine side by side (the submission queue's batch leaders, the pipelined packed calls) hold a Busy for
  // the whole of their operation, up to and including the wait for their stream.  A cached device handle (`aux`, `ac17_pk`) that is
  // evicted while any Busy is alive is parked, not destroyed -- another lane may have been handed the raw pointer and be about to
  // launch with it, or have kernels running on it -- and the parked handles are destroyed when the last Busy ends.
  struct Busy {
    Engine& e;
    uint64_t start;                                  // the engine's busy clock when this scope began
    explicit Busy(Engine& eng);
    ~Busy();
  };
  // Inside an ArenaScope, DBufs of the current lane are carved from one grow-only device block (no hipMalloc / hipFree -- the
  // latter synchronises the whole device -- per buffer); the block is recycled when the outermost scope ends (after a stream sync).
  // A request the block cannot hold falls back to hipMalloc and makes the block grow for the next call.
  struct ArenaScope {
    Engine& e;
    explicit ArenaScope(Engine& eng);
    ~ArenaScope();
  };
  // The current call stages secret material (master-key-derived scalars of a bulk keygen): when its outermost ArenaScope ends, what the
  // call used of the lane's device arena and of its pinned staging buffers is overwritten with zeros -- both outlive the call otherwise.
  void scrub_when_done();
  // the lighter form for calls that only hold SESSION secrets (the Gt values and AES key schedules of a seal / open, the encryption
  // scalars): what the call used of the device arena and of pinned slot 0 is zeroed -- not the staged record blobs (hundreds of MB of
  // public ciphertext whose host-side memset would cost more than the call)
  void scrub_session_when_done();
  // the lane's SIDE context (a second stream, created on first use): work that may run beside the main context's kernels
  rhip_ctx* side_ctx();
  void* arena_take(size_t bytes);                  // nullptr: no scope active or block full
  bool arena_owns(const void* p) const;
  void check(int32_t rc, const char* what) const;

  // Level E conveniences on host values (each a small batched launch)
  std::vector<G1> g1_mul(const std::vector<G1>& p, const std::vector<Fr>& k);
  std::vector<G2> g2_mul(const std::vector<G2>& p, const std::vector<Fr>& k);
  std::vector<Gt> gt_pow(const std::vector<Gt>& a, const std::vector<Fr>& k);
  std::vector<Gt> gt_mul(const std::vector<Gt>& a, const std::vector<Gt>& b);
  std::vector<Gt> pairing(const std::vector<G1>& p, const std::vector<G2>& q);
  // `rng.gen::<G1>()` = generator * Fr::random
  G1 random_g1(Rng& rng) { return g1_mul({g1_generator()}, {rng.next_fr()})[0]; }
  G2 random_g2(Rng& rng) { return g2_mul({g2_generator()}, {rng.next_fr()})[0]; }
  // `rng.gen::<Gt>()`: e(G1::one(), G2::one()) ^ Fr::random
  Gt random_gt(Rng& rng);
  const Gt& gt_generator();     // e(G1::one(), G2::one()), computed once
  // Calls whose operands repeat one base many times (the schemes multiply public-key elements by per-row scalars)
  // are served from per-base window tables built on first use and cached here: 32 mixed additions per element
  // instead of 254 doublings + ~127 additions.  Same group elements, hence the same bytes.  Building a table costs
  // 8 160 variable-base multiplications, so it pays for itself from a few thousand elements per base on (Gt: more).
  // device tables of AC17 public keys (g, h_a[3], e_gh_ka[2]; ~1.3 GB each with the 16-bit windows), built on first use
  rhip_ac17_pk* ac17_pk(const G1& g, const std::vector<G2>& h_a, const std::vector<Gt>& e_gh_ka);
  size_t fixed_base_min = 1024;   // elements sharing one base, in one call or accumulated over calls, before it gets a window table
  std::map<std::string, size_t> seen_[3];   // uses so far of the heavier bases (G1, G2, Gt)
  // grow-only pinned host staging buffers (slot 0..3) for the packed batch entry points: PCIe copies at full rate
  uint8_t* pinned(int slot, size_t bytes);
  // slot 3 as a bump allocator for the call's parameter packs (records.h: ParamPack): every take stays valid -- asynchronous copies
  // may still read it -- until the outermost ArenaScope of the call ends; a take the block cannot hold waits for the stream first
  uint8_t* pinned_bump(size_t bytes);
  // +1 / -1 around a helper thread that reads earlier takes of the current lane's block (records.h: PendingCopy): while the count is
  // not zero pinned_bump refuses to replace the block (callers with outstanding copies reserve up front: pinned_reserve)
  void pinned_hold(int delta);
  // makes sure the next takes of `bytes` in total do not move the block (a caller whose helper threads read earlier takes reserves first)
  void pinned_reserve(size_t bytes);
  // window table of the Gt generator e(G1::one(), G2::one()) (random Gt messages of a batch in one launch, on device)
  rhip_gt_table* gt_generator_table();
  // device-side key handles of the packed entry points (rhip_bsw_pk, rhip_lsw_pk, rhip_aw11_pk), cached by the key's bytes and
  // destroyed with the engine; at most `cap` entries per kind live at a time
  void* aux(const std::string& kind, const std::string& key, void* (*make)(Engine&, const void*), const void* arg, void (*destroy)(void*), size_t cap = 4);

 private:
  struct Lane {
    rhip_ctx* ctx = nullptr;
    rhip_ctx* side = nullptr;
    void* pin[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t pin_bytes[4] = {0, 0, 0, 0};
    uint8_t* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0, arena_want = 0;
    int arena_depth = 0;
    size_t pin3_used = 0;
    int pin3_hold = 0;
    size_t pin_touched[4] = {0, 0, 0, 0};          // bytes of each pinned slot handed out since the outermost scope began
    int scrub = 0;                                  // bit 0: device arena, bits 1..4: pinned slots 0..3
  };
  std::vector<std::unique_ptr<Lane>> lanes_;
  int device_ = 0;
  static int& tl_lane();
  size_t cur_lane() const { size_t l = (size_t)tl_lane(); return l < lanes_.size() ? l : 0; }
  mutable std::recursive_mutex mu_;                // tables, key handles, use counters: shared by the lanes
  bool have_e_ = false;
  Gt e_gen_;
  std::map<std::string, rhip_g1_table*> t1_;
  std::map<std::string, rhip_g2_table*> t2_;
  std::map<std::string, rhip_gt_table*> tt_;
  struct Pk17 { rhip_ac17_pk* h; uint64_t used; };
  std::map<std::string, Pk17> pk17_;
  struct Aux { void* h; void (*destroy)(void*); uint64_t used; };
  std::map<std::string, std::map<std::string, Aux>> aux_;
  uint64_t use_clock_ = 0;                         // least-recently-used eviction of the two caches above: ONE entry at a time
  int busy_ = 0;                                   // live Busy scopes
  // Evicted while busy_ > 0: parked with the busy clock of the eviction.  Only scopes that began BEFORE the eviction can still hold the handle
  // (it left the cache then), so a parked handle is destroyed as soon as every live scope is younger than it -- under sustained load from
  // overlapping queue lanes busy_ may never reach 0, and rotating public keys would otherwise park ~1.7 GB of tables per eviction without bound.
  struct Parked { uint64_t at; void* h; void (*destroy)(void*); };
  std::vector<Parked> parked_;
  uint64_t busy_clock_ = 0;
  std::multiset<uint64_t> busy_starts_;
  void retire(void* h, void (*destroy)(void*));    // destroy now (nobody else is working) or park
  rhip_gt_table* e_gen_tbl_ = nullptr;
  void destroy_table(rhip_g1_table* t);
  void destroy_table(rhip_g2_table* t);
  void destroy_table(rhip_gt_table* t);
  template <size_t N, class TBL, class CREATE, class MUL, class GENERIC>
  std::vector<std::array<uint8_t, N>> mul_grouped(const std::vector<std::array<uint8_t, N>>& p, const std::vector<Fr>& k,
                                                   std::map<std::string, TBL*>& cache, CREATE create, MUL mul, GENERIC generic);
};

// The batched decoding checks of untrusted records (coordinates < p, curve, subgroup: what rabe-bn's decoding establishes) on the lane's
// side context, BESIDE the scheme's kernels on the main one: the checks of a batch are small launches (one lane per element, a few
// hundred waves) that leave most of the GPU idle when they run alone, and the scheme's kernels do not depend on their verdicts -- they
// terminate on any input, and a failed item's result is discarded.  Construct after the staged uploads were enqueued on the main
// context (the side context waits for them), `add` the element arrays, launch the scheme's work, then `collect`.
// RABE_MEMBER_INLINE=1 runs the checks on the main context instead (A/B).
class MemberChecks {
 public:
  explicit MemberChecks(Engine& eng);
  // which: 1 G1 on curve, 2 G2 in the r-torsion, 3 Gt in the order-r subgroup, 4 G2 on the twist with canonical coordinates (no subgroup
  // test).  With `dev_seg_off` (n_seg + 1 device offsets, in rows of `scale` elements) the verdicts are folded per segment on the
  // device: ok(k) then has n_seg entries instead of `count`.
  void add(int which, const void* dev, size_t count, const uint32_t* dev_seg_off = nullptr, size_t n_seg = 0, uint32_t scale = 1);
  // the G2 subgroup test of the listed elements only: ok(k)[t] = verdict of element idx[t]
  void add_g2_at(const void* dev, const std::vector<uint32_t>& idx);
  rhip_ctx* ctx() const { return cx_; }
  size_t add_count() const { return flags_.size(); }          // the index the next add's verdicts will have
  // waits for the checks (not for the main context).  The Gt checks -- one lane per element, 64-thread blocks that would each take a
  // SIMD from a CU the Miller kernel's four-wave blocks need whole -- are launched HERE, i.e. after the caller queued its decrypt: the side
  // stream is released when the decrypt's Miller loops are done (rhip_ctx_release_after_miller) and the checks run beside its final
  // exponentiation, which leaves most of the chip idle below 65 536 items (measured, AC17 at 20 480 items: 29.8 -> 25 ms per checked decrypt)
  void collect();
  ~MemberChecks();
  const std::vector<uint32_t>& ok(size_t k) const { return flags_[k]; }      // verdicts of the k-th add
 private:
  Engine& eng_;
  rhip_ctx* cx_;
  std::vector<DBuf> dev_, scratch_;
  std::vector<std::vector<uint32_t>> flags_;
  struct Deferred { size_t k; const void* dev; size_t count; const uint32_t* seg; size_t n_seg; uint32_t scale; };
  std::vector<Deferred> later_;
  bool requested_ = false;
  void launch(int which, size_t k, const void* dev, size_t count, const uint32_t* dev_seg_off, size_t n_seg, uint32_t scale);
};

bool walk_checks();
// G2 elements of untrusted records whose subgroup membership the decrypt's OWN Miller loops establish (include/rabe_hip.h:
// rhip_ctx_collect_walk_verdicts): the decrypt walks the ciphertext's G2 elements anyway, and the point its last step leaves behind
// says whether the argument was a member -- 2.2 k field multiplications per element saved against the stand-alone test.  Everything a
// decoder has to establish stays established: curve equation + coordinate range of EVERY element (a cheap pass on the side context),
// the stand-alone subgroup test for the elements the decrypt does not walk (leaves the policy did not select), and for the elements of any
// item whose walk examined fewer arguments than expected (a skipped pair, a launch path without verdicts).
// Use: construct after MemberChecks (same place), `arm()` immediately before the scheme's decrypt call on the main context,
// `finish()` after that call was enqueued and MemberChecks::collect() ran: ok[j] per segment (item).
class WalkedG2 {
 public:
  // `count` elements in n_seg segments: rows [seg_off[j], seg_off[j + 1]) x scale elements (dev_seg_off: the same offsets on the device).
  // walked_off == nullptr: the decrypt walks every element; else element indices walked_idx[walked_off[j] .. walked_off[j + 1]) of item j.
  WalkedG2(Engine& eng, MemberChecks& mc, const void* dev_g2, size_t count, const uint32_t* dev_seg_off, const std::vector<uint32_t>& seg_off,
           uint32_t scale, const std::vector<uint32_t>* walked_idx = nullptr, const std::vector<uint32_t>* walked_off = nullptr,
           uint32_t extra_walks = 0 /* walking arguments per item that are not elements of this array (aw11: the sum of the C3 terms) */);
  void arm();
  void finish(std::vector<uint8_t>* ok);
  ~WalkedG2();          // an armed request no decrypt consumed (an exception on the way) is withdrawn: its arrays go back to the arena
 private:
  Engine& eng_;
  MemberChecks& mc_;
  const void* dev_;
  size_t count_, n_seg_;
  std::vector<uint32_t> seg_off_;
  uint32_t scale_, extra_ = 0;
  const std::vector<uint32_t>* widx_;
  const std::vector<uint32_t>* woff_;
  std::vector<uint32_t> rest_item_;          // item of every element tested stand-alone on the side context
  size_t k_curve_ = 0, k_rest_ = 0;
  bool has_rest_ = false, armed_ = false;
  DBuf d_verdicts_;                          // fail[n_seg] | count[n_seg]
};

template <class T, size_t N>
inline std::vector<uint8_t> flatten(const std::vector<std::array<T, N>>& v) {
  std::vector<uint8_t> o(v.size() * N);
  for (size_t i = 0; i < v.size(); i++) memcpy(o.data() + i * N, v[i].data(), N);
  return o;
}
inline std::vector<uint8_t> flatten_fr(const std::vector<Fr>& v) {
  std::vector<uint8_t> o(v.size() * 32);
  for (size_t i = 0; i < v.size(); i++) memcpy(o.data() + i * 32, v[i].l, 32);
  return o;
}

}  // namespace rabe
