// Pipelined packed batches.
//
// A packed entry point (schemes.cpp / packed.cpp) is a sequence of stages that keep different parts of the machine busy one after the
// other: host cores (record parsing, policy plans, draws), PCIe (staging copies), the GPU (membership pass, the scheme's kernels), PCIe
// again, host cores again (KDF + AES-GCM, record assembly).  Run as one piece, every part waits for the others: of the 70 ms of an AC17
// encrypt + decrypt of 20 480 items only ~25 are GPU time.  Here a batch is cut into chunks of items and the SAME entry point runs on
// every chunk, two chunks at a time, each on its own worker thread and engine lane (common.h: a context = stream + workspaces, pinned
// staging buffers, device arena) -- one chunk's parsing runs beside another's kernels and a third's copies.  OFF by default: see `cut`.
//
// Results are those of the unchunked call, byte for byte:
//  * records / plaintexts of chunk k land where the unchunked call puts them (producers: sized up front by the entry point's own sizing
//    pass; consumers: each chunk writes into the slice its well-formed records span, the slices are closed up afterwards);
//  * ordered randomness (a seeded source, the tests' tapes) is drawn chunk after chunk: the entry points bracket their draws with
//    Rng::begin_draws / end_draws, and a chunk's source waits at begin_draws until every earlier chunk has passed end_draws;
//  * per-item status / error texts are the chunk's, concatenated; a call-level exception of any chunk is rethrown after all have ended.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "schemes.h"

namespace rabe {
namespace pipeline {
namespace {

struct DrawGate {
  std::mutex mu;
  std::condition_variable cv;
  size_t turn = 0;
  void enter(size_t k) {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return turn >= k; });
  }
  void leave(size_t k) {
    std::lock_guard<std::mutex> g(mu);
    if (turn == k) turn = k + 1;
    cv.notify_all();
  }
};
// the source a chunk sees: the caller's, behind the gate
struct GatedRng : Rng {
  Rng& base;
  DrawGate& gate;
  size_t k;
  bool in = false, done = false;
  GatedRng(Rng& b, DrawGate& g, size_t chunk) : base(b), gate(g), k(chunk) {}
  ~GatedRng() override { pass(); }
  void take() { if (!in) { gate.enter(k); in = true; } }
  void pass() { if (!done) { take(); gate.leave(k); done = true; } }              // a chunk that ends without (further) draws
  Fr next_fr() override { guard(); return base.next_fr(); }
  void fill(uint8_t* out, size_t n) override { guard(); base.fill(out, n); }
  void guard() {
    if (done) throw RabeError("pipelined batch: the entry point drew outside its draw bracket (results would depend on the chunking)");
    take();
  }
  bool unordered() const override { return base.unordered(); }
  void begin_draws() override {
    // one bracket per call: a second one would draw after later chunks have already taken their turn
    if (done) throw RabeError("pipelined batch: the entry point opened a second draw bracket (results would depend on the chunking)");
    take();
    base.begin_draws();
  }
  void end_draws() override { base.end_draws(); pass(); }
};

size_t env_size(const char* name, size_t dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  long v = atol(e);
  return v > 0 ? (size_t)v : dflt;
}
// chunk k = items [lo(k), hi(k)): `base` items each, the first `extra` chunks one more (rabe_amd/shard.py: shard_range has the same rule)
struct Cut {
  size_t chunks, lanes, base, extra, n;
  size_t lo(size_t k) const { const size_t v = k * base + (k < extra ? k : extra); return v < n ? v : n; }
  size_t hi(size_t k) const { const size_t v = lo(k) + base + (k < extra ? 1 : 0); return v < n ? v : n; }
};
static Cut even_cut(size_t n, size_t chunks, size_t lanes) { return {chunks, lanes, n / chunks, n % chunks, n}; }
// Chunks of at least `min_chunk` items, one per lane.  Measured (tools/bench_packed_pipeline.py, AC17, 50 attributes, ops/s unchunked ->
// two half-size chunks on two lanes, after the unchunked call lost its per-buffer allocations, per-call line preparation and per-element
// verdict downloads): 20 480 items 339 k -> 311 k, 65 536 458 k -> 433 k, 131 072 406 k -> 402 k; finer chunks are worse still (20 480 as
// 5 x 4096 on three lanes: 224 k) -- every stage of a chunk has a fixed cost, the host stages already use all cores, small launches
// under-fill the GPU.  So host_abi.cpp passes min_chunk = "never"; RABE_PACKED_CHUNK / RABE_PACKED_LANES switch it on (the tests do).
// A device group is cut into one block per engine, whatever the environment says.
Cut cut(size_t n_engines, size_t n, size_t min_chunk) {
  if (n_engines > 1) {
    const size_t chunks = n < n_engines ? (n ? n : 1) : n_engines;
    return even_cut(n, chunks, chunks);
  }
  size_t lanes = env_size("RABE_PACKED_LANES", 2);
  if (lanes > 8) lanes = 8;
  const bool forced = getenv("RABE_PACKED_CHUNK") != nullptr;
  min_chunk = env_size("RABE_PACKED_CHUNK", min_chunk);
  size_t chunks = n / (min_chunk ? min_chunk : 1);
  const size_t cap = forced ? 2 * lanes : lanes;
  if (chunks > cap) chunks = cap;
  if (chunks < 2 || lanes < 2) return {1, 1, n, 0, n};
  const size_t per = (n + chunks - 1) / chunks;
  chunks = (n + per - 1) / per;
  Cut c{chunks, lanes < chunks ? lanes : chunks, per, 0, n};
  return c;
}
// run(k, engine) for every chunk; the first exception is rethrown after all workers ended.  One engine: `lanes` chunks at a time on its
// lanes, chunks handed out in order.  A group: chunk k on engine k, lane 0, one thread each, the engine's device current in that thread.
//
// In a group chunk w belongs to worker w alone, so a worker that fails BEFORE run(w) (its device cannot be made current, its lane cannot be
// set up) leaves a chunk nobody will run: `never_runs(w)` tells the caller, which must let the chunks behind it go on (produce: the draw
// gate would otherwise keep every later chunk waiting for w's turn and the join below would never return).  With one engine the other
// workers take the chunk.  RABE_FAULT_GROUP_WORKER=w injects exactly that failure (tests/test_gpu_device_group.py).
void fan_out(const std::vector<Engine*>& engines, const Cut& c, const std::function<void(size_t, Engine&)>& run,
             const std::function<void(size_t)>& never_runs = nullptr) {
  std::atomic<size_t> next{0};
  std::mutex mu;
  std::exception_ptr first;
  const bool group = engines.size() > 1;
  if (!group) engines[0]->ensure_lanes(c.lanes);
  const char* fault_env = group ? getenv("RABE_FAULT_GROUP_WORKER") : nullptr;
  const long fault_w = (fault_env && *fault_env) ? atol(fault_env) : -1;
  auto work = [&](size_t w) {
    Engine& eng = *engines[group ? w : 0];
    bool entered = false;
    try {
      if ((long)w == fault_w) throw RabeError("device group: worker " + std::to_string(w) + " failed before its block started (injected fault)");
      eng.make_current();
      Engine::Busy working(eng);
      Engine::LaneScope scope(group ? 0 : (int)w);
      for (;;) {
        const size_t k = group ? w : next.fetch_add(1);
        if (k >= c.chunks) return;
        entered = true;
        run(k, eng);
        if (group) return;
      }
    } catch (...) {
      {
        std::lock_guard<std::mutex> g(mu);
        if (!first) first = std::current_exception();
      }
      if (group && !entered && w < c.chunks && never_runs) {
        try { never_runs(w); } catch (...) {}
      }
    }
  };
  std::vector<std::thread> th;
  const size_t workers = group ? c.chunks : c.lanes;
  for (size_t w = 1; w < workers; w++) th.emplace_back(work, w);
  work(0);
  for (auto& t : th) t.join();
  if (group) engines[0]->make_current();            // the caller's thread goes on with the host's own engine
  if (first) std::rethrow_exception(first);
}

}  // namespace

bool produce(const std::vector<Engine*>& engines, Rng& rng, size_t n, size_t min_chunk, const ProduceFn& call, uint8_t* out_buf, size_t out_cap, uint64_t* out_off) {
  const Cut c = cut(engines.size(), n, min_chunk);
  if (c.chunks == 1) return call(*engines[0], 0, n, rng, out_buf, out_cap, out_off);
  // the entry point's own sizing pass (no buffer: it fills the offsets and returns before drawing anything)
  (void)call(*engines[0], 0, n, rng, nullptr, 0, out_off);
  if (!out_buf || out_cap < out_off[n]) return false;
  DrawGate gate;
  std::atomic<bool> ok{true};
  fan_out(engines, c, [&](size_t k, Engine& eng) {
    const size_t lo = c.lo(k), hi = c.hi(k);
    GatedRng r(rng, gate, k);
    std::vector<uint64_t> off(hi - lo + 1);
    if (!call(eng, lo, hi, r, out_buf + out_off[lo], (size_t)(out_off[hi] - out_off[lo]), off.data())) ok = false;
    else if (off[hi - lo] != out_off[hi] - out_off[lo]) throw RabeError("pipelined batch: a chunk's records do not have the announced size");
  }, [&](size_t k) { gate.enter(k); gate.leave(k); });          // a block that never started takes its turn at the draw gate and passes it on
  return ok;
}

bool consume(const std::vector<Engine*>& engines, size_t n, size_t min_chunk, const uint64_t* in_off, size_t in_len, const ConsumeFn& call, int32_t* status,
             uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors) {
  const Cut c = cut(engines.size(), n, min_chunk);
  if (c.chunks == 1 || !in_off) return call(*engines[0], 0, n, status, pt_buf, pt_cap, pt_off, errors);
  // what the entry points require of the plaintext buffer: the total size of the well-formed records (packed.cpp: check_offsets)
  std::vector<uint64_t> span(c.chunks + 1, 0);
  for (size_t k = 0; k < c.chunks; k++) {
    const size_t lo = c.lo(k), hi = c.hi(k);
    uint64_t s = 0;
    for (size_t i = lo; i < hi; i++) if (in_off[i] <= in_off[i + 1] && in_off[i + 1] <= in_len) s += in_off[i + 1] - in_off[i];
    span[k + 1] = span[k] + s;
  }
  if (!pt_buf || pt_cap < span[c.chunks]) return false;
  std::vector<std::vector<uint64_t>> off(c.chunks);
  std::vector<std::vector<std::string>> errs(c.chunks);
  std::atomic<bool> ok{true};
  fan_out(engines, c, [&](size_t k, Engine& eng) {
    const size_t lo = c.lo(k), hi = c.hi(k);
    off[k].assign(hi - lo + 1, 0);
    if (!call(eng, lo, hi, status + lo, pt_buf + span[k], (size_t)(span[k + 1] - span[k]), off[k].data(), &errs[k])) ok = false;
  });
  if (!ok) return false;
  // close the slices up: plaintexts contiguous in item order, as the unchunked call leaves them
  errors->assign(n, "");
  pt_off[0] = 0;
  uint64_t cur = 0;
  for (size_t k = 0; k < c.chunks; k++) {
    const size_t lo = c.lo(k), cnt = off[k].size() - 1;
    const uint64_t len = off[k][cnt];
    if (len && cur != span[k]) memmove(pt_buf + cur, pt_buf + span[k], (size_t)len);
    for (size_t i = 0; i < cnt; i++) {
      pt_off[lo + i + 1] = cur + off[k][i + 1];
      if (i < errs[k].size()) (*errors)[lo + i] = std::move(errs[k][i]);
    }
    cur += len;
  }
  return true;
}

}  // namespace pipeline
}  // namespace rabe
