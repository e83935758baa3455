// records.cpp -- device-side tails of the packed entry points: see records.h
#include "records.h"

#include <algorithm>
#include <chrono>
#include <functional>
#include <future>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace rabe {
void parallel_for(size_t n, const std::function<void(size_t)>& fn);          // schemes.cpp
namespace schemes {

namespace {
// RABE_HOST_TIMING: what the tails spend where (with a stream sync at every lap, so that device time is attributed to its stage)
struct TailTimer {
  bool on;
  Engine& eng;
  const char* what;
  std::chrono::steady_clock::time_point t0;
  TailTimer(Engine& e, const char* w) : on(getenv("RABE_HOST_TIMING") != nullptr), eng(e), what(w), t0(std::chrono::steady_clock::now()) {}
  void lap(const char* stage) {
    if (!on) return;
    rhip_sync(eng.ctx());
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[host-timing]   %s: %s %.1f ms\n", what, stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
}  // namespace

size_t ParamPack::add(const void* p, size_t bytes) {
  const size_t at = (host_.size() + 255) & ~(size_t)255;
  host_.resize(at + bytes);
  if (bytes) memcpy(host_.data() + at, p, bytes);
  return at;
}
void ParamPack::upload() {
  const size_t bytes = host_.size() ? host_.size() : 4;
  d_ = DBuf(&eng_, bytes);
  if (host_.empty()) return;
  uint8_t* pin = eng_.pinned_bump(host_.size());          // stays untouched until the call's ArenaScope ends (copies are asynchronous)
  memcpy(pin, host_.data(), host_.size());
  eng_.check(rhip_upload_async(eng_.ctx(), d_.ptr(), pin, host_.size()), "upload (parameters)");
}

void sym_shape(const std::vector<uint32_t>& len, std::vector<uint32_t>* blk_off, std::vector<uint32_t>* seg_off) {
  const size_t n = len.size();
  blk_off->assign(n + 1, 0);
  seg_off->assign(n + 1, 0);
  uint64_t b = 0, s = 0;
  for (size_t i = 0; i < n; i++) {
    const uint64_t blocks = ((uint64_t)len[i] + 15) / 16;
    b += blocks;
    s += (blocks + 63) / 64;
    if (b > 0xFFFFFFF0ull) throw RabeError("packed call: more than 2^32 AES blocks in one batch");
    (*blk_off)[i + 1] = (uint32_t)b;
    (*seg_off)[i + 1] = (uint32_t)s;
  }
}

void emit_sealed_records(Engine& eng, const std::vector<RecordLayout>& layouts, size_t n, const uint32_t* item_layout,
                         const std::vector<const void*>& dev_src, const std::vector<uint64_t>& src_item_off, const void* d_msg,
                         const uint8_t* nonces, const uint8_t* pt_blob, const uint64_t* pt_off, const uint64_t* out_off, uint8_t* out_buf,
                         PendingCopy* defer) {
  if (!n) return;
  TailTimer tm(eng, "emit_sealed_records");
  tm.lap("kernels queued before the tail");
  const size_t n_src = dev_src.size();
  if (src_item_off.size() != n_src * n) throw RabeError("emit_sealed_records: src_item_off has the wrong size");
  std::vector<uint32_t> layout_off{0}, map;
  for (const auto& l : layouts) {
    map.insert(map.end(), l.map.begin(), l.map.end());
    layout_off.push_back((uint32_t)map.size());
  }
  std::vector<uint32_t> len(n), blk_off, seg_off;
  std::vector<uint64_t> sealed_off(n);
  for (size_t i = 0; i < n; i++) {
    const uint64_t l = pt_off[i + 1] - pt_off[i];
    if (l > 0xFFFFFF00ull) throw RabeError("packed call: a plaintext of 4 GB or more");
    len[i] = (uint32_t)l;
    sealed_off[i] = out_off[i] + layouts[item_layout[i]].bytes() + 4;
    if (sealed_off[i] + l + 28 != out_off[i + 1]) throw RabeError("emit_sealed_records: record sizes do not add up");
  }
  sym_shape(len, &blk_off, &seg_off);
  rhip_ctx* cx = eng.ctx();
  auto pp_owner = std::make_shared<ParamPack>(eng);
  ParamPack& pp = *pp_owner;
  const size_t h_out_off = pp.add(out_off, n * 8), h_layout = pp.add(item_layout, n * 4), h_loff = pp.add(layout_off), h_map = pp.add(map),
               h_src = pp.add(dev_src), h_sio = pp.add(src_item_off), h_nonce = pp.add(nonces, n * 12), h_pt_off = pp.add(pt_off, n * 8),
               h_soff = pp.add(sealed_off), h_len = pp.add(len), h_blk = pp.add(blk_off), h_seg = pp.add(seg_off);
  const uint64_t pt_bytes = pt_off[n] - pt_off[0];
  const bool small_pt = pt_bytes <= (8u << 20);
  const size_t h_pt = small_pt ? pp.add(pt_blob + pt_off[0], (size_t)pt_bytes) : 0;
  pp.upload();
  tm.lap("tables + parameter pack");
  DBuf d_pt_big;
  const uint8_t* d_pt;
  if (small_pt) {
    d_pt = pp.dev<uint8_t>(h_pt);
  } else {                                      // large payloads go straight from the caller's memory
    d_pt_big = DBuf(&eng, (size_t)pt_bytes);
    eng.check(rhip_upload_async(cx, d_pt_big.ptr(), pt_blob + pt_off[0], (size_t)pt_bytes), "upload (plaintexts)");
    d_pt = d_pt_big.as<uint8_t>();
  }
  // the records of this call start at out_off[0] of the caller's buffer: the device block holds them from its first byte, the kernels
  // get a base pointer shifted by out_off[0]
  const size_t out_bytes = (size_t)(out_off[n] - out_off[0]);
  DBuf d_out_buf(&eng, out_bytes ? out_bytes : 4), d_ws(&eng, rhip_seal_workspace_bytes(n, seg_off[n]));
  uint8_t* const d_out = d_out_buf.as<uint8_t>() - out_off[0];
  eng.check(rhip_assemble_records(cx, n, d_out, pp.dev<uint64_t>(h_out_off), pp.dev<uint32_t>(h_layout), pp.dev<uint32_t>(h_loff),
                                  pp.dev<uint32_t>(h_map), (uint32_t)n_src, pp.dev<const uint8_t*>(h_src), pp.dev<uint64_t>(h_sio)),
            "rhip_assemble_records");
  tm.lap("rhip_assemble_records");
  // pt_off is relative to pt_blob; the device copy starts at pt_off[0]
  eng.scrub_session_when_done();          // Gt session keys, AES key schedules and the encryption scalars do not outlive the call
  eng.check(rhip_seal_batch(cx, n, (const rhip_gt*)d_msg, pp.dev<uint8_t>(h_nonce), d_pt - pt_off[0], pp.dev<uint64_t>(h_pt_off), d_out,
                            pp.dev<uint64_t>(h_soff), pp.dev<uint32_t>(h_len), pp.dev<uint32_t>(h_blk), blk_off[n], pp.dev<uint32_t>(h_seg), seg_off[n],
                            1, d_ws.ptr()),
            "rhip_seal_batch");
  tm.lap("rhip_seal_batch");
  if (defer) {
    // device -> pinned staging on the side stream, then staging -> the caller's buffer on a helper thread that waits for THIS copy only
    // (a copy straight into pageable memory occupies the runtime for its whole length).  Measured: the runtime
    // performs a large D2H copy as a blit KERNEL that covers the chip, so the next part's kernels slow down by what the copy takes --
    // no net overlap; callers therefore cut a batch into parts only when asked to (RABE_AC17_ENC_PARTS)
    rhip_ctx* const side = eng.side_ctx();
    eng.check(rhip_ctx_wait_for(side, cx), "rhip_ctx_wait_for");          // the copy starts when everything queued so far is done
    uint8_t* const dst = out_buf + out_off[0];
    uint8_t* const pin = eng.pinned_bump(out_bytes);
    eng.check(rhip_download_async(side, pin, d_out_buf.ptr(), out_bytes), "download (records)");
    rhip_event* ev = nullptr;
    eng.check(rhip_event_record(side, &ev), "rhip_event_record");
    auto fut = std::make_shared<std::future<int32_t>>(std::async(std::launch::async, [ev, dst, pin, out_bytes]() -> int32_t {
      const int32_t rc = rhip_event_wait(ev);
      if (rc) return rc;
      const size_t piece = (size_t)1 << 20, pieces = (out_bytes + piece - 1) / piece;
      parallel_for(pieces, [&](size_t k) { const size_t o = k * piece; memcpy(dst + o, pin + o, std::min(piece, out_bytes - o)); });
      return 0;
    }));
    defer->d_out = std::move(d_out_buf);
    defer->d_pt = std::move(d_pt_big);
    defer->d_ws = std::move(d_ws);
    defer->pp = pp_owner;
    defer->fut = fut;
    defer->eng = &eng;
    eng.pinned_hold(+1);               // the helper thread reads the lane's pinned block: it must not be replaced until wait()
    return;
  }
  eng.check(rhip_download(cx, out_buf + out_off[0], d_out_buf.ptr(), out_bytes), "download (records)");
  tm.lap("copy out");
}
void PendingCopy::wait(Engine& eng) {
  if (!fut) return;
  auto f = std::static_pointer_cast<std::future<int32_t>>(fut);
  const int32_t rc = f->get();
  fut.reset();
  if (this->eng) { this->eng->pinned_hold(-1); this->eng = nullptr; }
  eng.check(rc, "download (records)");
}
PendingCopy::~PendingCopy() {          // a caller that unwinds without wait(): the helper thread is joined before the buffers go
  if (fut) {
    auto f = std::static_pointer_cast<std::future<int32_t>>(fut);
    (void)f->get();
    fut.reset();
  }
  if (eng) { eng->pinned_hold(-1); eng = nullptr; }
}

void emit_plain_records(Engine& eng, const std::vector<RecordLayout>& layouts, size_t n, const uint32_t* item_layout,
                        const std::vector<const void*>& dev_src, const std::vector<uint64_t>& src_item_off, const uint64_t* out_off, uint8_t* out_buf) {
  if (!n) return;
  const size_t n_src = dev_src.size();
  if (src_item_off.size() != n_src * n) throw RabeError("emit_plain_records: src_item_off has the wrong size");
  std::vector<uint32_t> layout_off{0}, map;
  for (const auto& l : layouts) {
    map.insert(map.end(), l.map.begin(), l.map.end());
    layout_off.push_back((uint32_t)map.size());
  }
  for (size_t i = 0; i < n; i++)
    if (out_off[i] + layouts[item_layout[i]].bytes() != out_off[i + 1]) throw RabeError("emit_plain_records: record sizes do not add up");
  rhip_ctx* cx = eng.ctx();
  ParamPack pp(eng);
  const size_t h_out_off = pp.add(out_off, n * 8), h_layout = pp.add(item_layout, n * 4), h_loff = pp.add(layout_off), h_map = pp.add(map),
               h_src = pp.add(dev_src), h_sio = pp.add(src_item_off);
  pp.upload();
  DBuf d_out(&eng, (size_t)out_off[n]);
  eng.check(rhip_assemble_records(cx, n, d_out.as<uint8_t>(), pp.dev<uint64_t>(h_out_off), pp.dev<uint32_t>(h_layout), pp.dev<uint32_t>(h_loff),
                                  pp.dev<uint32_t>(h_map), (uint32_t)n_src, pp.dev<const uint8_t*>(h_src), pp.dev<uint64_t>(h_sio)),
            "rhip_assemble_records");
  eng.check(rhip_download(cx, out_buf + out_off[0], d_out.as<uint8_t>() + out_off[0], (size_t)(out_off[n] - out_off[0])), "download (records)");
}

struct BlobGather::Up { std::future<int32_t> f; };
BlobGather::BlobGather(Engine& eng, const uint8_t* blob, size_t len) : eng_(eng), d_blob_(&eng, len ? len : 4), up_(new Up) {
  // hipMemcpyAsync out of pageable memory occupies the calling thread for the length of the copy: a helper thread takes it -- and the
  // lane's SIDE stream, because calls on one stream serialise inside the runtime: on the main stream the host's own small uploads
  // (selection tables, parameter packs) waited for the blob (65 536 AC17 items: 12.5 ms) instead of running beside it
  rhip_ctx* const cx = eng.side_ctx();
  void* const dst = d_blob_.ptr();
  up_->f = std::async(std::launch::async, [cx, dst, blob, len]() -> int32_t { return len ? rhip_upload_async(cx, dst, blob, len) : RHIP_OK; });
}
BlobGather::~BlobGather() {
  if (up_->f.valid()) up_->f.wait();
  delete up_;
}
int BlobGather::find(const void* key) const {
  for (const auto& k : keys_) if (k.first == key) return (int)k.second;
  return -1;
}
uint32_t BlobGather::add_shape(const void* key, std::vector<RecordLayout::Part> parts) {
  const uint32_t id = (uint32_t)shape_off_.size() - 1;
  for (const auto& pt : parts) { part_src_.push_back(pt.rec_off); part_dst_.push_back(pt.part_off); part_len_.push_back(pt.len); part_k_.push_back(pt.k); }
  shape_off_.push_back((uint32_t)part_src_.size());
  if (key) keys_.push_back({key, id});
  return id;
}
void BlobGather::run(const std::vector<void*>& dst, const std::vector<uint64_t>& dst_item_off) {
  const size_t m = rec_off_.size();
  eng_.check(up_->f.get(), "upload (records)");
  eng_.check(rhip_ctx_wait_for(eng_.ctx(), eng_.side_ctx()), "rhip_ctx_wait_for");          // the gather starts when the blob is there
  if (!m) return;
  if (dst_item_off.size() != dst.size() * m) throw RabeError("BlobGather: dst_item_off has the wrong size");
  pp_.reset(new ParamPack(eng_));
  ParamPack& pp = *pp_;
  const size_t h_rec = pp.add(rec_off_), h_shape = pp.add(item_shape_), h_soff = pp.add(shape_off_), h_ps = pp.add(part_src_), h_pd = pp.add(part_dst_),
               h_pl = pp.add(part_len_), h_pk = pp.add(part_k_), h_dst = pp.add(dst), h_doff = pp.add(dst_item_off);
  pp.upload();
  eng_.check(rhip_gather_parts(eng_.ctx(), m, d_blob_.as<uint8_t>(), pp.dev<uint64_t>(h_rec), pp.dev<uint32_t>(h_shape), pp.dev<uint32_t>(h_soff),
                               pp.dev<uint32_t>(h_ps), pp.dev<uint32_t>(h_pd), pp.dev<uint32_t>(h_pl), pp.dev<uint32_t>(h_pk), pp.dev<uint8_t*>(h_dst),
                               pp.dev<uint64_t>(h_doff)),
             "rhip_gather_parts");
}

void open_sealed_records(Engine& eng, size_t n, const std::vector<size_t>& live, const void* d_gt, const uint8_t* d_blob,
                         const std::vector<uint64_t>& sealed_off, const std::vector<uint32_t>& sealed_len, int32_t* status, uint8_t* pt_buf,
                         uint64_t* pt_off, std::vector<std::string>* errors) {
  const size_t m = live.size();
  std::vector<uint32_t> slot_len(n, 0);
  std::vector<uint8_t> is_live(n, 0);
  for (size_t j = 0; j < m; j++) {
    is_live[live[j]] = 1;
    if ((*errors)[live[j]].empty() && sealed_len[j] >= 28) slot_len[live[j]] = sealed_len[j] - 28;
  }
  pt_off[0] = 0;
  for (size_t i = 0; i < n; i++) pt_off[i + 1] = pt_off[i] + slot_len[i];
  for (size_t i = 0; i < n; i++) status[i] = -1;
  // the items that reach AES: live, no earlier error, a sealed part that can hold nonce and tag
  std::vector<uint32_t> gt_idx, len;
  std::vector<uint64_t> s_off, p_off;
  std::vector<size_t> item;
  for (size_t j = 0; j < m; j++) {
    const size_t i = live[j];
    if (!(*errors)[i].empty()) continue;
    if (sealed_len[j] < 28) { (*errors)[i] = "decryption error: aead::Error"; continue; }
    gt_idx.push_back((uint32_t)j);
    len.push_back(sealed_len[j] - 28);
    s_off.push_back(sealed_off[j]);
    p_off.push_back(pt_off[i]);
    item.push_back(i);
  }
  const size_t q = item.size();
  if (!q) return;
  std::vector<uint32_t> blk_off, seg_off;
  sym_shape(len, &blk_off, &seg_off);
  rhip_ctx* cx = eng.ctx();
  ParamPack pp(eng);
  const size_t h_idx = pp.add(gt_idx), h_len = pp.add(len), h_soff = pp.add(s_off), h_poff = pp.add(p_off), h_blk = pp.add(blk_off),
               h_seg = pp.add(seg_off);
  pp.upload();
  const size_t total = (size_t)pt_off[n];
  DBuf d_pt(&eng, total ? total : 4), d_ok(&eng, q * 4), d_ws(&eng, rhip_seal_workspace_bytes(q, seg_off[q]));
  eng.scrub_session_when_done();          // the decrypted Gt values and the AES key schedules do not outlive the call
  eng.check(rhip_open_batch(cx, q, (const rhip_gt*)d_gt, pp.dev<uint32_t>(h_idx), d_blob, pp.dev<uint64_t>(h_soff), d_pt.as<uint8_t>(),
                            pp.dev<uint64_t>(h_poff), pp.dev<uint32_t>(h_len), pp.dev<uint32_t>(h_blk), blk_off[q], pp.dev<uint32_t>(h_seg),
                            seg_off[q], d_ok.as<uint32_t>(), d_ws.ptr()),
            "rhip_open_batch");
  std::vector<uint32_t> ok(q);
  // plaintext offsets on the device ARE the caller's offsets (an item that failed earlier holds no bytes): one copy, no scatter
  if (total) eng.check(rhip_download_async(cx, pt_buf, d_pt.ptr(), total), "download (plaintexts)");
  eng.check(rhip_download(cx, ok.data(), d_ok.ptr(), q * 4), "download (tag verdicts)");
  for (size_t k = 0; k < q; k++) {
    if (ok[k]) status[item[k]] = 0;
    else (*errors)[item[k]] = "decryption error: aead::Error";          // its plaintext bytes are zeros (k_sym_ctr)
  }
}

}  // namespace schemes
}  // namespace rabe
