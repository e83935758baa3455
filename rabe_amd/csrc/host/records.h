// records.h -- the device-side tails of the packed entry points.
//
// A packed encrypt ends with (a) records written from their parts ON THE DEVICE (rhip_assemble_records: the per-policy template of
// literal bytes -- policy text, names, counts -- with the elements the kernels just produced dropped in) and (b) the KEM -> DEM step ON
// THE DEVICE (rhip_seal_batch: key = SHA3-256(bytes(Gt)), AES-256-GCM; src/utils/aes/mod.rs:10-55), then ONE copy of the finished blob
// into the caller's buffer.  A packed decrypt uploads the caller's blob as it is, gathers the elements out of it on the device
// (rhip_gather_parts), and opens the sealed plaintexts there (rhip_open_batch): the Gt never leaves HBM on either side.
#pragma once
#include <stdint.h>
#include <string.h>
#include <memory>
#include <string>
#include <vector>
#include "common.h"

namespace rabe { namespace schemes {

// the fixed part of one record shape (everything but the sealed plaintext and its u32 length): byte b is a literal or byte `o` of
// the item's part in source k (include/rabe_hip.h: rhip_assemble_records)
struct RecordLayout {
  std::vector<uint32_t> map;
  struct Part { uint32_t rec_off, len, k, part_off; };
  std::vector<Part> parts;
  void lit(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) map.push_back(0xFF000000u | b[i]); }
  void u8(uint8_t v) { map.push_back(0xFF000000u | v); }
  void u32(uint32_t v) { for (int i = 0; i < 4; i++) map.push_back(0xFF000000u | ((v >> (8 * i)) & 0xFFu)); }
  void str(const std::string& s) { u32((uint32_t)s.size()); lit(s.data(), s.size()); }
  void src(uint32_t k, uint32_t off, uint32_t len) {
    parts.push_back({(uint32_t)map.size(), len, k, off});
    for (uint32_t i = 0; i < len; i++) map.push_back((k << 24) | (off + i));
  }
  size_t bytes() const { return map.size(); }
};

// many small host arrays -> one staging area -> ONE upload; device pointers by handle afterwards
class ParamPack {
 public:
  explicit ParamPack(Engine& e) : eng_(e) {}
  size_t add(const void* p, size_t bytes);
  template <class T> size_t add(const std::vector<T>& v) { return add(v.data(), v.size() * sizeof(T)); }
  void upload();                                       // asynchronous on the engine's stream, from the lane's pinned slot 3
  template <class T> T* dev(size_t h) const { return (T*)((uint8_t*)d_.ptr() + h); }
 private:
  Engine& eng_;
  std::vector<uint8_t> host_;
  DBuf d_;
};

// prefix sums of the 16-byte blocks and 64-block GHASH segments of plaintexts of these lengths (rabe_hip.h, Level S)
void sym_shape(const std::vector<uint32_t>& len, std::vector<uint32_t>* blk_off, std::vector<uint32_t>* seg_off);

// Tail of a packed encrypt.  layouts[l]: the record shape of layout l; item i has layout item_layout[i], its record starts at
// out_off[i] and is layouts[l].bytes() + 4 + (plaintext length + 28) bytes long.  dev_src[k] / src_item_off[k * n + i]: where item i's
// part in source k starts (device).  d_msg: the n Gt messages (device).  Writes the n finished records to out_buf.
// With `defer`, the copy of the finished records into out_buf is handed to a helper thread on the lane's SIDE stream (it starts when the
// kernels queued so far are done) and the call returns at once: a caller that cuts its batch into parts launches the next part's group
// arithmetic beside it (PendingCopy::wait before the buffers are read; the parts of one call share one ArenaScope).
struct PendingCopy {
  DBuf d_out;
  // everything the queued assemble / seal kernels still read when emit_sealed_records returns early: the parameter pack, a large
  // plaintext block, the sealing workspace (arena blocks outlive the call anyway; hipMalloc'd fallbacks would not)
  DBuf d_pt, d_ws;
  std::shared_ptr<void> pp;
  std::shared_ptr<void> fut;          // std::future<int32_t>
  Engine* eng = nullptr;              // set while the helper thread may still read the lane's pinned block (Engine::pinned_hold)
  void wait(Engine& eng);
  ~PendingCopy();
};
void emit_sealed_records(Engine& eng, const std::vector<RecordLayout>& layouts, size_t n, const uint32_t* item_layout,
                         const std::vector<const void*>& dev_src, const std::vector<uint64_t>& src_item_off, const void* d_msg,
                         const uint8_t* nonces /*[n][12]*/, const uint8_t* pt_blob, const uint64_t* pt_off /*[n+1]*/,
                         const uint64_t* out_off /*[n+1]*/, uint8_t* out_buf, PendingCopy* defer = nullptr);

// The same without a sealed part (bulk key issuing): a record is exactly its layout's bytes.
void emit_plain_records(Engine& eng, const std::vector<RecordLayout>& layouts, size_t n, const uint32_t* item_layout,
                        const std::vector<const void*>& dev_src, const std::vector<uint64_t>& src_item_off, const uint64_t* out_off /*[n+1]*/,
                        uint8_t* out_buf);

// Head of a packed decrypt: the caller's blob goes to the device as it is -- the copy starts at construction, on a helper thread, so it
// runs beside the host's parsing of the records -- and the elements are gathered out of it there (rhip_gather_parts).  A SHAPE is the part
// list of one record skeleton; items whose records share policy text and row names share a shape (same skeleton, same relative offsets).
class BlobGather {
 public:
  BlobGather(Engine& eng, const uint8_t* blob, size_t len);
  ~BlobGather();
  int find(const void* key) const;                                    // -1: no shape registered under this key
  uint32_t add_shape(const void* key /*or nullptr: not shared*/, std::vector<RecordLayout::Part> parts);
  void item(uint64_t rec_off, uint32_t shape) { rec_off_.push_back(rec_off); item_shape_.push_back(shape); }
  // dst[k] + dst_item_off[k * m + j] is where live item j's parts of destination k start; waits for the blob, launches the gather
  void run(const std::vector<void*>& dst, const std::vector<uint64_t>& dst_item_off);
  const uint8_t* dev_blob() const { return d_blob_.as<uint8_t>(); }
 private:
  Engine& eng_;
  DBuf d_blob_;
  struct Up;
  Up* up_;
  std::vector<std::pair<const void*, uint32_t>> keys_;
  std::vector<uint32_t> shape_off_{0}, part_src_, part_dst_, part_len_, part_k_, item_shape_;
  std::vector<uint64_t> rec_off_;
  std::unique_ptr<ParamPack> pp_;          // the gather's tables: alive as long as the kernel may read them
};

// Tail of a packed decrypt.  Live item j (j < m) = item live[j] of the call; its sealed bytes are blob[sealed_off[j] .. + sealed_len[j])
// of the DEVICE copy of the caller's blob, its Gt is d_gt[j].  Fills status / pt_buf / pt_off / errors for all n items as
// ac17::cp_decrypt_packed documents (an item whose errors[i] is already set stays failed).
void open_sealed_records(Engine& eng, size_t n, const std::vector<size_t>& live, const void* d_gt, const uint8_t* d_blob,
                         const std::vector<uint64_t>& sealed_off, const std::vector<uint32_t>& sealed_len, int32_t* status, uint8_t* pt_buf,
                         uint64_t* pt_off, std::vector<std::string>* errors);

// A verdict that arrives after open_sealed_records ran (the G2 membership a decrypt's own Miller loops establish, common.h: WalkedG2 --
// reading it earlier would make the host wait for the pairings before it prepares the open): the item fails like an item whose tag did not
// verify -- status -1, its plaintext slot zeroed -- with the decoder's error, which comes first in the reference's order of events.
inline void retract_item(size_t i, const char* msg, int32_t* status, uint8_t* pt_buf, const uint64_t* pt_off, std::vector<std::string>* errors) {
  if (status[i] == 0 && pt_off[i + 1] > pt_off[i]) memset(pt_buf + pt_off[i], 0, (size_t)(pt_off[i + 1] - pt_off[i]));
  status[i] = -1;
  (*errors)[i] = msg;
}

}}  // namespace rabe::schemes
