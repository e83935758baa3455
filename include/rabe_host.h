/*
 * rabe_host.h -- C interface of the C++ host layer (rabe::schemes::* mirror, rabe_amd/csrc/host/).
 *
 * Same convention as the reference's own C FFI (src/ffi/bsw.rs:22-163): opaque object pointers created by
 * one call and destroyed by a paired free, int32 status, out-buffers returned through out-pointers.
 * Status: 0 ok; 1 = `None` (the reference returns Option::None: bsw::keygen :132-134, aw11::authgen :127-129);
 * -1 = RabeError (text via rabe_host_last_error); -2 = a condition on which the reference panics
 * (malformed policy shapes, unwrap on None).  Group arithmetic always runs on the HIP engine.
 */
#ifndef RABE_HOST_H
#define RABE_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rabe_host rabe_host;

enum {   /* object kinds for rabe_obj_free / rabe_obj_serialize / rabe_obj_deserialize */
  RABE_AC17_PK = 1, RABE_AC17_MSK = 2, RABE_AC17_CP_SK = 3, RABE_AC17_CP_CT = 4, RABE_AC17_KP_SK = 5, RABE_AC17_KP_CT = 6,
  RABE_BSW_PK = 10, RABE_BSW_MSK = 11, RABE_BSW_SK = 12, RABE_BSW_CT = 13,
  RABE_LSW_PK = 20, RABE_LSW_MSK = 21, RABE_LSW_SK = 22, RABE_LSW_CT = 23,
  RABE_AW11_GK = 30, RABE_AW11_PK = 31, RABE_AW11_MSK = 32, RABE_AW11_SK = 33, RABE_AW11_CT = 34,
  RABE_GHW11_PK = 40, RABE_GHW11_MSK = 41, RABE_GHW11_SK = 42, RABE_GHW11_TK = 43, RABE_GHW11_RK = 44, RABE_GHW11_CT = 45, RABE_GHW11_TCT = 46,
  /* bdabe / mke08: public key, master key, secret AUTHORITY key, user key (with its attribute keys), public attribute key, ciphertext */
  RABE_BDABE_PK = 50, RABE_BDABE_MSK = 51, RABE_BDABE_SKA = 52, RABE_BDABE_UK = 53, RABE_BDABE_PKA = 54, RABE_BDABE_CT = 55,
  RABE_MKE08_PK = 60, RABE_MKE08_MSK = 61, RABE_MKE08_SKA = 62, RABE_MKE08_UK = 63, RABE_MKE08_PKA = 64, RABE_MKE08_CT = 65
};
enum { RABE_JSON_POLICY = 0, RABE_HUMAN_POLICY = 1 };   /* PolicyLanguage, src/utils/policy/pest/mod.rs:18-23 */

/* ABI revision of this header.  An argument list that changes keeps its symbol name only together with a bump of this number (revision 3
 * inserted ct_len and flags into the packed decrypts): a caller passes the revision it was COMPILED against -- rabe_host_open is the macro
 * for that -- and a library of another revision refuses with -3 instead of reading shifted arguments.  rabe_host_create does not check. */
#define RABE_HOST_ABI_VERSION 5
int32_t rabe_host_abi_version(void);
int32_t rabe_host_create_checked(int32_t abi_version, int32_t device, rabe_host** out);
#define rabe_host_open(device, out) rabe_host_create_checked(RABE_HOST_ABI_VERSION, (device), (out))
int32_t rabe_host_create(int32_t device, rabe_host** out);
/* ---- device group: ONE host over several GPUs of a node ----------------------------------------------------------------------
 * Every encrypt / keygen / decrypt call is independent (the reference's schemes are pure functions of their arguments and fresh
 * randomness), so a batch shards by item: the packed entry points of the four BASELINE schemes -- rabe_ac17_cp_{encrypt,decrypt}_packed,
 * rabe_bsw_{encrypt,decrypt}_packed, rabe_lsw_{keygen,decrypt}_packed, rabe_aw11_{encrypt,decrypt}_packed -- cut their n_items into one
 * contiguous block per device (sizes differ by at most one, blocks in device order), run every block on its own host thread and engine
 * (each device builds its replica of a key's window tables / prepared lines on first use and keeps it), and the records / plaintexts /
 * status entries land in the caller's buffers exactly where the single-device call puts them.  Randomness is drawn block after block
 * from the host's one source, so on a tape (rabe_host_set_tape) the bytes do not depend on the number of devices.  No data-path
 * collective: the only exchange is the results arriving in host memory.  A device may be listed more than once (two engines, i.e. two
 * streams with their own workspaces, on one GPU) -- how a one-GPU box tests the split.  Every other entry point runs on devices[0].
 * Status: as rabe_host_create_checked. */
int32_t rabe_host_open_group_checked(int32_t abi_version, size_t n_devices, const int32_t* devices, rabe_host** out);
#define rabe_host_open_group(n_devices, devices, out) rabe_host_open_group_checked(RABE_HOST_ABI_VERSION, (n_devices), (devices), (out))
int32_t rabe_host_group_size(rabe_host* h);          /* engines of the host: 1 for rabe_host_open */
void rabe_host_destroy(rabe_host* h);
/* ---- submission queue: one call at a time, many callers ------------------------------------------------------------------------
 * The reference's API is one call per ciphertext (src/schemes/ac17/mod.rs:274-279, :385-388; rabe-console/src/mod.rs:1098, 1310) and one
 * call is one small launch set: milliseconds of latency for microseconds of chip time.  With the queue ON, the one-call encrypt / decrypt
 * entry points of ac17 (CP), bsw, lsw and aw11 below may be called from MANY THREADS on one rabe_host: calls that arrive while a batch is
 * running are collected and run as one packed batch (group commit: the batch grows with the load; window_us > 0 additionally holds every
 * batch open for that long).  Randomness is drawn in arrival order: one caller on a fixed tape gets the bytes of the unqueued path.
 * Errors of a threaded caller: rabe_host_last_error(NULL) (per thread).  Every other entry point stays single-owner.
 * The *_submit forms queue a call and return at once; rabe_ticket_wait blocks until the result is there (and may run the queued batch
 * itself -- there is no service thread), hands out the ciphertext object (encrypt) or the plaintext (decrypt, rabe_bytes_free) and frees
 * the ticket.  Key objects and, for decrypts, the ciphertext object must stay alive until the wait returns; plaintexts are copied. */
typedef struct rabe_ticket rabe_ticket;
int32_t rabe_host_set_coalescing(rabe_host* h, int32_t on, uint32_t window_us);
/* what the queue has done so far: batches run, requests in them, groups (= packed calls), requests that ran singly, microseconds spent
 * inside batches, largest batch */
int32_t rabe_host_queue_stats(rabe_host* h, uint64_t out[6]);
int32_t rabe_ac17_cp_encrypt_submit(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* pt, size_t len, rabe_ticket** ticket);
int32_t rabe_ac17_cp_decrypt_submit(rabe_host* h, const void* sk, const void* ct, rabe_ticket** ticket);
int32_t rabe_bsw_encrypt_submit(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* pt, size_t len, rabe_ticket** ticket);
int32_t rabe_bsw_decrypt_submit(rabe_host* h, const void* sk, const void* ct, rabe_ticket** ticket);
int32_t rabe_lsw_encrypt_submit(rabe_host* h, const void* pk, const char* const* attributes, size_t n, const uint8_t* pt, size_t len, rabe_ticket** ticket);
int32_t rabe_lsw_decrypt_submit(rabe_host* h, const void* sk, const void* ct, rabe_ticket** ticket);
int32_t rabe_aw11_encrypt_submit(rabe_host* h, const void* gk, const void* const* pks, size_t n_pks, const char* policy, int32_t language,
                                 const uint8_t* data, size_t len, rabe_ticket** ticket);
int32_t rabe_aw11_decrypt_submit(rabe_host* h, const void* gk, const void* sk, const void* ct, rabe_ticket** ticket);
int32_t rabe_ticket_wait(rabe_host* h, rabe_ticket* ticket, void** obj /*encrypt*/, uint8_t** out, size_t* len /*decrypt*/);
/* measurement helper: `threads` native host threads issue rabe_ac17_cp_encrypt + rabe_ac17_cp_decrypt one call at a time on this host
 * (depth 1), or keep `depth` submitted calls in flight each, for `seconds`; *ops = verified encrypt+decrypt cycles, *bad = the others */
int32_t rabe_bench_ac17_threads(rabe_host* h, const void* pk, const void* sk, const char* const* policies, size_t n_policies, int32_t language,
                                uint32_t threads, uint32_t depth, double seconds, uint64_t* ops, uint64_t* bad);
const char* rabe_host_last_error(rabe_host* h);          /* h may be NULL: error of the last host-free call on this thread */
/* Explicit randomness: the next calls draw their Fr values from this tape, in the reference's
 * draw order, instead of the OS generator.  n = 0 switches back to OS randomness. */
int32_t rabe_host_set_tape(rabe_host* h, const uint8_t* fr_le32, size_t n);

/* G*Fr / Gt^Fr elements that share one base are served from a cached fixed-base window table once n of them have been seen,
 * in one call or accumulated over calls (default 1024); the results do not depend on it.  Tests set 1 / SIZE_MAX to force
 * either path. */
int32_t rabe_host_set_fixed_base_min(rabe_host* h, size_t n);

void rabe_obj_free(int32_t kind, void* obj);
/* canonical byte form of a key / ciphertext (layout: rabe_amd/csrc/host/host_abi.cpp); free with rabe_bytes_free */
int32_t rabe_obj_serialize(int32_t kind, const void* obj, uint8_t** out, size_t* len);
int32_t rabe_obj_deserialize(int32_t kind, const uint8_t* data, size_t len, void** obj);
/* rabe_obj_deserialize checks structure and ranges (lengths of the fixed-size AC17 vectors, scalars < r, coordinates < p);
 * the _checked form additionally establishes group membership of every element on the GPU (G1 on the curve, G2 in the r-torsion
 * of the twist, Gt in the order-r subgroup): use it for keys / ciphertexts that arrive from outside */
int32_t rabe_obj_deserialize_checked(rabe_host* h, int32_t kind, const uint8_t* data, size_t len, void** obj);
void rabe_bytes_free(void* p);

/* ---- ac17 (src/schemes/ac17/mod.rs:141-430) */
int32_t rabe_ac17_setup(rabe_host* h, void** pk, void** msk);
int32_t rabe_ac17_cp_keygen(rabe_host* h, const void* msk, const char* const* attributes, size_t n, void** sk);
int32_t rabe_ac17_cp_encrypt(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* plaintext, size_t len, void** ct);
int32_t rabe_ac17_cp_decrypt(rabe_host* h, const void* sk, const void* ct, uint8_t** plaintext, size_t* len);
int32_t rabe_ac17_cp_decrypt_gt(rabe_host* h, const void* sk, const void* ct, uint8_t out_gt[384]);
/* n independent cp_encrypt calls in one launch set; cts receives n object pointers */
int32_t rabe_ac17_cp_encrypt_batch(rabe_host* h, const void* pk, size_t n, const char* const* policies, int32_t language,
                                   const uint8_t* const* plaintexts, const size_t* lens, void** cts);
/* n independent cp_decrypt calls; status[i] = 0 ok / -1 error; plaintexts[i] malloc'd (rabe_bytes_free) */
int32_t rabe_ac17_cp_decrypt_batch(rabe_host* h, size_t n, const void* const* sks, const void* const* cts, int32_t* status,
                                   uint8_t** plaintexts, size_t* lens);
/* The same two batches with packed input and output -- no per-item objects on either side of the boundary, every buffer
 * caller-allocated (and reusable across calls).  Ciphertexts are ONE blob of canonical records (the byte form rabe_obj_serialize
 * gives an Ac17CpCiphertext) delimited by an offset array of n_items + 1 entries; plaintexts likewise.
 *   encrypt: item i uses policies[item_policy[i]] (distinct policy texts are parsed once, and cached across calls).  ct_off is
 *            always filled; return 1 (nothing done, no randomness drawn) when ct_cap < ct_off[n_items], the size the records need.
 *   decrypt: status[i] = 0 / -1 per item (a key that does not satisfy a policy, a malformed record or an authentication failure
 *            fails that item only; its plaintext is empty).  pt_cap >= the sum of the sizes of the records whose bounds are valid always suffices (<= ct_len unless records overlap);
 *            return 1 when it is smaller. */
int32_t rabe_ac17_cp_encrypt_packed(rabe_host* h, const void* pk, const char* const* policies, size_t n_policies, int32_t language, size_t n_items,
                                    const uint32_t* item_policy /*[n_items]*/, const uint8_t* pt_blob, const uint64_t* pt_off /*[n_items+1]*/,
                                    uint8_t* ct_buf, size_t ct_cap, uint64_t* ct_off /*[n_items+1]*/);
/* decrypt reads UNTRUSTED bytes: ct_len is the size of ct_blob, and ct_off must be monotone inside it (an item whose bounds are not
 * fails alone with status -1; nothing outside [0, ct_len) is read).  Unless RABE_PACKED_TRUSTED is set in `flags`, every decoded element
 * goes through one batched membership pass on the GPU -- coordinates < p, rows on the G1 curve, c_0 in the r-torsion of the twist, c_p in
 * the order-r subgroup of Fq12 (rabe-bn: FieldError::NotMember) -- and a non-member fails its item only.  Set the flag only for
 * ciphertexts this process produced itself. */
#define RABE_PACKED_TRUSTED 1u
/* Environment of the packed entry points (read per call; defaults are what the measurements in DESIGN.md section 7 favour):
 *   RABE_PACKED_CHUNK   cut a packed call into chunks of at least this many items that run RABE_PACKED_LANES (default 2) at a time, each on
 *                       its own engine lane; off by default -- measured no faster than one launch set once the unchunked call was tuned
 *   RABE_MEMBER_INLINE  run the decoding checks on the main stream before the kernels instead of beside them (A/B)
 *   RABE_G_WINDOW       signed window width of AC17's g table in this layer (default 20 = +0.44 GB per public key; 16 = none extra)
 *   RABE_NO_ARENA       every device buffer of a call its own hipMalloc again (diagnostics)
 *   RABE_ARENA_MAX_GB   cap of the grow-only device block a lane keeps between packed calls (default 16)
 *   RABE_HOST_TIMING    stage timings on stderr
 * Results never depend on any of them. */
/* A key authority issuing keys in bulk: n_items calls of ac17::cp_keygen (src/schemes/ac17/mod.rs:191-264) under one master key.  The
 * attribute lists are given once (n_sets lists, list s = the next counts[s] entries of `attributes`), item i gets list item_set[i].
 * sk_buf receives the Ac17CpSecretKey records (sk_off: n_items + 1 offsets, always filled; returns 1 when sk_cap is too small, before any
 * randomness is drawn).  Items that share a list are one launch of the Level B keygen kernels; the master key's window tables are kept. */
int32_t rabe_ac17_cp_keygen_packed(rabe_host* h, const void* msk, const char* const* attributes, const size_t* counts, size_t n_sets, size_t n_items,
                                   const uint32_t* item_set /*[n_items]*/, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off /*[n_items+1]*/);
int32_t rabe_ac17_cp_decrypt_packed(rabe_host* h, const void* sk, size_t n_items, const uint8_t* ct_blob, size_t ct_len,
                                    const uint64_t* ct_off /*[n_items+1]*/, uint32_t flags, int32_t* status /*[n_items]*/, uint8_t* pt_buf,
                                    size_t pt_cap, uint64_t* pt_off /*[n_items+1]*/);

/* KP-ABE variant (src/schemes/ac17/mod.rs:439-675) */
int32_t rabe_ac17_kp_keygen(rabe_host* h, const void* msk, const char* policy, int32_t language, void** sk);
/* n_items calls of ac17::kp_keygen (src/schemes/ac17/mod.rs:439-547) under one master key, item i under policies[item_policy[i]]: the triple
 * loop over rows x columns x (l, t) (:479-538) is Fr work on all host cores, the 3 rows + 3 elements per key are ONE fixed-base launch set,
 * the records (Ac17KpSecretKey: policy, k_0, rows, an empty k_p) are written on the device.  Buffers / return value as
 * rabe_ac17_cp_keygen_packed (1 = sk_cap too small, sk_off[n_items] says what is needed). */
int32_t rabe_ac17_kp_keygen_packed(rabe_host* h, const void* msk, const char* const* policies, size_t n_policies, int32_t language, size_t n_items,
                                   const uint32_t* item_policy /*[n_items]*/, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off /*[n_items+1]*/);
int32_t rabe_ac17_kp_encrypt(rabe_host* h, const void* pk, const char* const* attributes, size_t n, const uint8_t* data, size_t len, void** ct);
int32_t rabe_ac17_kp_decrypt(rabe_host* h, const void* sk, const void* ct, uint8_t** plaintext, size_t* len);
int32_t rabe_ac17_kp_decrypt_gt(rabe_host* h, const void* sk, const void* ct, uint8_t out_gt[384]);
/* The KP pair packed (conventions of rabe_ac17_cp_{encrypt,decrypt}_packed; records = Ac17KpCiphertext): n_items encrypts, item i under the
 * attribute list item_set[i] of the n_sets lists given once (list s = the next counts[s] entries of `attributes`); n_items decrypts
 * with ONE key (whose policy is checked against every ciphertext's attribute list).  Same kernels as the CP pair
 * (src/schemes/ac17/mod.rs:556-675). */
int32_t rabe_ac17_kp_encrypt_packed(rabe_host* h, const void* pk, const char* const* attributes, const size_t* counts, size_t n_sets, size_t n_items,
                                    const uint32_t* item_set /*[n_items]*/, const uint8_t* pt_blob, const uint64_t* pt_off /*[n_items+1]*/,
                                    uint8_t* ct_buf, size_t ct_cap, uint64_t* ct_off /*[n_items+1]*/);
int32_t rabe_ac17_kp_decrypt_packed(rabe_host* h, const void* sk, size_t n_items, const uint8_t* ct_blob, size_t ct_len,
                                    const uint64_t* ct_off /*[n_items+1]*/, uint32_t flags, int32_t* status /*[n_items]*/, uint8_t* pt_buf, size_t pt_cap,
                                    uint64_t* pt_off /*[n_items+1]*/);
/* n independent kp_encrypt / kp_decrypt calls in one launch set; item i's attributes are the next counts[i] entries of `attributes` */
int32_t rabe_ac17_kp_encrypt_batch(rabe_host* h, const void* pk, size_t n, const char* const* attributes, const size_t* counts,
                                   const uint8_t* const* datas, const size_t* lens, void** cts);
int32_t rabe_ac17_kp_decrypt_batch(rabe_host* h, size_t n, const void* const* sks, const void* const* cts, int32_t* status,
                                   uint8_t** plaintexts, size_t* lens);

/* ---- bsw (src/schemes/bsw/mod.rs:92-318) */
int32_t rabe_bsw_setup(rabe_host* h, void** pk, void** msk);
int32_t rabe_bsw_keygen(rabe_host* h, const void* pk, const void* msk, const char* const* attributes, size_t n, void** sk);
int32_t rabe_bsw_delegate(rabe_host* h, const void* pk, const void* sk, const char* const* subset, size_t n, void** out_sk);
int32_t rabe_bsw_encrypt(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* plaintext, size_t len, void** ct);
int32_t rabe_bsw_decrypt(rabe_host* h, const void* sk, const void* ct, uint8_t** plaintext, size_t* len);
int32_t rabe_bsw_decrypt_gt(rabe_host* h, const void* sk, const void* ct, uint8_t out_gt[384]);
/* n independent encrypt / decrypt calls, the group work of all items in one launch per operation type
 * (BASELINE config 3); conventions as rabe_ac17_cp_{encrypt,decrypt}_batch */
int32_t rabe_bsw_encrypt_batch(rabe_host* h, const void* pk, size_t n, const char* const* policies, int32_t language,
                               const uint8_t* const* plaintexts, const size_t* lens, void** cts);
int32_t rabe_bsw_decrypt_batch(rabe_host* h, size_t n, const void* const* sks, const void* const* cts, int32_t* status,
                               uint8_t** plaintexts, size_t* lens);
/* The same two batches packed (conventions of rabe_ac17_cp_{encrypt,decrypt}_packed: one blob of canonical CpAbeCiphertext records +
 * n_items + 1 offsets per side, caller-allocated buffers, per-item status, ct_len / RABE_PACKED_TRUSTED for untrusted input).  These
 * feed the device-resident path (rhip_bsw_{encrypt,decrypt}_batch): share generation, fixed-base multiplications, the folded
 * pairing product and one final exponentiation per item all happen on HBM-resident arrays (src/schemes/bsw/mod.rs:217-318). */
/* Bulk key issuing for bsw (conventions of rabe_ac17_cp_keygen_packed): n_items calls of bsw::keygen (src/schemes/bsw/mod.rs:125-152) under one
 * master key, records = CpAbeSecretKey.  Every key element is a fixed-base multiple of a generator (d = g2_alpha/beta + g2*(r/beta)):
 * three window-table launches for the whole batch.  An empty attribute list (for which bsw::keygen returns None) fails the call. */
int32_t rabe_bsw_keygen_packed(rabe_host* h, const void* pk, const void* msk, const char* const* attributes, const size_t* counts, size_t n_sets,
                               size_t n_items, const uint32_t* item_set /*[n_items]*/, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off /*[n_items+1]*/);
/* n_items calls of bsw::delegate (src/schemes/bsw/mod.rs:162-206) on ONE key: item i delegates `sk` to the attribute list item_subset[i] of the
 * n_subsets lists given once (flattened in `attributes`, `counts` each).  d' = d + f * r, rows D_j + (g1 * r_j, g2 * (h(j) r_j + r)): window-table
 * launches and batched additions for the whole batch, records (CpAbeSecretKey) written on the device.  A subset that is empty or not contained
 * in the key (delegate returns None) fails the call; buffers / return value as rabe_bsw_keygen_packed. */
int32_t rabe_bsw_delegate_packed(rabe_host* h, const void* pk, const void* sk, const char* const* attributes, const size_t* counts, size_t n_subsets,
                                 size_t n_items, const uint32_t* item_subset /*[n_items]*/, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off /*[n_items+1]*/);
int32_t rabe_bsw_encrypt_packed(rabe_host* h, const void* pk, const char* const* policies, size_t n_policies, int32_t language, size_t n_items,
                                const uint32_t* item_policy /*[n_items]*/, const uint8_t* pt_blob, const uint64_t* pt_off /*[n_items+1]*/,
                                uint8_t* ct_buf, size_t ct_cap, uint64_t* ct_off /*[n_items+1]*/);
int32_t rabe_bsw_decrypt_packed(rabe_host* h, const void* sk, size_t n_items, const uint8_t* ct_blob, size_t ct_len,
                                const uint64_t* ct_off /*[n_items+1]*/, uint32_t flags, int32_t* status /*[n_items]*/, uint8_t* pt_buf, size_t pt_cap,
                                uint64_t* pt_off /*[n_items+1]*/);

/* ---- lsw (src/schemes/lsw/mod.rs:86-290) */
int32_t rabe_lsw_setup(rabe_host* h, void** pk, void** msk);
int32_t rabe_lsw_keygen(rabe_host* h, const void* pk, const void* msk, const char* policy, int32_t language, void** sk);
int32_t rabe_lsw_encrypt(rabe_host* h, const void* pk, const char* const* attributes, size_t n, const uint8_t* plaintext, size_t len, void** ct);
int32_t rabe_lsw_decrypt(rabe_host* h, const void* sk, const void* ct, uint8_t** plaintext, size_t* len);
int32_t rabe_lsw_decrypt_gt(rabe_host* h, const void* sk, const void* ct, uint8_t out_gt[384]);
/* BASELINE config 4: n keygen / decrypt calls in one launch set */
int32_t rabe_lsw_keygen_batch(rabe_host* h, const void* pk, const void* msk, size_t n, const char* const* policies, int32_t language, void** sks);
int32_t rabe_lsw_decrypt_batch(rabe_host* h, size_t n, const void* const* sks, const void* const* cts, int32_t* status,
                               uint8_t** plaintexts, size_t* lens);
/* Packed forms over the device-resident path (rhip_lsw_{keygen,decrypt}_batch; src/schemes/lsw/mod.rs:121-290): keygen writes n
 * KpAbeSecretKey records (item i's policy = policies[item_policy[i]]) into one caller-allocated blob; decrypt takes n such records
 * and ONE ciphertext object (BASELINE config 4) and returns n plaintexts.  Size / status / ct_len / RABE_PACKED_TRUSTED conventions as
 * rabe_ac17_cp_{encrypt,decrypt}_packed.  keygen handles negative leaves ("!x", lsw/mod.rs:137-146) on the device as well; a decrypt
 * whose pruned selection reaches a negative attribute fails that item -- the reference's own decrypt has no negative branch (a TODO,
 * :265-278), and rabe_lsw_decrypt reproduces what it does instead. */
/* n_items calls of lsw::encrypt (src/schemes/lsw/mod.rs:180-219), item i under the attribute list item_set[i] of the n_sets lists given once;
 * records = KpAbeCiphertext (conventions of rabe_ac17_cp_encrypt_packed).  Every element is a fixed-base multiple of a public-key element
 * (the reference's sx[0] index quirk :197-200 included): window-table launches for the whole batch. */
int32_t rabe_lsw_encrypt_packed(rabe_host* h, const void* pk, const char* const* attributes, const size_t* counts, size_t n_sets, size_t n_items,
                                const uint32_t* item_set /*[n_items]*/, const uint8_t* pt_blob, const uint64_t* pt_off /*[n_items+1]*/,
                                uint8_t* ct_buf, size_t ct_cap, uint64_t* ct_off /*[n_items+1]*/);
int32_t rabe_lsw_keygen_packed(rabe_host* h, const void* pk, const void* msk, const char* const* policies, size_t n_policies, int32_t language,
                               size_t n_items, const uint32_t* item_policy /*[n_items]*/, uint8_t* sk_buf, size_t sk_cap, uint64_t* sk_off /*[n_items+1]*/);
int32_t rabe_lsw_decrypt_packed(rabe_host* h, const void* ct, size_t n_items, const uint8_t* sk_blob, size_t sk_len,
                                const uint64_t* sk_off /*[n_items+1]*/, uint32_t flags, int32_t* status /*[n_items]*/, uint8_t* pt_buf, size_t pt_cap,
                                uint64_t* pt_off /*[n_items+1]*/);

/* ---- aw11 (src/schemes/aw11/mod.rs:100-390) */
int32_t rabe_aw11_setup(rabe_host* h, void** gk);
int32_t rabe_aw11_authgen(rabe_host* h, const void* gk, const char* const* attributes, size_t n, void** pk, void** msk);
int32_t rabe_aw11_keygen(rabe_host* h, const void* gk, const void* msk, const char* name, const char* const* attributes, size_t n, void** sk);
int32_t rabe_aw11_add_to_attribute(rabe_host* h, const void* gk, const void* msk, const char* attribute, void* sk);
int32_t rabe_aw11_encrypt(rabe_host* h, const void* gk, const void* const* pks, size_t n_pks, const char* policy, int32_t language,
                          const uint8_t* data, size_t len, void** ct);
int32_t rabe_aw11_decrypt(rabe_host* h, const void* gk, const void* sk, const void* ct, uint8_t** plaintext, size_t* len);
int32_t rabe_aw11_decrypt_gt(rabe_host* h, const void* gk, const void* sk, const void* ct, uint8_t out_gt[384]);
/* BASELINE config 5: n encrypt / decrypt calls in one launch set (all items encrypt under the same authority keys) */
int32_t rabe_aw11_encrypt_batch(rabe_host* h, const void* gk, const void* const* pks, size_t n_pks, size_t n, const char* const* policies,
                                int32_t language, const uint8_t* const* datas, const size_t* lens, void** cts);
int32_t rabe_aw11_decrypt_batch(rabe_host* h, const void* gk, size_t n, const void* const* sks, const void* const* cts, int32_t* status,
                                uint8_t** plaintexts, size_t* lens);
/* Packed forms over the device-resident path (rhip_aw11_{encrypt,decrypt}_batch; src/schemes/aw11/mod.rs:241-366), conventions as
 * rabe_ac17_cp_{encrypt,decrypt}_packed.  Every policy leaf must name an attribute of one of the authority keys (the reference drops
 * other rows silently, :269-271; rabe_aw11_encrypt reproduces that). */
/* Bulk key issuing by one authority (conventions of rabe_ac17_cp_keygen_packed): n_items calls of aw11::keygen (src/schemes/aw11/mod.rs:165-231),
 * user gids[i] gets the attribute list item_set[i]; records = Aw11SecretKey.  No randomness: K_x = g1 * (alpha_x + h(gid) y_x), one
 * window-table launch for the whole batch. */
int32_t rabe_aw11_keygen_packed(rabe_host* h, const void* gk, const void* msk, const char* const* gids /*[n_items]*/, const char* const* attributes,
                                const size_t* counts, size_t n_sets, size_t n_items, const uint32_t* item_set /*[n_items]*/, uint8_t* sk_buf, size_t sk_cap,
                                uint64_t* sk_off /*[n_items+1]*/);
int32_t rabe_aw11_encrypt_packed(rabe_host* h, const void* gk, const void* const* pks, size_t n_pks, const char* const* policies, size_t n_policies,
                                 int32_t language, size_t n_items, const uint32_t* item_policy /*[n_items]*/, const uint8_t* pt_blob,
                                 const uint64_t* pt_off /*[n_items+1]*/, uint8_t* ct_buf, size_t ct_cap, uint64_t* ct_off /*[n_items+1]*/);
int32_t rabe_aw11_decrypt_packed(rabe_host* h, const void* gk, const void* sk, size_t n_items, const uint8_t* ct_blob, size_t ct_len,
                                 const uint64_t* ct_off /*[n_items+1]*/, uint32_t flags, int32_t* status /*[n_items]*/, uint8_t* pt_buf, size_t pt_cap,
                                 uint64_t* pt_off /*[n_items+1]*/);

/* ---- ghw11, CP-ABE with outsourced decryption (src/schemes/ghw11/mod.rs:92-305): `transform` is the server's part
 * (m + 2 pairings per ciphertext), `decrypt_out` the client's (one Gt power) */
int32_t rabe_ghw11_setup(rabe_host* h, void** pk, void** msk);
int32_t rabe_ghw11_keygen(rabe_host* h, const void* pk, const void* msk, const char* const* attributes, size_t n, void** sk);
int32_t rabe_ghw11_tkgen(rabe_host* h, const void* sk, void** tk, void** rk);
int32_t rabe_ghw11_encrypt(rabe_host* h, const void* pk, const char* policy, int32_t language, const uint8_t* plaintext, size_t len, void** ct);
int32_t rabe_ghw11_transform(rabe_host* h, const void* ct, const void* tk, void** tct);
/* n independent transforms in one launch set; status[i] = 0 ok / -1 (tk i does not satisfy ct i; tcts[i] = NULL) */
int32_t rabe_ghw11_transform_batch(rabe_host* h, size_t n, const void* const* cts, const void* const* tks, int32_t* status, void** tcts);
/* The outsourced half as a service: n_items Ghw11Ciphertext records in one blob (+ n_items + 1 offsets, ct_len; bounds and --
 * unless RABE_PACKED_TRUSTED -- group membership of every decoded element are checked, an item fails alone with status -1) transformed under
 * ONE transform key.  tct_buf + 768 i receives item i's Ghw11TransformCiphertext record (c | t; zeros where status[i] = -1); returns 1 when
 * tct_cap < 768 n_items.  Device-resident: every G2 argument is the key's, so all m + 2 Miller loops of an item replay the key's prepared
 * lines (kept across calls) and the rows that share l_z collapse into one pairing of a multi-scalar sum (ghw11/mod.rs:227-295). */
int32_t rabe_ghw11_transform_packed(rabe_host* h, const void* tk, size_t n_items, const uint8_t* ct_blob, size_t ct_len,
                                    const uint64_t* ct_off /*[n_items+1]*/, uint32_t flags, int32_t* status /*[n_items]*/, uint8_t* tct_buf, size_t tct_cap);
/* `data` of the reference's decrypt_out is the ciphertext's data field: pass the ciphertext object */
int32_t rabe_ghw11_decrypt_out(rabe_host* h, const void* tct, const void* rk, const void* ct, uint8_t** plaintext, size_t* len);
int32_t rabe_ghw11_decrypt_out_gt(rabe_host* h, const void* tct, const void* rk, uint8_t out_gt[384]);

/* ---- bdabe (src/schemes/bdabe/mod.rs:149-399) and mke08 (src/schemes/mke08/mod.rs:130-380): DNF policies, attributes named
 * "authority::attribute".  `request_attribute_sk` / `request_authority_sk` APPEND the new secret attribute key to the user key
 * (the reference returns it and its callers push it onto `sk.sk_a`: bdabe/mod.rs:21, mke08 tests).  decrypt: every item's
 * pairing factors on one accumulator, one final exponentiation (rhip_pairing_jobs). */
int32_t rabe_bdabe_setup(rabe_host* h, void** pk, void** msk);
int32_t rabe_bdabe_authgen(rabe_host* h, const void* pk, const void* msk, const char* name, void** ska);
int32_t rabe_bdabe_keygen(rabe_host* h, const void* pk, const void* ska, const char* name, void** uk);
int32_t rabe_bdabe_request_attribute_pk(rabe_host* h, const void* pk, const void* ska, const char* attribute, void** pka);
int32_t rabe_bdabe_request_attribute_sk(rabe_host* h, void* uk, const void* ska, const char* attribute);
int32_t rabe_bdabe_encrypt(rabe_host* h, const void* pk, const void* const* attr_pks, size_t n_pks, const char* policy, int32_t language,
                           const uint8_t* plaintext, size_t len, void** ct);
int32_t rabe_bdabe_decrypt(rabe_host* h, const void* uk, const void* ct, uint8_t** plaintext, size_t* len);
int32_t rabe_bdabe_decrypt_gt(rabe_host* h, const void* uk, const void* ct, uint8_t out_gt[384]);
int32_t rabe_bdabe_decrypt_batch(rabe_host* h, size_t n, const void* const* uks, const void* const* cts, int32_t* status, uint8_t** plaintexts,
                                 size_t* lens);
/* bdabe::decrypt (bdabe/mod.rs:359-399) / mke08::decrypt (mke08/mod.rs:343-380) of n_items serialized ciphertexts (one blob + n_items + 1
 * offsets, ct_len) under ONE user key -- same conventions as rabe_ac17_cp_decrypt_packed: bounds and, unless RABE_PACKED_TRUSTED, group
 * membership of every decoded element (one batched pass per group over the whole blob); status[i] = 0 / -1, plaintexts back to back in
 * pt_buf at pt_off[i] (nothing for a failed item); returns 1 when pt_cap is too small (sum of the sealed lengths always suffices).  The
 * pairings of the whole batch are one rhip_pairing_jobs launch set (<= 2 m + 3 Miller loops per item on a few accumulators, one final
 * exponentiation per item), the sealed plaintexts are opened on the device (Level S): no Gt crosses PCIe. */
int32_t rabe_bdabe_decrypt_packed(rabe_host* h, const void* uk, size_t n_items, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, uint32_t flags,
                                  int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off);
int32_t rabe_mke08_decrypt_packed(rabe_host* h, const void* uk, size_t n_items, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, uint32_t flags,
                                  int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off);
int32_t rabe_mke08_setup(rabe_host* h, void** pk, void** msk);
int32_t rabe_mke08_keygen(rabe_host* h, const void* pk, const void* msk, const char* name, void** uk);
int32_t rabe_mke08_authgen(rabe_host* h, const char* name, void** ska);
int32_t rabe_mke08_request_authority_pk(rabe_host* h, const void* pk, const char* attribute, const void* ska, void** pka);
int32_t rabe_mke08_request_authority_sk(rabe_host* h, void* uk, const char* attribute, const void* ska);
int32_t rabe_mke08_encrypt(rabe_host* h, const void* pk, const void* const* attr_pks, size_t n_pks, const char* policy, int32_t language,
                           const uint8_t* plaintext, size_t len, void** ct);
int32_t rabe_mke08_decrypt(rabe_host* h, const void* uk, const void* ct, uint8_t** plaintext, size_t* len);
int32_t rabe_mke08_decrypt_gt(rabe_host* h, const void* uk, const void* ct, uint8_t out_gt[384]);
int32_t rabe_mke08_decrypt_batch(rabe_host* h, size_t n, const void* const* uks, const void* const* cts, int32_t* status, uint8_t** plaintexts,
                                 size_t* lens);
/* dnf.rs:203-243 and :186-201 on attribute names only: result 1 / 0; the terms as JSON [[attr, ..], ..] (keys = one name per public attribute key) */
int32_t rabe_policy_in_dnf(const char* policy, int32_t language, int32_t* result);
int32_t rabe_policy_dnf_terms(const char* policy, int32_t language, const char* const* key_attrs, size_t n, char** out);

/* ---- host-only policy utilities (no GPU needed): results as small JSON texts, free with rabe_bytes_free.
 *   rabe_policy_parse      -> serialize_policy(parse(policy, language), out_language)       pest/mod.rs:40-114
 *   rabe_policy_msp        -> {"m": [[..]], "pi": [..], "c": n}                             msp.rs:78-147
 *   rabe_policy_pruned     -> {"match": bool, "list": [[name, name_col], ..]}               secretsharing/mod.rs:143-199
 *   rabe_policy_traverse   -> 1 / 0                                                         tools/mod.rs:31-61
 *   rabe_policy_shares     -> [[name_col, hex(fr_le)], ..] for secret + coefficient tape    secretsharing/mod.rs:82-141
 *   rabe_policy_coeffs     -> [[name_col, hex(fr_le)], ..]                                  secretsharing/mod.rs:9-72
 */
int32_t rabe_policy_parse(const char* policy, int32_t language, int32_t out_language, char** out);
int32_t rabe_policy_msp(const char* policy, int32_t language, char** out);
int32_t rabe_policy_pruned(const char* policy, int32_t language, const char* const* attributes, size_t n, char** out);
int32_t rabe_policy_traverse(const char* policy, int32_t language, const char* const* attributes, size_t n, int32_t* result);
int32_t rabe_policy_shares(const char* policy, int32_t language, const uint8_t secret_le32[32], const uint8_t* tape_le32, size_t n_tape, char** out);
int32_t rabe_policy_coeffs(const char* policy, int32_t language, char** out);
/* sha3_hash_fr (src/utils/hash/mod.rs:23-31): Fr::from_slice(SHA3-256(label)), little-endian canonical */
int32_t rabe_hash_fr(const char* label, uint8_t out_le32[32]);
/* test hook: x mod r of a 512-bit little-endian x by the fast path and by plain long division */
int32_t rabe_fr_reduce512(const uint8_t in_le64[64], uint8_t out_fast[32], uint8_t out_division[32]);
/* KDF + AES-256-GCM of src/utils/aes/mod.rs with an explicit nonce; out = nonce || ct || tag */
int32_t rabe_encrypt_symmetric(const uint8_t gt[384], const uint8_t* data, size_t len, const uint8_t nonce[12], uint8_t** out, size_t* out_len);
int32_t rabe_decrypt_symmetric(const uint8_t gt[384], const uint8_t* data, size_t len, uint8_t** out, size_t* out_len);
/* the raw primitives under the two functions above (sha3 0.10.8 / aes-gcm 0.10.3 in the reference, Cargo.toml:28,36), exported
 * so that public known-answer vectors can pin them; out of rabe_aes256_gcm_encrypt = ciphertext || tag(16) */
int32_t rabe_sha3_256(const uint8_t* data, size_t len, uint8_t out[32]);
int32_t rabe_aes256_gcm_encrypt(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* data, size_t len, uint8_t** out, size_t* out_len);
int32_t rabe_aes256_gcm_decrypt(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* data, size_t len, uint8_t** out, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
