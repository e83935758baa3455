/*
 * rabe_hip.h -- C ABI of the MI355X (gfx950) batched ABE pairing / scalar-multiplication engine.
 *
 * This is the drop-in boundary for the hot path of Fraunhofer-AISEC/rabe: the per-attribute loops of
 * rabe::schemes::{ac17,bsw,lsw,aw11}::{keygen,encrypt,decrypt}.  In the reference all of that math
 * enters through `use rabe_bn::{Group, Gt, G1, G2, Fr, pairing}` (src/schemes/ac17/mod.rs:42,
 * bsw/mod.rs:23, lsw/mod.rs:23, aw11/mod.rs:27); a Rust host keeps policy parsing, MSP, pruning, KDF
 * and AES and binds these symbols over FFI (INTEGRATION.md shows the `extern "C"` block).
 * The convention follows the reference's own (stale) C FFI, src/ffi/bsw.rs:22-163: opaque context
 * pointers created/destroyed by paired functions, int32 status (0 ok, <0 error).
 *
 * Two levels:
 *   Level E  element batches -- `rabe_bn` operator semantics on arrays (n independent operations),
 *            used for parity against the oracle and for a `rabe-bn`-shaped shim.
 *   Level B  scheme batches -- whole encrypt / keygen / decrypt group-arithmetic of n independent calls.
 *
 * Wire format (all little-endian, canonical = fully reduced, NOT Montgomery):
 *   rhip_fr  32 B   integer < r
 *   rhip_g1  64 B   affine x || y, each an integer < p; the point at infinity is all-zero
 *   rhip_g2 128 B   affine x.c0 || x.c1 || y.c0 || y.c1  (Fq2 = c0 + c1*u); infinity all-zero
 *   rhip_gt 384 B   12 integers < p in tower order c0.a0.c0, c0.a0.c1, c0.a1.c0, ... c1.a2.c1
 *                   (Fq12 = c0 + c1*w, Fq6 = a0 + a1*v + a2*v^2, Fq2 = c0 + c1*u)
 * Every `dev` pointer below is DEVICE memory (hipMalloc / rhip_malloc / a torch CUDA tensor);
 * the call is asynchronous on the context's stream; rhip_sync() waits.
 */
#ifndef RABE_HIP_H
#define RABE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint32_t l[8]; } rhip_fr;
typedef struct { uint32_t l[16]; } rhip_g1;
typedef struct { uint32_t l[32]; } rhip_g2;
typedef struct { uint32_t l[96]; } rhip_gt;

typedef struct rhip_ctx rhip_ctx;

#define RHIP_OK 0
#define RHIP_ERR_NO_DEVICE (-1)   /* no HIP device / kernels not loadable: the engine never falls back to the CPU */
#define RHIP_ERR_HIP (-2)         /* a HIP runtime call failed; see rhip_last_error */
#define RHIP_ERR_ARG (-3)
#define RHIP_ERR_NOT_MEMBER (-4)  /* an input point is not on the curve (rabe_bn::FieldError::NotMember, src/error.rs:60-69) */

/* ---- context (one per host thread / per GPU; single owner) ------------------------------------ */
/* The first context of a device runs a known-answer self-test of the BN254 field / tower arithmetic on every SIMD (adversarial
 * carry patterns through every multiply-accumulate and carry-chain form; ~1 ms; bn254/selftest.h) and refuses the device
 * (RHIP_ERR_HIP, message in rhip_last_error(NULL)) on any mismatch.  RABE_NO_SELFTEST=1 skips it. */
int32_t rhip_ctx_create(int32_t device, rhip_ctx** out);
/* number of distinct SIMDs the self-test of this context's device ran on (diagnostic) */
int32_t rhip_ctx_selftest_info(rhip_ctx* ctx, uint32_t* simds_checked);
void rhip_ctx_destroy(rhip_ctx* ctx);
/* use an existing hipStream_t (e.g. torch's current stream); NULL = the context's own stream */
int32_t rhip_ctx_set_stream(rhip_ctx* ctx, void* hip_stream);
/* make the context's device the calling thread's current HIP device (a host thread that serves several contexts / devices) */
int32_t rhip_ctx_make_current(rhip_ctx* ctx);
int32_t rhip_sync(rhip_ctx* ctx);
const char* rhip_last_error(rhip_ctx* ctx);
/* per-kernel timing with HIP events on the launch stream (measurement only; off by default).
 * rhip_ctx_timing_read drains the record as "kernel_name total_ms launches\n" lines. */
int32_t rhip_ctx_timing(rhip_ctx* ctx, int32_t enable);
int32_t rhip_ctx_timing_read(rhip_ctx* ctx, char* buf, size_t len);
/* pairing kernels: 0 = auto (cooperating lanes for launches that would under-fill the chip, one lane per item and chunk
 * otherwise), 1 = always one lane, 3 = three lanes per pairing (pairwise paths), 6 = six lanes per Fq12 accumulator
 * (k_miller_c6 / k_final_exp_c6: one Fq2 coefficient per lane), 29 = one lane per item and chunk on the reduced-radix field core
 * (k_miller_multi_rr: 9 x 29-bit limbs; what mode 0 takes for launches that fill the chip).  Results are identical.
 * 99 = cross-check: every multi-pairing launch runs as mode 0 would run it (the caller gets that result) and again with modes 1, 6 and 29
 * forced -- Miller loops and final exponentiation of each family on the same pair lists -- and the results are compared on the device; a
 * difference fails the call (RHIP_ERR_HIP, the families named in rhip_last_error).  Synchronous and about four times the pairing work: a
 * self-check for tests and suspect devices, not a production mode.
 * RABE_PAIRING_MODE in the environment of rhip_ctx_create presets the mode (A/B runs). */
int32_t rhip_ctx_set_pairing_mode(rhip_ctx* ctx, int32_t mode);
/* number of compute units / device name of the context's GPU */
int32_t rhip_device_info(rhip_ctx* ctx, int32_t* n_cu, char* name, size_t name_len);

/* device memory helpers for hosts without their own allocator */
int32_t rhip_malloc(rhip_ctx* ctx, size_t bytes, void** dev);
int32_t rhip_free(rhip_ctx* ctx, void* dev);
int32_t rhip_upload(rhip_ctx* ctx, void* dev, const void* host, size_t bytes);
int32_t rhip_download(rhip_ctx* ctx, void* host, const void* dev, size_t bytes);
/* Pinned host memory and copies that are only ordered on the context's stream (rhip_sync waits for them): a caller with
 * several contexts overlaps the PCIe traffic of one batch with the kernels of the others. */
int32_t rhip_host_alloc(rhip_ctx* ctx, size_t bytes, void** host);
int32_t rhip_host_free(rhip_ctx* ctx, void* host);
int32_t rhip_upload_async(rhip_ctx* ctx, void* dev, const void* host_pinned, size_t bytes);
int32_t rhip_download_async(rhip_ctx* ctx, void* host_pinned, const void* dev, size_t bytes);
/* stream-ordered fill (the host layer wipes staging areas that held master-key-derived scalars) */
int32_t rhip_memset_async(rhip_ctx* ctx, void* dev, int32_t byte, size_t bytes);
/* cross-context ordering without a host round trip: work submitted to `ctx` after this call starts only when everything
 * submitted to `other` so far has finished (event record + stream wait): e.g. a copy context that drains one batch's
 * outputs while the compute context already runs the next kernels */
int32_t rhip_ctx_wait_for(rhip_ctx* ctx, rhip_ctx* other);
/* a point of a context's stream that a HOST thread can wait for without waiting for the stream's later work (rhip_event_wait blocks until
 * everything submitted to ctx before rhip_event_record has finished, and frees the event) */
typedef struct rhip_event rhip_event;
int32_t rhip_event_record(rhip_ctx* ctx, rhip_event** out);
int32_t rhip_event_wait(rhip_event* ev);
/* Pipelining two launch sets on two contexts (streams): `waiter`'s stream is held until ctx's NEXT decrypt (any entry point that runs
 * the shared-accumulator Miller kernel) has issued its Miller loops; what is left on ctx then is its final exponentiation (one wave per
 * item), beside which the waiter's encrypt kernels can run.  One-shot.  Typical use: a small launch set on ctx, the next large one on
 * waiter -- submit ctx first, then enqueue waiter's work; make waiter's decrypt wait for all of ctx with rhip_ctx_wait_for.  The request
 * is consumed by ctx's next final-exponentiation launch; waiter = NULL withdraws a pending one (a context must not be destroyed while
 * it is a pending waiter).  The hold is bounded: the waiter's stream polls for at most a few tens of milliseconds. */
int32_t rhip_ctx_release_before_final_exp(rhip_ctx* ctx, rhip_ctx* waiter);
/* The same hold, released as soon as ctx's Miller loops are DONE (no wait for the final exponentiation's blocks to be resident): for a
 * waiter with little work of its own -- the Gt membership checks of a packed decrypt run beside the final exponentiation this way. */
int32_t rhip_ctx_release_after_miller(rhip_ctx* ctx, rhip_ctx* waiter);
/* The hold for a SMALL launch set on ctx beside the waiter's large one: the waiter's stream goes on as soon as the blocks of ctx's next Miller
 * launch are RESIDENT -- they own their CUs, the waiter's encrypt kernels take the CUs that are left -- instead of when that launch is done.
 * Served by the reduced-radix Miller kernel on uniform pair lists (pairing mode 0 / 29); any other launch behaves as under
 * rhip_ctx_release_before_final_exp.  One-shot; same bytes either way. */
int32_t rhip_ctx_release_when_miller_resident(rhip_ctx* ctx, rhip_ctx* waiter);

/* ---- Level E: element batches (n independent operations) --------------------------------------
 * rabe_bn surface replaced ("rabe_bn API surface actually used"):
 *   Fr: + - * neg inverse            src/utils/secretsharing/mod.rs:25-28,66,218; ac17/mod.rs:213,229,240
 *   Fr::from_slice(SHA3 digest)      src/utils/hash/mod.rs:16,27
 *   G1/G2: + - neg, * Fr             ac17/mod.rs:219,235-240,300-302,343-348,406-415
 *   Gt: * inverse pow                ac17/mod.rs:357-360,418; bsw/mod.rs:234,291-294,308
 *   pairing(G1,G2)                   ac17/mod.rs:148,415-416; bsw/mod.rs:108,292-293,308
 * Scalars (rhip_fr operands of * Fr, pow, the MSM / selection coefficients of Level B) are canonical records (< r).  A record
 * that is not is brought below r where it is loaded, so `k * P` is the group's answer for every 256-bit word; nothing relies on
 * the caller having checked.  Group elements are NOT validated by the arithmetic entry points: rhip_g1_on_curve /
 * rhip_g2_in_subgroup / rhip_gt_is_member (which also reject non-canonical coordinates >= p) are the decoding checks.
 */
/* RHIP_FR_POW: out = a^b with b read as an integer (`Fr::pow(Fr)`, src/utils/secretsharing/mod.rs:218) */
enum { RHIP_FR_ADD = 0, RHIP_FR_SUB = 1, RHIP_FR_MUL = 2, RHIP_FR_NEG = 3, RHIP_FR_INV = 4, RHIP_FR_POW = 5 };
int32_t rhip_fr_op(rhip_ctx* ctx, int32_t op, size_t n, const rhip_fr* dev_a, const rhip_fr* dev_b, rhip_fr* dev_out);
/* `Fr::from_slice` of n 32-byte BIG-endian digests: integer mod r */
int32_t rhip_fr_from_be32_reduce(rhip_ctx* ctx, size_t n, const uint8_t* dev_digests, rhip_fr* dev_out);

int32_t rhip_g1_add(rhip_ctx* ctx, size_t n, const rhip_g1* dev_a, const rhip_g1* dev_b, rhip_g1* dev_out);
int32_t rhip_g1_neg(rhip_ctx* ctx, size_t n, const rhip_g1* dev_a, rhip_g1* dev_out);
int32_t rhip_g1_mul(rhip_ctx* ctx, size_t n, const rhip_g1* dev_p, const rhip_fr* dev_k, rhip_g1* dev_out);
int32_t rhip_g2_add(rhip_ctx* ctx, size_t n, const rhip_g2* dev_a, const rhip_g2* dev_b, rhip_g2* dev_out);
int32_t rhip_g2_neg(rhip_ctx* ctx, size_t n, const rhip_g2* dev_a, rhip_g2* dev_out);
int32_t rhip_g2_mul(rhip_ctx* ctx, size_t n, const rhip_g2* dev_p, const rhip_fr* dev_k, rhip_g2* dev_out);
/* per-element curve membership: dev_ok[i] = 1 if on the curve (or infinity) */
int32_t rhip_g1_on_curve(rhip_ctx* ctx, size_t n, const rhip_g1* dev_p, uint32_t* dev_ok);
int32_t rhip_g2_on_curve(rhip_ctx* ctx, size_t n, const rhip_g2* dev_p, uint32_t* dev_ok);
/* membership in the groups proper (what decoding an untrusted element has to establish; G1 has cofactor 1, so on-curve suffices
 * there): G2 = the r-torsion of the twist (on the curve and r * P = O), Gt = the order-r subgroup of Fq12* */
int32_t rhip_g2_in_subgroup(rhip_ctx* ctx, size_t n, const rhip_g2* dev_p, uint32_t* dev_ok);
/* the same verdicts by the definition r * Q = O (rhip_g2_in_subgroup uses the BN-specific test [u+1]Q + psi([u]Q) + psi^2([u]Q) =
 * psi^3([2u]Q): one multiplication by the 63-bit curve parameter instead of the 254-bit order) -- kept for cross-checks */
int32_t rhip_g2_in_subgroup_by_order(rhip_ctx* ctx, size_t n, const rhip_g2* dev_p, uint32_t* dev_ok);
/* rhip_g2_in_subgroup of the listed elements only: dev_ok[k] = the verdict of dev_p[dev_idx[k]], k < n_idx */
int32_t rhip_g2_in_subgroup_at(rhip_ctx* ctx, size_t n_idx, const uint32_t* dev_idx, const rhip_g2* dev_p, uint32_t* dev_ok);
/* G2 membership as a by-product of a decrypt's Miller loops.  One-shot: the NEXT pair-list decrypt on this context (the shared-
 * accumulator paths of rhip_{ac17_cp,bsw,lsw,aw11}_decrypt_batch*, rhip_pairing_jobs) examines every G2 argument it WALKS -- the
 * untrusted side of a decrypt: c_0 (ac17), Cy.g2 (bsw), D2 (lsw), C2 (aw11) of the selected rows -- after its Miller loops:
 * the running point then holds [6u+2]Q + psi(Q) - psi^2(Q), which is -psi^3(Q) exactly for the points of the twist that lie
 * in G2 (engine_jobs.hip: k_walk_verdicts has the argument).  dev_fail[i] becomes 1 if an argument of item i fails,
 * dev_count[i] grows by the number of arguments examined (a pair with an argument at infinity is skipped and not counted; a
 * launch that takes another path counts nothing): the caller zeroes both arrays, compares the counts with what it expected to
 * be examined and runs rhip_g2_in_subgroup(_at) on whatever was not.  Curve equation and coordinate range are not part of this
 * verdict (rhip_g2_on_curve).  NULL, NULL withdraws the request. */
int32_t rhip_ctx_collect_walk_verdicts(rhip_ctx* ctx, uint32_t* dev_fail, uint32_t* dev_count);
int32_t rhip_gt_is_member(rhip_ctx* ctx, size_t n, const rhip_gt* dev_a, uint32_t* dev_ok);
/* the same verdicts with the order test by the definition f^r = 1 (rhip_gt_is_member uses f^p = f^(6u^2) after the cyclotomic test) */
int32_t rhip_gt_is_member_by_order(rhip_ctx* ctx, size_t n, const rhip_gt* dev_a, uint32_t* dev_ok);
/* verdicts folded per item on the device: dev_out[s] = AND of dev_flags[scale * dev_seg_off[s] .. scale * dev_seg_off[s + 1]),
 * s < n_seg (dev_seg_off has n_seg + 1 entries: an item's rows; scale = elements per row) */
int32_t rhip_flags_all(rhip_ctx* ctx, size_t n_seg, const uint32_t* dev_seg_off, uint32_t scale, const uint32_t* dev_flags, uint32_t* dev_out);

int32_t rhip_gt_mul(rhip_ctx* ctx, size_t n, const rhip_gt* dev_a, const rhip_gt* dev_b, rhip_gt* dev_out);
/* out[i] = product of a[off[i] .. off[i+1]) (1 for an empty segment): the Gt accumulation loops of aw11::decrypt :320-352 */
int32_t rhip_gt_product(rhip_ctx* ctx, size_t n_items, const uint32_t* dev_off /*[n_items+1]*/, const rhip_gt* dev_a, rhip_gt* dev_out);
int32_t rhip_gt_inv(rhip_ctx* ctx, size_t n, const rhip_gt* dev_a, rhip_gt* dev_out);
int32_t rhip_gt_pow(rhip_ctx* ctx, size_t n, const rhip_gt* dev_a, const rhip_fr* dev_k, rhip_gt* dev_out);

/* n independent pairings e(p_i, q_i) */
int32_t rhip_pairing(rhip_ctx* ctx, size_t n, const rhip_g1* dev_p, const rhip_g2* dev_q, rhip_gt* dev_out);
/* n_items products of pairings: out[i] = prod_{j in [off[i], off[i+1])} e(p_j, q_j), with ONE final
 * exponentiation per item (the restructuring of SURVEY.md Appendix B).  dev_off has n_items+1 entries. */
int32_t rhip_pairing_product(rhip_ctx* ctx, size_t n_items, const uint32_t* dev_off, size_t n_pairs,
                             const rhip_g1* dev_p, const rhip_g2* dev_q, rhip_gt* dev_out);

/* The general form the decrypts of bsw / lsw / aw11 / ghw11 reduce to (coefficient-folded products of pairings):
 *   out[i] = lead[i] * FE( prod_{j in [pair_off[i], pair_off[i+1])} ML(scal[j] * base[j], q[j])  *  ML( sum_{t in [sum_off[i], sum_off[i+1])} s_scal[t] * s_base[t], s_q[i] ) )
 * in one launch set: the G1 arguments are scaled over the NAF of their coefficients (the shorter of c and r - c), the sum runs with
 * shared doublings, all pairs of an item share a few Fq12 accumulators, one final exponentiation per item.  dev_scal NULL: all 1;
 * dev_sum_off NULL: no summed pair (dev_s_* ignored); dev_lead NULL: no leading factor.  max_pairs / max_terms: the largest per-item
 * counts (they size the launch). */
int32_t rhip_pairing_jobs(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t n_pairs, const uint32_t* dev_pair_off /*[n_items+1]*/,
                          const rhip_g1* dev_base, const rhip_fr* dev_scal, const rhip_g2* dev_q, size_t max_terms, size_t n_terms,
                          const uint32_t* dev_sum_off /*[n_items+1] or NULL*/, const rhip_g1* dev_s_base, const rhip_fr* dev_s_scal,
                          const rhip_g2* dev_s_q /*[n_items]*/, const rhip_gt* dev_lead /*[n_items] or NULL*/, rhip_gt* dev_out /*[n_items]*/);

/* ---- Level E, host-value forms: ONE element per call, host pointers (upload, launch, download).  These are what an
 * operator-overloading replacement of the `rabe_bn` crate binds (`impl Mul<Fr> for G1`, `pairing(p, q)`, ...; see
 * INTEGRATION.md section 2 and integration/rabe-bn-shim/): whole-program parity runs of unmodified rabe, not throughput. */
int32_t rhip_host_fr_op(rhip_ctx* ctx, int32_t op, const rhip_fr* a, const rhip_fr* b /* NULL for NEG / INV */, rhip_fr* out);
int32_t rhip_host_fr_pow(rhip_ctx* ctx, const rhip_fr* a, const rhip_fr* e, rhip_fr* out);
int32_t rhip_host_fr_from_be32_reduce(rhip_ctx* ctx, const uint8_t digest[32], rhip_fr* out);
/* curve membership of a decoded point (what makes `FieldError::NotMember`, src/error.rs:66): *ok = 1 on the curve or infinity */
int32_t rhip_host_g1_on_curve(rhip_ctx* ctx, const rhip_g1* p, int32_t* ok);
int32_t rhip_host_g2_on_curve(rhip_ctx* ctx, const rhip_g2* p, int32_t* ok);
/* the full decoding checks of one element: canonical coordinates, on the twist and r * P = O / in the order-r subgroup of Fq12 */
int32_t rhip_host_g2_in_subgroup(rhip_ctx* ctx, const rhip_g2* p, int32_t* ok);
int32_t rhip_host_gt_is_member(rhip_ctx* ctx, const rhip_gt* a, int32_t* ok);
int32_t rhip_host_g1_add(rhip_ctx* ctx, const rhip_g1* a, const rhip_g1* b, rhip_g1* out);
int32_t rhip_host_g1_neg(rhip_ctx* ctx, const rhip_g1* a, rhip_g1* out);
int32_t rhip_host_g1_mul(rhip_ctx* ctx, const rhip_g1* p, const rhip_fr* k, rhip_g1* out);
int32_t rhip_host_g2_add(rhip_ctx* ctx, const rhip_g2* a, const rhip_g2* b, rhip_g2* out);
int32_t rhip_host_g2_neg(rhip_ctx* ctx, const rhip_g2* a, rhip_g2* out);
int32_t rhip_host_g2_mul(rhip_ctx* ctx, const rhip_g2* p, const rhip_fr* k, rhip_g2* out);
int32_t rhip_host_gt_mul(rhip_ctx* ctx, const rhip_gt* a, const rhip_gt* b, rhip_gt* out);
int32_t rhip_host_gt_inv(rhip_ctx* ctx, const rhip_gt* a, rhip_gt* out);
int32_t rhip_host_gt_pow(rhip_ctx* ctx, const rhip_gt* a, const rhip_fr* k, rhip_gt* out);
int32_t rhip_host_pairing(rhip_ctx* ctx, const rhip_g1* p, const rhip_g2* q, rhip_gt* out);

/* ---- fixed-base tables (per public key; resident in HBM) --------------------------------------
 * Every G1/G2 element the schemes create is a known-scalar multiple of a generator from the public
 * key (hash-to-group is g * Fr(SHA3(label)), src/utils/hash/mod.rs:10-20), and every Gt power is of a
 * public-key constant, so the engine keeps 8-bit window tables of those bases. */
typedef struct rhip_g1_table rhip_g1_table;
typedef struct rhip_g2_table rhip_g2_table;
typedef struct rhip_gt_table rhip_gt_table;
int32_t rhip_g1_table_create(rhip_ctx* ctx, const rhip_g1* host_base, rhip_g1_table** out);
int32_t rhip_g2_table_create(rhip_ctx* ctx, const rhip_g2* host_base, rhip_g2_table** out);
int32_t rhip_gt_table_create(rhip_ctx* ctx, const rhip_gt* host_base, rhip_gt_table** out);
/* adds 16-bit windows (16 x 65535 entries, 67 MB) to a G1 table: halves the additions of rhip_g1_table_mul-style
 * kernels; rhip_ac17_pk_create does this for the public generator g */
int32_t rhip_g1_table_add_w16(rhip_ctx* ctx, rhip_g1_table* t);
/* adds signed w_bits-wide windows (17 <= w_bits <= 27): a fixed-base multiplication then costs ceil(254 / w_bits)
 * mixed additions (11 at 24 bits, 10 at 26) for 64 B x 2^(w_bits-1) x ceil(254 / w_bits) of HBM (5.4 GB at 24 bits,
 * 19 GB at 26) -- memory traded for work on a 288 GB part.  Results are the same group elements, hence the same bytes. */
int32_t rhip_g1_table_add_wide(rhip_ctx* ctx, rhip_g1_table* t, int32_t w_bits);
/* the same for a G2 base (16 x 65535 x 128 B = 134 MB; rhip_ac17_pk_create does this for h_a[0..2]) */
int32_t rhip_g2_table_add_w16(rhip_ctx* ctx, rhip_g2_table* t);
/* the same for a Gt base (16 x 65535 x 384 B = 402 MB): halves the multiplications of a fixed-base Gt power */
int32_t rhip_gt_table_add_w16(rhip_ctx* ctx, rhip_gt_table* t);
void rhip_g1_table_destroy(rhip_g1_table* t);
void rhip_g2_table_destroy(rhip_g2_table* t);
void rhip_gt_table_destroy(rhip_gt_table* t);
/* out[i] = base * k[i]  /  base ^ k[i] */
int32_t rhip_g1_table_mul(rhip_ctx* ctx, const rhip_g1_table* t, size_t n, const rhip_fr* dev_k, rhip_g1* dev_out);
int32_t rhip_g2_table_mul(rhip_ctx* ctx, const rhip_g2_table* t, size_t n, const rhip_fr* dev_k, rhip_g2* dev_out);
int32_t rhip_gt_table_pow(rhip_ctx* ctx, const rhip_gt_table* t, size_t n, const rhip_fr* dev_k, rhip_gt* dev_out);

/* ---- Level B: AC17 (FAME) CP-ABE -------------------------------------------------------------- */
/* Public key side of ac17::cp_encrypt (Ac17PublicKey, src/schemes/ac17/mod.rs:62-66). */
typedef struct rhip_ac17_pk rhip_ac17_pk;
int32_t rhip_ac17_pk_create(rhip_ctx* ctx, const rhip_g1* host_g, const rhip_g2* host_h_a /*[3]*/,
                            const rhip_gt* host_e_gh_ka /*[2]*/, rhip_ac17_pk** out);
/* switches the public generator's table of pk to signed w_bits-wide windows (rhip_g1_table_add_wide) */
int32_t rhip_ac17_pk_set_g_window(rhip_ctx* ctx, rhip_ac17_pk* pk, int32_t w_bits);
void rhip_ac17_pk_destroy(rhip_ac17_pk* pk);

/* Group arithmetic of n_items calls of ac17::cp_encrypt (src/schemes/ac17/mod.rs:274-376).
 * The host has parsed each policy and built its MSP; per distinct policy it supplies the Fr table
 *   A[row][l][t] = h(pi_row || l || t) + sum_j M[row][j] * h("0" || (j+1) || l || t)   (mod r)
 * (h = Fr::from_slice(SHA3-256), src/utils/hash/mod.rs:16) for row < n_rows(policy), l < 3, t < 2; the
 * tables of all policies used by the batch are concatenated in dev_A and item i points at the first row
 * of its policy with item_A_off[i].  Ciphertext i owns output rows [ct_row_off[i], ct_row_off[i+1])
 * (n_rows of its policy).  Per item the explicit randomness is s[i][0..2) (drawn at :292) and the Gt
 * `msg` (drawn at :362).  Outputs:
 *   c_0[i][0..3)  = (h_a0*s0, h_a1*s1, h_a2*(s0+s1))                       (:297-302)
 *   c[row][l]     = g * (s0*A[a][l][0] + s1*A[a][l][1])                    (:330-356)
 *   c_p[i]        = e_gh_ka0^s0 * e_gh_ka1^s1 * msg                        (:357-368)
 */
int32_t rhip_ac17_cp_encrypt_batch(rhip_ctx* ctx, const rhip_ac17_pk* pk, size_t n_items,
                                   const rhip_fr* dev_A /*[policy rows][3][2]*/, const uint32_t* dev_item_A_off /*[n_items]*/,
                                   const uint32_t* dev_ct_row_off /*[n_items+1]*/, size_t total_rows,
                                   const rhip_fr* dev_s /*[n_items][2]*/, const rhip_gt* dev_msg /*[n_items]*/,
                                   rhip_g2* dev_c0 /*[n_items][3]*/, rhip_g1* dev_c /*[total_rows][3]*/, rhip_gt* dev_cp /*[n_items]*/);

/* Group arithmetic of n_items calls of ac17::cp_keygen (src/schemes/ac17/mod.rs:191-264).
 * Host supplies per attribute y the hashes H[y][l][t] = h(y||l||t) and, once, H01[l][t] = h("01"||l||t);
 * per item the randomness r0, r1, sigma_y (one per attribute), sigma'.  a_inv[t] = a_t^-1, b[t] from the msk.
 * Outputs per item: k_0[3] (G2), k[y][3] (G1), k_p[3] (G1).  g_k[3] are the msk's G1 elements. */
int32_t rhip_ac17_cp_keygen_batch(rhip_ctx* ctx, const rhip_g1_table* g_table, const rhip_g2_table* h_table,
                                  const rhip_g1* dev_g_k /*[3]*/, const rhip_fr* dev_a_inv /*[2]*/, const rhip_fr* dev_b /*[2]*/,
                                  size_t n_items, size_t n_attrs, const rhip_fr* dev_H /*[n_attrs][3][2]*/,
                                  const rhip_fr* dev_H01 /*[3][2]*/, const rhip_fr* dev_r /*[n_items][2]*/,
                                  const rhip_fr* dev_sigma /*[n_items][n_attrs]*/, const rhip_fr* dev_sigma_p /*[n_items]*/,
                                  rhip_g2* dev_k0 /*[n_items][3]*/, rhip_g1* dev_k /*[n_items][n_attrs][3]*/,
                                  rhip_g1* dev_kp /*[n_items][3]*/);

/* Group arithmetic of n_items calls of ac17::cp_decrypt (src/schemes/ac17/mod.rs:385-430).
 * Ciphertext i occupies rows [ct_row_off[i], ct_row_off[i+1]) of dev_ct_c; it is decrypted with secret
 * key sk_idx[i], whose attribute rows are [sk_row_off[k], sk_row_off[k+1]) of dev_sk_k.  The host has run
 * traverse_policy / calc_pruned (string work) and hands over, per item, the matched row indices:
 * ct_sel / sk_sel hold row numbers relative to the item's ciphertext / key, delimited by sel_off
 * (one (ct,sk) index list pair per item; entries are summed once per occurrence exactly as the
 * reference's name-matching loops do, :403-414).  Output: the Gt handed to decrypt_symmetric (:418). */
int32_t rhip_ac17_cp_decrypt_batch(rhip_ctx* ctx, size_t n_items,
                                   const rhip_g2* dev_ct_c0 /*[n_items][3]*/, const rhip_g1* dev_ct_c /*[rows][3]*/,
                                   const uint32_t* dev_ct_row_off /*[n_items+1]*/, const rhip_gt* dev_ct_cp /*[n_items]*/,
                                   const rhip_g2* dev_sk_k0 /*[n_sk][3]*/, const rhip_g1* dev_sk_k /*[sk rows][3]*/,
                                   const uint32_t* dev_sk_row_off /*[n_sk+1]*/, const rhip_g1* dev_sk_kp /*[n_sk][3]*/,
                                   const uint32_t* dev_sk_idx /*[n_items]*/,
                                   const uint32_t* dev_ct_sel, const uint32_t* dev_ct_sel_off /*[n_items+1]*/,
                                   const uint32_t* dev_sk_sel, const uint32_t* dev_sk_sel_off /*[n_items+1]*/,
                                   rhip_gt* dev_out /*[n_items]*/);

/* Prepared secret keys.  k_0 is fixed per key, so the G2 side of the three pairings e(.., k_0[j]) of cp_decrypt (:416)
 * -- the Miller-loop line coefficients -- is computed once per key and replayed by every decryption with it (the
 * role of rabe-bn's G2 precomputation inside `pairing`; 3 x 88 x 192 B per key).  rhip_ac17_cp_decrypt_batch_prepared
 * takes the handle in place of dev_sk_k0 and returns exactly the values of rhip_ac17_cp_decrypt_batch; it runs the two
 * pairings of each index j on one accumulator (3 lanes per item, one Fq12 squaring per doubling step for both).
 * rhip_ac17_sk_prepare returns after the lines are complete, so the handle may be used from any context at once. */
typedef struct rhip_ac17_sk_lines rhip_ac17_sk_lines;
int32_t rhip_ac17_sk_prepare(rhip_ctx* ctx, size_t n_sk, const rhip_g2* dev_sk_k0 /*[n_sk][3]*/, rhip_ac17_sk_lines** out);
void rhip_ac17_sk_lines_destroy(rhip_ac17_sk_lines* p);
int32_t rhip_ac17_cp_decrypt_batch_prepared(rhip_ctx* ctx, size_t n_items,
                                            const rhip_g2* dev_ct_c0, const rhip_g1* dev_ct_c, const uint32_t* dev_ct_row_off,
                                            const rhip_gt* dev_ct_cp, const rhip_ac17_sk_lines* sk_lines,
                                            const rhip_g1* dev_sk_k, const uint32_t* dev_sk_row_off, const rhip_g1* dev_sk_kp,
                                            const uint32_t* dev_sk_idx, const uint32_t* dev_ct_sel, const uint32_t* dev_ct_sel_off,
                                            const uint32_t* dev_sk_sel, const uint32_t* dev_sk_sel_off, rhip_gt* dev_out);

/* ---- prepared G2 arguments (generic form of rhip_ac17_sk_prepare) -----------------------------------------------
 * The Miller-loop line coefficients of n G2 points (88 triples of Fq2 each, 16.9 KB per point): what `pairing` spends on
 * the G2 side, paid once for a G2 element that many pairings share (a secret key's components, a public key element).
 * Returns after the lines are complete, so the handle may be used from any context. */
typedef struct rhip_g2_lines rhip_g2_lines;
int32_t rhip_g2_lines_prepare(rhip_ctx* ctx, size_t n, const rhip_g2* dev_q, rhip_g2_lines** out);
void rhip_g2_lines_destroy(rhip_g2_lines* p);

/* ---- Level B: BSW CP-ABE (src/schemes/bsw/mod.rs) ---------------------------------------------------------------
 * Flattened policy trees (the host has parsed the policy text; everything below is numbers).  The leaves of a policy
 * are numbered in DFS order -- the order gen_shares_policy emits shares (src/utils/secretsharing/mod.rs:82-122) --
 * and the tables of all distinct policies of a batch are concatenated:
 *   path_off[leaf .. leaf+1]      the leaf's path entries, root first
 *   path_gate[e], path_x[e]       "child number x (1-based) of gate `gate`" (gate numbered relative to the policy's first gate)
 *   gate_k[g]                     threshold of gate g: its child count for AND, 1 for OR (:98-110)
 *   gate_coef_off[g]              where the gate's k-1 polynomial coefficients start in an item's draw list (DFS pre-order,
 *                                 the reference's draw order :128-134)
 *   leaf_hash[leaf]               Fr(SHA3(name)) of the leaf's attribute (src/utils/hash/mod.rs:23-31)
 * Item i uses the policy whose first leaf row / first gate row are item_tree_leaf[i] / item_tree_gate[i], owns output
 * leaf rows [item_leaf_off[i], item_leaf_off[i+1]) and the draws coef[item_coef_off[i] ..].
 */
typedef struct rhip_bsw_pk rhip_bsw_pk;       /* window tables of g1, g2, h, e_gg_alpha (CpAbePublicKey, bsw/mod.rs:43-49) */
int32_t rhip_bsw_pk_create(rhip_ctx* ctx, const rhip_g1* host_g1, const rhip_g2* host_g2, const rhip_g1* host_h,
                           const rhip_gt* host_e_gg_alpha, rhip_bsw_pk** out);
void rhip_bsw_pk_destroy(rhip_bsw_pk* pk);
/* Group arithmetic of n_items calls of bsw::encrypt (bsw/mod.rs:217-251): explicit randomness per item = secret (:228),
 * the Gt `msg` (:229) and the gate coefficients.  Outputs: c = h * secret, c_p = e_gg_alpha^secret * msg, and per leaf
 * row (g1 * q_y, (g2 * h(name_y)) * q_y) with q_y the leaf's share of `secret`. */
int32_t rhip_bsw_encrypt_batch(rhip_ctx* ctx, const rhip_bsw_pk* pk, size_t n_items, size_t total_leaves,
                               const uint32_t* dev_item_leaf_off /*[n_items+1]*/, const uint32_t* dev_item_tree_leaf /*[n_items]*/,
                               const uint32_t* dev_item_tree_gate /*[n_items]*/, const uint32_t* dev_path_off, const uint32_t* dev_path_gate,
                               const uint32_t* dev_path_x, const uint32_t* dev_gate_k, const uint32_t* dev_gate_coef_off,
                               const rhip_fr* dev_leaf_hash, const rhip_fr* dev_secret /*[n_items]*/, const rhip_fr* dev_coef,
                               const uint32_t* dev_item_coef_off /*[n_items]*/, const rhip_gt* dev_msg /*[n_items]*/,
                               rhip_g1* dev_c /*[n_items]*/, rhip_gt* dev_cp /*[n_items]*/, rhip_g1* dev_cy_g1 /*[total_leaves]*/,
                               rhip_g2* dev_cy_g2 /*[total_leaves]*/);
/* Group arithmetic of n_items calls of bsw::decrypt (bsw/mod.rs:260-318).  The host has run traverse_policy / calc_pruned /
 * calc_coefficients (string and Fr work, per distinct (policy, key attribute set)) and hands over selection tables: entry e
 * names a pruned leaf by its row in the ciphertext (sel_ct_leaf, relative to the item's first leaf row), the matching
 * attribute row of the key (sel_sk_attr, relative to the key's first row) and the leaf's Lagrange coefficient z
 * (sel_coeff).  Item i uses entries sel_start[i] .. sel_start[i] + m_i - 1 and owns pairs [pair_off[i], pair_off[i+1]),
 * pair_off[i+1] - pair_off[i] = 2 m_i + 1 (max_pairs = the largest such count).  Ciphertext i: c, c_p and leaf rows
 * [ct_leaf_off[i], ct_leaf_off[i+1]); key sk_idx[i]: d and attribute rows [sk_attr_off[k], sk_attr_off[k+1]).
 *   out[i] = c_p * FE( ML(-c, d) * prod_e ML(z_e Cy.g1, Dj.g2) ML(-z_e Dj.g1, Cy.g2) )     
 * i.e. the Gt handed to decrypt_symmetric (:308-311).  sk_lines (optional, rhip_bsw_sk_prepare): prepared lines of the
 * keys' G2 components -- the same values, without the G2 arithmetic of the key-side pairings. */
typedef struct rhip_bsw_sk_lines rhip_bsw_sk_lines;
int32_t rhip_bsw_sk_prepare(rhip_ctx* ctx, size_t n_sk, size_t total_attrs, const rhip_g2* dev_sk_d /*[n_sk]*/,
                            const rhip_g2* dev_sk_dj_g2 /*[total_attrs]*/, rhip_bsw_sk_lines** out);
void rhip_bsw_sk_lines_destroy(rhip_bsw_sk_lines* p);
int32_t rhip_bsw_decrypt_batch(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs,
                               const uint32_t* dev_pair_off /*[n_items+1]*/, const uint32_t* dev_sel_start /*[n_items]*/,
                               const uint32_t* dev_sel_ct_leaf, const uint32_t* dev_sel_sk_attr, const rhip_fr* dev_sel_coeff,
                               const rhip_g1* dev_ct_c /*[n_items]*/, const rhip_gt* dev_ct_cp /*[n_items]*/,
                               const rhip_g1* dev_ct_cy_g1, const rhip_g2* dev_ct_cy_g2, const uint32_t* dev_ct_leaf_off /*[n_items+1]*/,
                               const rhip_g2* dev_sk_d /*[n_sk]*/, const rhip_g1* dev_sk_dj_g1, const rhip_g2* dev_sk_dj_g2,
                               const uint32_t* dev_sk_attr_off /*[n_sk+1]*/, const uint32_t* dev_sk_idx /*[n_items]*/,
                               const rhip_bsw_sk_lines* sk_lines /* or NULL */, rhip_gt* dev_out /*[n_items]*/);

/* The same with ONE secret key for every item (dev_sk_d [1], dev_sk_dj_* = its attribute rows, dev_sk_attr_off = {0, n}) -- what
 * rabe_bsw_decrypt_packed does and BASELINE config 3 measures.  The key-side G1 arguments -z_e * Dj.g1 then depend on the selection
 * entry alone: ciphertexts that share a policy share their entries (dev_sel_start), so those scalings are computed once per ENTRY (n_sel
 * of them) instead of once per pair.  Results are those of rhip_bsw_decrypt_batch with dev_sk_idx = all zero. */
int32_t rhip_bsw_decrypt_batch_one_sk(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel,
                                      const uint32_t* dev_pair_off /*[n_items+1]*/, const uint32_t* dev_sel_start /*[n_items]*/,
                                      const uint32_t* dev_sel_ct_leaf, const uint32_t* dev_sel_sk_attr, const rhip_fr* dev_sel_coeff,
                                      const rhip_g1* dev_ct_c /*[n_items]*/, const rhip_gt* dev_ct_cp /*[n_items]*/,
                                      const rhip_g1* dev_ct_cy_g1, const rhip_g2* dev_ct_cy_g2, const uint32_t* dev_ct_leaf_off /*[n_items+1]*/,
                                      const rhip_g2* dev_sk_d /*[1]*/, const rhip_g1* dev_sk_dj_g1, const rhip_g2* dev_sk_dj_g2,
                                      const uint32_t* dev_sk_attr_off /*[2]*/, const rhip_bsw_sk_lines* sk_lines /* or NULL */, rhip_gt* dev_out /*[n_items]*/);

/* ---- Level B: LSW KP-ABE (src/schemes/lsw/mod.rs) ----------------------------------------------------------------
 * rhip_lsw_keygen_batch: positive leaves; rhip_lsw_keygen_batch_signed: positive and negative ("!x", :137-146).  The reference's
 * decrypt has no negative branch at all (a TODO, :265-278): the host layer reproduces what it does instead. */
typedef struct rhip_lsw_pk rhip_lsw_pk;       /* window tables of g1, g2 (KpAbePublicKey, lsw/mod.rs:44-51) */
int32_t rhip_lsw_pk_create(rhip_ctx* ctx, const rhip_g1* host_g1, const rhip_g2* host_g2, rhip_lsw_pk** out);
void rhip_lsw_pk_destroy(rhip_lsw_pk* pk);
/* Group arithmetic of n_items calls of lsw::keygen (lsw/mod.rs:121-170): shares q_y of alpha1 over the item's policy
 * (flattened trees as for bsw; draws = the gate coefficients), then per leaf the explicit `random` r_y (:136):
 *   d1[row] = g1 * (alpha2 * q_y + h(y) * r_y)   ( = g1*(alpha2 q_y) + (g1*h(y))*r_y, :149-154 ),   d2[row] = g2 * r_y
 * dev_alpha = (alpha1, alpha2) of the master key. */
int32_t rhip_lsw_keygen_batch(rhip_ctx* ctx, const rhip_lsw_pk* pk, size_t n_items, size_t total_leaves,
                              const uint32_t* dev_item_leaf_off /*[n_items+1]*/, const uint32_t* dev_item_tree_leaf, const uint32_t* dev_item_tree_gate,
                              const uint32_t* dev_path_off, const uint32_t* dev_path_gate, const uint32_t* dev_path_x, const uint32_t* dev_gate_k,
                              const uint32_t* dev_gate_coef_off, const rhip_fr* dev_leaf_hash, const rhip_fr* dev_alpha /*[2]*/,
                              const rhip_fr* dev_coef, const uint32_t* dev_item_coef_off /*[n_items]*/, const rhip_fr* dev_rand /*[total_leaves]*/,
                              rhip_g1* dev_d1 /*[total_leaves]*/, rhip_g2* dev_d2 /*[total_leaves]*/);
/* The same with negative leaves ("!x", lsw/mod.rs:137-146): leaf_neg[leaf] (per policy leaf, beside leaf_hash) != 0 marks them.
 * A positive row gets (d1, d2) as above and the identity in d3..d5; a negative row gets the identity in d1, d2 and
 *   d3 = g1 * q_y + g1_b2 * r_y,   d4 = g1_b * (h(y) r_y) + h_g1 * r_y,   d5 = g1 * (-r_y)
 * with g1_b = g1 * b, g1_b2 = g1 * b^2 (dev_b = the master key's b, host_h_g1 its h_g1; a window table of h_g1 is built per call). */
int32_t rhip_lsw_keygen_batch_signed(rhip_ctx* ctx, const rhip_lsw_pk* pk, size_t n_items, size_t total_leaves,
                                     const uint32_t* dev_item_leaf_off /*[n_items+1]*/, const uint32_t* dev_item_tree_leaf, const uint32_t* dev_item_tree_gate,
                                     const uint32_t* dev_path_off, const uint32_t* dev_path_gate, const uint32_t* dev_path_x, const uint32_t* dev_gate_k,
                                     const uint32_t* dev_gate_coef_off, const rhip_fr* dev_leaf_hash, const uint32_t* dev_leaf_neg,
                                     const rhip_fr* dev_alpha /*[2]*/, const rhip_fr* dev_b /*[1]*/, const rhip_g1* host_h_g1, const rhip_fr* dev_coef,
                                     const uint32_t* dev_item_coef_off /*[n_items]*/, const rhip_fr* dev_rand /*[total_leaves]*/,
                                     rhip_g1* dev_d1, rhip_g2* dev_d2, rhip_g1* dev_d3, rhip_g1* dev_d4, rhip_g1* dev_d5 /*[total_leaves] each*/);
/* Group arithmetic of n_items calls of lsw::decrypt (lsw/mod.rs:228-290).  Selection entry e: the key's leaf row
 * (sel_sk_leaf, relative to the key's first row), the ciphertext's attribute row (sel_ct_attr, relative) and the leaf's
 * coefficient c.  Item i: entries sel_start[i] .. + m_i - 1, pairs [pair_off[i], pair_off[i+1]) with m_i + 1 of them; its
 * key is sk_idx[i] (NULL: i) with leaf rows [sk_leaf_off[k], ..), its ciphertext ct_idx[i] (NULL: i) with attribute rows
 * [ct_attr_off[c], ..).  n_sel = total number of selection entries.
 *   out[i] = e1 * FE( ML( sum_e -c_e D1_e, e2 ) * prod_e ML( c_e E1_e, D2_e ) )               
 * (the factors that share e2 collapse into one pairing of a multi-scalar sum with shared doublings).
 * ct_e2_lines (optional): prepared lines of dev_ct_e2 (rhip_g2_lines_prepare over the same array). */
int32_t rhip_lsw_decrypt_batch(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel,
                               const uint32_t* dev_pair_off /*[n_items+1]*/, const uint32_t* dev_sel_start /*[n_items]*/,
                               const uint32_t* dev_sel_sk_leaf, const uint32_t* dev_sel_ct_attr, const rhip_fr* dev_sel_coeff,
                               const rhip_gt* dev_ct_e1 /*[n_items]: item i's e1*/, const rhip_g2* dev_ct_e2 /*[n_ct]*/,
                               const rhip_g1* dev_ct_e1j /*[ct attribute rows]: ej.1*/, const uint32_t* dev_ct_attr_off /*[n_ct+1]*/,
                               const uint32_t* dev_ct_idx /*[n_items] or NULL*/, const rhip_g1* dev_sk_d1, const rhip_g2* dev_sk_d2,
                               const uint32_t* dev_sk_leaf_off /*[n_sk+1]*/, const uint32_t* dev_sk_idx /*[n_items] or NULL*/,
                               const rhip_g2_lines* ct_e2_lines /* or NULL */, rhip_gt* dev_out /*[n_items]*/);

/* The same with ONE ciphertext for every item (dev_ct_e2 [1], dev_ct_e1j = its attribute rows) -- n fresh keys against one ciphertext, what
 * BASELINE config 4 measures and rabe_lsw_decrypt_packed does.  The scaled G1 arguments c_e * E1_e then depend on the selection entry
 * alone: items that share a policy share their entries (dev_sel_start), so the scalings are computed once per ENTRY (n_sel of them)
 * instead of once per pair.  Results are those of rhip_lsw_decrypt_batch with dev_ct_idx = all zero. */
int32_t rhip_lsw_decrypt_batch_one_ct(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel,
                                      const uint32_t* dev_pair_off /*[n_items+1]*/, const uint32_t* dev_sel_start /*[n_items]*/,
                                      const uint32_t* dev_sel_sk_leaf, const uint32_t* dev_sel_ct_attr, const rhip_fr* dev_sel_coeff,
                                      const rhip_gt* dev_ct_e1 /*[n_items]: e1 replicated*/, const rhip_g2* dev_ct_e2 /*[1]*/,
                                      const rhip_g1* dev_ct_e1j /*[the ciphertext's attribute rows]*/, const rhip_g1* dev_sk_d1, const rhip_g2* dev_sk_d2,
                                      const uint32_t* dev_sk_leaf_off /*[n_sk+1]*/, const uint32_t* dev_sk_idx /*[n_items] or NULL*/,
                                      const rhip_g2_lines* ct_e2_lines /* or NULL */, rhip_gt* dev_out /*[n_items]*/);

/* ---- Level B: GHW11 outsourced decryption (src/schemes/ghw11/mod.rs:227-295) ----------------------
 * Group arithmetic of n_items calls of ghw11::transform under ONE transform key.  tk_lines = rhip_g2_lines_prepare over the key's G2
 * elements in the order k_z, l_z, k_x[0], k_x[1], ... : every Miller loop of the batch replays prepared lines (no G2 arithmetic).
 * Selection entry e: the ciphertext row (sel_ct_row, relative to the item's first row), the key attribute (sel_tk_attr) and the leaf's
 * coefficient w.  Item i: entries sel_start[i] .. + m_i - 1, pairs [pair_off[i], pair_off[i+1]) with m_i + 2 of them.
 *   out[i] = t = e(c1, k_z) / ( prod_e e(w_e D_e, K_e) * e(sum_e w_e C_e, l_z) )   as ONE product of m_i + 2 Miller values and one
 * final exponentiation (the factors that share l_z collapse into one pairing of a multi-scalar sum). */
int32_t rhip_ghw11_transform_batch(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel,
                                   const uint32_t* dev_pair_off /*[n_items+1]*/, const uint32_t* dev_sel_start /*[n_items]*/,
                                   const uint32_t* dev_sel_ct_row, const uint32_t* dev_sel_tk_attr, const rhip_fr* dev_sel_coeff,
                                   const rhip_g1* dev_ct_c1 /*[n_items]*/, const rhip_g1* dev_ct_c /*[rows]: ci*/, const rhip_g1* dev_ct_d /*[rows]: di*/,
                                   const uint32_t* dev_ct_row_off /*[n_items+1]*/, const rhip_g2_lines* tk_lines, rhip_gt* dev_out /*[n_items]*/);

/* ---- Level B: AW11 multi-authority CP-ABE (src/schemes/aw11/mod.rs) -----------------------------------------------
 * rhip_aw11_pk: gk (g1, g2), the constant e(g1, g2) and, for each of the n_attrs attributes of the authorities in play,
 * (egg_alpha_x, g2 * y_x) (Aw11PublicKey.attr, :56-61) as window tables.  leaf_attr[leaf] (per policy leaf, beside the
 * flattened tree tables) = the attribute's index in those arrays.  The per-attribute tables are 8-bit windows (4.2 MB per
 * attribute) plus, by default, SIGNED 10-bit windows built from them (26 x 512 entries = 6.8 MB per attribute, 1.4 GB for 200: 26 instead
 * of 32 table entries per power, a negative digit is a conjugation in Gt / a negated y in G2; RABE_AW11_ATTR_BITS = 9 ... 14 picks another
 * width, 8 none; only built while they leave half of the device memory free).  16-bit windows (536 MB per attribute, 107 GB for 200: 16
 * entries per power) are OPT-IN through RABE_AW11_ATTR_W16=1, and even then only built while they leave a quarter of the device memory
 * free.  Results do not depend on any of it. */
typedef struct rhip_aw11_pk rhip_aw11_pk;
int32_t rhip_aw11_pk_create(rhip_ctx* ctx, const rhip_g1* host_g1, const rhip_g2* host_g2, size_t n_attrs,
                            const rhip_gt* host_egg_alpha /*[n_attrs]*/, const rhip_g2* host_g2_y /*[n_attrs]*/, rhip_aw11_pk** out);
void rhip_aw11_pk_destroy(rhip_aw11_pk* pk);
/* Group arithmetic of n_items calls of aw11::encrypt (aw11/mod.rs:241-289).  Explicit randomness per item, in the
 * reference's draw order: s (:257), the gate coefficients of the s-shares then of the 0-shares (item_n_coef[i] each,
 * consecutive in dev_coef from item_coef_off[i]; :259-260), the Gt `msg` (:262), and one r_x per row (dev_rand, :267).
 *   c_0 = msg * E^s ;  per row: c1 = E^lambda_x * egg_alpha_x^r_x,  c2 = g2 * r_x,  c3 = (g2*y_x) * r_x + g2 * omega_x */
int32_t rhip_aw11_encrypt_batch(rhip_ctx* ctx, const rhip_aw11_pk* pk, size_t n_items, size_t total_rows,
                                const uint32_t* dev_item_row_off /*[n_items+1]*/, const uint32_t* dev_item_tree_leaf, const uint32_t* dev_item_tree_gate,
                                const uint32_t* dev_item_n_coef /*[n_items]*/, const uint32_t* dev_path_off, const uint32_t* dev_path_gate,
                                const uint32_t* dev_path_x, const uint32_t* dev_gate_k, const uint32_t* dev_gate_coef_off,
                                const uint32_t* dev_leaf_attr, const rhip_fr* dev_s /*[n_items]*/, const rhip_fr* dev_coef,
                                const uint32_t* dev_item_coef_off /*[n_items]*/, const rhip_fr* dev_rand /*[total_rows]*/,
                                const rhip_gt* dev_msg /*[n_items]*/, rhip_gt* dev_c0 /*[n_items]*/, rhip_gt* dev_c1 /*[total_rows]*/,
                                rhip_g2* dev_c2 /*[total_rows]*/, rhip_g2* dev_c3 /*[total_rows]*/);
/* Group arithmetic of n_items calls of aw11::decrypt (aw11/mod.rs:298-366).  Selection entry e: ciphertext row (sel_ct_row,
 * relative to the item's first row), key attribute row (sel_sk_attr, relative to the key's first row), coefficient c.
 * dev_sk_hash[k] = g1 * h(gid_k) (the reference hashes the gid inside decrypt, :318).
 *   out[i] = c_0 * prod_e C1_e^(-c_e) * FE( ML( -H(gid), sum_e c_e C3_e ) * prod_e ML( c_e K_e, C2_e ) )   
 * The G2 sum and the Gt product run with shared doublings / squarings over the NAFs of the coefficients. */
int32_t rhip_aw11_decrypt_batch(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel,
                                const uint32_t* dev_pair_off /*[n_items+1]*/, const uint32_t* dev_sel_start /*[n_items]*/,
                                const uint32_t* dev_sel_ct_row, const uint32_t* dev_sel_sk_attr, const rhip_fr* dev_sel_coeff,
                                const rhip_gt* dev_ct_c0 /*[n_items]*/, const rhip_gt* dev_ct_c1, const rhip_g2* dev_ct_c2, const rhip_g2* dev_ct_c3,
                                const uint32_t* dev_ct_row_off /*[n_items+1]*/, const rhip_g1* dev_sk_hash /*[n_sk]*/, const rhip_g1* dev_sk_k,
                                const uint32_t* dev_sk_attr_off /*[n_sk+1]*/, const uint32_t* dev_sk_idx /*[n_items] or NULL*/,
                                rhip_gt* dev_out /*[n_items]*/);

/* ---- Level S: the KEM -> DEM step and label hashing on the device (rabe_amd/csrc/engine_sym.hip) ------------------------------
 * Replaces, for batches, src/utils/aes/mod.rs:10-55 (`encrypt_symmetric` :10, `decrypt_symmetric` :29, `kdf` :47: key = SHA3-256(bytes(Gt)),
 * AES-256-GCM, sealed form = nonce(12) || ciphertext || tag(16)) and src/utils/hash/mod.rs:10-31 (`sha3_hash_fr`: Fr::from_slice(SHA3-256)).
 * The Gt of an encrypt / decrypt stays in HBM; only plaintext and record bytes cross PCIe.  No memory table is indexed by a secret
 * (the AES S-box lives in a register of the wave and is read through the lane crossbar).
 *
 * Offsets are byte offsets into the named blob; dev_len[i] = plaintext length of item i.  The host states the shape of the batch:
 * dev_blk_off[n+1] = prefix sums of ceil(len/16) (one lane per 16-byte block), dev_seg_off[n+1] = prefix sums of ceil(blocks/64) (GHASH runs
 * per 64-block segment, folded with powers of H).  dev_ws: rhip_seal_workspace_bytes(n, total_segments) bytes of device scratch. */
/* SHA3-256 (FIPS 202) of n byte strings data[off[i] .. off[i+1]); digests 32 bytes each */
int32_t rhip_sha3_256_batch(rhip_ctx* ctx, size_t n, const uint8_t* dev_data, const uint64_t* dev_off /*[n+1]*/, uint8_t* dev_digest /*[n][32]*/);
/* hash/mod.rs:23-31 sha3_hash_fr: the digest as a big-endian integer reduced mod r, canonical little-endian rhip_fr */
int32_t rhip_sha3_fr_batch(rhip_ctx* ctx, size_t n, const uint8_t* dev_data, const uint64_t* dev_off /*[n+1]*/, rhip_fr* dev_out /*[n]*/);
/* aes/mod.rs:47-55 kdf: key_i = SHA3-256(bytes(gt[idx ? idx[i] : i])), bytes = 12 coefficients as 32 big-endian bytes each */
int32_t rhip_gt_kdf_batch(rhip_ctx* ctx, size_t n, const rhip_gt* dev_gt, const uint32_t* dev_gt_idx /*[n] or NULL*/, uint8_t* dev_keys /*[n][32]*/);
/* AES-256 (FIPS 197) of n blocks under n keys (known-answer tests) */
int32_t rhip_aes256_encrypt_blocks(rhip_ctx* ctx, size_t n, const uint8_t* dev_keys /*[n][32]*/, const uint8_t* dev_in /*[n][16]*/, uint8_t* dev_out /*[n][16]*/);
size_t rhip_seal_workspace_bytes(size_t n, size_t total_segments);
/* AES-256-GCM (SP 800-38D; 96-bit nonce, no AAD) with explicit keys.  seal != 0: in = plaintext at dev_in_off[i], out = nonce || ct || tag at
 * dev_out_off[i] (len_prefix != 0 also writes the u32 length len + 28 in the four bytes BEFORE it: the record's length field);
 * seal == 0: in = the sealed bytes at dev_in_off[i] (nonce read from there), out = plaintext at dev_out_off[i], dev_ok[i] = tag verified
 * (a failed item's plaintext bytes are zeros). */
int32_t rhip_aes256_gcm_batch(rhip_ctx* ctx, int32_t seal, size_t n, const uint8_t* dev_keys /*[n][32]*/, const uint8_t* dev_nonce /*[n][12], seal only*/,
                              const uint8_t* dev_in, const uint64_t* dev_in_off /*[n]*/, uint8_t* dev_out, const uint64_t* dev_out_off /*[n]*/,
                              const uint32_t* dev_len /*[n]*/, const uint32_t* dev_blk_off /*[n+1]*/, size_t total_blocks,
                              const uint32_t* dev_seg_off /*[n+1]*/, size_t total_segments, int32_t len_prefix, uint32_t* dev_ok /*[n], open only*/,
                              void* dev_ws);
/* n calls of encrypt_symmetric(gt_i, plaintext_i) with explicit nonces (the reference draws them from thread_rng, aes/mod.rs:17) */
int32_t rhip_seal_batch(rhip_ctx* ctx, size_t n, const rhip_gt* dev_gt /*[n]*/, const uint8_t* dev_nonce /*[n][12]*/, const uint8_t* dev_pt,
                        const uint64_t* dev_pt_off /*[n]*/, uint8_t* dev_out, const uint64_t* dev_sealed_off /*[n]*/, const uint32_t* dev_len /*[n]*/,
                        const uint32_t* dev_blk_off /*[n+1]*/, size_t total_blocks, const uint32_t* dev_seg_off /*[n+1]*/, size_t total_segments,
                        int32_t len_prefix, void* dev_ws);
/* n calls of decrypt_symmetric(gt[idx ? idx[i] : i], sealed_i); dev_len[i] = sealed length - 28 */
int32_t rhip_open_batch(rhip_ctx* ctx, size_t n, const rhip_gt* dev_gt, const uint32_t* dev_gt_idx /*[n] or NULL*/, const uint8_t* dev_blob,
                        const uint64_t* dev_sealed_off /*[n]*/, uint8_t* dev_pt, const uint64_t* dev_pt_off /*[n]*/, const uint32_t* dev_len /*[n]*/,
                        const uint32_t* dev_blk_off /*[n+1]*/, size_t total_blocks, const uint32_t* dev_seg_off /*[n+1]*/, size_t total_segments,
                        uint32_t* dev_ok /*[n]*/, void* dev_ws);
/* Records of a batch written from their parts on the device (the struct layouts of src/schemes/ac17/mod.rs:58-135 etc. in the canonical byte
 * form of rabe_obj_serialize): byte b of item i's record = dev_map[dev_layout_off[dev_layout[i]] + b], where a map word 0xFF0000vv is the
 * literal byte vv (policy text, names, counts -- one template per distinct policy) and k << 24 | o is byte o of item i's part in source
 * k: dev_src[k] + dev_src_item_off[k * n + i].  rhip_gather_parts is the reverse for a decrypt: part p of a layout copies dev_part_len[p]
 * bytes from record offset dev_part_src[p] to dev_dst[dev_part_k[p]] + dev_dst_item_off[k * n + i] + dev_part_dst[p]. */
int32_t rhip_assemble_records(rhip_ctx* ctx, size_t n, uint8_t* dev_out, const uint64_t* dev_out_off /*[n]*/, const uint32_t* dev_layout /*[n]*/,
                              const uint32_t* dev_layout_off, const uint32_t* dev_map, uint32_t n_src, const uint8_t* const* dev_src,
                              const uint64_t* dev_src_item_off /*[n_src][n]*/);
int32_t rhip_gather_parts(rhip_ctx* ctx, size_t n, const uint8_t* dev_blob, const uint64_t* dev_rec_off /*[n]*/, const uint32_t* dev_layout /*[n]*/,
                          const uint32_t* dev_layout_off, const uint32_t* dev_part_src, const uint32_t* dev_part_dst, const uint32_t* dev_part_len,
                          const uint32_t* dev_part_k, uint8_t* const* dev_dst, const uint64_t* dev_dst_item_off /*[n_dst][n]*/);

/* ---- measurement helper: integer-multiply issue-rate microbenchmark (the roofline denominator) --
 * Runs `iters` dependent-free v_mad_u64_u32 per lane on every CU and returns elapsed milliseconds
 * and the number of multiply-adds executed. */
int32_t rhip_calibrate_mad(rhip_ctx* ctx, int32_t variant, uint32_t iters, double* ms, double* n_ops);

#ifdef __cplusplus
}
#endif
#endif /* RABE_HIP_H */
