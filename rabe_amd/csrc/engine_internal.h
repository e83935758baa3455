// Shared between the engine's translation units (engine.hip: Level E, tables, AC17; engine_jobs.hip: the generic
// pairing-job kernels and the BSW / LSW / AW11 Level B entry points).  Device code is not linked across translation
// units (no -fgpu-rdc): every __device__ function here is static or inline and compiled into each unit that uses it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/rabe_hip.h"
#include "bn254/io.h"
#include "bn254/coop3.h"

using namespace rabe::bn254;
// Minimum resident waves per SIMD every kernel is compiled for (second __launch_bounds__ argument).
// Measured on MI355X: 1 is best -- the Miller / final-exponentiation state
// (Fq12 accumulator, G2 point, line, temporaries) then lives in the 512-entry VGPR+AGPR file instead of
// scratch; asking for 2..4 waves caps the budget at 256..128 registers and the extra scratch traffic costs
// more than the second wave hides (-2 % / -13 % / -20 %).
// G1-only kernels carry a small state (a Jacobian accumulator and a table entry); they are compiled for
// more resident waves so table-gather and dependent-issue latency overlap.
#ifndef RB_G1_WAVES
#define RB_G1_WAVES 4
#endif
#ifndef RB_MIN_WAVES
#define RB_MIN_WAVES 1
#endif
// the fixed-base / summing G2 kernels sit at the edge of 256 registers: two waves per SIMD hide their table gathers and the
// single-wave issue gaps (k_aw11_enc_c3: 15.4 ms with 256 registers, 21.5 ms with 260)
#ifndef RB_G2_WAVES
#define RB_G2_WAVES 2
#endif
// ------------------------------------------------------------------------------------------------
// context
struct rhip_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  int n_cu = 0;
  // grow-only device scratch (Miller values between k_miller and k_final_exp)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // grow-only workspace of k_final_exp (FE_SLOTS Fq12 per lane, [slot][word][lane])
  void* fe_ws = nullptr;
  size_t fe_ws_bytes = 0;
  // grow-only work arenas of the job kernels (engine_jobs.hip: pair lists, running G2 points, scalars)
  enum { N_WORK = 15 };          // 12, 13: the cross-check mode's second result buffer and its snapshot of an in-place factor (engine_jobs.hip); 14: the two-lane Miller kernel's workspace (its own slot: in the cross-check mode it alternates with the one-lane kernel's, and a grow-only slot shared by two sizes would be re-allocated -- a device-wide hipFree -- on every launch)
  void* work[N_WORK] = {};
  size_t work_bytes[N_WORK] = {};
  // optional per-kernel timing (HIP events on the launch stream), for bench.py's roofline leg
  int pairing_mode = 0;   // 0 auto, 1 one lane per pairing, 3 three cooperating lanes per pairing, 6 six lanes per Fq12 accumulator (engine_coop.hip),
                          // 29 reduced radix (engine_rr.hip), 99 cross-check: auto, and every other family of kernels on the same inputs, compared on the device
  void* fe_started = nullptr;      // device counter of resident final-exponentiation waves (rhip_ctx_release_before_final_exp)
  uint32_t* walk_fail = nullptr;    // one-shot (rhip_ctx_collect_walk_verdicts): per-item verdict / count arrays of the next pair-list launch
  uint32_t* walk_count = nullptr;
  bool fe_waiter_poll = true;      // false: the waiter is released when the Miller loops are done, without waiting for the final exponentiation's blocks
  rhip_ctx* fe_waiter = nullptr;   // one-shot (rhip_ctx_release_before_final_exp): released after this context's next Miller launch
  bool early_release = false;      // one-shot (rhip_ctx_release_when_miller_resident): fe_waiter is released when the next reduced-radix Miller launch's blocks are resident
  // two side streams for launch sets whose kernels are independent and too small to fill the chip one by one (rhip_fork / rhip_join,
  // engine.hip): created on first use
  hipStream_t fork[2] = {nullptr, nullptr};
  hipEvent_t fork_ev[3] = {nullptr, nullptr, nullptr};
  bool timing = false;
  struct Pending { std::string name; hipEvent_t e0, e1; };
  std::vector<Pending> pending;
};
// defined in engine.hip
void rhip_ktime_begin(rhip_ctx* ctx, const char* name);
void rhip_ktime_end(rhip_ctx* ctx);
int32_t rhip_fail(rhip_ctx* ctx, hipError_t e, const char* what);
// the mode the kernel-selection predicates see: the cross-check mode (99) selects as "auto" does (engine_jobs.hip: run_pair_lists forces the
// other families one after the other)
static inline int rhip_mode(const rhip_ctx* ctx) { return ctx->pairing_mode == 99 ? 0 : ctx->pairing_mode; }
// fork: the side streams wait for everything queued on ctx->stream so far; join: ctx->stream waits for what was queued on them since.
// Between the two, `RhipOnFork f(ctx, i)` makes ctx->stream side stream i for the launches in its scope.
int32_t rhip_fork(rhip_ctx* ctx);
int32_t rhip_join(rhip_ctx* ctx);
struct RhipOnFork {
  rhip_ctx* c; hipStream_t main;
  RhipOnFork(rhip_ctx* ctx, int i) : c(ctx), main(ctx->stream) { c->stream = c->fork[i]; }
  ~RhipOnFork() { c->stream = main; }
};
int32_t rhip_ensure_scratch(rhip_ctx* ctx, size_t bytes);
int32_t rhip_ensure_fe_ws(rhip_ctx* ctx, size_t bytes);
int32_t rhip_ensure_work(rhip_ctx* ctx, int slot, size_t bytes, void** out);
bool rhip_use_c3(const rhip_ctx* ctx, size_t n_pairs);
#define ktime_begin rhip_ktime_begin
#define ktime_end rhip_ktime_end
#define fail rhip_fail
#define ensure_scratch rhip_ensure_scratch
#define ensure_fe_ws rhip_ensure_fe_ws
#define KLAUNCH(ctx, NAME, ...)            \
  do {                                     \
    ktime_begin(ctx, NAME);                \
    hipLaunchKernelGGL(__VA_ARGS__);       \
    ktime_end(ctx);                        \
    LAUNCH_CHECK(ctx, NAME);               \
  } while (0)

#define HIP_TRY(ctx, call)                                   \
  do {                                                       \
    hipError_t e__ = (call);                                 \
    if (e__ != hipSuccess) return fail(ctx, e__, #call);     \
  } while (0)
#define LAUNCH_CHECK(ctx, name)                              \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return fail(ctx, e__, name);      \
  } while (0)

#define NEED(ctx) do { if (!(ctx)) return RHIP_ERR_ARG; } while (0)
static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

// ------------------------------------------------------------------------------------------------
// internal (Montgomery) storage of points / Gt values in HBM
struct G1M { uint32_t l[16]; };   // x, y Montgomery
struct G2M { uint32_t l[32]; };
struct G1JM { uint32_t l[24]; };  // Jacobian x, y, z Montgomery
struct GtM { uint32_t l[96]; };

__device__ __forceinline__ Fp ld_fp_m(const uint32_t* p) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = p[i];
  return r;
}
__device__ __forceinline__ void st_fp_m(uint32_t* p, const Fp& a) {
#pragma unroll
  for (int i = 0; i < 8; i++) p[i] = a.v[i];
}
__device__ __forceinline__ Fp2 ld_fp2_m(const uint32_t* p) { return Fp2{ld_fp_m(p), ld_fp_m(p + 8)}; }
__device__ __forceinline__ void st_fp2_m(uint32_t* p, const Fp2& a) { st_fp_m(p, a.c0); st_fp_m(p + 8, a.c1); }
__device__ __forceinline__ G1Aff ld_g1_m(const G1M* p) { return G1Aff{ld_fp_m(p->l), ld_fp_m(p->l + 8)}; }
__device__ __forceinline__ void st_g1_m(G1M* p, const G1Aff& a) { st_fp_m(p->l, a.x); st_fp_m(p->l + 8, a.y); }
__device__ __forceinline__ G2Aff ld_g2_m(const G2M* p) { return G2Aff{ld_fp2_m(p->l), ld_fp2_m(p->l + 16)}; }
__device__ __forceinline__ void st_g2_m(G2M* p, const G2Aff& a) { st_fp2_m(p->l, a.x); st_fp2_m(p->l + 16, a.y); }
static __device__ __noinline__ Fp12 ld_gt_m(const GtM* p) {
  Fp12 r;
  r.c0.a0 = ld_fp2_m(p->l);      r.c0.a1 = ld_fp2_m(p->l + 16); r.c0.a2 = ld_fp2_m(p->l + 32);
  r.c1.a0 = ld_fp2_m(p->l + 48); r.c1.a1 = ld_fp2_m(p->l + 64); r.c1.a2 = ld_fp2_m(p->l + 80);
  return r;
}
static __device__ __noinline__ void st_gt_m(GtM* p, const Fp12& a) {
  st_fp2_m(p->l, a.c0.a0);      st_fp2_m(p->l + 16, a.c0.a1); st_fp2_m(p->l + 32, a.c0.a2);
  st_fp2_m(p->l + 48, a.c1.a0); st_fp2_m(p->l + 64, a.c1.a1); st_fp2_m(p->l + 80, a.c1.a2);
}
// A scalar record is a canonical Fr (< r); every chain below it (NAF masks compute 3k in 256 bits, window digits, GLV) relies on
// k < 2^254.  The raw C ABI takes rhip_fr arrays from anywhere, so the load itself brings any 256-bit word below r: at most five
// conditional subtractions (2^256 / r < 5.3), a no-op on canonical input -- k * P is then the group's answer for every input.
__device__ __forceinline__ void ld_scalar(uint32_t k[8], const rhip_fr* p) {
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = p->l[i];
  if (k[7] >= FrParams::mod(7)) {          // only words with a top limb >= r's can be >= r: canonical input almost never enters
    for (int t = 0; t < 5; t++) {
      uint32_t d[8], borrow = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) d[i] = subb32(k[i], FrParams::mod(i), borrow);
      if (borrow) break;
#pragma unroll
      for (int i = 0; i < 8; i++) k[i] = d[i];
    }
  }
}

// out[item] = (mul_in ? mul_in[item] : 1) * FE( prod_{j in [off[item], off[item+1])} mill[j] ), canonical.
// The exponentiation's Fq12 values live in the context's workspace (final_exponentiation_ws, bn254/pairing.h).
// The home of the Fq12 value a lane is working on (bn254/pairing.h: the FA interface).  The Fq12 kernels run one wave per
// SIMD -- blocks of ONE wave, 512 registers -- so a wave owns a quarter of the CU's 160 KB of LDS: 576 bytes per lane hold the
// value's two halves and one parked Fq6, laid out [quad][lane of the wave] (conflict-free 16-byte accesses).  What would
// otherwise be evicted to scratch (HBM) between the Fq6-level steps is fetched from here instead.  Only kernels launched with
// 64-thread blocks may use it.
static __shared__ uint4 rb_facc_lds[36 * 64];
// the same home for blocks of FOUR waves (k_final_exp): a block then owns a whole CU -- its four SIMDs and 144 KB of the LDS -- so
// that a launch of few items packs into few CUs and leaves the others to whatever else is running (rhip_ctx_release_before_final_exp:
// workgroups of the fixed-base kernels need a wave slot on every SIMD of a CU, and one resident final-exponentiation wave per CU
// would lock them out of the whole chip)
static __shared__ uint4 rb_facc_lds4[4 * 36 * 64];
template <int W> __device__ __forceinline__ uint4* rb_facc_base();
template <> __device__ __forceinline__ uint4* rb_facc_base<1>() { return rb_facc_lds + threadIdx.x; }
template <> __device__ __forceinline__ uint4* rb_facc_base<4>() { return rb_facc_lds4 + (threadIdx.x >> 6) * (36 * 64) + (threadIdx.x & 63); }
template <int W>
struct LdsHomeT {
  __device__ __forceinline__ Fp ld_fp(int q0) const {
    const uint4* p = rb_facc_base<W>() + q0 * 64;
    const uint4 a = p[0], b = p[64];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
  }
  __device__ __forceinline__ void st_fp(int q0, const Fp& a) const {
    uint4* p = rb_facc_base<W>() + q0 * 64;
    p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    p[64] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
  }
  // coefficient i (an Fq2) of the value: quads [4 i, 4 i + 4); the parked Fq6: quads [24, 36)
  __device__ __forceinline__ Fp2 ld_q4(int q0) const { return Fp2{ld_fp(q0), ld_fp(q0 + 2)}; }
  __device__ __forceinline__ void st_q4(int q0, const Fp2& a) const { st_fp(q0, a.c0); st_fp(q0 + 2, a.c1); }
  __device__ __forceinline__ Fp6 ld_q12(int q0) const { return Fp6{ld_q4(q0), ld_q4(q0 + 4), ld_q4(q0 + 8)}; }
  __device__ __forceinline__ void st_q12(int q0, const Fp6& a) const { st_q4(q0, a.a0); st_q4(q0 + 4, a.a1); st_q4(q0 + 8, a.a2); }
  __device__ __forceinline__ Fp2 ld_f2(int i) const { return ld_q4(4 * i); }
  __device__ __forceinline__ void st_f2(int i, const Fp2& v) const { st_q4(4 * i, v); }
  __device__ __forceinline__ Fp6 ld_f6(int h) const { return ld_q12(12 * h); }
  __device__ __forceinline__ void st_f6(int h, const Fp6& v) const { st_q12(12 * h, v); }
  __device__ __forceinline__ Fp6 ld_x() const { return ld_q12(24); }
  __device__ __forceinline__ void st_x(const Fp6& v) const { st_q12(24, v); }
  __device__ __forceinline__ void fence() const { asm volatile("" ::: "memory"); }
};
typedef LdsHomeT<1> LdsHome;
template <class HOME>
struct DevWsT {
  uint4* base;         // + lane; [slot][quad of words][lane]: one 16-byte access per lane, contiguous over the wave
  size_t stride;       // lanes (padded to 64)
  __device__ __forceinline__ Fp2 ld2(const uint4* p) const {
    const uint4 q0 = p[0], q1 = p[stride], q2 = p[2 * stride], q3 = p[3 * stride];
    Fp2 c;
    c.c0.v[0] = q0.x; c.c0.v[1] = q0.y; c.c0.v[2] = q0.z; c.c0.v[3] = q0.w;
    c.c0.v[4] = q1.x; c.c0.v[5] = q1.y; c.c0.v[6] = q1.z; c.c0.v[7] = q1.w;
    c.c1.v[0] = q2.x; c.c1.v[1] = q2.y; c.c1.v[2] = q2.z; c.c1.v[3] = q2.w;
    c.c1.v[4] = q3.x; c.c1.v[5] = q3.y; c.c1.v[6] = q3.z; c.c1.v[7] = q3.w;
    return c;
  }
  __device__ __forceinline__ void st2(uint4* p, const Fp2& c) const {
    p[0] = make_uint4(c.c0.v[0], c.c0.v[1], c.c0.v[2], c.c0.v[3]);
    p[stride] = make_uint4(c.c0.v[4], c.c0.v[5], c.c0.v[6], c.c0.v[7]);
    p[2 * stride] = make_uint4(c.c1.v[0], c.c1.v[1], c.c1.v[2], c.c1.v[3]);
    p[3 * stride] = make_uint4(c.c1.v[4], c.c1.v[5], c.c1.v[6], c.c1.v[7]);
  }
  // half h of slot: quads [12 h, 12 h + 12)
  __device__ __forceinline__ Fp6 ld6(int slot, int h) const {
    const uint4* p = base + ((size_t)slot * 24 + 12 * h) * stride;
    return Fp6{ld2(p), ld2(p + 4 * stride), ld2(p + 8 * stride)};
  }
  __device__ __forceinline__ void st6(int slot, int h, const Fp6& a) const {
    uint4* p = base + ((size_t)slot * 24 + 12 * h) * stride;
    st2(p, a.a0); st2(p + 4 * stride, a.a1); st2(p + 8 * stride, a.a2);
  }
  __device__ __forceinline__ Fp12 ld(int slot) const { return Fp12{ld6(slot, 0), ld6(slot, 1)}; }
  __device__ __forceinline__ void st(int slot, const Fp12& a) const { st6(slot, 0, a.c0); st6(slot, 1, a.c1); }
  __device__ __forceinline__ HOME home() const { return HOME{}; }
};
typedef DevWsT<LdsHome> DevWs;
struct GtM;
int32_t rhip_build_w16_gt(rhip_ctx* ctx, const GtM* t8, GtM* t16);
struct G2M;
int32_t rhip_build_w16_g2(rhip_ctx* ctx, const G2M* t8, G2M* t16);
int32_t rhip_launch_final_exp(rhip_ctx* ctx, size_t n_items, const uint32_t* off, uint32_t stride, const GtM* mill, const rhip_gt* mul_in,
                              rhip_gt* out);
#define launch_final_exp rhip_launch_final_exp
// ------------------------------------------------------------------------------------------------
// fixed-base tables: T[w][d-1] = (d * 256^w) * base, d = 1..255, w = 0..31, affine Montgomery.
#define TBL_WINDOWS 32
#define TBL_DIGITS 255
#define TBL16_WINDOWS 16
#define TBL16_DIGITS 65535
// dev: 8-bit windows; dev16: optional 16-bit windows; wide: optional signed w-bit windows (17 <= w <= 27)
struct rhip_g1_table { rhip_ctx* ctx; G1M* dev; G1M* dev16; G1M* wide; int wide_bits; };
// signed w-bit windows: k = sum_i d_i 2^(w i), -2^(w-1) < d_i <= 2^(w-1); T[i][|d|-1] = (|d| 2^(w i)) * base, the sign is
// applied to y on the fly.  n = ceil(254 / w) windows of 2^(w-1) entries (the top one only needs 2^(254-w(n-1)) of them).
__host__ __device__ inline int wide_windows(int w) { return (254 + w - 1) / w; }
__host__ __device__ inline size_t wide_count(int w, int i) {
  const int n = wide_windows(w);
  return (i < n - 1) ? ((size_t)1 << (w - 1)) : ((size_t)1 << (254 - w * (n - 1)));
}
__host__ __device__ inline size_t wide_offset(int w, int i) { return (size_t)i << (w - 1); }   // windows below the top are full
struct rhip_g2_table { rhip_ctx* ctx; G2M* dev; G2M* dev16; };   // dev16: optional 16-bit windows (134 MB)
struct rhip_gt_table { rhip_ctx* ctx; GtM* dev; GtM* dev16; };   // dev16: optional 16-bit windows (402 MB)

// ---- fixed-base Gt powers on the lane's home value (LdsHome above): home *= entry for every non-zero window digit.
// `started` false: the first entry is copied instead of multiplied (and stays false when the scalar is 0).
struct GtMOperand {
  const GtM* e;
  bool conj = false;          // the entry's inverse (members of Gt are unitary: the inverse is the conjugate)
  __device__ __forceinline__ Fp6 half(int h) const {
    const uint32_t* p = e->l + 48 * h;
    const Fp6 v{ld_fp2_m(p), ld_fp2_m(p + 16), ld_fp2_m(p + 32)};
    return (conj && h == 1) ? fp6_neg(v) : v;
  }
};
__device__ __forceinline__ void home_mul_entry(const GtM* e) { facc_mul(LdsHome{}, GtMOperand{e}); }
__device__ __forceinline__ void home_take(bool& started, const GtM* e) {
  if (started) home_mul_entry(e);
  else { facc_set(LdsHome{}, GtMOperand{e}); started = true; }
}
__device__ __forceinline__ void home_take_signed(bool& started, const GtM* e, bool inverse) {
  if (started) facc_mul(LdsHome{}, GtMOperand{e, inverse});
  else { facc_set(LdsHome{}, GtMOperand{e, inverse}); started = true; }
}
static __device__ __noinline__ void home_table_pow_gt_w16(bool& started, const GtM* tbl, const uint32_t k[8]) {
#pragma unroll 1
  for (int w = 0; w < TBL16_WINDOWS; w++) {
    uint32_t word;
    switch (w >> 1) {
      case 0: word = k[0]; break;
      case 1: word = k[1]; break;
      case 2: word = k[2]; break;
      case 3: word = k[3]; break;
      case 4: word = k[4]; break;
      case 5: word = k[5]; break;
      case 6: word = k[6]; break;
      default: word = k[7]; break;
    }
    const uint32_t d = (w & 1) ? (word >> 16) : (word & 0xffffu);
    if (d) home_take(started, tbl + (size_t)w * TBL16_DIGITS + (d - 1));
  }
}
__device__ __forceinline__ uint32_t scalar_byte(const uint32_t k[8], int w) {
  uint32_t word;
  switch (w >> 2) {
    case 0: word = k[0]; break;
    case 1: word = k[1]; break;
    case 2: word = k[2]; break;
    case 3: word = k[3]; break;
    case 4: word = k[4]; break;
    case 5: word = k[5]; break;
    case 6: word = k[6]; break;
    default: word = k[7]; break;
  }
  return (word >> (8 * (w & 3))) & 255u;
}
// sum of <= 32 table entries selected by the bytes of the canonical scalar k
static __device__ __noinline__ G1Jac table_mul_g1(const G1M* tbl, const uint32_t k[8]) {
  G1Jac acc = jac_inf<Fp>();
  for (int w = 0; w < TBL_WINDOWS; w++) {
    uint32_t d = scalar_byte(k, w);
    if (d) acc = jac_add_aff(acc, ld_g1_m(tbl + w * TBL_DIGITS + (d - 1)));
  }
  return acc;
}
// 16-bit digits: 16 mixed additions.  The entry is fetched where it is used: holding the next entry across the addition (a
// software prefetch) costs 16 of the 128 registers these kernels run with and came out slower than the gather latency it hid
// (four waves per SIMD cover it): row kernel 13.0 -> 12.1 ms.
// The *_inl forms are for kernels whose lane walks several tables in a row (the AC17 row kernels): as out-of-line functions
// these return their point through a stack slot, and the compiler then keeps the ACCUMULATOR in that slot -- 96 bytes of scratch
// written per window (profiles/r02k_pmc_traffic.txt: 14 GB of write-back per launch of k_ac17_enc_rows, 22 x its results).
static __device__ __forceinline__ G1Jac table_mul_g1_w16_inl(const G1M* tbl, const uint32_t k[8]) {
  G1Jac acc = jac_inf<Fp>();
#pragma unroll 1
  for (int w = 0; w < TBL16_WINDOWS; w++) {
    uint32_t word;
    switch (w >> 1) {
      case 0: word = k[0]; break;
      case 1: word = k[1]; break;
      case 2: word = k[2]; break;
      case 3: word = k[3]; break;
      case 4: word = k[4]; break;
      case 5: word = k[5]; break;
      case 6: word = k[6]; break;
      default: word = k[7]; break;
    }
    const uint32_t d = (w & 1) ? (word >> 16) : (word & 0xffffu);
    if (d) acc = g1_madd_inl(acc, ld_g1_m(tbl + (size_t)w * TBL16_DIGITS + (d - 1)));
  }
  return acc;
}
static __device__ __noinline__ G1Jac table_mul_g1_w16(const G1M* tbl, const uint32_t k[8]) { return table_mul_g1_w16_inl(tbl, k); }
// signed w-bit digits: ceil(254/w) mixed additions (11 for w = 24, 10 for w = 26); the digit's sign flips y
__device__ __forceinline__ uint32_t scalar_bits(const uint32_t k[8], int b, int w) {
  const int word = b >> 5, sh = b & 31;
  uint32_t lo, hi;
  switch (word) {
    case 0: lo = k[0]; hi = k[1]; break;
    case 1: lo = k[1]; hi = k[2]; break;
    case 2: lo = k[2]; hi = k[3]; break;
    case 3: lo = k[3]; hi = k[4]; break;
    case 4: lo = k[4]; hi = k[5]; break;
    case 5: lo = k[5]; hi = k[6]; break;
    case 6: lo = k[6]; hi = k[7]; break;
    default: lo = k[7]; hi = 0; break;
  }
  const uint64_t v = (((uint64_t)hi << 32) | lo) >> sh;
  return (uint32_t)v & ((1u << w) - 1u);
}
static __device__ __forceinline__ G1Jac table_mul_g1_wide_inl(const G1M* tbl, const uint32_t k[8], int w) {
  const int n = wide_windows(w);
  const uint32_t half = 1u << (w - 1);
  G1Jac acc = jac_inf<Fp>();
  uint32_t carry = 0;
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    const uint32_t raw = scalar_bits(k, w * i, w) + carry;
    carry = raw > half ? 1u : 0u;
    const uint32_t mag = carry ? (1u << w) - raw : raw;
    if (mag) {
      G1Aff e = ld_g1_m(tbl + wide_offset(w, i) + (mag - 1));        // fetched where it is used (see table_mul_g1_w16)
      if (carry) e.y = neg(e.y);
      acc = g1_madd_inl(acc, e);
    }
  }
  return acc;
}
static __device__ __noinline__ G1Jac table_mul_g1_wide(const G1M* tbl, const uint32_t k[8], int w) { return table_mul_g1_wide_inl(tbl, k, w); }
// The same signed w-bit recoding for tables with FULL windows (8 <= w <= 14; per-attribute bases of AW11): n = sgn_windows(w) windows
// of 2^(w-1) entries each, T[i][|d|-1] = (|d| 2^(w i)) * base.  n w >= 255, so the last carry always lands in a window.
__host__ __device__ inline int sgn_windows(int w) { return (255 + w - 1) / w; }
__host__ __device__ inline size_t sgn_entries(int w) { return (size_t)sgn_windows(w) << (w - 1); }
static __device__ __noinline__ G2Jac table_mul_g2_signed(G2Jac acc, const G2M* tbl, const uint32_t k[8], int w) {
  const int n = sgn_windows(w);
  const uint32_t half = 1u << (w - 1);
  uint32_t carry = 0;
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    const int bit = w * i;
    const uint32_t raw = scalar_bits(k, bit, w) + carry;          // bit <= w (n - 1) < 255; bits past 255 read as zero
    carry = raw > half ? 1u : 0u;
    const uint32_t mag = carry ? (1u << w) - raw : raw;
    if (mag) {
      G2Aff e = ld_g2_m(tbl + ((size_t)i << (w - 1)) + (mag - 1));
      if (carry) e.y = fp2_neg(e.y);
      acc = jac_add_aff(acc, e);
    }
  }
  return acc;
}
static __device__ __noinline__ G2Jac table_mul_g2(const G2M* tbl, const uint32_t k[8]) {
  G2Jac acc = jac_inf<Fp2>();
  for (int w = 0; w < TBL_WINDOWS; w++) {
    uint32_t d = scalar_byte(k, w);
    if (d) acc = jac_add_aff(acc, ld_g2_m(tbl + w * TBL_DIGITS + (d - 1)));
  }
  return acc;
}
static __device__ __noinline__ G2Jac table_mul_g2_w16(const G2M* tbl, const uint32_t k[8]) {
  G2Jac acc = jac_inf<Fp2>();
#pragma unroll 1
  for (int w = 0; w < TBL16_WINDOWS; w++) {
    uint32_t word;
    switch (w >> 1) {
      case 0: word = k[0]; break;
      case 1: word = k[1]; break;
      case 2: word = k[2]; break;
      case 3: word = k[3]; break;
      case 4: word = k[4]; break;
      case 5: word = k[5]; break;
      case 6: word = k[6]; break;
      default: word = k[7]; break;
    }
    const uint32_t d = (w & 1) ? (word >> 16) : (word & 0xffffu);
    if (d) acc = jac_add_aff(acc, ld_g2_m(tbl + (size_t)w * TBL16_DIGITS + (d - 1)));
  }
  return acc;
}
static __device__ __noinline__ void home_table_pow_gt(bool& started, const GtM* tbl, const uint32_t k[8]) {
#pragma unroll 1
  for (int w = 0; w < TBL_WINDOWS; w++) {
    const uint32_t d = scalar_byte(k, w);
    if (d) home_take(started, tbl + w * TBL_DIGITS + (d - 1));
  }
}
static __device__ __noinline__ void home_table_pow_gt_signed(bool& started, const GtM* tbl, const uint32_t k[8], int w) {
  const int n = sgn_windows(w);
  const uint32_t half = 1u << (w - 1);
  uint32_t carry = 0;
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    const uint32_t raw = scalar_bits(k, w * i, w) + carry;
    carry = raw > half ? 1u : 0u;
    const uint32_t mag = carry ? (1u << w) - raw : raw;
    if (mag) home_take_signed(started, tbl + ((size_t)i << (w - 1)) + (mag - 1), carry != 0);
  }
}
// the value forms (64-thread blocks only: they go through the home)
__device__ __forceinline__ Fp12 home_result(bool started) { return started ? facc_get(LdsHome{}) : fp12_one(); }
__device__ __forceinline__ void home_put(const Fp12& v) {
  const LdsHome h{};
  h.st_f6(0, v.c0);
  h.st_f6(1, v.c1);
  h.fence();
}
// ---- batched inversion across a 256-thread block (Montgomery's trick over LDS + one wave-level scan).
// Every thread of the block contributes one non-zero Fp value and gets its inverse back; the block pays ONE
// field inversion (fp.h: inv -- a binary extended Euclid, ~45 multiplication-equivalents, computed on a wave-uniform value by
// wave 0 while the other waves wait at the barrier and free their issue slots for other resident blocks) instead of one per thread.
__device__ __forceinline__ Fp shfl_up_fp(const Fp& x, int d) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl_up((int)x.v[i], d);
  return r;
}
__device__ __forceinline__ Fp shfl_down_fp(const Fp& x, int d) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl_down((int)x.v[i], d);
  return r;
}
__device__ __forceinline__ Fp shfl_fp(const Fp& x, int src) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl((int)x.v[i], src);
  return r;
}
__device__ __forceinline__ Fp sel_fp(bool c, const Fp& a, const Fp& b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
// sh: 8 x 256 words (limb-major, conflict-free).  All 256 threads must call this.
static __device__ __noinline__ Fp block_batch_inverse_256(uint32_t (*sh)[256], const Fp& mine) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; i++) sh[i][tid] = mine.v[i];
  __syncthreads();
  if (tid < 64) {
    Fp p[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int i = 0; i < 8; i++) p[e].v[i] = sh[i][4 * tid + e];
    const Fp q1 = mul(p[0], p[1]);
    const Fp q2 = mul(q1, p[2]);
    const Fp q3 = mul(q2, p[3]);
    // inclusive prefix / suffix products of q3 over the 64 lanes
    Fp pre = q3, suf = q3;
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
      Fp u = shfl_up_fp(pre, d);
      Fp w = shfl_down_fp(suf, d);
      Fp pu = mul(pre, u);
      Fp sw = mul(suf, w);
      pre = sel_fp(tid >= d, pu, pre);
      suf = sel_fp(tid + d < 64, sw, suf);
    }
    const Fp total_inv = inv(shfl_fp(pre, 63));
    // exclusive prefix / suffix
    Fp ex_pre = shfl_up_fp(pre, 1), ex_suf = shfl_down_fp(suf, 1);
    ex_pre = sel_fp(tid >= 1, ex_pre, one<FpParams>());
    ex_suf = sel_fp(tid < 63, ex_suf, one<FpParams>());
    const Fp iq3 = mul(total_inv, mul(ex_pre, ex_suf));       // 1 / q3 of this lane
    const Fp ip3 = mul(iq3, q2);
    const Fp iq2 = mul(iq3, p[3]);
    const Fp ip2 = mul(iq2, q1);
    const Fp iq1 = mul(iq2, p[2]);
    const Fp ip1 = mul(iq1, p[0]);
    const Fp ip0 = mul(iq1, p[1]);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      sh[i][4 * tid] = ip0.v[i];
      sh[i][4 * tid + 1] = ip1.v[i];
      sh[i][4 * tid + 2] = ip2.v[i];
      sh[i][4 * tid + 3] = ip3.v[i];
    }
  }
  __syncthreads();
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = sh[i][tid];
  return r;
}

// Block size of the G1 row kernels = the group that shares one field inversion (computed by wave 0 while the block's other
// waves wait at the barrier).  With the Fermat inversion (361 multiplications) that fixed cost asked for 512-thread blocks; the
// binary-Euclid inversion (~45 multiplication-equivalents, fp.h) makes it cheap, and smaller blocks overlap their barrier phases
// better: 256 threads 11.8 ms, 512 threads 12.2 ms per 16-step launch of k_ac17_enc_rows.
#ifndef RB_ROWS_BLOCK
#define RB_ROWS_BLOCK 256
#endif
// Generalisation of block_batch_inverse_256 to NT = 64 E threads: lane j of wave 0 owns elements j, j + 64, ...,
// j + 64 (E - 1) (conflict-free), keeps their running products in a second LDS array, joins the 64-lane scan, and
// back-substitutes.  lds: 2 x 8 x NT words.  All NT threads must call this.
template <int NT>
__device__ __noinline__ Fp block_batch_inverse_n(uint32_t* lds, const Fp& mine) {
  uint32_t (*val)[NT] = (uint32_t (*)[NT])lds;
  uint32_t (*pre)[NT] = (uint32_t (*)[NT])(lds + 8 * NT);
  constexpr int E = NT / 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; i++) val[i][tid] = mine.v[i];
  __syncthreads();
  if (tid < 64) {
    Fp run;
#pragma unroll
    for (int i = 0; i < 8; i++) run.v[i] = val[i][tid];
#pragma unroll 1
    for (int e = 1; e < E; e++) {
      Fp v;
#pragma unroll
      for (int i = 0; i < 8; i++) { pre[i][tid + 64 * (e - 1)] = run.v[i]; v.v[i] = val[i][tid + 64 * e]; }
      run = mul(run, v);
    }
    Fp prefix = run, suffix = run;
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
      Fp u = shfl_up_fp(prefix, d);
      Fp w = shfl_down_fp(suffix, d);
      Fp pu = mul(prefix, u);
      Fp sw = mul(suffix, w);
      prefix = sel_fp(tid >= d, pu, prefix);
      suffix = sel_fp(tid + d < 64, sw, suffix);
    }
    const Fp total_inv = inv(shfl_fp(prefix, 63));
    Fp ex_pre = shfl_up_fp(prefix, 1), ex_suf = shfl_down_fp(suffix, 1);
    ex_pre = sel_fp(tid >= 1, ex_pre, one<FpParams>());
    ex_suf = sel_fp(tid < 63, ex_suf, one<FpParams>());
    Fp inv_run = mul(total_inv, mul(ex_pre, ex_suf));       // 1 / (product of this lane's E elements)
#pragma unroll 1
    for (int e = E - 1; e >= 1; e--) {
      Fp v, p;
#pragma unroll
      for (int i = 0; i < 8; i++) { v.v[i] = val[i][tid + 64 * e]; p.v[i] = pre[i][tid + 64 * (e - 1)]; }
      const Fp r = mul(inv_run, p);                          // 1 / v_e
      inv_run = mul(inv_run, v);                             // 1 / (v_0 .. v_{e-1})
#pragma unroll
      for (int i = 0; i < 8; i++) val[i][tid + 64 * e] = r.v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) val[i][tid] = inv_run.v[i];
  }
  __syncthreads();
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = val[i][tid];
  return r;
}
// G2 Jacobian -> affine canonical with ONE Fp inversion per 128-thread block: 1 / (a + b u) = (a - b u) / (a^2 + b^2),
// and the norms a^2 + b^2 of the block's 128 z's are inverted together.  All 128 threads must call this.
static __device__ __noinline__ void store_g2_block128(uint32_t* lds, bool active, rhip_g2* out, const G2Jac& r) {
  const bool inf = !active || jac_is_inf(r);
  const Fp norm = inf ? one<FpParams>() : add(sqr(r.z.c0), sqr(r.z.c1));
  const Fp ninv = block_batch_inverse_n<128>(lds, norm);
  if (!active) return;
  if (inf) { store_g2(out->l, aff_inf<Fp2>()); return; }
  const Fp2 zinv{mul(r.z.c0, ninv), neg(mul(r.z.c1, ninv))};
  store_g2(out->l, jac_to_aff_with_zinv(r, zinv));
}
// ---- prepared secret keys: the Miller-loop line coefficients of k_0[j] (fixed per key) are computed once
// (rhip_ac17_sk_prepare) and replayed by every decryption with that key.
struct LineM { uint32_t l[48]; };   // cy, cx, c0 (Fq2 each), Montgomery
struct rhip_ac17_sk_lines {
  rhip_ctx* ctx;
  size_t n_sk;
  LineM* lines;      // [n_sk * 3][RB_MILLER_LINES]
  uint8_t* q_inf;    // [n_sk * 3]
  void* lines29 = nullptr;      // the same triples in the reduced-radix form k_miller_multi_rr replays (engine_rr.hip: rhip_lines_to_rr)
};
// prepared lines of n arbitrary G2 points (rhip_g2_lines_prepare): lines[i * RB_MILLER_LINES + k]
struct rhip_g2_lines {
  rhip_ctx* ctx;
  size_t n;
  LineM* lines;
  uint8_t* q_inf;    // [n]
  void* lines29 = nullptr;      // reduced-radix form (engine_rr.hip: rhip_lines_to_rr)
};
struct DevLineLoad {
  const LineM* base;
  __device__ __forceinline__ LineCoeffs operator()(int k) const {
    const uint32_t* p = base[k].l;
    return LineCoeffs{ld_fp2_m(p), ld_fp2_m(p + 16), ld_fp2_m(p + 32)};
  }
};

// the multi-pairing kernels' launch geometry and the device-made plan of a ragged batch (engine_jobs.hip: k_plan_*, k_miller_multi;
// engine_coop.hip: k_miller_c6 maps its groups to (item, chunk) exactly as k_miller_multi maps its lanes)
#define RB_MILLER_BLOCK 256
struct MillerPlan {
  uint32_t C, W, L, pad;
  uint32_t base[66];            // base[s]: where the entries of s pairs start in the list (sizes descending)
  uint32_t cursor[66];
};
#define RHIP_Q_WALK 0xFFFFFFFFu
#define RHIP_Q_SKIP 0xFFFFFFFEu
// six-lane cooperative pairing kernels (engine_coop.hip, bn254/coop6.h)
bool rhip_use_c6(const rhip_ctx* ctx, size_t n_items, size_t max_pairs);
void rhip_choose_chunks_c6(const rhip_ctx* ctx, size_t n_items, size_t max_pairs, uint32_t* L, uint32_t* C);
int32_t rhip_launch_miller_c6(rhip_ctx* ctx, size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const void* P, const void* Q,
                              const uint32_t* qref, const void* lines, void* ws, void* mill, const MillerPlan* plan, const void* work, const uint32_t* chunk_off,
                              size_t groups);
int32_t rhip_launch_final_exp_c6(rhip_ctx* ctx, size_t n_items, const uint32_t* off, uint32_t stride, const void* mill, const rhip_gt* mul_in, rhip_gt* out,
                                 uint32_t* started);
// reduced-radix Miller kernel (engine_rr.hip, bn254/fp29.h): the drop-in for k_miller_multi on launches that fill the chip; ws / ws_bytes: the
// workspace in k_miller_multi's layout (the walk verdicts read it), its own workspace is sized from it
bool rhip_use_rr(const rhip_ctx* ctx);
bool rhip_want_lines29(const rhip_ctx* ctx);
// engine_rr2.hip: the reduced-radix Miller kernel with one (item, chunk) unit on two lanes (two waves per SIMD)
bool rhip_use_rr2(const rhip_ctx* ctx, uint32_t c_max);
int32_t rhip_launch_miller_rr2(rhip_ctx* ctx, size_t n_items, uint32_t L, uint32_t C, uint32_t c_max, const uint32_t* pair_off, uint32_t uniform, const void* P, const void* Q,
                               const uint32_t* qref, const void* lines, const void* lines29, void* ws, void* mill, const MillerPlan* plan,
                               const void* work, const uint32_t* chunk_off, size_t units, uint32_t* started, size_t plan_pairs, uint32_t plan_c_lo);
int32_t rhip_launch_miller_rr(rhip_ctx* ctx, size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const void* P, const void* Q,
                              const uint32_t* qref, const void* lines, const void* lines29, void* ws, size_t ws_bytes, void* mill, const MillerPlan* plan,
                              const void* work, const uint32_t* chunk_off, size_t lanes, uint32_t* started = nullptr);
int32_t rhip_launch_final_exp_rr(rhip_ctx* ctx, size_t n_items, const uint32_t* off, uint32_t stride, const void* mill, const rhip_gt* mul_in, rhip_gt* out,
                                 uint32_t* started);
// defined in engine.hip: consumes a pending rhip_ctx_release_before_final_exp request in front of the launch of `blocks` blocks
int32_t rhip_take_waiter(rhip_ctx* ctx, size_t blocks, uint32_t** started_out);
// prepared line triples (LineM, 8 x 32-bit limbs) converted once into the records k_miller_multi_rr replays; *out is hipMalloc'ed
int32_t rhip_lines_to_rr(rhip_ctx* ctx, size_t n_lines, const void* lines, void** out);

int32_t rhip_launch_gt_is_member_c6(rhip_ctx* ctx, size_t n, const rhip_gt* a, uint32_t* ok);
bool rhip_use_c6_gt_pow(const rhip_ctx* ctx, size_t n_items);
int32_t rhip_launch_gt_table_pow_c6(rhip_ctx* ctx, const void* t0, const void* t1, int w16, size_t n_items, const rhip_fr* k, uint32_t kstride, const rhip_gt* mul_in,
                                    rhip_gt* out);
// known-answer check of the field arithmetic on every SIMD (engine_coop.hip; once per device and process)
int32_t rhip_device_selftest(rhip_ctx* ctx, uint32_t* simds, uint32_t* mismatches);

// the pairwise AC17 decrypt paths of engine.hip (one lane per pairing / per pairing couple), behind the public entry points
// of engine_jobs.hip
struct rhip_ac17_sk_lines;
int32_t rhip_ac17_cp_decrypt_batch_lanes6(rhip_ctx* ctx, size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c, const uint32_t* ct_row_off,
                                          const rhip_gt* ct_cp, const rhip_g2* sk_k0, const rhip_g1* sk_k, const uint32_t* sk_row_off,
                                          const rhip_g1* sk_kp, const uint32_t* sk_idx, const uint32_t* ct_sel, const uint32_t* ct_sel_off,
                                          const uint32_t* sk_sel, const uint32_t* sk_sel_off, rhip_gt* out);
int32_t rhip_ac17_cp_decrypt_batch_prepared_lanes3(rhip_ctx* ctx, size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c,
                                                   const uint32_t* ct_row_off, const rhip_gt* ct_cp, const rhip_ac17_sk_lines* sk_lines,
                                                   const rhip_g1* sk_k, const uint32_t* sk_row_off, const rhip_g1* sk_kp, const uint32_t* sk_idx,
                                                   const uint32_t* ct_sel, const uint32_t* ct_sel_off, const uint32_t* sk_sel,
                                                   const uint32_t* sk_sel_off, rhip_gt* out);
