// rabe_amd engine, third translation unit: the SIX-LANE COOPERATIVE pairing kernels (bn254/coop6.h).
//
//   k_miller_c6      group of six lanes = (item, chunk of its pairs): all pairs of the chunk on ONE Fq12 accumulator whose six Fq2
//                    coefficients live one per lane; ten groups per wave (lanes 60-63 idle)
//   k_final_exp_c6   group = item: product of its chunk values + final exponentiation, the same chain value by value
//
// They compute what k_miller_multi (engine_jobs.hip) and k_final_exp (engine.hip) compute -- the same field elements, hence the same
// bytes -- with the work of one item spread over six lanes: the dependency chain of a lone ciphertext's decrypt
// (`ac17::cp_decrypt`, src/schemes/ac17/mod.rs:385-430; bsw/mod.rs:260-318; lsw/mod.rs:228-290; aw11/mod.rs:298-366) is ~six times
// shorter, a lane needs ~1/3 of the registers (two waves per SIMD instead of one), and nothing of the accumulator's arithmetic
// is Fq6 / Fq12 Karatsuba glue (coop6.h: one lazy reduction per dot product).
// The group's slots are LDS rows [row][quad][lane] (16-byte quads: conflict-free for the lane-permuted and the broadcast reads);
// a wave's LDS traffic is ordered by the hardware, so the exchange inside a group needs no barrier instruction.
// There is no CPU fallback in this file.
#include "engine_internal.h"
#include "bn254/coop6.h"
#include "bn254/selftest.h"
#include <mutex>

// This file is compiled TWICE (engine_coop_w1.hip includes it with RB_C6_W1_UNIT): the kernels below exist in a variant built for two
// resident waves per SIMD (256 registers: launches that fill the chip) and one built for ONE (the whole 512-entry file: nothing of the
// Miller loop spills -- 12 % faster when a launch puts at most one wave on a SIMD anyway).  Two translation units because device
// functions are not linked across units: every out-of-line function (c6_dot, fp2_mul_regs, fp6_inv, ...) then exists once per variant
// with that variant's register budget -- in one unit the shared callees take the larger budget and the two-wave kernels lose their occupancy.
#ifdef RB_C6_W1_UNIT
#define RB_C6_WAVES 1
#define C6K(name) name##_w1
#else
#ifndef RB_C6_WAVES
#define RB_C6_WAVES 2
#endif
#define C6K(name) name
#endif
#define C6_GROUPS 10                                   // groups of six lanes per wave
#define C6_LDS_QUADS (C6_ROWS * 4 * 64)                // 20 KB per wave: eight waves per CU

// LDS pointers keep their address space across the out-of-line dot-product function (a generic pointer would turn every slot
// access into a flat_load / flat_store)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_quad;
struct DevCX6 {
  lds_quad* base;       // the wave's rows + the first lane of my group (reads)
  lds_quad* own;        // the wave's rows + my lane (writes)
  int k;
  int first;            // lane of the wave my group starts at
  __device__ __forceinline__ int role() const { return k; }
  __device__ __forceinline__ bool all(bool v) const { return ((__ballot(v) >> first) & 63ull) == 63ull; }
  __device__ __forceinline__ Fp2 ld(int row, int lane) const {
    const lds_quad* p = base + row * 256 + lane;
    const u32x4 a = p[0], b = p[64], c = p[128], d = p[192];
    Fp2 r;
    r.c0.v[0] = a.x; r.c0.v[1] = a.y; r.c0.v[2] = a.z; r.c0.v[3] = a.w;
    r.c0.v[4] = b.x; r.c0.v[5] = b.y; r.c0.v[6] = b.z; r.c0.v[7] = b.w;
    r.c1.v[0] = c.x; r.c1.v[1] = c.y; r.c1.v[2] = c.z; r.c1.v[3] = c.w;
    r.c1.v[4] = d.x; r.c1.v[5] = d.y; r.c1.v[6] = d.z; r.c1.v[7] = d.w;
    return r;
  }
  __device__ __forceinline__ void st(int row, const Fp2& v) const {
    lds_quad* p = own + row * 256;
    p[0] = u32x4{v.c0.v[0], v.c0.v[1], v.c0.v[2], v.c0.v[3]};
    p[64] = u32x4{v.c0.v[4], v.c0.v[5], v.c0.v[6], v.c0.v[7]};
    p[128] = u32x4{v.c1.v[0], v.c1.v[1], v.c1.v[2], v.c1.v[3]};
    p[192] = u32x4{v.c1.v[4], v.c1.v[5], v.c1.v[6], v.c1.v[7]};
  }
  // lockstep lanes + in-order LDS: only the compiler has to be kept from moving slot accesses across this point
  __device__ __forceinline__ void sync() const {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
};
// lanes 60..63 of a wave belong to no group: they run along on group 9's slots (reads) and write their own, unused, slots
__device__ __forceinline__ DevCX6 dev_cx6(uint4* rows, int lane, int g, int k) {
  lds_quad* r = (lds_quad*)rows;
  return DevCX6{r + (g < C6_GROUPS ? 6 * g : 58), r + lane, k, g < C6_GROUPS ? 6 * g : 58};
}

__device__ __forceinline__ Fp ld_fp_q6(const uint4* p) {
  const uint4 a = p[0], b = p[1];
  Fp r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
// the group's pairs (the ACC interface of bn254/pairing.h: miller_multi_line); running points in the workspace layout of k_miller_multi
// ([wave of 64 (item, chunk) slots][pair slot][quad][slot]), so that k_walk_verdicts reads them unchanged
struct DevAcc6 {
  const G1M* P;
  const G2M* Q;
  const uint32_t* qref;
  const LineM* lines;
  int cnt;
  uint4* ws;
  __device__ __forceinline__ int count() const { return cnt; }
  __device__ __forceinline__ int kind(int j) const {
    const uint32_t v = qref[j];
    return v == RHIP_Q_WALK ? MP_WALK : v == RHIP_Q_SKIP ? MP_SKIP : MP_LINES;
  }
  __device__ __forceinline__ MillerP p(int j) const {
    const uint4* q = (const uint4*)(P + j);
    const Fp x = ld_fp_q6(q), y = ld_fp_q6(q + 2);
    return MillerP{x, y, y, false};
  }
  __device__ __forceinline__ G2Aff q(int j) const {
    const uint4* p = (const uint4*)(Q + j);
    return G2Aff{Fp2{ld_fp_q6(p), ld_fp_q6(p + 2)}, Fp2{ld_fp_q6(p + 4), ld_fp_q6(p + 6)}};
  }
  __device__ __forceinline__ LineCoeffs line(int j, int n) const {
    const uint4* p = (const uint4*)(lines + (size_t)qref[j] * RB_MILLER_LINES + n);
    return LineCoeffs{Fp2{ld_fp_q6(p), ld_fp_q6(p + 2)}, Fp2{ld_fp_q6(p + 4), ld_fp_q6(p + 6)}, Fp2{ld_fp_q6(p + 8), ld_fp_q6(p + 10)}};
  }
  __device__ __forceinline__ Fp ld1(const uint4* p) const {
    const uint4 a = p[0], b = p[64];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
  }
  __device__ __forceinline__ void st1(uint4* p, const Fp& a) const {
    p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    p[64] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
  }
  // one coordinate (c = 0, 1, 2: X, Y, Z) of pair j's running point: quads [4 c, 4 c + 4) of its twelve
  __device__ __forceinline__ Fp2 ld_tc(int j, int c) const {
    const uint4* p = ws + (size_t)(12 * j + 4 * c) * 64;
    return Fp2{ld1(p), ld1(p + 2 * 64)};
  }
  __device__ __forceinline__ void st_tc(int j, int c, const Fp2& v) const {
    uint4* p = ws + (size_t)(12 * j + 4 * c) * 64;
    st1(p, v.c0); st1(p + 2 * 64, v.c1);
  }
  __device__ __forceinline__ G2Hom ld_t(int j) const {
    const uint4* p = ws + (size_t)(12 * j) * 64;
    G2Hom t;
    t.x = Fp2{ld1(p), ld1(p + 2 * 64)};
    t.y = Fp2{ld1(p + 4 * 64), ld1(p + 6 * 64)};
    t.z = Fp2{ld1(p + 8 * 64), ld1(p + 10 * 64)};
    return t;
  }
  __device__ __forceinline__ void st_t(int j, const G2Hom& t) const {
    uint4* p = ws + (size_t)(12 * j) * 64;
    st1(p, t.x.c0); st1(p + 2 * 64, t.x.c1);
    st1(p + 4 * 64, t.y.c0); st1(p + 6 * 64, t.y.c1);
    st1(p + 8 * 64, t.z.c0); st1(p + 10 * 64, t.z.c1);
  }
};
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int o = __shfl_xor(v, d);
    v = o > v ? o : v;
  }
  return v;
}

// group G = chunk * n_items + item, the lane -> (item, chunk) map of k_miller_multi (engine_jobs.hip) with "lane" read as "group";
// plan != NULL: entry G of the device-made work list of a ragged batch.  Output: mill[item * L + c] / mill[chunk_off[item] + c].
__global__ void __launch_bounds__(64, RB_C6_WAVES) C6K(k_miller_c6)(size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const G1M* P,
                                                               const G2M* Q, const uint32_t* qref, const LineM* lines, uint4* ws, GtM* mill,
                                                               const MillerPlan* plan, const uint2* work, const uint32_t* chunk_off) {
  __shared__ uint4 rows[C6_LDS_QUADS];
  const int lane = threadIdx.x, g = lane / 6, k = lane - 6 * g;
  const size_t G = (size_t)blockIdx.x * C6_GROUPS + g;
  const size_t total = plan ? (size_t)plan->W : n_items * L;
  const bool active = g < C6_GROUPS && G < total;
  uint64_t first = 0;
  int cnt = 0;
  size_t out_idx = 0;
  uint32_t Cw = plan ? plan->C : C;
  if (active) {
    size_t item;
    uint32_t cc;
    if (plan) {
      const uint2 w = work[G];
      item = w.x;
      cc = w.y;
      out_idx = (size_t)chunk_off[item] + cc;
    } else {
      const size_t c = G / n_items;
      item = (G % n_items + c * RB_MILLER_BLOCK) % n_items;
      cc = (uint32_t)c;
      out_idx = item * L + c;
    }
    const uint64_t lo = pair_off ? pair_off[item] : (uint64_t)item * uniform, hi = pair_off ? pair_off[item + 1] : (uint64_t)(item + 1) * uniform;
    const uint32_t p_item = (uint32_t)(hi - lo);
    const uint32_t nch = p_item ? (p_item + Cw - 1) / Cw : 0;
    if (cc < nch) {
      const uint32_t base = p_item / nch, rem = p_item % nch;
      first = lo + (uint64_t)cc * base + (cc < rem ? cc : rem);
      cnt = (int)(base + (cc < rem ? 1u : 0u));
    }
  }
  const DevCX6 cx = dev_cx6(rows, lane, g, k);
  const int cmax = wave_max_i32(cnt);
  Fp2 r = k == 0 ? fp2_one() : fp2_zero();
  if (cmax > 0) {
    const DevAcc6 acc{P + first, Q + first, qref + first, lines, cnt, ws + (G >> 6) * ((size_t)Cw * 12 * 64) + (G & 63)};
    r = c6_miller_loop_multi(cx, acc, cmax);
  }
  if (active) st_fp2_m(mill[out_idx].l + 16 * c6_tower_index(k), r);
}

// out[item] = (mul_in ? mul_in[item] : 1) * FE( prod_{j in [off[item], off[item+1])} mill[j] ), canonical -- k_final_exp's contract
__global__ void __launch_bounds__(64, RB_C6_WAVES) C6K(k_final_exp_c6)(size_t n_items, const uint32_t* off, uint32_t stride, const GtM* mill, const rhip_gt* mul_in,
                                                                  rhip_gt* out, uint32_t* started) {
  __shared__ uint4 rows[C6_LDS_QUADS];
  if (started && threadIdx.x == 0) { atomicAdd(started, 1u); __threadfence(); }
  const int lane = threadIdx.x, g = lane / 6, k = lane - 6 * g, ti = c6_tower_index(k);
  const size_t item = (size_t)blockIdx.x * C6_GROUPS + g;
  const bool active = g < C6_GROUPS && item < n_items;
  size_t lo = 0, hi = 0;
  if (active) { lo = off ? off[item] : item * stride; hi = off ? off[item + 1] : (item + 1) * stride; }
  const DevCX6 cx = dev_cx6(rows, lane, g, k);
  const int nmax = wave_max_i32((int)(hi - lo));
  Fp2 acc = k == 0 ? fp2_one() : fp2_zero();
#pragma unroll 1
  for (int jj = 0; jj < nmax; jj++) {
    const bool on = lo + jj < hi;
    Fp2 v = acc;
    if (on) v = ld_fp2_m(mill[lo + jj].l + 16 * ti);
    if (jj == 0) { acc = v; continue; }
    const Fp2 m = c6_mul(cx, acc, v);
    acc = fp2_select(on, m, acc);
  }
  Fp2 r = c6_final_exponentiation(cx, acc);
  if (mul_in) {
    Fp2 m = fp2_zero();
    if (active) m = load_fp2(mul_in[item].l + 16 * ti);
    r = c6_mul(cx, m, r);
  }
  if (active) store_fp2(out[item].l + 16 * ti, r);
}

// out[i] = (mul_in ? mul_in[i] : 1) * t0^k[stride i] * (t1 ? t1^k[stride i + 1] : 1): the fixed-base Gt powers of an encrypt (`c_p = msg *
// e_gh_ka[0]^s0 * e_gh_ka[1]^s1`, src/schemes/ac17/mod.rs:357-360; the Gt messages of a batch) as ONE running product per group: a
// window digit's table entry goes into the multiplier row, the accumulator is multiplied by it.  w16: 16 windows of 65535 entries,
// else 32 of 255 (engine_internal.h: TBL16_* / TBL_*).  The products of a wave's groups run in lockstep; a zero digit (no entry) only
// skips the commit.
__global__ void __launch_bounds__(64, RB_C6_WAVES) C6K(k_gt_table_pow_c6)(const GtM* t0, const GtM* t1, int w16, size_t n_items, const rhip_fr* k, uint32_t kstride,
                                                                     const rhip_gt* mul_in, rhip_gt* out) {
  __shared__ uint4 rows[C6_LDS_QUADS];
  const int lane = threadIdx.x, g = lane / 6, r = lane - 6 * g, ti = c6_tower_index(r);
  const size_t item = (size_t)blockIdx.x * C6_GROUPS + g;
  const bool active = g < C6_GROUPS && item < n_items;
  const DevCX6 cx = dev_cx6(rows, lane, g, r);
  Fp2 acc = r == 0 ? fp2_one() : fp2_zero();
  if (active && mul_in) acc = load_fp2(mul_in[item].l + 16 * ti);
  c6_put_f(cx, acc);
  const int n_win = w16 ? TBL16_WINDOWS : TBL_WINDOWS;
#pragma unroll 1
  for (int t = 0; t < 2; t++) {
    const GtM* tbl = t ? t1 : t0;
    if (!tbl) break;
    uint32_t kk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) ld_scalar(kk, k + item * kstride + t);
#pragma unroll 1
    for (int w = 0; w < n_win; w++) {
      uint32_t d;
      if (w16) {
        uint32_t word;
        switch (w >> 1) {
          case 0: word = kk[0]; break;
          case 1: word = kk[1]; break;
          case 2: word = kk[2]; break;
          case 3: word = kk[3]; break;
          case 4: word = kk[4]; break;
          case 5: word = kk[5]; break;
          case 6: word = kk[6]; break;
          default: word = kk[7]; break;
        }
        d = (w & 1) ? (word >> 16) : (word & 0xffffu);
      } else {
        d = scalar_byte(kk, w);
      }
      Fp2 e = fp2_zero();
      if (d) e = ld_fp2_m((tbl + (size_t)w * (w16 ? TBL16_DIGITS : TBL_DIGITS) + (d - 1))->l + 16 * ti);
      c6_put(cx, C6_B, e);
      const Fp2 m = c6_dot(cx, C6_OP_MUL, 0);
      if (d) c6_put_f(cx, m);
    }
  }
  if (active) store_fp2(out[item].l + 16 * ti, c6_mine(cx));
}

// ok[i] = a[i] is a canonical encoding of a member of Gt (rhip_gt_is_member: k_gt_is_member's mode 0 on six lanes)
__global__ void __launch_bounds__(64, RB_C6_WAVES) C6K(k_gt_is_member_c6)(size_t n, const rhip_gt* a, uint32_t* ok) {
  __shared__ uint4 rows[C6_LDS_QUADS];
  const int lane = threadIdx.x, g = lane / 6, r = lane - 6 * g, ti = c6_tower_index(r);
  const size_t item = (size_t)blockIdx.x * C6_GROUPS + g;
  const bool active = g < C6_GROUPS && item < n;
  const DevCX6 cx = dev_cx6(rows, lane, g, r);
  Fp2 f = r == 0 ? fp2_one() : fp2_zero();
  bool canon = true;
  if (active) {
    canon = wire_words_canonical(a[item].l + 16 * ti, 2);
    f = load_fp2(a[item].l + 16 * ti);
  }
  const bool good = c6_gt_is_member(cx, f, canon);
  if (active && r == 0) ok[item] = good ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------------ host side
// raw launches of this unit's variant
int32_t C6K(rhip_c6_raw_gt_is_member)(rhip_ctx* ctx, size_t n, const rhip_gt* a, uint32_t* ok) {
  KLAUNCH(ctx, "k_gt_is_member_c6", C6K(k_gt_is_member_c6), dim3(blocks_for(n, C6_GROUPS)), dim3(64), 0, ctx->stream, n, a, ok);
  return RHIP_OK;
}
int32_t C6K(rhip_c6_raw_miller)(rhip_ctx* ctx, size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const void* P, const void* Q,
                                const uint32_t* qref, const void* lines, void* ws, void* mill, const MillerPlan* plan, const void* work, const uint32_t* chunk_off,
                                size_t groups) {
  KLAUNCH(ctx, "k_miller_c6", C6K(k_miller_c6), dim3(blocks_for(groups, C6_GROUPS)), dim3(64), 0, ctx->stream, n_items, L, C, pair_off, uniform, (const G1M*)P,
          (const G2M*)Q, qref, (const LineM*)lines, (uint4*)ws, (GtM*)mill, plan, (const uint2*)work, chunk_off);
  return RHIP_OK;
}
int32_t C6K(rhip_c6_raw_final_exp)(rhip_ctx* ctx, size_t n_items, const uint32_t* off, uint32_t stride, const void* mill, const rhip_gt* mul_in, rhip_gt* out,
                                   uint32_t* started) {
  KLAUNCH(ctx, "k_final_exp_c6", C6K(k_final_exp_c6), dim3(blocks_for(n_items, C6_GROUPS)), dim3(64), 0, ctx->stream, n_items, off, stride, (const GtM*)mill, mul_in,
          out, started);
  return RHIP_OK;
}
int32_t C6K(rhip_c6_raw_gt_table_pow)(rhip_ctx* ctx, const void* t0, const void* t1, int w16, size_t n_items, const rhip_fr* k, uint32_t kstride, const rhip_gt* mul_in,
                                      rhip_gt* out) {
  KLAUNCH(ctx, "k_gt_table_pow_c6", C6K(k_gt_table_pow_c6), dim3(blocks_for(n_items, C6_GROUPS)), dim3(64), 0, ctx->stream, (const GtM*)t0, (const GtM*)t1, w16, n_items,
          k, kstride, mul_in, out);
  return RHIP_OK;
}
#ifndef RB_C6_W1_UNIT
int32_t rhip_c6_raw_miller_w1(rhip_ctx* ctx, size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const void* P, const void* Q,
                              const uint32_t* qref, const void* lines, void* ws, void* mill, const MillerPlan* plan, const void* work, const uint32_t* chunk_off, size_t groups);
int32_t rhip_c6_raw_final_exp_w1(rhip_ctx* ctx, size_t n_items, const uint32_t* off, uint32_t stride, const void* mill, const rhip_gt* mul_in, rhip_gt* out, uint32_t* started);
int32_t rhip_c6_raw_gt_table_pow_w1(rhip_ctx* ctx, const void* t0, const void* t1, int w16, size_t n_items, const rhip_fr* k, uint32_t kstride, const rhip_gt* mul_in,
                                    rhip_gt* out);
int32_t rhip_c6_raw_gt_is_member_w1(rhip_ctx* ctx, size_t n, const rhip_gt* a, uint32_t* ok);
// a launch of at most one wave per SIMD takes the one-wave variant
static bool c6_one_wave(const rhip_ctx* ctx, size_t groups) { return blocks_for(groups, C6_GROUPS) <= (unsigned)ctx->n_cu * 4; }
// When the six-lane kernels run.  Mode 6 (rhip_ctx_set_pairing_mode, or RABE_PAIRING_MODE=6 in the environment of rhip_ctx_create):
// always.  Mode 0 (auto; RABE_C6_AUTO=0 turns it off): where they are faster, by the measured instruction counts
// (profiles/r05a_pmc_sq_lone_c6.txt; a lone wave issues one VALU instruction per ~5.2 cycles, two or more per SIMD one per ~4.5):
//   Miller loops    a group's chain is 65 S + 88 (C Ln + ceil(C / 6) Pt) instructions (S = 3.4 k the shared squaring, Ln = 2.7 k one
//                   line, Pt = 12 k one G2 step, six pairs at a time), ~1.6 x the lane-instructions of the one-lane kernel per pair --
//                   worth it while the launch leaves SIMDs idle: when its groups at the best C fit the chip once, one wave per SIMD;
//   final exp.      0.67 M instructions per wave of ten items against one lane's 2.6 M for 64: faster up to ~48 k items.
static double c6_miller_chain(size_t c) { return 65 * 3400.0 + 88.0 * (2700.0 * (double)c + 12000.0 * (double)((c + 5) / 6)); }
static double c6_miller_cycles(const rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t c) {
  const size_t l = (max_pairs + c - 1) / c, c_eff = (max_pairs + l - 1) / l;
  const double simds = (double)ctx->n_cu * 4, waves = (double)((n_items * l + C6_GROUPS - 1) / C6_GROUPS);
  const double per_simd = waves <= simds ? 1.0 : (double)(size_t)((waves + simds - 1) / simds);      // the fullest SIMD sets the time
  return c6_miller_chain(c_eff) * (per_simd <= 1.0 ? 5.2 : 4.5 * per_simd);
}
void rhip_choose_chunks_c6(const rhip_ctx* ctx, size_t n_items, size_t max_pairs, uint32_t* L, uint32_t* C) {
  if (max_pairs < 1) max_pairs = 1;
  double best = 0;
  size_t best_l = 1;
  for (size_t c = 1; c <= 64; c++) {
    const size_t l = (max_pairs + c - 1) / c;
    const double cost = c6_miller_cycles(ctx, n_items, max_pairs, c);
    if (best == 0 || cost < best) { best = cost; best_l = l; }
    if (l == 1) break;
  }
  *L = (uint32_t)best_l;
  *C = (uint32_t)((max_pairs + best_l - 1) / best_l);
}
// max_pairs == 0: the question is about the final exponentiation of n_items items
bool rhip_use_c6(const rhip_ctx* ctx, size_t n_items, size_t max_pairs) {
  if (rhip_mode(ctx) == 6) return true;
  if (rhip_mode(ctx) != 0) return false;
  static const int auto_on = getenv("RABE_C6_AUTO") ? atoi(getenv("RABE_C6_AUTO")) : 1;
  if (!auto_on) return false;
  const size_t simds = (size_t)ctx->n_cu * 4;
  static const long fe_max = getenv("RABE_C6_FE_MAX") ? atol(getenv("RABE_C6_FE_MAX")) : -1;          // tuning runs
  static const int miller_on = getenv("RABE_C6_MILLER") ? atoi(getenv("RABE_C6_MILLER")) : 1;
  if (!max_pairs) return n_items <= (fe_max >= 0 ? (size_t)fe_max : 47 * simds);
  if (!miller_on) return false;
  uint32_t L, C;
  rhip_choose_chunks_c6(ctx, n_items, max_pairs, &L, &C);
  return (n_items * L + C6_GROUPS - 1) / C6_GROUPS <= simds;
}
int32_t rhip_launch_miller_c6(rhip_ctx* ctx, size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const void* P, const void* Q,
                              const uint32_t* qref, const void* lines, void* ws, void* mill, const MillerPlan* plan, const void* work, const uint32_t* chunk_off,
                              size_t groups) {
  return (c6_one_wave(ctx, groups) ? rhip_c6_raw_miller_w1 : rhip_c6_raw_miller)(ctx, n_items, L, C, pair_off, uniform, P, Q, qref, lines, ws, mill, plan, work, chunk_off,
                                                                                 groups);
}
int32_t rhip_launch_final_exp_c6(rhip_ctx* ctx, size_t n_items, const uint32_t* off, uint32_t stride, const void* mill, const rhip_gt* mul_in, rhip_gt* out,
                                 uint32_t* started) {
  return (c6_one_wave(ctx, n_items) ? rhip_c6_raw_final_exp_w1 : rhip_c6_raw_final_exp)(ctx, n_items, off, stride, mill, mul_in, out, started);
}
int32_t rhip_launch_gt_table_pow_c6(rhip_ctx* ctx, const void* t0, const void* t1, int w16, size_t n_items, const rhip_fr* k, uint32_t kstride, const rhip_gt* mul_in,
                                    rhip_gt* out) {
  return (c6_one_wave(ctx, n_items) ? rhip_c6_raw_gt_table_pow_w1 : rhip_c6_raw_gt_table_pow)(ctx, t0, t1, w16, n_items, k, kstride, mul_in, out);
}
// ------------------------------------------------------------------------------------------------ known-answer self-test
// bn254/selftest.h on every SIMD of the device: every wave computes the 64 lanes' digests and compares them with the compiled-in
// expectation; lane 0 records which SIMD the wave ran on (XCC_ID and the SE / SH / CU / SIMD fields of HW_ID).
#define RB_GETREG(id) __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (id))
__global__ void __launch_bounds__(64) k_selftest(uint32_t corrupt, uint32_t* fail, uint32_t* seen) {
  constexpr uint32_t expect[64] = RB_SELFTEST_EXPECT;
  const int lane = threadIdx.x;
  const uint32_t d = selftest_digest(lane);
  if (d != (expect[lane] ^ corrupt)) atomicAdd(fail, 1u);
  if (lane == 0) {
    const uint32_t hw = RB_GETREG(4) /* HW_REG_HW_ID */, xcc = RB_GETREG(20) /* HW_REG_XCC_ID */;
    const uint32_t key = ((hw >> 4) & 3u) | (((hw >> 8) & 0xFFu) << 2) | ((xcc & 0xFu) << 10);        // simd | cu, sh, se | xcc
    atomicOr(seen + (key >> 5), 1u << (key & 31));
  }
}
// once per device and process (rhip_ctx_create calls it for every context).  RABE_NO_SELFTEST=1 skips it; RABE_SELFTEST_CORRUPT=1
// flips the expectation so that the refusal itself can be tested.  *simds: distinct SIMDs the check ran on (diagnostic).
int32_t rhip_device_selftest(rhip_ctx* ctx, uint32_t* simds, uint32_t* mismatches) {
  static std::mutex mu;
  static uint64_t passed = 0;
  static uint32_t covered[64] = {};
  std::lock_guard<std::mutex> g(mu);
  const int dev = ctx->device & 63;
  if (simds) *simds = covered[dev];
  if (mismatches) *mismatches = 0;
  const char* corrupt = getenv("RABE_SELFTEST_CORRUPT");
  if (((passed >> dev) & 1) && !corrupt) return RHIP_OK;
  if (getenv("RABE_NO_SELFTEST")) return RHIP_OK;
  uint32_t* d = nullptr;
  const size_t words = 1 + 512;
  HIP_TRY(ctx, hipMalloc((void**)&d, words * 4));
  hipError_t e = hipMemsetAsync(d, 0, words * 4, ctx->stream);
  if (e == hipSuccess) {
    // four waves for every SIMD: the one-wave blocks spread over all of them (the coverage is reported, not assumed)
    hipLaunchKernelGGL(k_selftest, dim3((unsigned)ctx->n_cu * 16), dim3(64), 0, ctx->stream, corrupt ? 1u : 0u, d, d + 1);
    e = hipGetLastError();
  }
  uint32_t h[1 + 512] = {};
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, words * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(ctx, e, "device self-test");
  uint32_t n = 0;
  for (size_t i = 1; i < words; i++) n += (uint32_t)__builtin_popcount(h[i]);
  covered[dev] = n;
  if (simds) *simds = n;
  if (mismatches) *mismatches = h[0];
  if (h[0]) {
    ctx->err = "device self-test failed: " + std::to_string(h[0]) + " lane digests of the BN254 field arithmetic differ from their known answers on this GPU ("
               + std::to_string(n) + " SIMDs checked); refusing to compute with it (build with `python -m rabe_amd.build --safe` for the RB_SAFE_CARRY objects)";
    return RHIP_ERR_HIP;
  }
  passed |= 1ull << dev;
  return RHIP_OK;
}
extern "C" int32_t rhip_ctx_selftest_info(rhip_ctx* ctx, uint32_t* simds_checked) {
  if (!ctx || !simds_checked) return RHIP_ERR_ARG;
  return rhip_device_selftest(ctx, simds_checked, nullptr);
}

int32_t rhip_launch_gt_is_member_c6(rhip_ctx* ctx, size_t n, const rhip_gt* a, uint32_t* ok) {
  return (c6_one_wave(ctx, n) ? rhip_c6_raw_gt_is_member_w1 : rhip_c6_raw_gt_is_member)(ctx, n, a, ok);
}
// the fixed-base Gt kernels: 32 (16-bit windows) or 64 dependent products per item in one lane against ~5 k instructions each here
bool rhip_use_c6_gt_pow(const rhip_ctx* ctx, size_t n_items) {
  if (rhip_mode(ctx) == 6) return true;
  if (rhip_mode(ctx) != 0) return false;
  static const int auto_on = getenv("RABE_C6_AUTO") ? atoi(getenv("RABE_C6_AUTO")) : 1;
  return auto_on && n_items <= (size_t)ctx->n_cu * 4 * 32;
}
#endif  // !RB_C6_W1_UNIT
