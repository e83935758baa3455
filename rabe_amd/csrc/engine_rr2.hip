// rabe_amd engine, translation unit of the TWO-LANE reduced-radix Miller kernel (bn254/pairing29p.h), round 6.
//
//   k_miller_pair_rr     what k_miller_multi_rr (engine_rr.hip) computes -- unit = (item, chunk of its pairs), all pairs of the chunk on one Fq12
//                        accumulator, 9 signed 29-bit limbs -- with a unit spread over TWO adjacent lanes: f = c0 + c1 w, lane 0 owns c0, lane 1
//                        owns c1.  The home is 216 B of LDS per lane (+ one shared operand slot and the side slot: 308 B), eight waves fit a
//                        CU, the kernel is held to 256 registers: every SIMD runs TWO waves, and each hides the other's call boundaries, LDS
//                        latencies and global-memory round trips (tools/ubench_rr29_2w.hip: the dot-product routine 6 750 -> 4 940 SIMD cycles
//                        per call, the Fq2 multiplication 3 460 -> 2 610).  Same unit map, same inputs (pair lists of 8 x 32-bit Montgomery
//                        records, prepared unit-y lines in the 9-quad form), same outputs as k_miller_multi_rr: the Miller values and the
//                        points the walking pairs end on, in the canonical 8 x 32-bit form -- everything downstream is unchanged.
// Pairings of `ac17::cp_decrypt` (src/schemes/ac17/mod.rs:415-418), bsw/mod.rs:291-294,308, lsw/mod.rs:275-280, aw11/mod.rs:340-350.
// There is no CPU fallback in this file.
#include "engine_internal.h"
#include "bn254/pairing29p.h"

using rr::F;
using rr::F2;
using rr::F6;

// ---- LDS of a four-wave block (78 848 B; two blocks per CU):
//   home:  per wave [12 quads][64 lanes] + [6 dwords][64 lanes] -- a lane's three Fq2 coefficients = six Fp, element e = quads 2 e, 2 e + 1
//          (limbs 0..7) + dword e (limb 8)
//   slot:  per wave [4 quads][64 lanes] + [2 dwords][64 lanes] -- ONE Fq2 per lane; operand slot s (0, 1) of a pair lives in its lane s
//   side:  fp29.h's rr_side (5 dwords per lane: the overflow arguments / results of the out-of-line Fq2 routines)
static __shared__ uint4 p2_home_q[4 * 12 * 64];
static __shared__ uint32_t p2_home_d[4 * 6 * 64];
static __shared__ uint4 p2_slot_q[4 * 4 * 64];
static __shared__ uint32_t p2_slot_d[4 * 2 * 64];

__device__ __forceinline__ Fp p2_ld_fp_q(const uint4* p) {
  const uint4 a = p[0], b = p[1];
  Fp r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ F p2_from_quads(const uint4& a, const uint4& b, uint32_t c) {
  F r;
  r.l[0] = (int32_t)a.x; r.l[1] = (int32_t)a.y; r.l[2] = (int32_t)a.z; r.l[3] = (int32_t)a.w;
  r.l[4] = (int32_t)b.x; r.l[5] = (int32_t)b.y; r.l[6] = (int32_t)b.z; r.l[7] = (int32_t)b.w;
  r.l[8] = (int32_t)c;
  return r;
}
__device__ __forceinline__ uint4 p2_quad(const F& a, int h) {
  return make_uint4((uint32_t)a.l[4 * h], (uint32_t)a.l[4 * h + 1], (uint32_t)a.l[4 * h + 2], (uint32_t)a.l[4 * h + 3]);
}
__device__ __forceinline__ rr::i32x9 p2_lds_elem(const uint4* q, const uint32_t* d) {
  const uint4 a = q[0], b = q[64];
  rr::i32x9 r;
  r[0] = (int32_t)a.x; r[1] = (int32_t)a.y; r[2] = (int32_t)a.z; r[3] = (int32_t)a.w;
  r[4] = (int32_t)b.x; r[5] = (int32_t)b.y; r[6] = (int32_t)b.z; r[7] = (int32_t)b.w;
  r[8] = (int32_t)d[0];
  return r;
}
// DPP moves inside a lane pair: 0xB1 = quad_perm [1, 0, 3, 2] (the partner), 0xA0 = [0, 0, 2, 2] (lane 0 of the pair), 0xF5 = [1, 1, 3, 3]
template <int CTRL> __device__ __forceinline__ int32_t p2_dpp(int32_t v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ F p2_dpp9(const F& a) {
  F r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = p2_dpp<CTRL>(a.l[i]);
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__)          // (the host pass of this file only needs the kernels' signatures)
// f[is0] slot0 + f[ir] yr + f[is1] slot1 for this lane: the pair's coefficients and slots from LDS (coefficient i lives in lane base + i / 3,
// element pair i % 3; slot s in lane base + s), yr in registers; the result as the Fq2 routines of fp29.h return theirs.
// REGISTERS: the routine is written to stay inside the calling convention's 144 caller-saved VGPRs, so that it saves nothing and the loop body
// keeps ~110 values in callee-saved registers across its calls without a spill -- at two waves per SIMD a wave has 256 registers, not 512.
// One column set at a time (the real part, then the imaginary part), the operands of one term at a time, re-read from LDS for the second
// set; compiler barriers keep the loads where they are written (left alone the scheduler hoists all 22 operand loads to the top: 248
// registers, and every call of the loop body turned into ~100 moves and scratch accesses: 21.2 ms against the one-lane kernel's 16.6).
#define P2_CO(i, part) p2_lds_elem(hq + (4 * ((i) % 3) + 2 * (part)) * 64 + (i) / 3, hd + (2 * ((i) % 3) + (part)) * 64 + (i) / 3)
#define P2_BARRIER asm volatile("" ::: "memory")
template <bool UNIT>
__device__ __forceinline__ rr::Out16 rr2_dot_body(const rr::i32x9& ya, const rr::i32x9& yb, int is0, int ir, int is1) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, base = ln & ~1;
  const uint4* hq = p2_home_q + wv * (12 * 64) + base;
  const uint32_t* hd = p2_home_d + wv * (6 * 64) + base;
  const uint4* sq = p2_slot_q + wv * (4 * 64) + base;
  const uint32_t* sd = p2_slot_d + wv * (2 * 64) + base;
  rr::i32x9 c0, c1;
  {          // real part: x0a s0a - x0b s0b + x1a ya - x1b yb + x2a s1a - x2b s1b
    int64_t t[18];
    rr::cols_init(t);
    {
      const rr::i32x9 xa = P2_CO(is0, 0), sa = p2_lds_elem(sq, sd);
      rr::cols_mac(t, xa, sa);
      if (!UNIT) { const rr::i32x9 xb = P2_CO(is0, 1), sb = p2_lds_elem(sq + 2 * 64, sd + 64); rr::cols_mac(t, -xb, sb); }
    }
    P2_BARRIER;
    { const rr::i32x9 xa = P2_CO(ir, 0); rr::cols_mac(t, xa, ya); const rr::i32x9 xb = P2_CO(ir, 1); rr::cols_mac(t, -xb, yb); }
    P2_BARRIER;
    {
      const rr::i32x9 xa = P2_CO(is1, 0), sa = p2_lds_elem(sq + 1, sd + 1);
      rr::cols_mac(t, xa, sa);
      const rr::i32x9 xb = P2_CO(is1, 1), sb = p2_lds_elem(sq + 2 * 64 + 1, sd + 64 + 1);
      rr::cols_mac(t, -xb, sb);
    }
    c0 = rr::redc(t);
  }
  P2_BARRIER;
  {          // imaginary part: x0a s0b + x0b s0a + x1a yb + x1b ya + x2a s1b + x2b s1a
    int64_t t[18];
    rr::cols_init(t);
    if (UNIT) { const rr::i32x9 xb = P2_CO(is0, 1), sa = p2_lds_elem(sq, sd); rr::cols_mac(t, xb, sa); }
    else {
      const rr::i32x9 xa = P2_CO(is0, 0), sb = p2_lds_elem(sq + 2 * 64, sd + 64);
      rr::cols_mac(t, xa, sb);
      const rr::i32x9 xb = P2_CO(is0, 1), sa = p2_lds_elem(sq, sd);
      rr::cols_mac(t, xb, sa);
    }
    P2_BARRIER;
    { const rr::i32x9 xa = P2_CO(ir, 0); rr::cols_mac(t, xa, yb); const rr::i32x9 xb = P2_CO(ir, 1); rr::cols_mac(t, xb, ya); }
    P2_BARRIER;
    {
      const rr::i32x9 xa = P2_CO(is1, 0), sb = p2_lds_elem(sq + 2 * 64 + 1, sd + 64 + 1);
      rr::cols_mac(t, xa, sb);
      const rr::i32x9 xb = P2_CO(is1, 1), sa = p2_lds_elem(sq + 1, sd + 1);
      rr::cols_mac(t, xb, sa);
    }
    c1 = rr::redc(t);
  }
  return rr::side_ret(c0, c1, rr::rr_side + threadIdx.x);
}
#undef P2_CO
__device__ __attribute__((noinline)) rr::Out16 rr2_dot_core(rr::i32x9 ya, rr::i32x9 yb, int is0, int ir, int is1) { return rr2_dot_body<false>(ya, yb, is0, ir, is1); }
// the same with slot 0 in Fq (the unit-y lines of prepared pairs): f[is0] s + f[ir] yr + f[is1] slot1
__device__ __attribute__((noinline)) rr::Out16 rr2_dots_core(rr::i32x9 ya, rr::i32x9 yb, int is0, int ir, int is1) { return rr2_dot_body<true>(ya, yb, is0, ir, is1); }
#endif

// A lane's column of the global workspace (quads at stride 64: one coalesced 1 KB access per quad and wave), for chunks of up to Cw pairs:
//   quads [5 j, 5 j + 5), j < Cw:  the converted G1 argument of pair j (px, py: 4 quads of limbs 0..7 + 1 quad of top limbs) -- both lanes keep
//                                  their own copy, so nothing one lane writes to global memory is ever read by the other
//   then 23 quads per walking pair THIS lane walks (every second walking pair of the chunk): the running point T (12 quads + 2 of top
//   limbs), the converted G2 argument (8 quads + 1)
#define RR2_P_QUADS 5
#define RR2_WALK_QUADS 23
__host__ __device__ static inline size_t rr2_col_quads(size_t Cw) { return RR2_P_QUADS * Cw + RR2_WALK_QUADS * ((Cw + 1) / 2); }
#define RR_LINE_QUADS 9
struct DevPairAcc29 {
  const G1M* P;
  const G2M* Q;
  const uint32_t* qref;
  const uint4* lines29;      // prepared lines, unit y-coefficient, 9 quads each (engine_rr.hip: k_lines_to_rr)
  int cnt;
  uint32_t Cw;
  uint4* ws;                 // this lane's column
  uint64_t walk_m, skip_m;
  __device__ __forceinline__ int count() const { return cnt; }
  __device__ __forceinline__ bool hi() const { return (threadIdx.x & 1) != 0; }
  __device__ __forceinline__ int kind(int j) const { return ((walk_m >> j) & 1ull) ? MP_WALK : ((skip_m >> j) & 1ull) ? MP_SKIP : MP_LINES; }
  __device__ __forceinline__ int walk_ordinal(int j) const { return __popcll(walk_m & ((1ull << j) - 1ull)); }
  // ---- home
  __device__ __forceinline__ F ld_elem(int lane, int e) const {
    const uint4* q = p2_home_q + (threadIdx.x >> 6) * (12 * 64) + lane + (2 * e) * 64;
    return p2_from_quads(q[0], q[64], p2_home_d[(threadIdx.x >> 6) * (6 * 64) + e * 64 + lane]);
  }
  __device__ __forceinline__ F2 ld_co(int i) const {
    const int lane = ((threadIdx.x & 63) & ~1) + i / 3, e = 2 * (i % 3);
    return rr::mk2(ld_elem(lane, e), ld_elem(lane, e + 1));
  }
  __device__ __forceinline__ void st_elem(int e, const F& a) const {
    const int lane = threadIdx.x & 63;
    uint4* q = p2_home_q + (threadIdx.x >> 6) * (12 * 64) + lane + (2 * e) * 64;
    q[0] = p2_quad(a, 0); q[64] = p2_quad(a, 1);
    p2_home_d[(threadIdx.x >> 6) * (6 * 64) + e * 64 + lane] = (uint32_t)a.l[8];
  }
  __device__ __forceinline__ void st_own(int i, const F2& a) const { st_elem(2 * i, a.c0); st_elem(2 * i + 1, a.c1); }
  __device__ __forceinline__ F ld_own_elem(int e) const { return ld_elem(threadIdx.x & 63, e); }
  // ---- the pair's operand slots
  __device__ __forceinline__ void set_slot(int s, const F2& a, bool w) const {
    if (!w) return;
    const int lane = ((threadIdx.x & 63) & ~1) + s;
    uint4* q = p2_slot_q + (threadIdx.x >> 6) * (4 * 64) + lane;
    uint32_t* d = p2_slot_d + (threadIdx.x >> 6) * (2 * 64) + lane;
    q[0] = p2_quad(a.c0, 0); q[64] = p2_quad(a.c0, 1); q[128] = p2_quad(a.c1, 0); q[192] = p2_quad(a.c1, 1);
    d[0] = (uint32_t)a.c0.l[8]; d[64] = (uint32_t)a.c1.l[8];
  }
  __device__ __forceinline__ void set_slot0_fp(const F& a, bool w) const {
    if (!w) return;
    const int lane = (threadIdx.x & 63) & ~1;
    uint4* q = p2_slot_q + (threadIdx.x >> 6) * (4 * 64) + lane;
    q[0] = p2_quad(a, 0); q[64] = p2_quad(a, 1);
    p2_slot_d[(threadIdx.x >> 6) * (2 * 64) + lane] = (uint32_t)a.l[8];
  }
  __device__ __forceinline__ F2 dotp(const F2& yr, int is0, int ir, int is1) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const rr::Out16 o = rr2_dot_core(yr.c0.l, yr.c1.l, is0, ir, is1);
    uint32_t* side = rr::rr_side + threadIdx.x;
    RB29_TAKE(o, c0, c1, side)
    return rr::mk2(rr::mk<1, 1>(c0), rr::mk<1, 1>(c1));
#else
    return yr;
#endif
  }
  __device__ __forceinline__ F2 dotps(const F2& yr, int is0, int ir, int is1) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const rr::Out16 o = rr2_dots_core(yr.c0.l, yr.c1.l, is0, ir, is1);
    uint32_t* side = rr::rr_side + threadIdx.x;
    RB29_TAKE(o, c0, c1, side)
    return rr::mk2(rr::mk<1, 1>(c0), rr::mk<1, 1>(c1));
#else
    return yr;
#endif
  }
  __device__ __forceinline__ F2 other2(const F2& a) const { return rr::mk2(p2_dpp9<0xB1>(a.c0), p2_dpp9<0xB1>(a.c1)); }
  template <int O> __device__ __forceinline__ F2 from2(const F2& a) const {
    return rr::mk2(p2_dpp9<(O ? 0xF5 : 0xA0)>(a.c0), p2_dpp9<(O ? 0xF5 : 0xA0)>(a.c1));
  }
  __device__ __forceinline__ void fence() const { asm volatile("" ::: "memory"); }
  // ---- workspace: NF consecutive Fp starting at quad q0 of the column, their top limbs packed in the quads from qt on
  template <int NF> __device__ __forceinline__ void ld_n(size_t q0, size_t qt, F* out) const {
    const uint4* p = ws + q0 * 64;
    uint32_t tops[8];
#pragma unroll
    for (int k = 0; k < (NF + 3) / 4; k++) {
      const uint4 t = ws[(qt + k) * 64];
      tops[4 * k] = t.x; tops[4 * k + 1] = t.y; tops[4 * k + 2] = t.z; tops[4 * k + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < NF; i++) out[i] = p2_from_quads(p[(size_t)(2 * i) * 64], p[(size_t)(2 * i + 1) * 64], tops[i]);
  }
  template <int NF> __device__ __forceinline__ void st_n(size_t q0, size_t qt, const F* in) const {
    uint4* p = ws + q0 * 64;
#pragma unroll
    for (int i = 0; i < NF; i++) { p[(size_t)(2 * i) * 64] = p2_quad(in[i], 0); p[(size_t)(2 * i + 1) * 64] = p2_quad(in[i], 1); }
#pragma unroll
    for (int k = 0; k < (NF + 3) / 4; k++) {
      uint32_t t[4];
#pragma unroll
      for (int e = 0; e < 4; e++) t[e] = (4 * k + e < NF) ? (uint32_t)in[4 * k + e].l[8] : 0u;
      ws[(qt + k) * 64] = make_uint4(t[0], t[1], t[2], t[3]);
    }
  }
  __device__ __forceinline__ size_t walk_base(int j) const { return (size_t)RR2_P_QUADS * Cw + (size_t)RR2_WALK_QUADS * (size_t)(walk_ordinal(j) >> 1); }
  __device__ __forceinline__ rr::G2Hom29 ld_t(int j) const {
    F e[6];
    const size_t b = walk_base(j);
    ld_n<6>(b, b + 12, e);
    return rr::G2Hom29{rr::mk2(e[0], e[1]), rr::mk2(e[2], e[3]), rr::mk2(e[4], e[5])};
  }
  __device__ __forceinline__ void st_t(int j, const rr::G2Hom29& t) const {
    const F e[6] = {t.x.c0, t.x.c1, t.y.c0, t.y.c1, t.z.c0, t.z.c1};
    const size_t b = walk_base(j);
    st_n<6>(b, b + 12, e);
  }
  __device__ __forceinline__ rr::G2Aff29 q(int j) const {
    F e[4];
    const size_t b = walk_base(j) + 14;
    ld_n<4>(b, b + 8, e);
    return rr::G2Aff29{rr::mk2(e[0], e[1]), rr::mk2(e[2], e[3])};
  }
  __device__ __forceinline__ rr::MillerP29 p(int j) const {
    F e[2];
    ld_n<2>((size_t)RR2_P_QUADS * j, (size_t)RR2_P_QUADS * j + 4, e);
    return rr::MillerP29{e[0], e[1]};
  }
  __device__ __forceinline__ rr::LineU29 line_u(int j, int n) const {
    const uint4* p = lines29 + ((size_t)qref[j] * RB_MILLER_LINES + n) * RR_LINE_QUADS;
    const uint4 t0 = p[8];
    rr::LineU29 r;
    r.cx = rr::mk2(p2_from_quads(p[0], p[1], t0.x), p2_from_quads(p[2], p[3], t0.y));
    r.c0 = rr::mk2(p2_from_quads(p[4], p[5], t0.z), p2_from_quads(p[6], p[7], t0.w));
    return r;
  }
  // once, before the loop: the arguments this lane will use, in the field core's representation -- every pair's G1 argument, and the G2
  // argument + running point of the walking pairs this lane walks (walking pair number w of the chunk belongs to lane w & 1)
  __device__ __forceinline__ void begin() const {
    for (int j = 0; j < cnt; j++) {
      const int k = kind(j);
      if (k == MP_SKIP) continue;
      {
        const uint4* g = (const uint4*)(P + j);
        const F e[2] = {rr::from_fp(p2_ld_fp_q(g)), rr::from_fp(p2_ld_fp_q(g + 2))};
        st_n<2>((size_t)RR2_P_QUADS * j, (size_t)RR2_P_QUADS * j + 4, e);
      }
      if (k == MP_WALK && ((walk_ordinal(j) & 1) != 0) == hi()) {
        const uint4* g = (const uint4*)(Q + j);
        const F e[4] = {rr::from_fp(p2_ld_fp_q(g)), rr::from_fp(p2_ld_fp_q(g + 2)), rr::from_fp(p2_ld_fp_q(g + 4)), rr::from_fp(p2_ld_fp_q(g + 6))};
        const size_t b = walk_base(j) + 14;
        st_n<4>(b, b + 8, e);
        st_t(j, rr::G2Hom29{rr::mk2(e[0], e[1]), rr::mk2(e[2], e[3]), rr::one2()});
      }
    }
  }
};

// Same arguments and unit map as k_miller_multi_rr (engine_rr.hip) with unit = (global lane) / 2; ws2: the workspace in this kernel's layout;
// ws: the one k_walk_verdicts reads ([wave of 64 units][pair slot][12 quads][unit], 8 x 32-bit Montgomery limbs) -- written once, at the end.
__global__ void __launch_bounds__(RB_MILLER_BLOCK, 2) k_miller_pair_rr(size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const G1M* P,
                                                                      const G2M* Q, const uint32_t* qref, const uint4* lines29, uint4* ws, uint4* ws2, GtM* mill,
                                                                      const MillerPlan* plan, const uint2* work, const uint32_t* chunk_off, uint32_t* started) {
  if (started && threadIdx.x == 0) { atomicAdd(started, 1u); __threadfence(); }          // rhip_ctx_release_when_miller_resident
  const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t t = gl >> 1;
  const bool hi = (gl & 1) != 0;
  uint64_t first;
  int cnt;
  uint32_t Cw;
  GtM* out;
  if (plan) {
    if (t >= plan->W) return;
    Cw = plan->C;
    const uint2 w = work[t];
    const uint64_t lo = pair_off[w.x], hi_ = pair_off[w.x + 1];
    const uint32_t p_item = (uint32_t)(hi_ - lo), nch = (p_item + Cw - 1) / Cw;
    const uint32_t base = p_item / nch, rem = p_item % nch, cc = w.y;
    first = lo + (uint64_t)cc * base + (cc < rem ? cc : rem);
    cnt = (int)(base + (cc < rem ? 1u : 0u));
    out = mill + chunk_off[w.x] + cc;
  } else {
    if (t >= n_items * L) return;
    Cw = C;
    const size_t c = t / n_items, item = (t % n_items + c * RB_MILLER_BLOCK) % n_items;
    const uint64_t lo = pair_off ? pair_off[item] : (uint64_t)item * uniform, hi_ = pair_off ? pair_off[item + 1] : (uint64_t)(item + 1) * uniform;
    const uint32_t p_item = (uint32_t)(hi_ - lo);
    const uint32_t nch = (p_item + C - 1) / C;
    out = mill + item * L + c;
    if (c >= nch) {
      if (!hi) st_gt_m(out, fp12_one());
      return;
    }
    const uint32_t base = p_item / nch, rem = p_item % nch, cc = (uint32_t)c;
    first = lo + (uint64_t)cc * base + (cc < rem ? cc : rem);
    cnt = (int)(base + (cc < rem ? 1u : 0u));
  }
  uint64_t walk_m = 0, skip_m = 0;
  for (int j = 0; j < cnt && j < 64; j++) {
    const uint32_t v = qref[first + j];
    if (v == RHIP_Q_WALK) walk_m |= 1ull << j;
    else if (v == RHIP_Q_SKIP) skip_m |= 1ull << j;
  }
  const DevPairAcc29 acc{P + first, Q + first, qref + first, lines29, cnt, Cw, ws2 + (gl >> 6) * (rr2_col_quads(Cw) * 64) + (gl & 63), walk_m, skip_m};
  rr::miller_loop_pair(acc);
  // the value, back in the canonical Montgomery form of the 8 x 32-bit core: this lane's six Fp
  {
    uint32_t* o = out->l + (hi ? 48 : 0);
#pragma unroll 1
    for (int e = 0; e < 6; e++) st_fp_m(o + 8 * e, rr::to_fp(acc.ld_own_elem(e)));
  }
  // the points this lane's walking pairs ended on, where k_walk_verdicts looks for them
  uint4* wl = ws + (t >> 6) * ((size_t)Cw * 12 * 64) + (t & 63);
  for (int j = 0; j < cnt; j++) {
    if (acc.kind(j) != MP_WALK || ((acc.walk_ordinal(j) & 1) != 0) != hi) continue;
    F e[6];
    const size_t b = acc.walk_base(j);
    acc.ld_n<6>(b, b + 12, e);
    uint4* p = wl + (size_t)(12 * j) * 64;
#pragma unroll 1
    for (int k = 0; k < 6; k++) {
      const Fp v = rr::to_fp(e[k]);
      p[(size_t)(2 * k) * 64] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
      p[(size_t)(2 * k + 1) * 64] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
    }
  }
}

// When it runs: pairing mode 58 (rhip_ctx_set_pairing_mode / RABE_PAIRING_MODE) -- every multi-pairing launch whose chunks hold at most 64
// pairs -- and, with RABE_RR2=1 in the environment, the launches of mode 0 (auto) that the reduced-radix kernels take.  NOT the default:
// measured (docs/tried.md, "two waves per SIMD"; profiles/r06*) the second wave buys 1.36 x in CYCLES and the chip answers with a lower
// clock (2.3 -> 1.7-1.9 GHz at this VALU density: the board is power-limited), so that the kernel ends at 19.4 ms per 65 536 AC17 items
// against the one-lane kernel's 16.6 -- its 16 % more lane-instructions (selections, DPP moves, the second round trip of the running
// points, one idle lane when a chunk holds an odd number of walking pairs) are no longer paid for.  Kept as a fourth, independently written
// family of pairing kernels: the cross-check mode (99) runs it beside the others on every launch of the GPU suite.
bool rhip_use_rr2(const rhip_ctx* ctx, uint32_t c_max) {
  static const int on = getenv("RABE_RR2") ? atoi(getenv("RABE_RR2")) : 0;
  if (c_max > 64) return false;
  return rhip_mode(ctx) == 58 || (on != 0 && rhip_mode(ctx) == 0);
}
// `units` = (item, chunk) units of the launch (the `lanes` of rhip_launch_miller_rr); c_max: the largest chunk the launch can hold (the
// chunk size C, or the upper bound of a device-side plan)
// plan_pairs / plan_c_lo: for a device-side plan (plan != NULL) the total number of pairs and the smallest chunk size the plan may choose -- the
// workspace is sized for the worst chunk size of the range (a plan with chunks of C pairs has at most plan_pairs / C + n_items units of
// rr2_col_quads(C) quads per lane), not for the most units times the largest column
int32_t rhip_launch_miller_rr2(rhip_ctx* ctx, size_t n_items, uint32_t L, uint32_t C, uint32_t c_max, const uint32_t* pair_off, uint32_t uniform, const void* P, const void* Q,
                               const uint32_t* qref, const void* lines, const void* lines29, void* ws, void* mill, const MillerPlan* plan,
                               const void* work, const uint32_t* chunk_off, size_t units, uint32_t* started, size_t plan_pairs, uint32_t plan_c_lo) {
  if (lines && !lines29) {
    ctx->err = "reduced-radix pairing kernels: this prepared-lines handle was made with RABE_RR=0 and carries no converted lines; prepare it again";
    return RHIP_ERR_ARG;
  }
  if (c_max > 64 || !units) return RHIP_ERR_ARG;
  void* ws2 = nullptr;
  size_t quads = 0;
  if (plan) {
    for (uint32_t c = plan_c_lo ? plan_c_lo : 1; c <= c_max; c++) {
      size_t u = plan_pairs / c + n_items + 64;
      if (u > units) u = units;
      const size_t q = (2 * u + 63) / 64 * 64 * rr2_col_quads(c);
      if (q > quads) quads = q;
    }
  } else {
    quads = (2 * units + 63) / 64 * 64 * rr2_col_quads(c_max);
  }
  const int32_t rc = rhip_ensure_work(ctx, 14, quads * sizeof(uint4), &ws2);
  if (rc) return rc;
  KLAUNCH(ctx, "k_miller_pair_rr", k_miller_pair_rr, dim3(blocks_for(2 * units, RB_MILLER_BLOCK)), dim3(RB_MILLER_BLOCK), 0, ctx->stream, n_items, L, C, pair_off, uniform,
          (const G1M*)P, (const G2M*)Q, qref, (const uint4*)lines29, (uint4*)ws, (uint4*)ws2, (GtM*)mill, plan, (const uint2*)work, chunk_off, started);
  return RHIP_OK;
}
