// rabe_amd engine, translation unit of the REDUCED-RADIX pairing kernels (bn254/fp29.h, bn254/pairing29.h).
//
//   k_miller_multi_rr    what k_miller_multi (engine_jobs.hip) computes -- lane = (item, chunk of its pairs), all pairs of the chunk on one
//                        Fq12 accumulator -- with the field elements held as 9 signed 29-bit limbs: column sums without carry instructions,
//                        unreduced additions, one normalisation where a bound asks for it.  Same lane -> (item, chunk) map, same inputs
//                        (pair lists of 8 x 32-bit Montgomery records, prepared lines), same outputs: the Miller values and -- for the walk
//                        verdicts -- the points the walking pairs end on are converted back to the canonical 8 x 32-bit form, so everything
//                        downstream (k_final_exp, k_walk_verdicts) is unchanged and the bytes are identical.
// Pairings of `ac17::cp_decrypt` (src/schemes/ac17/mod.rs:415-418), bsw/mod.rs:291-294,308, lsw/mod.rs:275-280, aw11/mod.rs:340-350.
// There is no CPU fallback in this file.
#include "engine_internal.h"
#include "bn254/pairing29.h"

using rr::F;
using rr::F2;
using rr::F6;

// ---- the accumulator's home: per wave 24 quads + 12 dwords per lane ([quad][lane], [dword][lane]: conflict-free), element i of the
// twelve Fp coefficients = quads 2 i, 2 i + 1 (limbs 0..7) + dword i (limb 8).  110.6 KB per four-wave block: a block owns a CU.
static __shared__ uint4 rr_home_q[4 * 24 * 64];
static __shared__ uint32_t rr_home_d[4 * 12 * 64];
// the second and third right-hand operand of the line products' dot products (pairing29.h: facc_mul_by_line): two Fq2 per lane, same layout
// (slot s, coefficient c: element 2 s + c).  36.9 KB per block; with the home and fp29.h's side slot 152.5 of the CU's 160 KB.
static __shared__ uint4 rr_y_q[4 * 8 * 64];
static __shared__ uint32_t rr_y_d[4 * 4 * 64];

__device__ __forceinline__ Fp ld_fp_q(const uint4* p) {
  const uint4 a = p[0], b = p[1];
  Fp r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ F rr_from_quads(const uint4& a, const uint4& b, uint32_t c) {
  F r;
  r.l[0] = (int32_t)a.x; r.l[1] = (int32_t)a.y; r.l[2] = (int32_t)a.z; r.l[3] = (int32_t)a.w;
  r.l[4] = (int32_t)b.x; r.l[5] = (int32_t)b.y; r.l[6] = (int32_t)b.z; r.l[7] = (int32_t)b.w;
  r.l[8] = (int32_t)c;
  return r;
}
__device__ __forceinline__ uint4 rr_quad(const F& a, int h) {
  return make_uint4((uint32_t)a.l[4 * h], (uint32_t)a.l[4 * h + 1], (uint32_t)a.l[4 * h + 2], (uint32_t)a.l[4 * h + 3]);
}

__device__ __forceinline__ rr::i32x9 rr_lds_elem(const uint4* q, const uint32_t* d) {
  const uint4 a = q[0], b = q[64];
  rr::i32x9 r;
  r[0] = (int32_t)a.x; r[1] = (int32_t)a.y; r[2] = (int32_t)a.z; r[3] = (int32_t)a.w;
  r[4] = (int32_t)b.x; r[5] = (int32_t)b.y; r[6] = (int32_t)b.z; r[7] = (int32_t)b.w;
  r[8] = (int32_t)d[0];
  return r;
}
// f[ia] y0 + f[ib] y[1] + f[ic] y[2]: the accumulator's coefficients and y[1], y[2] come from the LDS, y0 in registers (18 of the 31 argument
// registers), the result as the other Fq2 routines return theirs (16 dwords + the two top limbs through the side slot)
#if defined(__HIP_DEVICE_COMPILE__)          // (the host pass of this file only needs the kernels' signatures)
__device__ __attribute__((noinline)) rr::Out16 rr_dot3_core(rr::i32x9 y0a, rr::i32x9 y0b, int ia, int ib, int ic) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const uint4* hq = rr_home_q + wv * (24 * 64) + ln;
  const uint32_t* hd = rr_home_d + wv * (12 * 64) + ln;
  const uint4* yq = rr_y_q + wv * (8 * 64) + ln;
  const uint32_t* yd = rr_y_d + wv * (4 * 64) + ln;
  const rr::i32x9 x0a = rr_lds_elem(hq + (4 * ia) * 64, hd + (2 * ia) * 64), x0b = rr_lds_elem(hq + (4 * ia + 2) * 64, hd + (2 * ia + 1) * 64);
  const rr::i32x9 x1a = rr_lds_elem(hq + (4 * ib) * 64, hd + (2 * ib) * 64), x1b = rr_lds_elem(hq + (4 * ib + 2) * 64, hd + (2 * ib + 1) * 64);
  const rr::i32x9 x2a = rr_lds_elem(hq + (4 * ic) * 64, hd + (2 * ic) * 64), x2b = rr_lds_elem(hq + (4 * ic + 2) * 64, hd + (2 * ic + 1) * 64);
  const rr::i32x9 y1a = rr_lds_elem(yq, yd), y1b = rr_lds_elem(yq + 2 * 64, yd + 64);
  const rr::i32x9 y2a = rr_lds_elem(yq + 4 * 64, yd + 2 * 64), y2b = rr_lds_elem(yq + 6 * 64, yd + 3 * 64);
  rr::i32x9 c0, c1;
  rr::dot3_raw(c0, c1, x0a, x0b, y0a, y0b, x1a, x1b, y1a, y1b, x2a, x2b, y2a, y2b);
  return rr::side_ret(c0, c1, rr::rr_side + threadIdx.x);
}
// f[ia] s + f[ib] y[1] + f[ic] y[2] with s in Fq (the unit-y lines of prepared pairs: pairing29.h facc_mul_by_line_s)
__device__ __attribute__((noinline)) rr::Out16 rr_dot3s_core(rr::i32x9 s, int ia, int ib, int ic) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const uint4* hq = rr_home_q + wv * (24 * 64) + ln;
  const uint32_t* hd = rr_home_d + wv * (12 * 64) + ln;
  const uint4* yq = rr_y_q + wv * (8 * 64) + ln;
  const uint32_t* yd = rr_y_d + wv * (4 * 64) + ln;
  const rr::i32x9 x0a = rr_lds_elem(hq + (4 * ia) * 64, hd + (2 * ia) * 64), x0b = rr_lds_elem(hq + (4 * ia + 2) * 64, hd + (2 * ia + 1) * 64);
  const rr::i32x9 x1a = rr_lds_elem(hq + (4 * ib) * 64, hd + (2 * ib) * 64), x1b = rr_lds_elem(hq + (4 * ib + 2) * 64, hd + (2 * ib + 1) * 64);
  const rr::i32x9 x2a = rr_lds_elem(hq + (4 * ic) * 64, hd + (2 * ic) * 64), x2b = rr_lds_elem(hq + (4 * ic + 2) * 64, hd + (2 * ic + 1) * 64);
  const rr::i32x9 y1a = rr_lds_elem(yq, yd), y1b = rr_lds_elem(yq + 2 * 64, yd + 64);
  const rr::i32x9 y2a = rr_lds_elem(yq + 4 * 64, yd + 2 * 64), y2b = rr_lds_elem(yq + 6 * 64, yd + 3 * 64);
  rr::i32x9 c0, c1;
  rr::dot3s_raw(c0, c1, x0a, x0b, s, x1a, x1b, y1a, y1b, x2a, x2b, y2a, y2b);
  return rr::side_ret(c0, c1, rr::rr_side + threadIdx.x);
}
#endif

// a lane's slice of the global workspace: per pair slot RR_SLOT_QUADS quads at stride 64 (one coalesced 1 KB access per quad and wave)
//   quads [0, 12): the running point T (six Fp: limbs 0..7), quads [12, 14): its six top limbs (+ 2 unused dwords)
//   quads [14, 22) + 22: the converted G2 argument (four Fp + their top limbs), quads [23, 27) + 27: the converted G1 argument
#define RR_SLOT_QUADS 28
#define RR_LINE_QUADS 9
#ifdef RB_MILLER_PROF
__device__ unsigned long long rb_miller_prof[8];          // summed over the lanes 0 of every wave: regions 0..4, [5] = whole loop, [6] = waves
#endif
struct DevMultiAcc29 {
#ifdef RB_MILLER_PROF
  unsigned long long* prof;
#endif
  const G1M* P;
  const G2M* Q;
  const uint32_t* qref;
  const uint4* lines29;      // prepared lines in this core's form, unit y-coefficient: 9 quads each (cx, c0: 4 x 8 limbs, then the 4 top limbs)
  int cnt;
  uint4* ws;                 // + lane
  F6* x;                     // the parked Fq6 (a local of the kernel)
  __device__ __forceinline__ int count() const { return cnt; }
  // the kinds of the lane's pairs, read once (kernel prologue): a global-memory round trip per pair and line event otherwise
  uint64_t walk_m, skip_m;
  __device__ __forceinline__ int kind(int j) const {
    if (cnt > 64) {
      const uint32_t v = qref[j];
      return v == RHIP_Q_WALK ? MP_WALK : v == RHIP_Q_SKIP ? MP_SKIP : MP_LINES;
    }
    return ((walk_m >> j) & 1ull) ? MP_WALK : ((skip_m >> j) & 1ull) ? MP_SKIP : MP_LINES;
  }
  // home
  __device__ __forceinline__ F ld_h(int i) const {
    const uint4* q = rr_home_q + (threadIdx.x >> 6) * (24 * 64) + (threadIdx.x & 63) + (2 * i) * 64;
    return rr_from_quads(q[0], q[64], rr_home_d[(threadIdx.x >> 6) * (12 * 64) + i * 64 + (threadIdx.x & 63)]);
  }
  __device__ __forceinline__ void st_h(int i, const F& a) const {
    uint4* q = rr_home_q + (threadIdx.x >> 6) * (24 * 64) + (threadIdx.x & 63) + (2 * i) * 64;
    q[0] = rr_quad(a, 0); q[64] = rr_quad(a, 1);
    rr_home_d[(threadIdx.x >> 6) * (12 * 64) + i * 64 + (threadIdx.x & 63)] = (uint32_t)a.l[8];
  }
  __device__ __forceinline__ F2 ld_h2(int i) const { return rr::mk2(ld_h(2 * i), ld_h(2 * i + 1)); }
  __device__ __forceinline__ void st_h2(int i, const F2& a) const { st_h(2 * i, a.c0); st_h(2 * i + 1, a.c1); }
  __device__ __forceinline__ F6 ld_f6(int h) const { return rr::mk6(ld_h2(3 * h), ld_h2(3 * h + 1), ld_h2(3 * h + 2)); }
  __device__ __forceinline__ void st_f6(int h, const F6& v) const { st_h2(3 * h, v.a0); st_h2(3 * h + 1, v.a1); st_h2(3 * h + 2, v.a2); }
  __device__ __forceinline__ void st_f2(int i, const F2& a) const { st_h2(i, a); }
  __device__ __forceinline__ void set_y(int slot, const F2& a) const {
    uint4* q = rr_y_q + (threadIdx.x >> 6) * (8 * 64) + (threadIdx.x & 63) + (4 * (slot - 1)) * 64;
    uint32_t* d = rr_y_d + (threadIdx.x >> 6) * (4 * 64) + (threadIdx.x & 63) + (2 * (slot - 1)) * 64;
    q[0] = rr_quad(a.c0, 0); q[64] = rr_quad(a.c0, 1); q[128] = rr_quad(a.c1, 0); q[192] = rr_quad(a.c1, 1);
    d[0] = (uint32_t)a.c0.l[8]; d[64] = (uint32_t)a.c1.l[8];
  }
  __device__ __forceinline__ F2 dot3(const F2& y0, int ia, int ib, int ic) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const rr::Out16 o = rr_dot3_core(y0.c0.l, y0.c1.l, ia, ib, ic);
    uint32_t* side = rr::rr_side + threadIdx.x;
    RB29_TAKE(o, c0, c1, side)
    return rr::mk2(rr::mk<1, 1>(c0), rr::mk<1, 1>(c1));
#else
    return y0;
#endif
  }
  __device__ __forceinline__ F2 dot3s(const F& y, int ia, int ib, int ic) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const rr::Out16 o = rr_dot3s_core(y.l, ia, ib, ic);
    uint32_t* side = rr::rr_side + threadIdx.x;
    RB29_TAKE(o, c0, c1, side)
    return rr::mk2(rr::mk<1, 1>(c0), rr::mk<1, 1>(c1));
#else
    return rr::mk2(y, y);
#endif
  }
  __device__ __forceinline__ F6 ld_x() const { return *x; }
  __device__ __forceinline__ void st_x(const F6& v) const { *x = v; }
  __device__ __forceinline__ void fence() const { asm volatile("" ::: "memory"); }
  // workspace: NF consecutive Fp starting at quad q0 of slot j, their top limbs packed in the quads from qt on
  template <int NF> __device__ __forceinline__ void ld_n(int j, int q0, int qt, F* out) const {
    const uint4* p = ws + ((size_t)j * RR_SLOT_QUADS + q0) * 64;
    uint32_t tops[8];
#pragma unroll
    for (int k = 0; k < (NF + 3) / 4; k++) {
      const uint4 t = ws[((size_t)j * RR_SLOT_QUADS + qt + k) * 64];
      tops[4 * k] = t.x; tops[4 * k + 1] = t.y; tops[4 * k + 2] = t.z; tops[4 * k + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < NF; i++) out[i] = rr_from_quads(p[(size_t)(2 * i) * 64], p[(size_t)(2 * i + 1) * 64], tops[i]);
  }
  template <int NF> __device__ __forceinline__ void st_n(int j, int q0, int qt, const F* in) const {
    uint4* p = ws + ((size_t)j * RR_SLOT_QUADS + q0) * 64;
#pragma unroll
    for (int i = 0; i < NF; i++) { p[(size_t)(2 * i) * 64] = rr_quad(in[i], 0); p[(size_t)(2 * i + 1) * 64] = rr_quad(in[i], 1); }
#pragma unroll
    for (int k = 0; k < (NF + 3) / 4; k++) {
      uint32_t t[4];
#pragma unroll
      for (int e = 0; e < 4; e++) t[e] = (4 * k + e < NF) ? (uint32_t)in[4 * k + e].l[8] : 0u;
      ws[((size_t)j * RR_SLOT_QUADS + qt + k) * 64] = make_uint4(t[0], t[1], t[2], t[3]);
    }
  }
  __device__ __forceinline__ rr::G2Hom29 ld_t(int j) const {
    F e[6];
    ld_n<6>(j, 0, 12, e);
    return rr::G2Hom29{rr::mk2(e[0], e[1]), rr::mk2(e[2], e[3]), rr::mk2(e[4], e[5])};
  }
  __device__ __forceinline__ void st_t(int j, const rr::G2Hom29& t) const {
    const F e[6] = {t.x.c0, t.x.c1, t.y.c0, t.y.c1, t.z.c0, t.z.c1};
    st_n<6>(j, 0, 12, e);
  }
  __device__ __forceinline__ rr::G2Aff29 q(int j) const {
    F e[4];
    ld_n<4>(j, 14, 22, e);
    return rr::G2Aff29{rr::mk2(e[0], e[1]), rr::mk2(e[2], e[3])};
  }
  __device__ __forceinline__ rr::MillerP29 p(int j) const {
    F e[2];
    ld_n<2>(j, 23, 27, e);
    return rr::MillerP29{e[0], e[1]};
  }
  __device__ __forceinline__ rr::LineU29 line_u(int j, int n) const {
    const uint4* p = lines29 + ((size_t)qref[j] * RB_MILLER_LINES + n) * RR_LINE_QUADS;
    const uint4 t0 = p[8];
    rr::LineU29 r;
    r.cx = rr::mk2(rr_from_quads(p[0], p[1], t0.x), rr_from_quads(p[2], p[3], t0.y));
    r.c0 = rr::mk2(rr_from_quads(p[4], p[5], t0.z), rr_from_quads(p[6], p[7], t0.w));
    return r;
  }
  // once, before the loop: the lane's arguments in the field core's representation
  __device__ __forceinline__ void begin() const {
    for (int j = 0; j < cnt; j++) {
      const int k = kind(j);
      if (k == MP_SKIP) continue;
      {
        const uint4* g = (const uint4*)(P + j);
        const F e[2] = {rr::from_fp(ld_fp_q(g)), rr::from_fp(ld_fp_q(g + 2))};
        st_n<2>(j, 23, 27, e);
      }
      if (k == MP_WALK) {
        const uint4* g = (const uint4*)(Q + j);
        const F e[4] = {rr::from_fp(ld_fp_q(g)), rr::from_fp(ld_fp_q(g + 2)), rr::from_fp(ld_fp_q(g + 4)), rr::from_fp(ld_fp_q(g + 6))};
        st_n<4>(j, 14, 22, e);
        st_t(j, rr::G2Hom29{rr::mk2(e[0], e[1]), rr::mk2(e[2], e[3]), rr::one2()});
      }
    }
  }
};

// Same arguments and lane map as k_miller_multi (engine_jobs.hip); ws29: the workspace in this kernel's layout; ws: the one k_walk_verdicts
// reads ([wave][pair slot][12 quads][lane], 8 x 32-bit Montgomery limbs) -- written once, at the end, for the walking pairs.
__device__ __forceinline__ void miller_multi_rr_lane(size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const G1M* P,
                                                     const G2M* Q, const uint32_t* qref, const uint4* lines29, uint4* ws, uint4* ws29, GtM* mill,
                                                     const MillerPlan* plan, const uint2* work, const uint32_t* chunk_off) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t first;
  int cnt;
  uint32_t Cw;
  GtM* out;
  if (plan) {
    if (t >= plan->W) return;
    Cw = plan->C;
    const uint2 w = work[t];
    const uint64_t lo = pair_off[w.x], hi = pair_off[w.x + 1];
    const uint32_t p_item = (uint32_t)(hi - lo), nch = (p_item + Cw - 1) / Cw;
    const uint32_t base = p_item / nch, rem = p_item % nch, cc = w.y;
    first = lo + (uint64_t)cc * base + (cc < rem ? cc : rem);
    cnt = (int)(base + (cc < rem ? 1u : 0u));
    out = mill + chunk_off[w.x] + cc;
  } else {
    if (t >= n_items * L) return;
    Cw = C;
    const size_t c = t / n_items, item = (t % n_items + c * RB_MILLER_BLOCK) % n_items;
    const uint64_t lo = pair_off ? pair_off[item] : (uint64_t)item * uniform, hi = pair_off ? pair_off[item + 1] : (uint64_t)(item + 1) * uniform;
    const uint32_t p_item = (uint32_t)(hi - lo);
    const uint32_t nch = (p_item + C - 1) / C;
    out = mill + item * L + c;
    if (c >= nch) {
      st_gt_m(out, fp12_one());
      return;
    }
    const uint32_t base = p_item / nch, rem = p_item % nch, cc = (uint32_t)c;
    first = lo + (uint64_t)cc * base + (cc < rem ? cc : rem);
    cnt = (int)(base + (cc < rem ? 1u : 0u));
  }
  F6 parked;
  uint64_t walk_m = 0, skip_m = 0;
  for (int j = 0; j < cnt && j < 64; j++) {
    const uint32_t v = qref[first + j];
    if (v == RHIP_Q_WALK) walk_m |= 1ull << j;
    else if (v == RHIP_Q_SKIP) skip_m |= 1ull << j;
  }
#ifdef RB_MILLER_PROF
  unsigned long long prof[5] = {0, 0, 0, 0, 0};
  const unsigned long long prof_t0 = clock64();
  const DevMultiAcc29 acc{prof, P + first, Q + first, qref + first, lines29, cnt, ws29 + (t >> 6) * ((size_t)Cw * RR_SLOT_QUADS * 64) + (t & 63), &parked, walk_m, skip_m};
#else
  const DevMultiAcc29 acc{P + first, Q + first, qref + first, lines29, cnt, ws29 + (t >> 6) * ((size_t)Cw * RR_SLOT_QUADS * 64) + (t & 63), &parked, walk_m, skip_m};
#endif
  rr::miller_loop_multi(acc);
#ifdef RB_MILLER_PROF
  if ((threadIdx.x & 63) == 0) {
    for (int k = 0; k < 5; k++) atomicAdd(&rb_miller_prof[k], prof[k]);
    atomicAdd(&rb_miller_prof[5], clock64() - prof_t0);
    atomicAdd(&rb_miller_prof[6], 1ull);
  }
#endif
  // the value, back in the canonical Montgomery form of the 8 x 32-bit core
  {
    uint32_t* o = out->l;
#pragma unroll 1
    for (int i = 0; i < 12; i++) st_fp_m(o + 8 * i, rr::to_fp(acc.ld_h(i)));
  }
  // the points the walking pairs ended on, where k_walk_verdicts looks for them
  uint4* wl = ws + (t >> 6) * ((size_t)Cw * 12 * 64) + (t & 63);
  for (int j = 0; j < cnt; j++) {
    if (acc.kind(j) != MP_WALK) continue;
    F e[6];
    acc.ld_n<6>(j, 0, 12, e);
    uint4* p = wl + (size_t)(12 * j) * 64;
#pragma unroll 1
    for (int k = 0; k < 6; k++) {
      const Fp v = rr::to_fp(e[k]);
      p[(size_t)(2 * k) * 64] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
      p[(size_t)(2 * k + 1) * 64] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
    }
  }
}

__global__ void __launch_bounds__(RB_MILLER_BLOCK, 1) k_miller_multi_rr(size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const G1M* P,
                                                                       const G2M* Q, const uint32_t* qref, const uint4* lines29, uint4* ws, uint4* ws29, GtM* mill,
                                                                       const MillerPlan* plan, const uint2* work, const uint32_t* chunk_off, uint32_t* started) {
  if (started && threadIdx.x == 0) { atomicAdd(started, 1u); __threadfence(); }          // rhip_ctx_release_when_miller_resident
  miller_multi_rr_lane(n_items, L, C, pair_off, uniform, P, Q, qref, lines29, ws, ws29, mill, plan, work, chunk_off);
}

// ------------------------------------------------------------------------------------------------ final exponentiation
// k_final_exp (engine.hip) on the reduced-radix core: out[item] = (mul_in ? mul_in[item] : 1) * FE(prod of the item's Miller values), the chain of
// pairing29.h: final_exponentiation_ws value by value.  The Fq12 values of the chain live in the context's workspace, FE_SLOTS slots per lane of
// RR_FE_QUADS quads ([slot][quad][lane]: 24 quads of limbs 0..7, 3 quads of top limbs); the value being multiplied into lives in the LDS home.
#define RR_FE_QUADS 27
struct DevHome29 {
  F6* x;
  __device__ __forceinline__ F ld_h(int i) const {
    const uint4* q = rr_home_q + (threadIdx.x >> 6) * (24 * 64) + (threadIdx.x & 63) + (2 * i) * 64;
    return rr_from_quads(q[0], q[64], rr_home_d[(threadIdx.x >> 6) * (12 * 64) + i * 64 + (threadIdx.x & 63)]);
  }
  __device__ __forceinline__ void st_h(int i, const F& a) const {
    uint4* q = rr_home_q + (threadIdx.x >> 6) * (24 * 64) + (threadIdx.x & 63) + (2 * i) * 64;
    q[0] = rr_quad(a, 0); q[64] = rr_quad(a, 1);
    rr_home_d[(threadIdx.x >> 6) * (12 * 64) + i * 64 + (threadIdx.x & 63)] = (uint32_t)a.l[8];
  }
  __device__ __forceinline__ F2 ld_h2(int i) const { return rr::mk2(ld_h(2 * i), ld_h(2 * i + 1)); }
  __device__ __forceinline__ void st_h2(int i, const F2& a) const { st_h(2 * i, a.c0); st_h(2 * i + 1, a.c1); }
  __device__ __forceinline__ F6 ld_f6(int h) const { return rr::mk6(ld_h2(3 * h), ld_h2(3 * h + 1), ld_h2(3 * h + 2)); }
  __device__ __forceinline__ void st_f6(int h, const F6& v) const { st_h2(3 * h, v.a0); st_h2(3 * h + 1, v.a1); st_h2(3 * h + 2, v.a2); }
  __device__ __forceinline__ F6 ld_x() const { return *x; }
  __device__ __forceinline__ void st_x(const F6& v) const { *x = v; }
  __device__ __forceinline__ void fence() const { asm volatile("" ::: "memory"); }
};
struct DevWs29 {
  uint4* base;         // + lane
  size_t stride;       // lanes (padded to 64)
  F6* x;
  __device__ __forceinline__ F6 ld6(int slot, int h) const {
    const uint4* p = base + ((size_t)slot * RR_FE_QUADS + 12 * h) * stride;
    const uint4* tp = base + ((size_t)slot * RR_FE_QUADS + 24) * stride;
    uint32_t tops[6];
    if (h == 0) { const uint4 a = tp[0], b = tp[stride]; tops[0] = a.x; tops[1] = a.y; tops[2] = a.z; tops[3] = a.w; tops[4] = b.x; tops[5] = b.y; }
    else { const uint4 b = tp[stride], c = tp[2 * stride]; tops[0] = b.z; tops[1] = b.w; tops[2] = c.x; tops[3] = c.y; tops[4] = c.z; tops[5] = c.w; }
    F e[6];
#pragma unroll
    for (int i = 0; i < 6; i++) e[i] = rr_from_quads(p[(size_t)(2 * i) * stride], p[(size_t)(2 * i + 1) * stride], tops[i]);
    return rr::mk6(rr::mk2(e[0], e[1]), rr::mk2(e[2], e[3]), rr::mk2(e[4], e[5]));
  }
  // the top limbs of a half fill one and a half quads: the shared middle quad is written by whoever stores second, so a half is only ever
  // stored together with the other one (st) or after it (st6 h = 0 then h = 1 -- wsx_from_home's order)
  __device__ __forceinline__ void st(int slot, const rr::F12& a) const {
    uint4* p = base + (size_t)slot * RR_FE_QUADS * stride;
    const F e[12] = {a.c0.a0.c0, a.c0.a0.c1, a.c0.a1.c0, a.c0.a1.c1, a.c0.a2.c0, a.c0.a2.c1, a.c1.a0.c0, a.c1.a0.c1, a.c1.a1.c0, a.c1.a1.c1, a.c1.a2.c0, a.c1.a2.c1};
#pragma unroll
    for (int i = 0; i < 12; i++) { p[(size_t)(2 * i) * stride] = rr_quad(e[i], 0); p[(size_t)(2 * i + 1) * stride] = rr_quad(e[i], 1); }
#pragma unroll
    for (int k = 0; k < 3; k++)
      p[(size_t)(24 + k) * stride] = make_uint4((uint32_t)e[4 * k].l[8], (uint32_t)e[4 * k + 1].l[8], (uint32_t)e[4 * k + 2].l[8], (uint32_t)e[4 * k + 3].l[8]);
  }
  __device__ __forceinline__ rr::F12 ld(int slot) const { return rr::F12{ld6(slot, 0), ld6(slot, 1)}; }
  __device__ __forceinline__ void st6(int slot, int h, const F6& v) const {
    uint4* p = base + ((size_t)slot * RR_FE_QUADS + 12 * h) * stride;
    uint4* tp = base + ((size_t)slot * RR_FE_QUADS + 24) * stride;
    const F e[6] = {v.a0.c0, v.a0.c1, v.a1.c0, v.a1.c1, v.a2.c0, v.a2.c1};
#pragma unroll
    for (int i = 0; i < 6; i++) { p[(size_t)(2 * i) * stride] = rr_quad(e[i], 0); p[(size_t)(2 * i + 1) * stride] = rr_quad(e[i], 1); }
    if (h == 0) {
      tp[0] = make_uint4((uint32_t)e[0].l[8], (uint32_t)e[1].l[8], (uint32_t)e[2].l[8], (uint32_t)e[3].l[8]);
      uint2* mid = (uint2*)(tp + stride);
      mid[0] = make_uint2((uint32_t)e[4].l[8], (uint32_t)e[5].l[8]);
    } else {
      uint2* mid = (uint2*)(tp + stride);
      mid[1] = make_uint2((uint32_t)e[0].l[8], (uint32_t)e[1].l[8]);
      tp[2 * stride] = make_uint4((uint32_t)e[2].l[8], (uint32_t)e[3].l[8], (uint32_t)e[4].l[8], (uint32_t)e[5].l[8]);
    }
  }
  __device__ __forceinline__ DevHome29 home() const { return DevHome29{x}; }
};
__device__ __forceinline__ void final_exp_rr_lane(size_t n_items, const uint32_t* off, uint32_t stride, const GtM* mill, const rhip_gt* mul_in, rhip_gt* out,
                                                  uint4* ws_base, size_t ws_stride) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  const size_t lo = off ? off[i] : i * stride, hi = off ? off[i + 1] : (i + 1) * stride;
  F6 parked;
  const DevWs29 ws{ws_base + i, ws_stride, &parked};
  if (lo == hi) ws.st(FE_T0, rr::from_fp12(fp12_one()));
  for (size_t j = lo; j < hi; j++) {
    ws.st(j == lo ? FE_T0 : FE_T1, rr::from_fp12(ld_gt_m(mill + j)));
    if (j != lo) rr::wsx_mul(ws, FE_T0, FE_T0, false, FE_T1, false);
  }
  rr::final_exponentiation_ws(ws);
  if (mul_in) {
    ws.st(FE_T0, rr::from_fp12(load_gt(mul_in[i].l)));
    rr::wsx_mul(ws, FE_T1, FE_T0, false, FE_T1, false);
  }
  store_gt(out[i].l, rr::to_fp12(ws.ld(FE_T1)));
}
__global__ void __launch_bounds__(256, 1) k_final_exp_rr(size_t n_items, const uint32_t* off, uint32_t stride, const GtM* mill, const rhip_gt* mul_in, rhip_gt* out,
                                                       uint4* ws_base, size_t ws_stride, uint32_t* started) {
  if (started && threadIdx.x == 0) { atomicAdd(started, 1u); __threadfence(); }
  final_exp_rr_lane(n_items, off, stride, mill, mul_in, out, ws_base, ws_stride);
}
int32_t rhip_launch_final_exp_rr(rhip_ctx* ctx, size_t n_items, const uint32_t* off, uint32_t stride, const void* mill, const rhip_gt* mul_in, rhip_gt* out,
                                 uint32_t* started) {
  const size_t lanes = (n_items + 63) / 64 * 64;
  const int32_t rc = rhip_ensure_fe_ws(ctx, lanes * FE_SLOTS * RR_FE_QUADS * sizeof(uint4));
  if (rc) return rc;
  KLAUNCH(ctx, "k_final_exp_rr", k_final_exp_rr, dim3(blocks_for(n_items, 256)), dim3(256), 0, ctx->stream, n_items, off, stride, (const GtM*)mill, mul_in, out,
          (uint4*)ctx->fe_ws, lanes, started);
  return RHIP_OK;
}

// ---- micro-benchmark of the out-of-line routines at the kernels' own occupancy (one wave per SIMD, the 152 KB home: a block owns a CU): shader
// cycles per call, measured inside the kernel (s_memtime), per wave.  which: 0 rr_dot3_core, 1 rr_dot3s_core, 2 mul2_core, 3 sqr2_sd_core-free
// baseline (empty loop).  tools/ubench_cores.py prints the table; not part of the product path.
#ifdef RB_UBENCH_CORES          // diagnostic builds only (tools/ubench_cores.py: RABE_HIPCC_FLAGS=-DRB_UBENCH_CORES); not in the product library
__global__ void __launch_bounds__(RB_MILLER_BLOCK, 1) k_ubench_cores(uint32_t iters, int which, uint64_t* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  for (int i = 0; i < 24; i++) rr_home_q[wv * (24 * 64) + i * 64 + ln] = make_uint4(0x0123456u + i + ln, 0x0abcdefu ^ (i * 77), 0x0fedcbau - ln, 0x07777777u + i);
  for (int i = 0; i < 12; i++) rr_home_d[wv * (12 * 64) + i * 64 + ln] = 1000u + i;
  for (int i = 0; i < 8; i++) rr_y_q[wv * (8 * 64) + i * 64 + ln] = make_uint4(0x0333333u + i, 0x0444444u + ln, 0x0555555u, 0x0666666u - i);
  for (int i = 0; i < 4; i++) rr_y_d[wv * (4 * 64) + i * 64 + ln] = 2000u + i;
  rr::i32x9 a, b;
  for (int i = 0; i < 9; i++) { a[i] = 0x0111111 + i + ln; b[i] = 0x0222222 - i; }
  a[8] = 1234; b[8] = 2345;
  uint32_t* side = rr::rr_side + threadIdx.x;
  __syncthreads();
  const uint64_t t0 = clock64();
  for (uint32_t it = 0; it < iters; it++) {
    const int ia = (int)(it % 6), ib = (int)((it + 1) % 6), ic = (int)((it + 3) % 6);
    rr::Out16 o;
    if (which == 0) o = rr_dot3_core(a, b, ia, ib, ic);
    else if (which == 1) o = rr_dot3s_core(a, ia, ib, ic);
    else if (which == 2) { rr::i32x4 lo; lo[0] = b[0]; lo[1] = b[1]; lo[2] = b[2]; lo[3] = b[3]; o = rr::mul2_core(a, b, a, lo); }
    else { o.lo0 = (rr::i32x8)(0); o.lo1 = (rr::i32x8)(0); }
    a[0] = (a[0] ^ (o.lo0[0] & 1)) & 0x0fffffff;          // a dependence from call to call, as in the loop
    b[1] = (b[1] ^ ((int32_t)side[0] & 1)) & 0x0fffffff;
  }
  const uint64_t t1 = clock64();
  if (ln == 0) out[(size_t)blockIdx.x * 4 + wv] = (t1 - t0) + (uint64_t)((a[0] ^ b[1]) & 1);
#endif
}
#endif
#ifdef RB_MILLER_PROF
// diagnostic build only: reads and clears the region sums of the k_miller_multi_rr launches since the last call
extern "C" int32_t rhip_debug_miller_prof(rhip_ctx* ctx, unsigned long long out[8]) {
  if (!ctx || !out) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(rb_miller_prof), 8 * sizeof(unsigned long long)));
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(rb_miller_prof), z, sizeof z));
  return RHIP_OK;
}
#endif
#ifdef RB_UBENCH_CORES
extern "C" int32_t rhip_debug_ubench_cores(rhip_ctx* ctx, uint32_t iters, int32_t which, uint32_t blocks, uint64_t* d_out) {
  if (!ctx || !d_out) return RHIP_ERR_ARG;
  KLAUNCH(ctx, "k_ubench_cores", k_ubench_cores, dim3(blocks), dim3(RB_MILLER_BLOCK), 0, ctx->stream, iters, (int)which, d_out);
  return RHIP_OK;
}
#endif

// When it runs: pairing mode 29 (rhip_ctx_set_pairing_mode / RABE_PAIRING_MODE) -- every multi-pairing launch; mode 0 (auto) -- the launches
// the six-lane kernels do not take (the caller asks rhip_use_c6 first), unless RABE_RR=0 (A/B runs, and the conservative switch).
bool rhip_use_rr(const rhip_ctx* ctx) {
  static const int on = getenv("RABE_RR") ? atoi(getenv("RABE_RR")) : 1;
  return rhip_mode(ctx) == 29 || rhip_mode(ctx) == 58 || (on != 0 && rhip_mode(ctx) == 0);
}
// Does a handle that carries prepared lines need their converted form?  Always, unless the reduced-radix kernels are switched off for the
// process (RABE_RR=0, the conservative switch) AND the context is not in a mode that forces them: then the conversion (+ 75 % of the lines'
// memory) is skipped, and a later launch that wants the converted lines of such a handle fails with a message instead of reading nothing.
bool rhip_want_lines29(const rhip_ctx* ctx) {
  static const int on = getenv("RABE_RR") ? atoi(getenv("RABE_RR")) : 1;
  return on != 0 || ctx->pairing_mode == 29 || ctx->pairing_mode == 58 || ctx->pairing_mode == 99;
}
// one lane per prepared triple (cy, cx, c0): divided by its y-coefficient -- cx / cy, c0 / cy, the unit-y form pairing29.h's facc_ell_u takes (the
// factor lies in Fq2 and dies in the final exponentiation) -- and converted: the 9-quad record line_u() reads (4 x 8 limbs, then the 4 top limbs)
__global__ void __launch_bounds__(256) k_lines_to_rr(size_t n, const LineM* in, uint4* out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint4* p = (const uint4*)(in + t);
  const Fp2 cy{ld_fp_q(p), ld_fp_q(p + 2)}, cx{ld_fp_q(p + 4), ld_fp_q(p + 6)}, c0{ld_fp_q(p + 8), ld_fp_q(p + 10)};
  const Fp2 iy = fp2_inv(cy);          // cy = 0 only for arguments outside G2 (the chord / tangent of a point of order <= 2): the line stays 0
  const Fp2 ux = fp2_mul(cx, iy), u0 = fp2_mul(c0, iy);
  const Fp e[4] = {ux.c0, ux.c1, u0.c0, u0.c1};
  uint4* o = out + t * RR_LINE_QUADS;
  uint32_t top[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const F v = rr::from_fp(e[k]);
    o[2 * k] = rr_quad(v, 0);
    o[2 * k + 1] = rr_quad(v, 1);
    top[k] = (uint32_t)v.l[8];
  }
  o[8] = make_uint4(top[0], top[1], top[2], top[3]);
}
int32_t rhip_lines_to_rr(rhip_ctx* ctx, size_t n_lines, const void* lines, void** out) {
  HIP_TRY(ctx, hipMalloc(out, n_lines * RR_LINE_QUADS * sizeof(uint4)));
  KLAUNCH(ctx, "k_lines_to_rr", k_lines_to_rr, dim3(blocks_for(n_lines, 256)), dim3(256), 0, ctx->stream, n_lines, (const LineM*)lines, (uint4*)*out);
  return RHIP_OK;
}
int32_t rhip_launch_miller_rr(rhip_ctx* ctx, size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const void* P, const void* Q,
                              const uint32_t* qref, const void* lines, const void* lines29, void* ws, size_t ws_bytes, void* mill, const MillerPlan* plan,
                              const void* work, const uint32_t* chunk_off, size_t lanes, uint32_t* started) {
  if (lines && !lines29) {          // a handle prepared while the reduced-radix kernels were switched off (rhip_want_lines29) has no converted lines
    ctx->err = "reduced-radix pairing kernels: this prepared-lines handle was made with RABE_RR=0 and carries no converted lines; prepare it again";
    return RHIP_ERR_ARG;
  }
  void* ws29 = nullptr;
  const int32_t rc = rhip_ensure_work(ctx, 11, ws_bytes / 12 * RR_SLOT_QUADS, &ws29);
  if (rc) return rc;
  KLAUNCH(ctx, "k_miller_multi_rr", k_miller_multi_rr, dim3(blocks_for(lanes, RB_MILLER_BLOCK)), dim3(RB_MILLER_BLOCK), 0, ctx->stream, n_items, L, C, pair_off, uniform,
          (const G1M*)P, (const G2M*)Q, qref, (const uint4*)lines29, (uint4*)ws, (uint4*)ws29, (GtM*)mill, plan, (const uint2*)work, chunk_off, started);
  return RHIP_OK;
}
