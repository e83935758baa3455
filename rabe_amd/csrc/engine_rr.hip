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

// a lane's slice of the global workspace: per pair slot RR_SLOT_QUADS quads at stride 64 (one coalesced 1 KB access per quad and wave)
//   quads [0, 12): the running point T (six Fp: limbs 0..7), quads [12, 14): its six top limbs (+ 2 unused dwords)
//   quads [14, 22) + 22: the converted G2 argument (four Fp + their top limbs), quads [23, 27) + 27: the converted G1 argument
#define RR_SLOT_QUADS 28
struct DevMultiAcc29 {
  const G1M* P;
  const G2M* Q;
  const uint32_t* qref;
  const LineM* lines;
  int cnt;
  uint4* ws;                 // + lane
  F6* x;                     // the parked Fq6 (a local of the kernel)
  __device__ __forceinline__ int count() const { return cnt; }
  __device__ __forceinline__ int kind(int j) const {
    const uint32_t v = qref[j];
    return v == RHIP_Q_WALK ? MP_WALK : v == RHIP_Q_SKIP ? MP_SKIP : MP_LINES;
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
  __device__ __forceinline__ rr::Line29 line(int j, int n) const {
    const uint4* p = (const uint4*)(lines + (size_t)qref[j] * RB_MILLER_LINES + n);
    rr::Line29 r;
    r.cy = rr::mk2(rr::from_fp(ld_fp_q(p)), rr::from_fp(ld_fp_q(p + 2)));
    r.cx = rr::mk2(rr::from_fp(ld_fp_q(p + 4)), rr::from_fp(ld_fp_q(p + 6)));
    r.c0 = rr::mk2(rr::from_fp(ld_fp_q(p + 8)), rr::from_fp(ld_fp_q(p + 10)));
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
__global__ void __launch_bounds__(RB_MILLER_BLOCK, 1) k_miller_multi_rr(size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const G1M* P,
                                                                       const G2M* Q, const uint32_t* qref, const LineM* lines, uint4* ws, uint4* ws29, GtM* mill,
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
  const DevMultiAcc29 acc{P + first, Q + first, qref + first, lines, cnt, ws29 + (t >> 6) * ((size_t)Cw * RR_SLOT_QUADS * 64) + (t & 63), &parked};
  rr::miller_loop_multi(acc);
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

// When it runs: pairing mode 29 (rhip_ctx_set_pairing_mode / RABE_PAIRING_MODE) -- every multi-pairing launch; mode 0 (auto) -- the launches
// the six-lane kernels do not take (the caller asks rhip_use_c6 first), unless RABE_RR=0 (A/B runs, and the conservative switch).
bool rhip_use_rr(const rhip_ctx* ctx) {
  static const int on = getenv("RABE_RR") ? atoi(getenv("RABE_RR")) : 1;
  return ctx->pairing_mode == 29 || (on != 0 && ctx->pairing_mode == 0);
}
int32_t rhip_launch_miller_rr(rhip_ctx* ctx, size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const void* P, const void* Q,
                              const uint32_t* qref, const void* lines, void* ws, size_t ws_bytes, void* mill, const MillerPlan* plan, const void* work,
                              const uint32_t* chunk_off, size_t lanes) {
  void* ws29 = nullptr;
  const int32_t rc = rhip_ensure_work(ctx, 11, ws_bytes / 12 * RR_SLOT_QUADS, &ws29);
  if (rc) return rc;
  KLAUNCH(ctx, "k_miller_multi_rr", k_miller_multi_rr, dim3(blocks_for(lanes, RB_MILLER_BLOCK)), dim3(RB_MILLER_BLOCK), 0, ctx->stream, n_items, L, C, pair_off, uniform,
          (const G1M*)P, (const G2M*)Q, qref, (const LineM*)lines, (uint4*)ws, (uint4*)ws29, (GtM*)mill, plan, (const uint2*)work, chunk_off);
  return RHIP_OK;
}
