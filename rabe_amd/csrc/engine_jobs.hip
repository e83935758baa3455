// rabe_amd engine, second translation unit: the device-resident Level B paths of bsw / lsw / aw11.
//
// Every decrypt of those schemes is a product of pairings with coefficient-folded G1 arguments (SURVEY.md Appendix
// B.4 / B.5), so one set of kernels carries all three:
//   k_*_dec_pairs        scheme-specific gather: base point + Lagrange coefficient of every pair of every item,
//                        variable-base multiplication over the NAF of the coefficient (bn254/curve.h: jac_mul_naf),
//                        affine Montgomery output with one field inversion per block; the pair's G2 argument is either
//                        copied (a walking pair) or named by its prepared-line block
//   k_msm_g1 / k_msm_g2  sums  sum_j k_j P_j  with shared doublings (jac_msm_naf): lsw's e(sum -c_y D1_y, e2),
//                        aw11's e(-H(gid), sum c_x C3_x)
//   k_miller_multi       lane = (item, chunk of its pairs): ALL pairs of the chunk on one Fq12 accumulator
//                        (bn254/pairing.h: miller_loop_multi); running G2 points in a coalesced global workspace
//   k_final_exp          (engine.hip) product of the item's few chunk values + ONE final exponentiation
// and the encrypt / keygen sides are Fr kernels (share generation over flattened policy trees) in front of the
// fixed-base table kernels of engine.hip.
// There is no CPU fallback anywhere in this file.
#include "engine_internal.h"

// ------------------------------------------------------------------------------------------------ small device helpers
__device__ __forceinline__ Fp ld_fp_q(const uint4* p) {
  const uint4 a = p[0], b = p[1];
  Fp r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ void st_fp_q(uint4* p, const Fp& a) {
  p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  p[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
// Montgomery records are 64 / 128 / 192 bytes and 16-byte aligned: 128-bit accesses
__device__ __forceinline__ G1Aff ld_g1_q(const G1M* p) { const uint4* q = (const uint4*)p; return G1Aff{ld_fp_q(q), ld_fp_q(q + 2)}; }
__device__ __forceinline__ void st_g1_q(G1M* p, const G1Aff& a) { uint4* q = (uint4*)p; st_fp_q(q, a.x); st_fp_q(q + 2, a.y); }
__device__ __forceinline__ G2Aff ld_g2_q(const G2M* p) {
  const uint4* q = (const uint4*)p;
  return G2Aff{Fp2{ld_fp_q(q), ld_fp_q(q + 2)}, Fp2{ld_fp_q(q + 4), ld_fp_q(q + 6)}};
}
__device__ __forceinline__ void st_g2_q(G2M* p, const G2Aff& a) {
  uint4* q = (uint4*)p;
  st_fp_q(q, a.x.c0); st_fp_q(q + 2, a.x.c1); st_fp_q(q + 4, a.y.c0); st_fp_q(q + 6, a.y.c1);
}
// item owning flat index t: the i with off[i] <= t < off[i+1] (off non-decreasing, off[n] > t)
__device__ __forceinline__ size_t owner_of(const uint32_t* off, size_t n, size_t t) {
  size_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) >> 1;
    if (off[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}
// k > (r-1)/2 for a canonical k < r: then r - k is the shorter scalar and the base is negated instead
__device__ __forceinline__ bool fr_above_half(const uint32_t k[8]) {
  // (r-1)/2 = r >> 1 (r is odd)
  bool gt = false, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    const uint32_t h = (FrParams::mod(i) >> 1) | (i < 7 ? (FrParams::mod(i + 1) << 31) : 0u);
    if (!decided && k[i] != h) { gt = k[i] > h; decided = true; }
  }
  return gt;
}
__device__ __forceinline__ void fr_negate_canon(uint32_t k[8]) {      // k <- r - k for 0 < k < r
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = subb32(FrParams::mod(i), k[i], borrow);
}
// scalar and sign in the shorter form: returns true when the base has to be negated
__device__ __forceinline__ bool fr_shorten(uint32_t k[8]) {
  if (!fr_above_half(k)) return false;
  fr_negate_canon(k);
  return true;
}

// ------------------------------------------------------------------------------------------------ pair lists
// A batch's pairs after the gather kernel: P (affine Montgomery), Q (affine Montgomery, walking pairs only) and
// qref: RHIP_Q_WALK, RHIP_Q_SKIP (an argument at infinity: the pair contributes 1) or the prepared-line block of Q.
struct PairLists {
  G1M* P;
  G2M* Q;
  uint32_t* qref;
};
static int32_t alloc_pair_lists(rhip_ctx* ctx, size_t total_pairs, PairLists* pl) {
  void* p = nullptr;
  int32_t rc = rhip_ensure_work(ctx, 0, total_pairs * sizeof(G1M), &p);
  if (rc) return rc;
  pl->P = (G1M*)p;
  rc = rhip_ensure_work(ctx, 1, total_pairs * sizeof(G2M), &p);
  if (rc) return rc;
  pl->Q = (G2M*)p;
  rc = rhip_ensure_work(ctx, 2, total_pairs * sizeof(uint32_t), &p);
  if (rc) return rc;
  pl->qref = (uint32_t*)p;
  return RHIP_OK;
}

// k * base (or k * -base), affine Montgomery, one field inversion per 256-thread block.  All threads must call this.
#ifndef RB_PAIRS_BLOCK
#define RB_PAIRS_BLOCK 256
#endif
__device__ __forceinline__ void scale_and_store(uint32_t* lds, bool active, G1Aff base, uint32_t k[8], bool negate, G1M* out, bool* is_inf) {
  if (fr_shorten(k)) negate = !negate;
  if (negate) base.y = neg(base.y);
  const G1Jac r = jac_mul_naf(base, k);
  const bool inf = !active || jac_is_inf(r);
  const Fp zinv = block_batch_inverse_n<RB_PAIRS_BLOCK>(lds, inf ? one<FpParams>() : r.z);
  *is_inf = inf;
  if (active && !inf) st_g1_q(out, jac_to_aff_with_zinv(r, zinv));
}

// lane -> pair of the gather kernels.  ppi != 0 (every item of the batch owns ppi pairs, pair_off[i] = i ppi): a WAVE holds pair j of
// 64 consecutive items.  Items that share a policy share the scalar of their j-th pair, so the NAF chains of a wave run without
// divergence -- lanes with different scalars execute the union of their additions, ~1.7x the work (k_bsw_dec_pairs 22.6 -> see
// DESIGN.md section 8).  ppi == 0 (ragged batch): lane = pair.
__device__ __forceinline__ void pair_lane(size_t v, size_t n_items, size_t total_pairs, const uint32_t* pair_off, uint32_t ppi, const uint32_t* tile_off, size_t* t,
                                          size_t* item, bool* active) {
  if (ppi) {
    const size_t per_tile = (size_t)64 * ppi;
    const size_t tile = v / per_tile, r = v % per_tile;
    size_t it = tile * 64 + (r & 63);
    *active = it < n_items;
    if (!*active) it = n_items - 1;
    *item = it;
    *t = it * ppi + (r >> 6);
  } else if (tile_off) {
    // ragged batch, tiled (k_tile_offsets): tile = 64 consecutive items, its lanes = 64 x (largest pair count in the tile), lane r of the
    // tile = pair r / 64 of item r % 64 -- a wave holds the j-th pair of 64 neighbouring items, like the uniform map, so that in a batch
    // grouped by shape its lanes share scalar and kind of work (a lane = pair map puts scaled and copied pairs, and the scalars of one
    // item's leaves, side by side in a wave: 2.4 x the time per scaling, measured on the mixed-shape config-3 batch)
    const size_t n_tiles = (n_items + 63) / 64;
    size_t lo = 0, hi = n_tiles;                    // the tile with tile_off[tile] <= v < tile_off[tile + 1]
    while (hi - lo > 1) {
      const size_t mid = (lo + hi) >> 1;
      if (tile_off[mid] <= v) lo = mid; else hi = mid;
    }
    const size_t r = v - tile_off[lo];
    size_t it = lo * 64 + (r & 63);
    const uint32_t j = (uint32_t)(r >> 6);
    bool act = v < tile_off[n_tiles] && it < n_items;
    if (it >= n_items) it = n_items - 1;
    const uint32_t p_it = pair_off[it + 1] - pair_off[it];
    act = act && j < p_it;
    *active = act;
    *item = it;
    *t = act ? (size_t)pair_off[it] + j : (p_it ? (size_t)pair_off[it] : total_pairs - 1);
    if (!act && !p_it) *item = owner_of(pair_off, n_items, *t);
  } else {
    *active = v < total_pairs;
    *t = *active ? v : total_pairs - 1;
    *item = owner_of(pair_off, n_items, *t);
  }
}
// tile_off[k] = first lane of tile k, k <= n_tiles (one block; a thread owns a run of tiles)
__global__ void __launch_bounds__(1024) k_tile_offsets(size_t n_items, const uint32_t* pair_off, uint32_t* tile_off) {
  __shared__ uint32_t part[1024];
  const size_t n_tiles = (n_items + 63) / 64;
  const size_t per = (n_tiles + 1023) / 1024, lo = (size_t)threadIdx.x * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
  auto lanes_of = [&](size_t tile) {
    uint32_t mx = 0;
    const size_t i1 = tile * 64 + 64 < n_items ? tile * 64 + 64 : n_items;
    for (size_t i = tile * 64; i < i1; i++) { const uint32_t p = pair_off[i + 1] - pair_off[i]; mx = p > mx ? p : mx; }
    return 64u * mx;
  };
  uint32_t sum = 0;
  for (size_t k = lo; k < hi; k++) sum += lanes_of(k);
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t at = part[threadIdx.x] - sum;
  for (size_t k = lo; k < hi; k++) { tile_off[k] = at; at += lanes_of(k); }
  if (threadIdx.x == 1023) tile_off[n_tiles] = part[1023];
}
static inline uint32_t uniform_ppi(size_t n_items, size_t max_pairs, size_t total_pairs) {
  return (max_pairs && total_pairs == n_items * max_pairs) ? (uint32_t)max_pairs : 0u;
}
static inline size_t pair_lanes(size_t n_items, size_t total_pairs, uint32_t ppi) { return ppi ? (n_items + 63) / 64 * 64 * (size_t)ppi : total_pairs; }
// lanes and tile table of a gather kernel's launch: uniform batches need none; a ragged one gets the tiled map (the grid is sized for the
// bound 64 x max_pairs lanes per tile -- the lanes beyond the table's end leave at once).  RABE_NO_PAIR_TILES=1: lane = pair (A/B runs).
static int32_t gather_lanes(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, const uint32_t* pair_off, uint32_t ppi, const uint32_t** tile_off,
                            size_t* lanes) {
  *tile_off = nullptr;
  *lanes = pair_lanes(n_items, total_pairs, ppi);
  static const bool off = getenv("RABE_NO_PAIR_TILES") != nullptr;
  if (ppi || off || !pair_off || !max_pairs || n_items < 64) return RHIP_OK;
  const size_t n_tiles = (n_items + 63) / 64;
  if (n_tiles * 64 * max_pairs >= ((size_t)1 << 32)) return RHIP_OK;          // the table holds 32-bit lane numbers
  void* w = nullptr;
  const int32_t rc = rhip_ensure_work(ctx, 10, (n_tiles + 1) * sizeof(uint32_t), &w);
  if (rc) return rc;
  KLAUNCH(ctx, "k_tile_offsets", k_tile_offsets, dim3(1), dim3(1024), 0, ctx->stream, n_items, pair_off, (uint32_t*)w);
  *tile_off = (const uint32_t*)w;
  *lanes = n_tiles * 64 * max_pairs;
  return RHIP_OK;
}

// ------------------------------------------------------------------------------------------------ the multi-pairing kernel
// The Fq12 accumulator of a lane lives in LDS (LdsHomeT, engine_internal.h) -- nothing of the loop goes to scratch.  Blocks of four
// waves: a block owns one CU (see rb_facc_lds4), so that a launch smaller than the chip leaves WHOLE CUs to whatever else is running
// (launch sets in flight on other streams: their 256-thread blocks need a wave slot on every SIMD of a CU).
struct DevMultiAcc : LdsHomeT<4> {
  const G1M* P;
  const G2M* Q;
  const uint32_t* qref;
  const LineM* lines;
  int cnt;
  uint4* ws;          // this lane's column of its wave's block; running point of pair j: quads [12 j, 12 j + 12) at stride `stride`
  size_t stride;      // 64: a wave's running points are one contiguous block (C x 12 KB), quad-major inside it
  __device__ __forceinline__ int count() const { return cnt; }
  __device__ __forceinline__ int kind(int j) const {
    const uint32_t v = qref[j];
    return v == RHIP_Q_WALK ? MP_WALK : v == RHIP_Q_SKIP ? MP_SKIP : MP_LINES;
  }
  __device__ __forceinline__ MillerP p(int j) const {
    const G1Aff a = ld_g1_q(P + j);
    return MillerP{a.x, a.y, a.y, false};
  }
  __device__ __forceinline__ G2Aff q(int j) const { return ld_g2_q(Q + j); }
  __device__ __forceinline__ LineCoeffs line(int j, int n) const {
    const uint4* p = (const uint4*)(lines + (size_t)qref[j] * RB_MILLER_LINES + n);
    return LineCoeffs{Fp2{ld_fp_q(p), ld_fp_q(p + 2)}, Fp2{ld_fp_q(p + 4), ld_fp_q(p + 6)}, Fp2{ld_fp_q(p + 8), ld_fp_q(p + 10)}};
  }
  __device__ __forceinline__ Fp ld1(const uint4* p) const {
    const uint4 a = p[0], b = p[stride];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
  }
  __device__ __forceinline__ void st1(uint4* p, const Fp& a) const {
    p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    p[stride] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
  }
  __device__ __forceinline__ G2Hom ld_t(int j) const {
    const uint4* p = ws + (size_t)(12 * j) * stride;
    G2Hom t;
    t.x = Fp2{ld1(p), ld1(p + 2 * stride)};
    t.y = Fp2{ld1(p + 4 * stride), ld1(p + 6 * stride)};
    t.z = Fp2{ld1(p + 8 * stride), ld1(p + 10 * stride)};
    return t;
  }
  __device__ __forceinline__ void st_t(int j, const G2Hom& t) const {
    uint4* p = ws + (size_t)(12 * j) * stride;
    st1(p, t.x.c0); st1(p + 2 * stride, t.x.c1);
    st1(p + 4 * stride, t.y.c0); st1(p + 6 * stride, t.y.c1);
    st1(p + 8 * stride, t.z.c0); st1(p + 10 * stride, t.z.c1);
  }
};
// lane t = chunk * n_items + item: a wave holds 64 items at the same chunk position, so that batches of equally shaped
// items run without divergence and read the same prepared lines.  An item of p pairs uses ceil(p / C) of its L chunks and splits its pairs
// EVENLY over them (a ragged batch: 29 pairs at C = 13 are 10 + 10 + 9, not 13 + 13 + 3); the chunks it does not use write the unit and
// leave at once (a Miller loop without pairs would still square its accumulator 65 times).  Output: mill[item * L + c].
//
// A RAGGED batch (plan != NULL) runs from a work list instead: the chunks that exist, largest first (k_plan_*, below), lane t = entry t of
// the list.  Every block of the launch then has work, the blocks of one size are spread over the XCDs by the round-robin of consecutive
// blocks, and the longest ones start first; C was chosen ON THE DEVICE from the batch's histogram of pair counts.  Output: the compact
// mill[chunk_off[item] + c].
__global__ void __launch_bounds__(RB_MILLER_BLOCK, RB_MIN_WAVES) k_miller_multi(size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const G1M* P,
                                                                  const G2M* Q, const uint32_t* qref, const LineM* lines, uint4* ws, size_t ws_stride,
                                                                  GtM* mill, const MillerPlan* plan, const uint2* work, const uint32_t* chunk_off) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (plan) {
    if (t >= plan->W) return;
    const uint32_t Cp = plan->C;
    const uint2 w = work[t];
    const uint64_t lo = pair_off[w.x], hi = pair_off[w.x + 1];
    const uint32_t p_item = (uint32_t)(hi - lo), nch = (p_item + Cp - 1) / Cp;
    const uint32_t base = p_item / nch, rem = p_item % nch, cc = w.y;
    const uint64_t first = lo + (uint64_t)cc * base + (cc < rem ? cc : rem);
    const int cnt = (int)(base + (cc < rem ? 1u : 0u));
    const DevMultiAcc acc{{}, P + first, Q + first, qref + first, lines, cnt, ws + (t >> 6) * ((size_t)Cp * 12 * 64) + (t & 63), ws_stride};
    const Fp12 f = miller_loop_multi(acc);
    st_gt_m(mill + chunk_off[w.x] + cc, f);
    return;
  }
  if (t >= n_items * L) return;
  // Chunk row c is rotated by c blocks: consecutive blocks go to consecutive XCDs (8 of them, 32 CUs each), and in a batch grouped by shape
  // block g of EVERY row holds the same policy -- unrotated, the rows of a many-leaf policy pile up on one XCD (measured: a ragged BSW
  // batch whose largest policy sat on XCD 0 in every row took two rounds there, 103 ms instead of ~50).
  const size_t c = t / n_items, item = (t % n_items + c * RB_MILLER_BLOCK) % n_items;
  // pair_off == NULL: every item owns exactly `uniform` pairs
  const uint64_t lo = pair_off ? pair_off[item] : (uint64_t)item * uniform, hi = pair_off ? pair_off[item + 1] : (uint64_t)(item + 1) * uniform;
  const uint32_t p_item = (uint32_t)(hi - lo);
  const uint32_t nch = (p_item + C - 1) / C;
  if (c >= nch) {                                // nothing for this chunk (whole waves, when the batch is grouped by shape)
    st_gt_m(mill + item * L + c, fp12_one());
    return;
  }
  const uint32_t base = p_item / nch, rem = p_item % nch, cc = (uint32_t)c;
  const uint64_t first = lo + (uint64_t)cc * base + (cc < rem ? cc : rem);
  const int cnt = (int)(base + (cc < rem ? 1u : 0u));
  // workspace: [wave][pair slot][quad][lane of the wave] -- one coalesced 1 KB access per quad, and everything a wave touches
  // during its whole run sits in one contiguous C x 12 KB block (page locality)
  const DevMultiAcc acc{{}, P + first, Q + first, qref + first, lines, cnt, ws + (t >> 6) * ((size_t)C * 12 * 64) + (t & 63), ws_stride};
  const Fp12 f = miller_loop_multi(acc);
  st_gt_m(mill + item * L + c, f);
}
// chunking of a batch: C pairs per lane (lines are merged two by two; an odd C leaves one unmerged), L lanes per item.  The kernel runs one
// wave per SIMD, so a launch of W waves takes ceil(W / #SIMDs) rounds of one lane's time; a lane's time is the shared
// squarings (65 x 36 Fp multiplications) plus ~5.5 k per pair (tests/count_muls.py).  Pick the C that minimises rounds x
// lane time: e.g. 4096 items x 201 pairs -> C = 14, L = 15: 960 waves, one round (C = 12 would need a second round for 64 waves).
// A ragged batch (total_pairs < n_items x max_pairs; items grouped by shape, so that the chunks an item does not use are whole waves
// that leave at once) is priced at the waves that do work: sum_i ceil(p_i / c) / 64 ~ (total_pairs / c + n_items / 2) / 64.
static void choose_chunks(const rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, uint32_t* L, uint32_t* C) {
  if (max_pairs < 1) max_pairs = 1;
  const bool ragged = total_pairs && total_pairs < n_items * max_pairs;
  const size_t simds = (size_t)ctx->n_cu * 4;
  double best = 0;
  size_t best_c = 2;
  for (size_t c = 1; c <= 64; c++) {
    const size_t l = (max_pairs + c - 1) / c;
    const size_t c_eff = ragged ? c : (max_pairs + l - 1) / l;
    const size_t waves = ragged ? (total_pairs / c + n_items / 2 + 63) / 64 : (n_items * l + 63) / 64;
    const size_t rounds = (waves + simds - 1) / simds;
    // an odd chunk leaves one line unmerged (13 instead of 11.5 Fq2 products for it): small launches still prefer it -- a lone batch
    // of 4096 six-pair items runs as 384 waves of one pair each instead of 192 of two
    static const double sq_env = getenv("RABE_CHUNK_SQ") ? atof(getenv("RABE_CHUNK_SQ")) : 0.0;          // tuning runs
    // the reduced-radix kernel (engine_rr.hip) does not merge lines two by two (no penalty for an odd chunk) and its shared squaring weighs
    // more against a pair (11 k against 15 k instructions per line event; 2.3 k against 5.5 k Fp multiplications in the 8 x 32-bit kernel)
    const bool rr = rhip_use_rr(ctx);
    const double sq = sq_env > 0 ? sq_env : (rr ? 4000.0 : 2340.0);
    const double cost = (double)rounds * (sq + 5500.0 * (double)c_eff + ((!rr && (c_eff & 1)) ? 600.0 : 0.0));
    if (best == 0 || cost < best) { best = cost; best_c = c; }
    if (l == 1) break;
  }
  const size_t l = (max_pairs + best_c - 1) / best_c;
  const size_t c = ragged ? best_c : (max_pairs + l - 1) / l;
  *L = (uint32_t)l;
  *C = (uint32_t)c;
}
// ---- membership of the WALKING G2 arguments as a by-product of the Miller loops (rhip_ctx_collect_walk_verdicts).
// When k_miller_multi is done, the workspace slot of a walking pair holds R = [6u+2]Q + psi(Q) - psi^2(Q): the loop's last two
// additions are the Frobenius steps.  On G2 the twist's Frobenius psi acts as multiplication by p, and 6u+2 + p - p^2 + p^3 = 0 mod r
// (the relation the optimal ate pairing rests on), so a member satisfies  R = -psi^3(Q).  Conversely, for Q on the twist
// E'(Fp2) (order r * h2) the endomorphism f(psi) = (6u+2) + psi - psi^2 + psi^3 and the characteristic equation
// chi(psi) = psi^2 - t psi + p = 0 give Res(f, chi) * Q = O, and Res(f, chi) = r * m with gcd(m, r * h2) = 1
// (tests/test_walk_relation.py checks it with exact integers, and the same computation
// reproduces the soundness of the El Housni-Guillevic test k_g2_in_subgroup uses): the order of Q divides r, Q is in G2.
// Degenerate steps cannot forge the relation: the doubling and addition formulas (bn254/pairing.h) give Z3 a factor Z, the chord of
// T = +-Q gives Z3 = 0, so a walk that leaves the formulas' domain ends with Z = 0 and is rejected.  The curve equation and the
// coordinate range of Q are NOT established here (rhip_g2_on_curve does both); a pair that was skipped (an argument at infinity)
// is not counted -- the caller compares the count with the number of elements it expected to see examined.
// Same lane -> (item, chunk) map as k_miller_multi.
__global__ void __launch_bounds__(256) k_walk_verdicts(size_t n_items, uint32_t L, uint32_t C, const uint32_t* pair_off, uint32_t uniform, const G2M* Q,
                                                       const uint32_t* qref, const uint4* ws, const MillerPlan* plan, const uint2* work, uint32_t* fail,
                                                       uint32_t* count) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t item;
  uint64_t first;
  uint32_t cnt, Cw;
  if (plan) {
    if (t >= plan->W) return;
    Cw = plan->C;
    const uint2 w = work[t];
    item = w.x;
    const uint64_t lo = pair_off[item], hi = pair_off[item + 1];
    const uint32_t p_item = (uint32_t)(hi - lo), nch = (p_item + Cw - 1) / Cw, base = p_item / nch, rem = p_item % nch;
    first = lo + (uint64_t)w.y * base + (w.y < rem ? w.y : rem);
    cnt = base + (w.y < rem ? 1u : 0u);
  } else {
    if (t >= n_items * L) return;
    Cw = C;
    const size_t c = t / n_items;
    item = (t % n_items + c * RB_MILLER_BLOCK) % n_items;
    const uint64_t lo = pair_off ? pair_off[item] : (uint64_t)item * uniform, hi = pair_off ? pair_off[item + 1] : (uint64_t)(item + 1) * uniform;
    const uint32_t p_item = (uint32_t)(hi - lo), nch = (p_item + C - 1) / C;
    if (c >= nch) return;
    const uint32_t base = p_item / nch, rem = p_item % nch, cc = (uint32_t)c;
    first = lo + (uint64_t)cc * base + (cc < rem ? cc : rem);
    cnt = base + (cc < rem ? 1u : 0u);
  }
  const uint4* wl = ws + (t >> 6) * ((size_t)Cw * 12 * 64) + (t & 63);
  uint32_t seen = 0;
  bool bad = false;
  for (uint32_t j = 0; j < cnt; j++) {
    if (qref[first + j] != RHIP_Q_WALK) continue;
    const uint4* p = wl + (size_t)(12 * j) * 64;
    Fp e[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const uint4 a = p[(size_t)(2 * k) * 64], b = p[(size_t)(2 * k + 1) * 64];
      e[k].v[0] = a.x; e[k].v[1] = a.y; e[k].v[2] = a.z; e[k].v[3] = a.w;
      e[k].v[4] = b.x; e[k].v[5] = b.y; e[k].v[6] = b.z; e[k].v[7] = b.w;
    }
    const Fp2 rx{e[0], e[1]}, ry{e[2], e[3]}, rz{e[4], e[5]};
    const G2Aff s = aff_neg(g2_frob1(g2_frob2(ld_g2_q(Q + first + j))));
    const bool ok = !fp2_is_zero(rz) && fp2_eq(rx, fp2_mul(s.x, rz)) && fp2_eq(ry, fp2_mul(s.y, rz));
    seen++;
    bad |= !ok;
  }
  if (seen) atomicAdd(count + item, seen);
  if (bad) fail[item] = 1u;
}
// ---- the plan of a ragged batch, made on the device (the pair counts live there; nothing comes back to the host)
// hist[p] = number of items with p pairs
__global__ void k_plan_hist(size_t n_items, const uint32_t* pair_off, uint32_t max_pairs, uint32_t* hist) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  uint32_t p = pair_off[i + 1] - pair_off[i];
  if (p > max_pairs) p = max_pairs;                 // cannot happen for a caller that states max_pairs truthfully; keeps the table in bounds
  // a batch grouped by shape: whole waves agree -- one atomic per distinct value of the wave
  uint64_t todo = __ballot(1);
  while (todo) {
    const int lead = __ffsll((long long)todo) - 1;
    const uint32_t v = (uint32_t)__shfl((int)p, lead);
    const uint64_t same = __ballot(p == v) & todo;
    if ((int)(threadIdx.x & 63) == lead) atomicAdd(hist + v, (uint32_t)__popcll(same));
    todo &= ~same;
  }
}
// One block of 1024: candidate C = 1 + (tid & 63), sixteen threads per candidate share the histogram.  For each candidate the list's
// size distribution cnt[s] (an item of p pairs has ceil(p / C) chunks, p mod nch of them one pair longer), then the launch priced as
// rounds of n_cu blocks over the list sorted by size -- a round costs what its FIRST (largest) block costs: the shared squarings (2340
// Fp multiplications) + 5500 per pair (+ 600 for an unmerged odd line) -- plus the product of an item's chunk values in k_final_exp.
__global__ void __launch_bounds__(1024) k_plan_choose(const uint32_t* hist, uint32_t max_pairs, uint32_t n_items, uint32_t c_lo, uint32_t c_hi, uint32_t n_cu,
                                                      MillerPlan* plan) {
  __shared__ uint32_t cnt[64][66];
  __shared__ float cost[64];
  const int tid = threadIdx.x, j = tid & 63, part = tid >> 6;
  const uint32_t C = (uint32_t)j + 1;
  for (int k = tid; k < 64 * 66; k += 1024) (&cnt[0][0])[k] = 0;
  __syncthreads();
  const bool live = C >= c_lo && C <= c_hi;
  if (live)
    for (uint32_t p = 1 + (uint32_t)part; p <= max_pairs; p += 16) {
      const uint32_t h = hist[p];
      if (!h) continue;
      const uint32_t nch = (p + C - 1) / C, base = p / nch, rem = p % nch;
      if (rem) atomicAdd(&cnt[j][base + 1], h * rem);
      atomicAdd(&cnt[j][base], h * (nch - rem));
    }
  __syncthreads();
  if (tid < 64) {
    float c = 3.0e38f;
    if (live) {
      const uint64_t R = (uint64_t)n_cu * RB_MILLER_BLOCK;
      uint64_t pos = 0, next = 0;
      c = 0;
      for (int sz = 64; sz >= 1; sz--) {
        const uint32_t n = cnt[j][sz];
        if (!n) continue;
        while (next < pos + n) { c += 2340.0f + 5500.0f * (float)sz + ((sz & 1) ? 600.0f : 0.0f); next += R; }
        pos += n;
      }
      const uint32_t fe_rounds = (n_items + n_cu * 256u - 1) / (n_cu * 256u)   /* k_final_exp: blocks of 256 */;
      c += 60.0f * (float)((max_pairs + C - 1) / C) * (float)fe_rounds;
      cnt[j][65] = (uint32_t)pos;
    }
    cost[j] = c;
  }
  __syncthreads();
  if (tid == 0) {
    int best = (int)c_lo - 1;
    for (int k = 0; k < 64; k++)
      if (cost[k] < cost[best] || (cost[k] == cost[best] && k > best)) best = k;          // ties: the larger chunk (fewer lanes)
    plan->C = (uint32_t)best + 1;
    plan->W = cnt[best][65];
    plan->L = (max_pairs + (uint32_t)best) / ((uint32_t)best + 1);
    uint32_t at = 0;
    for (int sz = 65; sz >= 0; sz--) {
      plan->base[sz] = at;
      plan->cursor[sz] = at;
      if (sz >= 1 && sz <= 64) at += cnt[best][sz];
    }
  }
}
// chunk_off[i] = number of chunks of the items before i (one block; a thread owns a run of items)
__global__ void __launch_bounds__(1024) k_plan_scan(size_t n_items, const uint32_t* pair_off, const MillerPlan* plan, uint32_t* chunk_off) {
  __shared__ uint32_t part[1024];
  const uint32_t C = plan->C;
  const size_t per = (n_items + 1023) / 1024, lo = (size_t)threadIdx.x * per, hi = lo + per < n_items ? lo + per : n_items;
  uint32_t sum = 0;
  for (size_t i = lo; i < hi; i++) sum += (pair_off[i + 1] - pair_off[i] + C - 1) / C;
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t at = part[threadIdx.x] - sum;
  for (size_t i = lo; i < hi; i++) {
    chunk_off[i] = at;
    at += (pair_off[i + 1] - pair_off[i] + C - 1) / C;
  }
  if (threadIdx.x == 1023) chunk_off[n_items] = part[1023];
}
// The list: candidate t = chunk row c x item (64 neighbouring items at the same chunk position stay neighbours: they replay the same
// prepared lines), appended to the run of its size; one atomic per distinct size of a wave.  Grid-stride: the number of rows is the plan's.
__global__ void __launch_bounds__(256) k_plan_fill(size_t n_items, const uint32_t* pair_off, MillerPlan* plan, uint2* work) {
  const uint32_t C = plan->C, L = plan->L;
  const size_t n_pad = (n_items + 63) / 64 * 64, total = n_pad * L, step = (size_t)gridDim.x * blockDim.x;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += step) {
    const size_t c = t / n_pad, item = t % n_pad;
    uint32_t sz = 0;
    if (item < n_items) {
      const uint32_t p = pair_off[item + 1] - pair_off[item], nch = (p + C - 1) / C;
      if (c < nch) sz = p / nch + (c < p % nch ? 1u : 0u);
    }
    uint64_t todo = __ballot(sz != 0);
    const int lane = (int)(threadIdx.x & 63);
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const uint32_t v = (uint32_t)__shfl((int)sz, lead);
      const uint64_t same = __ballot(sz == v) & todo;
      uint32_t at = 0;
      if (lane == lead) at = atomicAdd(&plan->cursor[v], (uint32_t)__popcll(same));
      at = (uint32_t)__shfl((int)at, lead);
      if (sz == v && sz) work[at + (uint32_t)__popcll(same & ((1ull << lane) - 1))] = make_uint2((uint32_t)item, (uint32_t)c);
      todo &= ~same;
    }
  }
}
// Miller values of all items' pairs + final exponentiation: out[i] = mul_in[i] * FE(prod_j ML(P_j, Q_j))
static int32_t run_pair_lists_mode(rhip_ctx* ctx, size_t n_items, const uint32_t* pair_off, size_t max_pairs, size_t total_pairs, const PairLists& pl,
                                   const LineM* lines, const void* lines29, const rhip_gt* mul_in, rhip_gt* out);
// ---- pairing mode 99, the cross-check: the launch runs as "auto" would run it (that result is the one the caller gets, walk verdicts and early
// releases included), then AGAIN with each family of pairing kernels forced -- one lane per accumulator on 8 x 32-bit limbs (1), six lanes (6),
// reduced radix (29), reduced radix with a unit on two lanes (58): Miller loops AND final exponentiation of that family -- on the same pair lists, and every result is compared with the
// first on the device, byte for byte.  A difference fails the call.  The three families compute the same field elements by construction
// (docs/coop6.md, docs/rr29.md); this mode is how the GPU test suite holds them to it on every pairing launch of every scheme
// (tests/conftest.py) and what an operator can switch on to have a suspect device check itself.
// RABE_XCHECK_FAULT=<family> flips one bit of that family's result before the comparison (tests/test_gpu_xcheck.py: the check must notice).
__global__ void __launch_bounds__(256) k_xcheck_compare(size_t n_quads, const uint4* a, uint4* b, uint32_t bit, uint32_t flip, uint32_t* flag) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_quads) return;
  uint4 y = b[t];
  if (flip && t == n_quads / 2) y.y ^= 0x10000u;
  const uint4 x = a[t];
  if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) atomicOr(flag, bit);
}
static int32_t run_pair_lists(rhip_ctx* ctx, size_t n_items, const uint32_t* pair_off, size_t max_pairs, size_t total_pairs, const PairLists& pl,
                              const LineM* lines, const void* lines29, const rhip_gt* mul_in, rhip_gt* out) {
  if (ctx->pairing_mode != 99) return run_pair_lists_mode(ctx, n_items, pair_off, max_pairs, total_pairs, pl, lines, lines29, mul_in, out);
  struct Restore { rhip_ctx* c; ~Restore() { c->pairing_mode = 99; } } restore{ctx};
  const size_t bytes = n_items * sizeof(rhip_gt);
  void* tmp = nullptr;
  int32_t rc = rhip_ensure_work(ctx, 12, bytes + 256, &tmp);
  if (rc) return rc;
  uint32_t* flag = (uint32_t*)((uint8_t*)tmp + bytes);
  HIP_TRY(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
  const rhip_gt* factor = mul_in;
  if (mul_in && (const void*)mul_in == (const void*)out) {          // in place: the other families need the factor as it was
    void* keep = nullptr;
    rc = rhip_ensure_work(ctx, 13, bytes, &keep);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(keep, mul_in, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    factor = (const rhip_gt*)keep;
  }
  ctx->pairing_mode = 0;
  rc = run_pair_lists_mode(ctx, n_items, pair_off, max_pairs, total_pairs, pl, lines, lines29, factor, out);
  if (rc) return rc;
  const int fault = getenv("RABE_XCHECK_FAULT") ? atoi(getenv("RABE_XCHECK_FAULT")) : 0;
  static const int families[4] = {1, 6, 29, 58};
  for (int f = 0; f < 4; f++) {
    ctx->pairing_mode = families[f];
    rc = run_pair_lists_mode(ctx, n_items, pair_off, max_pairs, total_pairs, pl, lines, lines29, factor, (rhip_gt*)tmp);
    if (rc) return rc;
    KLAUNCH(ctx, "k_xcheck_compare", k_xcheck_compare, dim3(blocks_for(bytes / 16, 256)), dim3(256), 0, ctx->stream, bytes / 16, (const uint4*)out, (uint4*)tmp,
            1u << f, (uint32_t)(fault == families[f]), flag);
  }
  uint32_t h = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (h) {
    ctx->err = std::string("pairing cross-check: the kernel families disagree on this launch (differing from the automatic choice:") + ((h & 1) ? " one-lane 8x32" : "") +
               ((h & 2) ? " six-lane" : "") + ((h & 4) ? " reduced-radix" : "") + ((h & 8) ? " two-lane" : "") + ")";
    return RHIP_ERR_HIP;
  }
  return RHIP_OK;
}
static int32_t run_pair_lists_mode(rhip_ctx* ctx, size_t n_items, const uint32_t* pair_off, size_t max_pairs, size_t total_pairs, const PairLists& pl,
                              const LineM* lines, const void* lines29, const rhip_gt* mul_in, rhip_gt* out) {          // pair_off == NULL: every item owns max_pairs pairs
  if (pair_off && total_pairs && total_pairs < n_items * max_pairs && !getenv("RABE_NO_MILLER_PLAN")) {
    // ragged: plan on the device.  Bounds of what the plan may choose: no fewer pairs per chunk than four rounds' worth of lanes need
    // (more lanes only add shared squarings), no more than 64; the buffers are sized for the worst of the range.
    const size_t R = (size_t)ctx->n_cu * RB_MILLER_BLOCK;
    size_t c_lo = total_pairs / (4 * R);
    if (c_lo < 1) c_lo = 1;
    size_t c_hi = max_pairs < 64 ? max_pairs : 64;
    if (c_lo > c_hi) c_lo = c_hi;
    const size_t w_max = total_pairs / c_lo + n_items + 64;
    const size_t hist_words = (max_pairs + 2 + 3) / 4 * 4;
    const size_t plan_words = (sizeof(MillerPlan) / 4 + 3) / 4 * 4;
    const size_t off_words = (n_items + 1 + 3) / 4 * 4;
    void* pb = nullptr;
    int32_t rc = rhip_ensure_work(ctx, 9, (plan_words + hist_words + off_words) * 4 + w_max * sizeof(uint2), &pb);
    if (rc) return rc;
    MillerPlan* plan = (MillerPlan*)pb;
    uint32_t* hist = (uint32_t*)pb + plan_words;
    uint32_t* chunk_off = hist + hist_words;
    uint2* work = (uint2*)(chunk_off + off_words);
    void* ws = nullptr;
    const size_t ws_bytes = (total_pairs + n_items * (c_hi - 1) + 64 * c_hi + 64) * 12 * sizeof(uint4);
    rc = rhip_ensure_work(ctx, 3, ws_bytes, &ws);
    if (rc) return rc;
    rc = ensure_scratch(ctx, w_max * sizeof(GtM));
    if (rc) return rc;
    GtM* mill = (GtM*)ctx->scratch;
    HIP_TRY(ctx, hipMemsetAsync(hist, 0, hist_words * 4, ctx->stream));
    KLAUNCH(ctx, "k_plan_hist", k_plan_hist, dim3(blocks_for(n_items, 256)), dim3(256), 0, ctx->stream, n_items, pair_off, (uint32_t)max_pairs, hist);
    KLAUNCH(ctx, "k_plan_choose", k_plan_choose, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)hist, (uint32_t)max_pairs, (uint32_t)n_items, (uint32_t)c_lo,
            (uint32_t)c_hi, (uint32_t)ctx->n_cu, plan);
    KLAUNCH(ctx, "k_plan_scan", k_plan_scan, dim3(1), dim3(1024), 0, ctx->stream, n_items, pair_off, (const MillerPlan*)plan, chunk_off);
    KLAUNCH(ctx, "k_plan_fill", k_plan_fill, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, n_items, pair_off, plan, work);
    if (rhip_use_c6(ctx, n_items, max_pairs)) {
      rc = rhip_launch_miller_c6(ctx, n_items, 0u, 0u, pair_off, (uint32_t)max_pairs, pl.P, pl.Q, pl.qref, lines, ws, mill, plan, work, chunk_off, w_max);
      if (rc) return rc;
    } else if (rhip_use_rr2(ctx, (uint32_t)c_hi)) {          // one unit on two lanes, two waves per SIMD (engine_rr2.hip)
      rc = rhip_launch_miller_rr2(ctx, n_items, 0u, 0u, (uint32_t)c_hi, pair_off, (uint32_t)max_pairs, pl.P, pl.Q, pl.qref, lines, lines29, ws, mill, plan, work, chunk_off, w_max,
                                  nullptr, total_pairs, (uint32_t)c_lo);
      if (rc) return rc;
    } else if (rhip_use_rr(ctx)) {
      rc = rhip_launch_miller_rr(ctx, n_items, 0u, 0u, pair_off, (uint32_t)max_pairs, pl.P, pl.Q, pl.qref, lines, lines29, ws, ws_bytes, mill, plan, work, chunk_off, w_max);
      if (rc) return rc;
    } else
    KLAUNCH(ctx, "k_miller_multi", k_miller_multi, dim3(blocks_for(w_max, RB_MILLER_BLOCK)), dim3(RB_MILLER_BLOCK), 0, ctx->stream, n_items, 0u, 0u, pair_off, (uint32_t)max_pairs,
            (const G1M*)pl.P, (const G2M*)pl.Q, (const uint32_t*)pl.qref, lines, (uint4*)ws, (size_t)64, mill, (const MillerPlan*)plan, (const uint2*)work,
            (const uint32_t*)chunk_off);
    if (ctx->walk_fail) {
      KLAUNCH(ctx, "k_walk_verdicts", k_walk_verdicts, dim3(blocks_for(w_max, 256)), dim3(256), 0, ctx->stream, n_items, 0u, 0u, pair_off, (uint32_t)max_pairs, (const G2M*)pl.Q,
              (const uint32_t*)pl.qref, (const uint4*)ws, (const MillerPlan*)plan, (const uint2*)work, ctx->walk_fail, ctx->walk_count);
      ctx->walk_fail = ctx->walk_count = nullptr;
    }
    return launch_final_exp(ctx, n_items, (const uint32_t*)chunk_off, 1u, (const GtM*)mill, mul_in, out);
  }
  uint32_t L, C;
  const bool c6 = rhip_use_c6(ctx, n_items, max_pairs);
  if (c6) rhip_choose_chunks_c6(ctx, n_items, max_pairs, &L, &C);
  else choose_chunks(ctx, n_items, max_pairs, pair_off ? total_pairs : 0, &L, &C);
  const size_t lanes = n_items * L;
  const size_t lanes_pad = (lanes + 63) / 64 * 64;
  void* ws = nullptr;
  const size_t ws_bytes = lanes_pad * C * 12 * sizeof(uint4);
  int32_t rc = rhip_ensure_work(ctx, 3, ws_bytes, &ws);
  if (rc) return rc;
  rc = ensure_scratch(ctx, lanes * sizeof(GtM));
  if (rc) return rc;
  GtM* mill = (GtM*)ctx->scratch;
  if (c6) {       // six lanes per (item, chunk): engine_coop.hip; same (item, chunk) map and workspace layout, so the walk verdicts below apply unchanged
    rc = rhip_launch_miller_c6(ctx, n_items, L, C, pair_off, (uint32_t)max_pairs, pl.P, pl.Q, pl.qref, lines, ws, mill, nullptr, nullptr, nullptr, lanes);
    if (rc) return rc;
  } else if (rhip_use_rr2(ctx, C)) {          // one unit on two lanes, two waves per SIMD (engine_rr2.hip)
    uint32_t* started = nullptr;
    if (ctx->early_release) {          // rhip_ctx_release_when_miller_resident: the waiter goes on beside the Miller loops already
      rc = rhip_take_waiter(ctx, blocks_for(2 * lanes, RB_MILLER_BLOCK), &started);
      if (rc) return rc;
    }
    rc = rhip_launch_miller_rr2(ctx, n_items, L, C, C, pair_off, (uint32_t)max_pairs, pl.P, pl.Q, pl.qref, lines, lines29, ws, mill, nullptr, nullptr, nullptr, lanes, started, 0, 0u);
    if (rc) return rc;
  } else if (rhip_use_rr(ctx)) {
    uint32_t* started = nullptr;
    if (ctx->early_release) {          // rhip_ctx_release_when_miller_resident: the waiter goes on beside the Miller loops already
      rc = rhip_take_waiter(ctx, blocks_for(lanes, RB_MILLER_BLOCK), &started);
      if (rc) return rc;
    }
    rc = rhip_launch_miller_rr(ctx, n_items, L, C, pair_off, (uint32_t)max_pairs, pl.P, pl.Q, pl.qref, lines, lines29, ws, ws_bytes, mill, nullptr, nullptr, nullptr, lanes, started);
    if (rc) return rc;
  } else
  KLAUNCH(ctx, "k_miller_multi", k_miller_multi, dim3(blocks_for(lanes, RB_MILLER_BLOCK)), dim3(RB_MILLER_BLOCK), 0, ctx->stream, n_items, L, C, pair_off, (uint32_t)max_pairs, (const G1M*)pl.P,
          (const G2M*)pl.Q, (const uint32_t*)pl.qref, lines, (uint4*)ws, (size_t)64, mill, (const MillerPlan*)nullptr, (const uint2*)nullptr, (const uint32_t*)nullptr);
  if (ctx->walk_fail) {
    KLAUNCH(ctx, "k_walk_verdicts", k_walk_verdicts, dim3(blocks_for(lanes, 256)), dim3(256), 0, ctx->stream, n_items, L, C, pair_off, (uint32_t)max_pairs, (const G2M*)pl.Q,
            (const uint32_t*)pl.qref, (const uint4*)ws, (const MillerPlan*)nullptr, (const uint2*)nullptr, ctx->walk_fail, ctx->walk_count);
    ctx->walk_fail = ctx->walk_count = nullptr;
  }
  return launch_final_exp(ctx, n_items, (const uint32_t*)nullptr, L, (const GtM*)mill, mul_in, out);
}

// ------------------------------------------------------------------------------------------------ share generation
// gen_shares_policy (src/utils/secretsharing/mod.rs:82-141) over a flattened policy: leaf `y` of a policy owns the
// path entries [path_off[y], path_off[y+1]) from the root down; entry e says "child number x[e] (1-based) of gate
// gate[e]".  Gate g has threshold k[g] (AND over n children: n, OR: 1) and consumed k[g] - 1 coefficient draws starting
// at coef_off[g] of the item's draw list (DFS pre-order, the reference's draw order :128-134).  The share handed to
// child x of a gate with secret s is  s + a_1 x + ... + a_(k-1) x^(k-1)  (:135-138, `polynomial` :215-221).
struct TreeTables {
  const uint32_t* path_off;     // [leaves + 1]
  const uint32_t* path_gate;    // gate index relative to the policy's first gate
  const uint32_t* path_x;
  const uint32_t* gate_k;       // [gates]
  const uint32_t* gate_coef_off;
};
__device__ __noinline__ Fr share_of_leaf(const TreeTables& tt, uint32_t leaf /* global leaf row of the policy tables */, uint32_t gate0,
                                         const rhip_fr* coef /* the item's draws */, Fr secret) {
  Fr s = secret;
  for (uint32_t e = tt.path_off[leaf]; e < tt.path_off[leaf + 1]; e++) {
    const uint32_t g = gate0 + tt.path_gate[e];
    const uint32_t k = tt.gate_k[g];
    if (k <= 1) continue;                                   // OR: every child gets the secret
    uint32_t xs[8] = {tt.path_x[e], 0, 0, 0, 0, 0, 0, 0};
    const Fr x = to_mont<FrParams>(xs);
    const rhip_fr* a = coef + tt.gate_coef_off[g];          // a_1 .. a_(k-1)
    Fr acc = load_fr(a[k - 2].l);
    for (int j = (int)k - 3; j >= 0; j--) acc = add(mul(acc, x), load_fr(a[j].l));
    s = add(mul(acc, x), s);
  }
  return s;
}

// ------------------------------------------------------------------------------------------------ BSW CP-ABE
struct rhip_bsw_pk {
  rhip_ctx* ctx;
  rhip_g1_table* g1;
  rhip_g2_table* g2;
  rhip_g1_table* h;
  rhip_gt_table* e;
};
extern "C" void rhip_bsw_pk_destroy(rhip_bsw_pk* pk) {
  if (!pk) return;
  rhip_g1_table_destroy(pk->g1);
  rhip_g2_table_destroy(pk->g2);
  rhip_g1_table_destroy(pk->h);
  rhip_gt_table_destroy(pk->e);
  delete pk;
}
extern "C" int32_t rhip_bsw_pk_create(rhip_ctx* ctx, const rhip_g1* g1, const rhip_g2* g2, const rhip_g1* h, const rhip_gt* e_gg_alpha,
                                      rhip_bsw_pk** out) {
  if (!ctx || !g1 || !g2 || !h || !e_gg_alpha || !out) return RHIP_ERR_ARG;
  *out = nullptr;
  rhip_bsw_pk* pk = new rhip_bsw_pk{ctx, nullptr, nullptr, nullptr, nullptr};
  int32_t rc = rhip_g1_table_create(ctx, g1, &pk->g1);
  if (!rc) rc = rhip_g1_table_add_w16(ctx, pk->g1);
  if (!rc) rc = rhip_g2_table_create(ctx, g2, &pk->g2);
  if (!rc) rc = rhip_g2_table_add_w16(ctx, pk->g2);
  if (!rc) rc = rhip_g1_table_create(ctx, h, &pk->h);
  if (!rc) rc = rhip_g1_table_add_w16(ctx, pk->h);
  if (!rc) rc = rhip_gt_table_create(ctx, e_gg_alpha, &pk->e);
  if (!rc) rc = rhip_gt_table_add_w16(ctx, pk->e);
  if (rc) { rhip_bsw_pk_destroy(pk); return rc; }
  *out = pk;
  return RHIP_OK;
}
// one lane per (item, leaf): q_y and h(name_y) * q_y  (bsw/mod.rs:236-243: g1 * q_y, (g2 * h(name)) * q_y)
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_bsw_enc_scalars(size_t n_items, size_t total_leaves, const uint32_t* item_leaf_off,
                                                                      const uint32_t* item_tree_leaf, const uint32_t* item_tree_gate, TreeTables tt,
                                                                      const rhip_fr* leaf_hash, const rhip_fr* secret, const rhip_fr* coef,
                                                                      const uint32_t* item_coef_off, rhip_fr* kq, rhip_fr* khq) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_leaves) return;
  const size_t item = owner_of(item_leaf_off, n_items, t);
  const uint32_t leaf = item_tree_leaf[item] + (uint32_t)(t - item_leaf_off[item]);
  const Fr q = share_of_leaf(tt, leaf, item_tree_gate[item], coef + item_coef_off[item], load_fr(secret[item].l));
  store_fr(kq[t].l, q);
  store_fr(khq[t].l, mul(q, load_fr(leaf_hash[leaf].l)));
}
// out[i] = base^k[i] * m[i] from the 16-bit (or 8-bit) window table
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_table_pow_gt_mul(const GtM* tbl, int w16, size_t n, const rhip_fr* k, const rhip_gt* m, rhip_gt* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t kk[8];
  ld_scalar(kk, k + i);
  home_put(load_gt(m[i].l));
  bool started = true;
  if (w16) home_table_pow_gt_w16(started, tbl, kk); else home_table_pow_gt(started, tbl, kk);
  store_gt(out[i].l, home_result(true));
}
extern "C" int32_t rhip_bsw_encrypt_batch(rhip_ctx* ctx, const rhip_bsw_pk* pk, size_t n_items, size_t total_leaves, const uint32_t* item_leaf_off,
                                          const uint32_t* item_tree_leaf, const uint32_t* item_tree_gate, const uint32_t* path_off,
                                          const uint32_t* path_gate, const uint32_t* path_x, const uint32_t* gate_k, const uint32_t* gate_coef_off,
                                          const rhip_fr* leaf_hash, const rhip_fr* secret, const rhip_fr* coef, const uint32_t* item_coef_off,
                                          const rhip_gt* msg, rhip_g1* c, rhip_gt* cp, rhip_g1* cy_g1, rhip_g2* cy_g2) {
  NEED(ctx);
  if (!pk) return RHIP_ERR_ARG;
  if (!n_items) return RHIP_OK;
  void* w = nullptr;
  int32_t rc = rhip_ensure_work(ctx, 4, (total_leaves ? total_leaves : 1) * 2 * sizeof(rhip_fr), &w);
  if (rc) return rc;
  rhip_fr* kq = (rhip_fr*)w;
  rhip_fr* khq = kq + total_leaves;
  if (total_leaves) {
    const TreeTables tt{path_off, path_gate, path_x, gate_k, gate_coef_off};
    KLAUNCH(ctx, "k_bsw_enc_scalars", k_bsw_enc_scalars, dim3(blocks_for(total_leaves, 256)), dim3(256), 0, ctx->stream, n_items, total_leaves,
            item_leaf_off, item_tree_leaf, item_tree_gate, tt, leaf_hash, secret, coef, item_coef_off, kq, khq);
    rc = rhip_g1_table_mul(ctx, pk->g1, total_leaves, kq, cy_g1);
    if (rc) return rc;
    rc = rhip_g2_table_mul(ctx, pk->g2, total_leaves, khq, cy_g2);
    if (rc) return rc;
  }
  rc = rhip_g1_table_mul(ctx, pk->h, n_items, secret, c);
  if (rc) return rc;
  const rhip_gt_table* et = pk->e;
  KLAUNCH(ctx, "k_table_pow_gt_mul", k_table_pow_gt_mul, dim3(blocks_for(n_items, 64)), dim3(64), 0, ctx->stream,
          (const GtM*)(et->dev16 ? et->dev16 : et->dev), et->dev16 ? 1 : 0, n_items, secret, msg, cp);
  return RHIP_OK;
}

// decrypt: the pairs of item i are [pair_off[i], pair_off[i+1]) = 2 m_i + 1 of them; pair 2s / 2s+1 belong to selected
// leaf s (entry e = sel_start[i] + s of the selection tables), the last one is e(-c, d):
//   2s   : P =  z_e * Cy.g1,  Q = Dj.g2     (key side: prepared lines when the key was prepared)
//   2s+1 : P = -z_e * Dj.g1,  Q = Cy.g2
//   2m   : P = -c,            Q = d
// (msg = c_p * FE(prod), bsw/mod.rs:282-308 restated in SURVEY.md Appendix B.4)
__global__ void __launch_bounds__(RB_PAIRS_BLOCK, 2) k_bsw_dec_pairs(size_t n_items, size_t total_pairs, const uint32_t* pair_off, uint32_t ppi, const uint32_t* tile_off, const uint32_t* sel_start,
                                                                    const uint32_t* sel_ct_leaf, const uint32_t* sel_sk_attr, const rhip_fr* sel_coeff,
                                                                    const rhip_g1* ct_c, const rhip_g1* ct_cy_g1, const rhip_g2* ct_cy_g2,
                                                                    const uint32_t* ct_leaf_off, const rhip_g2* sk_d, const rhip_g1* sk_dj_g1,
                                                                    const rhip_g2* sk_dj_g2, const uint32_t* sk_attr_off, const uint32_t* sk_idx,
                                                                    uint32_t lines_d_base, int prepared, const uint8_t* line_inf, G1M* P, G2M* Q,
                                                                    uint32_t* qref, const G1M* psel, const uint8_t* psel_inf, int compact) {
  __shared__ uint32_t lds[2 * 8 * RB_PAIRS_BLOCK];
  size_t t, item;
  bool active;
  uint32_t j, m;
  if (compact) {      // uniform batch, one key: only the pairs that still need a scaling get a lane -- positions 0, 2, .. 2m-2 and 2m
    const uint32_t mm = (ppi - 1) >> 1;
    pair_lane((size_t)blockIdx.x * blockDim.x + threadIdx.x, n_items, n_items * (size_t)(mm + 1), pair_off, mm + 1, (const uint32_t*)nullptr, &t, &item, &active);
    const uint32_t jj = (uint32_t)(t - item * (size_t)(mm + 1));
    m = mm;
    j = jj < mm ? 2 * jj : 2 * mm;
    t = item * (size_t)ppi + j;
  } else {
    pair_lane((size_t)blockIdx.x * blockDim.x + threadIdx.x, n_items, total_pairs, pair_off, ppi, tile_off, &t, &item, &active);
    j = (uint32_t)(t - pair_off[item]);
    m = (pair_off[item + 1] - pair_off[item] - 1) >> 1;
  }
  const uint32_t sk = sk_idx ? sk_idx[item] : 0u;           // NULL: one key for the whole batch
  G1Aff base;
  uint32_t k[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  bool negate;
  const rhip_g2* qsrc;
  uint32_t line = RHIP_Q_WALK, entry = 0;
  bool from_entry = false;
  if (j == 2 * m) {
    base = load_g1(ct_c[item].l);
    negate = true;
    qsrc = sk_d + sk;
    if (prepared) line = lines_d_base + sk;
  } else {
    const uint32_t e = sel_start[item] + (j >> 1);
    const uint32_t leaf = ct_leaf_off[item] + sel_ct_leaf[e];
    const uint32_t attr = sk_attr_off[sk] + sel_sk_attr[e];
    ld_scalar(k, sel_coeff + e);
    if ((j & 1) == 0) {
      base = load_g1(ct_cy_g1[leaf].l);
      negate = false;
      qsrc = sk_dj_g2 + attr;
      if (prepared) line = attr;
    } else {
      base = load_g1(sk_dj_g1[attr].l);
      negate = true;
      qsrc = ct_cy_g2 + leaf;
      if (psel) {                               // one key: -z_e * Dj.g1 depends on the selection entry alone (k_bsw_scale_entries)
        from_entry = true;
        entry = e;
        base = aff_inf<Fp>();                   // the lane still takes part in the block's inversion, with nothing to scale
        k[0] = 1; k[1] = k[2] = k[3] = k[4] = k[5] = k[6] = k[7] = 0;
      }
    }
  }
  bool p_inf;
  scale_and_store(lds, active && !from_entry, base, k, negate, P + t, &p_inf);
  if (from_entry) {
    p_inf = psel_inf[entry] != 0;
    if (active && !p_inf) st_g1_q(P + t, ld_g1_q(psel + entry));
  }
  if (!active) return;
  if (line != RHIP_Q_WALK) {
    qref[t] = (p_inf || line_inf[line]) ? RHIP_Q_SKIP : line;
  } else {
    const G2Aff q = load_g2(qsrc->l);
    const bool skip = p_inf || aff_is_inf(q);
    if (!skip) st_g2_q(Q + t, q);
    qref[t] = skip ? RHIP_Q_SKIP : RHIP_Q_WALK;
  }
}
// prepared keys: lines of every key's d_j.g2 (blocks 0 .. total_attrs-1, the index space of dev_sk_dj_g2) and d (blocks
// total_attrs + key)
struct rhip_bsw_sk_lines {
  rhip_g2_lines* l;
  size_t total_attrs;
  size_t n_sk;
};
extern "C" void rhip_bsw_sk_lines_destroy(rhip_bsw_sk_lines* p) {
  if (!p) return;
  rhip_g2_lines_destroy(p->l);
  delete p;
}
extern "C" int32_t rhip_bsw_sk_prepare(rhip_ctx* ctx, size_t n_sk, size_t total_attrs, const rhip_g2* sk_d, const rhip_g2* sk_dj_g2,
                                       rhip_bsw_sk_lines** out) {
  NEED(ctx);
  if (!out || !n_sk || !sk_d || (total_attrs && !sk_dj_g2)) return RHIP_ERR_ARG;
  *out = nullptr;
  rhip_g2* cat = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&cat, (total_attrs + n_sk) * sizeof(rhip_g2)));
  hipError_t e = hipSuccess;
  if (total_attrs) e = hipMemcpyAsync(cat, sk_dj_g2, total_attrs * sizeof(rhip_g2), hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(cat + total_attrs, sk_d, n_sk * sizeof(rhip_g2), hipMemcpyDeviceToDevice, ctx->stream);
  if (e != hipSuccess) { (void)hipFree(cat); return fail(ctx, e, "rhip_bsw_sk_prepare: copy"); }
  rhip_g2_lines* l = nullptr;
  const int32_t rc = rhip_g2_lines_prepare(ctx, total_attrs + n_sk, cat, &l);      // synchronises the stream
  (void)hipFree(cat);
  if (rc) return rc;
  *out = new rhip_bsw_sk_lines{l, total_attrs, n_sk};
  return RHIP_OK;
}
// the odd pairs of a uniform one-key batch: P = the entry's scaled key element, Q = the ciphertext's Cy.g2 (a walking pair)
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_bsw_entry_pairs(size_t n_items, uint32_t ppi, const uint32_t* sel_start, const uint32_t* sel_ct_leaf,
                                                                      const rhip_g2* ct_cy_g2, const uint32_t* ct_leaf_off, const G1M* psel,
                                                                      const uint8_t* psel_inf, G1M* P, G2M* Q, uint32_t* qref) {
  const uint32_t m = (ppi - 1) >> 1;
  const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_items * (size_t)m) return;
  const size_t item = v / m;
  const uint32_t s_ = (uint32_t)(v % m);
  const size_t t = item * (size_t)ppi + 2 * s_ + 1;
  const uint32_t e = sel_start[item] + s_;
  const G2Aff q = load_g2(ct_cy_g2[ct_leaf_off[item] + sel_ct_leaf[e]].l);
  const bool skip = psel_inf[e] != 0 || aff_is_inf(q);
  if (!skip) { st_g1_q(P + t, ld_g1_q(psel + e)); st_g2_q(Q + t, q); }
  qref[t] = skip ? RHIP_Q_SKIP : RHIP_Q_WALK;
}
// one key for the whole batch: entry e -> -z_e * Dj.g1[sel_sk_attr[e]]
__global__ void __launch_bounds__(RB_PAIRS_BLOCK, 2) k_bsw_scale_entries(size_t n_sel, const uint32_t* sel_sk_attr, const rhip_fr* sel_coeff, const rhip_g1* sk_dj_g1,
                                                                        G1M* psel, uint8_t* psel_inf) {
  __shared__ uint32_t lds[2 * 8 * RB_PAIRS_BLOCK];
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = e < n_sel;
  if (!active) e = n_sel - 1;
  uint32_t k[8];
  ld_scalar(k, sel_coeff + e);
  bool inf;
  scale_and_store(lds, active, load_g1(sk_dj_g1[sel_sk_attr[e]].l), k, true, psel + e, &inf);
  if (active) psel_inf[e] = inf ? 1 : 0;
}
static int32_t bsw_decrypt_impl(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel, const uint32_t* pair_off,
                                const uint32_t* sel_start, const uint32_t* sel_ct_leaf, const uint32_t* sel_sk_attr,
                                const rhip_fr* sel_coeff, const rhip_g1* ct_c, const rhip_gt* ct_cp, const rhip_g1* ct_cy_g1,
                                const rhip_g2* ct_cy_g2, const uint32_t* ct_leaf_off, const rhip_g2* sk_d, const rhip_g1* sk_dj_g1,
                                const rhip_g2* sk_dj_g2, const uint32_t* sk_attr_off, const uint32_t* sk_idx,
                                const rhip_bsw_sk_lines* sk_lines, rhip_gt* out);
extern "C" int32_t rhip_bsw_decrypt_batch(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, const uint32_t* pair_off,
                                          const uint32_t* sel_start, const uint32_t* sel_ct_leaf, const uint32_t* sel_sk_attr,
                                          const rhip_fr* sel_coeff, const rhip_g1* ct_c, const rhip_gt* ct_cp, const rhip_g1* ct_cy_g1,
                                          const rhip_g2* ct_cy_g2, const uint32_t* ct_leaf_off, const rhip_g2* sk_d, const rhip_g1* sk_dj_g1,
                                          const rhip_g2* sk_dj_g2, const uint32_t* sk_attr_off, const uint32_t* sk_idx,
                                          const rhip_bsw_sk_lines* sk_lines, rhip_gt* out) {
  if (!sk_idx) return RHIP_ERR_ARG;
  return bsw_decrypt_impl(ctx, n_items, max_pairs, total_pairs, 0, pair_off, sel_start, sel_ct_leaf, sel_sk_attr, sel_coeff, ct_c, ct_cp, ct_cy_g1, ct_cy_g2,
                          ct_leaf_off, sk_d, sk_dj_g1, sk_dj_g2, sk_attr_off, sk_idx, sk_lines, out);
}
extern "C" int32_t rhip_bsw_decrypt_batch_one_sk(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel, const uint32_t* pair_off,
                                                 const uint32_t* sel_start, const uint32_t* sel_ct_leaf, const uint32_t* sel_sk_attr,
                                                 const rhip_fr* sel_coeff, const rhip_g1* ct_c, const rhip_gt* ct_cp, const rhip_g1* ct_cy_g1,
                                                 const rhip_g2* ct_cy_g2, const uint32_t* ct_leaf_off, const rhip_g2* sk_d, const rhip_g1* sk_dj_g1,
                                                 const rhip_g2* sk_dj_g2, const uint32_t* sk_attr_off, const rhip_bsw_sk_lines* sk_lines, rhip_gt* out) {
  return bsw_decrypt_impl(ctx, n_items, max_pairs, total_pairs, n_sel, pair_off, sel_start, sel_ct_leaf, sel_sk_attr, sel_coeff, ct_c, ct_cp, ct_cy_g1, ct_cy_g2,
                          ct_leaf_off, sk_d, sk_dj_g1, sk_dj_g2, sk_attr_off, (const uint32_t*)nullptr, sk_lines, out);
}
static int32_t bsw_decrypt_impl(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel, const uint32_t* pair_off,
                                const uint32_t* sel_start, const uint32_t* sel_ct_leaf, const uint32_t* sel_sk_attr,
                                const rhip_fr* sel_coeff, const rhip_g1* ct_c, const rhip_gt* ct_cp, const rhip_g1* ct_cy_g1,
                                const rhip_g2* ct_cy_g2, const uint32_t* ct_leaf_off, const rhip_g2* sk_d, const rhip_g1* sk_dj_g1,
                                const rhip_g2* sk_dj_g2, const uint32_t* sk_attr_off, const uint32_t* sk_idx,
                                const rhip_bsw_sk_lines* sk_lines, rhip_gt* out) {
  NEED(ctx);
  if (!n_items) return RHIP_OK;
  if (!total_pairs || !pair_off) return RHIP_ERR_ARG;
  PairLists pl;
  int32_t rc = alloc_pair_lists(ctx, total_pairs, &pl);
  if (rc) return rc;
  const uint32_t ppi = uniform_ppi(n_items, max_pairs, total_pairs);
  G1M* psel = nullptr;
  uint8_t* psel_inf = nullptr;
  if (!sk_idx && n_sel && 2 * n_sel < total_pairs - n_items) {     // one key and shared entries: its scaled Dj.g1 once per entry
    void* w_sel = nullptr;
    rc = rhip_ensure_work(ctx, 8, n_sel * (sizeof(G1M) + 1) + 64, &w_sel);
    if (rc) return rc;
    psel = (G1M*)w_sel;
    psel_inf = (uint8_t*)(psel + n_sel);
    KLAUNCH(ctx, "k_bsw_scale_entries", k_bsw_scale_entries, dim3(blocks_for(n_sel, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_sel, sel_sk_attr, sel_coeff,
            sk_dj_g1, psel, psel_inf);
  }
  const int compact = (psel && ppi >= 3) ? 1 : 0;        // uniform batch: the odd pairs are copies (k_bsw_entry_pairs), the others get the lanes
  size_t lanes = 0;
  const uint32_t* tile_off = nullptr;
  if (compact) lanes = pair_lanes(n_items, n_items * (size_t)((ppi + 1) / 2), (ppi + 1) / 2);
  else if ((rc = gather_lanes(ctx, n_items, max_pairs, total_pairs, pair_off, ppi, &tile_off, &lanes)) != RHIP_OK) return rc;
  KLAUNCH(ctx, "k_bsw_dec_pairs", k_bsw_dec_pairs, dim3(blocks_for(lanes, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_items,
          total_pairs, pair_off, ppi, tile_off, sel_start, sel_ct_leaf, sel_sk_attr, sel_coeff, ct_c, ct_cy_g1, ct_cy_g2, ct_leaf_off, sk_d, sk_dj_g1, sk_dj_g2,
          sk_attr_off, sk_idx, (uint32_t)(sk_lines ? sk_lines->total_attrs : 0), sk_lines ? 1 : 0,
          (const uint8_t*)(sk_lines ? sk_lines->l->q_inf : nullptr), pl.P, pl.Q, pl.qref, (const G1M*)psel, (const uint8_t*)psel_inf, compact);
  if (compact)
    KLAUNCH(ctx, "k_bsw_entry_pairs", k_bsw_entry_pairs, dim3(blocks_for(n_items * (size_t)((ppi - 1) / 2), 256)), dim3(256), 0, ctx->stream, n_items, ppi, sel_start,
            sel_ct_leaf, ct_cy_g2, ct_leaf_off, (const G1M*)psel, (const uint8_t*)psel_inf, pl.P, pl.Q, pl.qref);
  return run_pair_lists(ctx, n_items, pair_off, max_pairs, total_pairs, pl, sk_lines ? (const LineM*)sk_lines->l->lines : (const LineM*)nullptr, sk_lines ? sk_lines->l->lines29 : nullptr, ct_cp, out);
}

// ------------------------------------------------------------------------------------------------ shared-doubling sums
// NAF masks of a batch's selection coefficients, once per entry: masks[e] = pos[8] | neg[8] of the SHORTER of c and r - c,
// with the sign folded in (negating a scalar swaps its masks), so sum_j c_j P_j needs 254 doublings for the whole sum.
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_naf_masks(size_t n, const rhip_fr* k, uint32_t* masks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t kk[8], pos[8], neg_[8];
  ld_scalar(kk, k + i);
  const bool flip = fr_shorten(kk);
  naf_masks(kk, pos, neg_);
#pragma unroll
  for (int w = 0; w < 8; w++) {
    masks[16 * i + w] = flip ? neg_[w] : pos[w];
    masks[16 * i + 8 + w] = flip ? pos[w] : neg_[w];
  }
}
struct G2JM { uint32_t l[48]; };
__device__ __forceinline__ void st_jac_q(G1JM* p, const G1Jac& a) { uint4* q = (uint4*)p; st_fp_q(q, a.x); st_fp_q(q + 2, a.y); st_fp_q(q + 4, a.z); }
__device__ __forceinline__ G1Jac ld_jac_q(const G1JM* p) { const uint4* q = (const uint4*)p; return G1Jac{ld_fp_q(q), ld_fp_q(q + 2), ld_fp_q(q + 4)}; }
__device__ __forceinline__ void st_jac_q(G2JM* p, const G2Jac& a) {
  uint4* q = (uint4*)p;
  st_fp_q(q, a.x.c0); st_fp_q(q + 2, a.x.c1); st_fp_q(q + 4, a.y.c0); st_fp_q(q + 6, a.y.c1); st_fp_q(q + 8, a.z.c0); st_fp_q(q + 10, a.z.c1);
}
__device__ __forceinline__ G2Jac ld_jac_q(const G2JM* p) {
  const uint4* q = (const uint4*)p;
  return G2Jac{Fp2{ld_fp_q(q), ld_fp_q(q + 2)}, Fp2{ld_fp_q(q + 4), ld_fp_q(q + 6)}, Fp2{ld_fp_q(q + 8), ld_fp_q(q + 10)}};
}
__device__ __forceinline__ G1Aff ld_aff_q(const G1M* p) { return ld_g1_q(p); }
__device__ __forceinline__ G2Aff ld_aff_q(const G2M* p) { return ld_g2_q(p); }
// terms of one lane: bases pts[0 .. cnt) (Montgomery), masks of entries e0 .. e0 + cnt - 1; `flip` negates every scalar
template <class F, class AFFM>
struct DevTerms {
  const AFFM* pts;
  const uint32_t* masks;      // masks + 16 e0
  int cnt;
  bool flip;
  __device__ __forceinline__ int count() const { return cnt; }
  __device__ __forceinline__ Aff<F> base(int j) const { return ld_aff_q(pts + j); }
  __device__ __forceinline__ uint32_t pos_word(int j, int w) const { return masks[16 * j + (flip ? 8 : 0) + w]; }
  __device__ __forceinline__ uint32_t neg_word(int j, int w) const { return masks[16 * j + (flip ? 0 : 8) + w]; }
};
// lane t = chunk * n_items + item: the partial sum over terms [term_off[item] + c C, ...) of the item, Jacobian
template <class F, class AFFM, class JACM>
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_msm_partial(size_t n_items, uint32_t L, uint32_t C, const uint32_t* term_off, const uint32_t* sel_start,
                                                                 const AFFM* pts, const uint32_t* masks, int flip, JACM* part) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_items * L) return;
  const size_t c = t / n_items, item = t % n_items;
  const uint32_t lo = term_off[item], hi = term_off[item + 1];
  const uint64_t first = (uint64_t)lo + (uint64_t)c * C;
  int cnt = 0;
  if (first < hi) cnt = (int)((hi - first < C) ? (hi - first) : C);
  const DevTerms<F, AFFM> terms{pts + first, masks + 16 * ((size_t)sel_start[item] + c * C), cnt, flip != 0};
  st_jac_q(part + item * L + c, jac_msm_naf<F>(terms));
}
// lane = item: sum of its L partial sums -> the G1 argument of the item's last pair (affine Montgomery; one inversion per block)
__global__ void __launch_bounds__(RB_PAIRS_BLOCK, 2) k_msm_finish_g1(size_t n_items, uint32_t L, const G1JM* part, const uint32_t* pair_off, G1M* P, uint32_t* qref) {
  __shared__ uint32_t lds[2 * 8 * RB_PAIRS_BLOCK];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n_items;
  if (!active) i = n_items - 1;
  G1Jac acc = ld_jac_q(part + i * L);
  for (uint32_t c = 1; c < L; c++) acc = jac_add(acc, ld_jac_q(part + i * L + c));
  const bool inf = !active || jac_is_inf(acc);
  const Fp zinv = block_batch_inverse_n<RB_PAIRS_BLOCK>(lds, inf ? one<FpParams>() : acc.z);
  if (!active) return;
  const size_t last = (size_t)pair_off[i + 1] - 1;
  if (inf) qref[last] = RHIP_Q_SKIP;
  else st_g1_q(P + last, jac_to_aff_with_zinv(acc, zinv));
}
// the same for a G2 sum -> the G2 argument of the item's last pair (a walking pair)
__global__ void __launch_bounds__(128, RB_G2_WAVES) k_msm_finish_g2(size_t n_items, uint32_t L, const G2JM* part, const uint32_t* pair_off, G2M* Q, uint32_t* qref) {
  __shared__ uint32_t lds[2 * 8 * 128];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n_items;
  if (!active) i = n_items - 1;
  G2Jac acc = ld_jac_q(part + i * L);
  for (uint32_t c = 1; c < L; c++) acc = jac_add(acc, ld_jac_q(part + i * L + c));
  const bool inf = !active || jac_is_inf(acc);
  const Fp norm = inf ? one<FpParams>() : add(sqr(acc.z.c0), sqr(acc.z.c1));
  const Fp ninv = block_batch_inverse_n<128>(lds, norm);
  if (!active) return;
  const size_t last = (size_t)pair_off[i + 1] - 1;
  if (inf) { qref[last] = RHIP_Q_SKIP; return; }
  const Fp2 zinv{mul(acc.z.c0, ninv), neg(mul(acc.z.c1, ninv))};
  st_g2_q(Q + last, jac_to_aff_with_zinv(acc, zinv));
}
// chunking of the shared-doubling sums: a lane pays its own 254 doublings / squarings (~1.8 k in G1, ~4 k in G2, ~4.6 k in Gt)
// plus ~1 k (G1) .. 4.6 k (Gt) per term; same rounds x lane-time model as choose_chunks with a 2 : 1 doubling-to-term ratio
static void choose_msm_chunks(const rhip_ctx* ctx, size_t n_items, size_t max_terms, uint32_t* L, uint32_t* C) {
  if (max_terms < 1) max_terms = 1;
  const size_t simds = (size_t)ctx->n_cu * 4;
  double best = 0;
  size_t best_c = 1;
  for (size_t c = 1; c <= 256; c++) {
    const size_t l = (max_terms + c - 1) / c;
    const size_t c_eff = (max_terms + l - 1) / l;
    const size_t waves = (n_items * l + 63) / 64;
    const size_t rounds = (waves + simds - 1) / simds;
    const double cost = (double)rounds * (2.0 + (double)c_eff);
    if (best == 0 || cost < best) { best = cost; best_c = c; }
    if (l == 1) break;
  }
  const size_t l = (max_terms + best_c - 1) / best_c;
  *C = (uint32_t)((max_terms + l - 1) / l);
  *L = (uint32_t)l;
}

// ------------------------------------------------------------------------------------------------ LSW KP-ABE
struct rhip_lsw_pk {
  rhip_ctx* ctx;
  rhip_g1_table* g1;
  rhip_g2_table* g2;
};
extern "C" void rhip_lsw_pk_destroy(rhip_lsw_pk* pk) {
  if (!pk) return;
  rhip_g1_table_destroy(pk->g1);
  rhip_g2_table_destroy(pk->g2);
  delete pk;
}
extern "C" int32_t rhip_lsw_pk_create(rhip_ctx* ctx, const rhip_g1* g1, const rhip_g2* g2, rhip_lsw_pk** out) {
  if (!ctx || !g1 || !g2 || !out) return RHIP_ERR_ARG;
  *out = nullptr;
  rhip_lsw_pk* pk = new rhip_lsw_pk{ctx, nullptr, nullptr};
  int32_t rc = rhip_g1_table_create(ctx, g1, &pk->g1);
  if (!rc) rc = rhip_g1_table_add_w16(ctx, pk->g1);
  if (!rc) rc = rhip_g2_table_create(ctx, g2, &pk->g2);
  if (!rc) rc = rhip_g2_table_add_w16(ctx, pk->g2);
  if (rc) { rhip_lsw_pk_destroy(pk); return rc; }
  *out = pk;
  return RHIP_OK;
}
// keygen, positive leaves (lsw/mod.rs:147-160): D1 = g1 * (alpha2 q_y + h(y) r_y), D2 = g2 * r_y, q_y the leaf's share of alpha1
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_lsw_keygen_scalars(size_t n_items, size_t total_leaves, const uint32_t* item_leaf_off,
                                                                         const uint32_t* item_tree_leaf, const uint32_t* item_tree_gate, TreeTables tt,
                                                                         const rhip_fr* leaf_hash, const rhip_fr* alpha /*[2]*/, const rhip_fr* coef,
                                                                         const uint32_t* item_coef_off, const rhip_fr* rand, rhip_fr* k1) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_leaves) return;
  const size_t item = owner_of(item_leaf_off, n_items, t);
  const uint32_t leaf = item_tree_leaf[item] + (uint32_t)(t - item_leaf_off[item]);
  const Fr q = share_of_leaf(tt, leaf, item_tree_gate[item], coef + item_coef_off[item], load_fr(alpha[0].l));
  store_fr(k1[t].l, add(mul(load_fr(alpha[1].l), q), mul(load_fr(leaf_hash[leaf].l), load_fr(rand[t].l))));
}
extern "C" int32_t rhip_lsw_keygen_batch(rhip_ctx* ctx, const rhip_lsw_pk* pk, size_t n_items, size_t total_leaves, const uint32_t* item_leaf_off,
                                         const uint32_t* item_tree_leaf, const uint32_t* item_tree_gate, const uint32_t* path_off,
                                         const uint32_t* path_gate, const uint32_t* path_x, const uint32_t* gate_k, const uint32_t* gate_coef_off,
                                         const rhip_fr* leaf_hash, const rhip_fr* alpha, const rhip_fr* coef, const uint32_t* item_coef_off,
                                         const rhip_fr* rand, rhip_g1* d1, rhip_g2* d2) {
  NEED(ctx);
  if (!pk) return RHIP_ERR_ARG;
  if (!n_items || !total_leaves) return RHIP_OK;
  void* w = nullptr;
  int32_t rc = rhip_ensure_work(ctx, 4, total_leaves * sizeof(rhip_fr), &w);
  if (rc) return rc;
  const TreeTables tt{path_off, path_gate, path_x, gate_k, gate_coef_off};
  KLAUNCH(ctx, "k_lsw_keygen_scalars", k_lsw_keygen_scalars, dim3(blocks_for(total_leaves, 256)), dim3(256), 0, ctx->stream, n_items, total_leaves,
          item_leaf_off, item_tree_leaf, item_tree_gate, tt, leaf_hash, alpha, coef, item_coef_off, rand, (rhip_fr*)w);
  rc = rhip_g1_table_mul(ctx, pk->g1, total_leaves, (const rhip_fr*)w, d1);
  if (rc) return rc;
  return rhip_g2_table_mul(ctx, pk->g2, total_leaves, rand, d2);
}
// keygen with negative leaves ("!x", lsw/mod.rs:137-146) beside positive ones: per leaf row six scalars --
//   positive:  k1 = alpha2 q + h(y) r   (D1 = g1 * k1),   k2 = r  (D2 = g2 * r),   k3 = k4a = k4b = k5 = 0
//   negative:  k1 = k2 = 0,   k3 = q + b^2 r   (D3 = g1 * q + g1_b2 * r = g1 * k3),   k4a = b h(y) r,  k4b = r
//              (D4 = g1_b * (h r) + h_g1 * r = g1 * k4a + h_g1 * k4b),   k5 = -r   (D5 = g1 * (-r))
// a zero scalar gives the identity, which is what the reference's struct holds in the unused slots of a row (None -> G::zero()).
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_lsw_keygen_scalars_signed(size_t n_items, size_t total_leaves, const uint32_t* item_leaf_off,
                                                                                const uint32_t* item_tree_leaf, const uint32_t* item_tree_gate, TreeTables tt,
                                                                                const rhip_fr* leaf_hash, const uint32_t* leaf_neg, const rhip_fr* alpha /*[2]*/,
                                                                                const rhip_fr* b, const rhip_fr* coef, const uint32_t* item_coef_off,
                                                                                const rhip_fr* rand, rhip_fr* k /* [6][total_leaves] */) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_leaves) return;
  const size_t item = owner_of(item_leaf_off, n_items, t);
  const uint32_t leaf = item_tree_leaf[item] + (uint32_t)(t - item_leaf_off[item]);
  const Fr q = share_of_leaf(tt, leaf, item_tree_gate[item], coef + item_coef_off[item], load_fr(alpha[0].l));
  const Fr r = load_fr(rand[t].l), h = load_fr(leaf_hash[leaf].l), z = zero<FrParams>();
  const bool neg_leaf = leaf_neg[leaf] != 0;
  Fr k1 = z, k2 = z, k3 = z, k4a = z, k4b = z, k5 = z;
  if (neg_leaf) {
    const Fr bb = load_fr(b[0].l);
    k3 = add(q, mul(mul(bb, bb), r));
    k4a = mul(mul(bb, h), r);
    k4b = r;
    k5 = neg(r);
  } else {
    k1 = add(mul(load_fr(alpha[1].l), q), mul(h, r));
    k2 = r;
  }
  store_fr(k[t].l, k1);
  store_fr(k[total_leaves + t].l, k2);
  store_fr(k[2 * total_leaves + t].l, k3);
  store_fr(k[3 * total_leaves + t].l, k4a);
  store_fr(k[4 * total_leaves + t].l, k4b);
  store_fr(k[5 * total_leaves + t].l, k5);
}
extern "C" int32_t rhip_lsw_keygen_batch_signed(rhip_ctx* ctx, const rhip_lsw_pk* pk, size_t n_items, size_t total_leaves, const uint32_t* item_leaf_off,
                                                const uint32_t* item_tree_leaf, const uint32_t* item_tree_gate, const uint32_t* path_off,
                                                const uint32_t* path_gate, const uint32_t* path_x, const uint32_t* gate_k, const uint32_t* gate_coef_off,
                                                const rhip_fr* leaf_hash, const uint32_t* leaf_neg, const rhip_fr* alpha, const rhip_fr* b,
                                                const rhip_g1* host_h_g1, const rhip_fr* coef, const uint32_t* item_coef_off, const rhip_fr* rand,
                                                rhip_g1* d1, rhip_g2* d2, rhip_g1* d3, rhip_g1* d4, rhip_g1* d5) {
  NEED(ctx);
  if (!pk || !host_h_g1) return RHIP_ERR_ARG;
  if (!n_items || !total_leaves) return RHIP_OK;
  void* w = nullptr;
  int32_t rc = rhip_ensure_work(ctx, 4, total_leaves * (6 * sizeof(rhip_fr) + sizeof(rhip_g1)), &w);
  if (rc) return rc;
  rhip_fr* k = (rhip_fr*)w;
  rhip_g1* tmp = (rhip_g1*)(k + 6 * total_leaves);
  const TreeTables tt{path_off, path_gate, path_x, gate_k, gate_coef_off};
  KLAUNCH(ctx, "k_lsw_keygen_scalars_signed", k_lsw_keygen_scalars_signed, dim3(blocks_for(total_leaves, 256)), dim3(256), 0, ctx->stream, n_items,
          total_leaves, item_leaf_off, item_tree_leaf, item_tree_gate, tt, leaf_hash, leaf_neg, alpha, b, coef, item_coef_off, rand, k);
  rhip_g1_table* th = nullptr;
  rc = rhip_g1_table_create(ctx, host_h_g1, &th);
  if (!rc) rc = rhip_g1_table_mul(ctx, pk->g1, total_leaves, k, d1);
  if (!rc) rc = rhip_g2_table_mul(ctx, pk->g2, total_leaves, k + total_leaves, d2);
  if (!rc) rc = rhip_g1_table_mul(ctx, pk->g1, total_leaves, k + 2 * total_leaves, d3);
  if (!rc) rc = rhip_g1_table_mul(ctx, pk->g1, total_leaves, k + 3 * total_leaves, tmp);
  if (!rc) rc = rhip_g1_table_mul(ctx, th, total_leaves, k + 4 * total_leaves, d4);
  if (!rc) rc = rhip_g1_add(ctx, total_leaves, tmp, d4, d4);
  if (!rc) rc = rhip_g1_table_mul(ctx, pk->g1, total_leaves, k + 5 * total_leaves, d5);
  if (!rc) rc = rhip_sync(ctx);                       // the table of h_g1 lives for this call only
  rhip_g1_table_destroy(th);
  return rc;
}
// decrypt (lsw/mod.rs:228-290 restated in SURVEY.md Appendix B.4): item i owns pairs [pair_off[i], pair_off[i+1]) = m_i + 1:
//   s < m : P = c_e * E1[ct attr],                  Q = D2[key leaf]          (e = sel_start[i] + s)
//   m     : P = sum_e (-c_e) * D1[key leaf]  (MSM),  Q = e2
// This kernel does the scaled pairs and gathers the MSM's bases (D1, Montgomery) at terms[pair index - item].
__global__ void __launch_bounds__(RB_PAIRS_BLOCK, 2) k_lsw_dec_pairs(size_t n_items, size_t total_pairs, const uint32_t* pair_off, uint32_t ppi, const uint32_t* tile_off, const uint32_t* sel_start,
                                                                    const uint32_t* sel_sk_leaf, const uint32_t* sel_ct_attr, const rhip_fr* sel_coeff,
                                                                    const rhip_g2* ct_e2, const rhip_g1* ct_e1j, const uint32_t* ct_attr_off,
                                                                    const uint32_t* ct_idx, const rhip_g1* sk_d1, const rhip_g2* sk_d2,
                                                                    const uint32_t* sk_leaf_off, const uint32_t* sk_idx, const uint8_t* e2_line_inf,
                                                                    G1M* P, G2M* Q, uint32_t* qref, G1M* terms, const G1M* psel, const uint8_t* psel_inf,
                                                                    int one_ct) {
  __shared__ uint32_t lds[2 * 8 * RB_PAIRS_BLOCK];
  size_t t, item;
  bool active;
  pair_lane((size_t)blockIdx.x * blockDim.x + threadIdx.x, n_items, total_pairs, pair_off, ppi, tile_off, &t, &item, &active);
  const uint32_t j = (uint32_t)(t - pair_off[item]);
  const uint32_t m = pair_off[item + 1] - pair_off[item] - 1;
  const uint32_t ct = one_ct ? 0u : ct_idx ? ct_idx[item] : (uint32_t)item;
  const uint32_t sk = sk_idx ? sk_idx[item] : (uint32_t)item;
  const bool last = (j == m);
  G1Aff base = aff_inf<Fp>();
  uint32_t k[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  uint32_t leaf = 0, e = 0;
  if (!last) {
    e = sel_start[item] + j;
    leaf = sk_leaf_off[sk] + sel_sk_leaf[e];
  }
  bool p_inf;
  if (psel) {                                  // one ciphertext for all items: c_e * E1_e was computed once per selection entry (k_lsw_scale_entries)
    p_inf = last || psel_inf[e] != 0;
    if (active && !p_inf) st_g1_q(P + t, ld_g1_q(psel + e));
  } else {
    if (!last) {
      base = load_g1(ct_e1j[(one_ct ? 0u : ct_attr_off[ct]) + sel_ct_attr[e]].l);
      ld_scalar(k, sel_coeff + e);
    }
    scale_and_store(lds, active && !last, base, k, false, P + t, &p_inf);
  }
  if (!active) return;
  if (last) {
    if (e2_line_inf) {                                   // prepared lines of the ciphertexts' e2: block = ciphertext index
      qref[t] = e2_line_inf[ct] ? RHIP_Q_SKIP : ct;
    } else {
      const G2Aff q = load_g2(ct_e2[ct].l);
      if (!aff_is_inf(q)) st_g2_q(Q + t, q);
      qref[t] = aff_is_inf(q) ? RHIP_Q_SKIP : RHIP_Q_WALK;
    }
    return;
  }
  st_g1_q(terms + (t - item), load_g1(sk_d1[leaf].l));
  const G2Aff q = load_g2(sk_d2[leaf].l);
  const bool skip = p_inf || aff_is_inf(q);
  if (!skip) st_g2_q(Q + t, q);
  qref[t] = skip ? RHIP_Q_SKIP : RHIP_Q_WALK;
}
// term offsets of the MSM: item i has m_i = pair_off[i+1] - pair_off[i] - 1 terms starting at pair_off[i] - i
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_term_off(size_t n_items, const uint32_t* pair_off, uint32_t* term_off, uint32_t other = 1) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_items) term_off[i] = pair_off[i] - other * (uint32_t)i;          // `other`: pairs of an item that are not terms of its sum
}
// one ciphertext for the whole batch: the scaled G1 arguments depend on the selection entry alone -- entry e: c_e * E1[sel_ct_attr[e]]
__global__ void __launch_bounds__(RB_PAIRS_BLOCK, 2) k_lsw_scale_entries(size_t n_sel, const uint32_t* sel_ct_attr, const rhip_fr* sel_coeff, const rhip_g1* ct_e1j,
                                                                        G1M* psel, uint8_t* psel_inf) {
  __shared__ uint32_t lds[2 * 8 * RB_PAIRS_BLOCK];
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = e < n_sel;
  if (!active) e = n_sel - 1;
  uint32_t k[8];
  ld_scalar(k, sel_coeff + e);
  bool inf;
  scale_and_store(lds, active, load_g1(ct_e1j[sel_ct_attr[e]].l), k, false, psel + e, &inf);
  if (active) psel_inf[e] = inf ? 1 : 0;
}
static int32_t lsw_decrypt_impl(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel, const uint32_t* pair_off,
                                const uint32_t* sel_start, const uint32_t* sel_sk_leaf, const uint32_t* sel_ct_attr, const rhip_fr* sel_coeff,
                                const rhip_gt* ct_e1, const rhip_g2* ct_e2, const rhip_g1* ct_e1j, const uint32_t* ct_attr_off,
                                const uint32_t* ct_idx, const rhip_g1* sk_d1, const rhip_g2* sk_d2, const uint32_t* sk_leaf_off,
                                const uint32_t* sk_idx, const rhip_g2_lines* ct_e2_lines, rhip_gt* out, bool one_ct);
extern "C" int32_t rhip_lsw_decrypt_batch(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel, const uint32_t* pair_off,
                                          const uint32_t* sel_start, const uint32_t* sel_sk_leaf, const uint32_t* sel_ct_attr, const rhip_fr* sel_coeff,
                                          const rhip_gt* ct_e1, const rhip_g2* ct_e2, const rhip_g1* ct_e1j, const uint32_t* ct_attr_off,
                                          const uint32_t* ct_idx, const rhip_g1* sk_d1, const rhip_g2* sk_d2, const uint32_t* sk_leaf_off,
                                          const uint32_t* sk_idx, const rhip_g2_lines* ct_e2_lines, rhip_gt* out) {
  return lsw_decrypt_impl(ctx, n_items, max_pairs, total_pairs, n_sel, pair_off, sel_start, sel_sk_leaf, sel_ct_attr, sel_coeff, ct_e1, ct_e2, ct_e1j,
                          ct_attr_off, ct_idx, sk_d1, sk_d2, sk_leaf_off, sk_idx, ct_e2_lines, out, false);
}
extern "C" int32_t rhip_lsw_decrypt_batch_one_ct(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel, const uint32_t* pair_off,
                                                 const uint32_t* sel_start, const uint32_t* sel_sk_leaf, const uint32_t* sel_ct_attr, const rhip_fr* sel_coeff,
                                                 const rhip_gt* ct_e1, const rhip_g2* ct_e2, const rhip_g1* ct_e1j, const rhip_g1* sk_d1,
                                                 const rhip_g2* sk_d2, const uint32_t* sk_leaf_off, const uint32_t* sk_idx,
                                                 const rhip_g2_lines* ct_e2_lines, rhip_gt* out) {
  return lsw_decrypt_impl(ctx, n_items, max_pairs, total_pairs, n_sel, pair_off, sel_start, sel_sk_leaf, sel_ct_attr, sel_coeff, ct_e1, ct_e2, ct_e1j,
                          (const uint32_t*)nullptr, (const uint32_t*)nullptr, sk_d1, sk_d2, sk_leaf_off, sk_idx, ct_e2_lines, out, true);
}
static int32_t lsw_decrypt_impl(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel, const uint32_t* pair_off,
                                const uint32_t* sel_start, const uint32_t* sel_sk_leaf, const uint32_t* sel_ct_attr, const rhip_fr* sel_coeff,
                                const rhip_gt* ct_e1, const rhip_g2* ct_e2, const rhip_g1* ct_e1j, const uint32_t* ct_attr_off,
                                const uint32_t* ct_idx, const rhip_g1* sk_d1, const rhip_g2* sk_d2, const uint32_t* sk_leaf_off,
                                const uint32_t* sk_idx, const rhip_g2_lines* ct_e2_lines, rhip_gt* out, bool one_ct) {
  NEED(ctx);
  if (!n_items) return RHIP_OK;
  if (!total_pairs || !pair_off || !n_sel) return RHIP_ERR_ARG;
  PairLists pl;
  int32_t rc = alloc_pair_lists(ctx, total_pairs, &pl);
  if (rc) return rc;
  const size_t total_terms = total_pairs - n_items;
  void *w_terms = nullptr, *w_masks = nullptr, *w_part = nullptr, *w_off = nullptr;
  rc = rhip_ensure_work(ctx, 4, (total_terms ? total_terms : 1) * sizeof(G1M), &w_terms);
  if (!rc) rc = rhip_ensure_work(ctx, 5, n_sel * 16 * sizeof(uint32_t), &w_masks);
  uint32_t L, C;
  choose_msm_chunks(ctx, n_items, max_pairs - 1, &L, &C);
  if (!rc) rc = rhip_ensure_work(ctx, 6, n_items * L * sizeof(G1JM), &w_part);
  if (!rc) rc = rhip_ensure_work(ctx, 7, (n_items + 1) * sizeof(uint32_t), &w_off);
  if (rc) return rc;
  KLAUNCH(ctx, "k_naf_masks", k_naf_masks, dim3(blocks_for(n_sel, 256)), dim3(256), 0, ctx->stream, n_sel, sel_coeff, (uint32_t*)w_masks);
  KLAUNCH(ctx, "k_term_off", k_term_off, dim3(blocks_for(n_items + 1, 256)), dim3(256), 0, ctx->stream, n_items, pair_off, (uint32_t*)w_off);
  const uint32_t ppi = uniform_ppi(n_items, max_pairs, total_pairs);
  G1M* psel = nullptr;
  uint8_t* psel_inf = nullptr;
  if (one_ct && n_sel < total_pairs - n_items) {          // fewer entries than pairs: items share entries, scale per entry
    void* w_sel = nullptr;
    rc = rhip_ensure_work(ctx, 8, n_sel * (sizeof(G1M) + 1) + 64, &w_sel);
    if (rc) return rc;
    psel = (G1M*)w_sel;
    psel_inf = (uint8_t*)(psel + n_sel);
    KLAUNCH(ctx, "k_lsw_scale_entries", k_lsw_scale_entries, dim3(blocks_for(n_sel, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_sel, sel_ct_attr, sel_coeff,
            ct_e1j, psel, psel_inf);
  }
  size_t g_lanes = 0;
  const uint32_t* tile_off = nullptr;
  if ((rc = gather_lanes(ctx, n_items, max_pairs, total_pairs, pair_off, ppi, &tile_off, &g_lanes)) != RHIP_OK) return rc;
  KLAUNCH(ctx, "k_lsw_dec_pairs", k_lsw_dec_pairs, dim3(blocks_for(g_lanes, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_items, total_pairs,
          pair_off, ppi, tile_off, sel_start, sel_sk_leaf, sel_ct_attr, sel_coeff, ct_e2, ct_e1j, ct_attr_off, ct_idx, sk_d1, sk_d2, sk_leaf_off, sk_idx,
          (const uint8_t*)(ct_e2_lines ? ct_e2_lines->q_inf : nullptr), pl.P, pl.Q, pl.qref, (G1M*)w_terms, (const G1M*)psel, (const uint8_t*)psel_inf, one_ct ? 1 : 0);
  KLAUNCH(ctx, "k_msm_partial_g1", (k_msm_partial<Fp, G1M, G1JM>), dim3(blocks_for(n_items * L, 64)), dim3(64), 0, ctx->stream, n_items, L, C,
          (const uint32_t*)w_off, sel_start, (const G1M*)w_terms, (const uint32_t*)w_masks, 1, (G1JM*)w_part);
  KLAUNCH(ctx, "k_msm_finish_g1", k_msm_finish_g1, dim3(blocks_for(n_items, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_items, L,
          (const G1JM*)w_part, pair_off, pl.P, pl.qref);
  return run_pair_lists(ctx, n_items, pair_off, max_pairs, total_pairs, pl, ct_e2_lines ? (const LineM*)ct_e2_lines->lines : (const LineM*)nullptr, ct_e2_lines ? ct_e2_lines->lines29 : nullptr, ct_e1, out);
}

// ------------------------------------------------------------------------------------------------ GHW11 outsourced decryption
// transform (ghw11/mod.rs:227-295; SURVEY.md 8f-1: "decrypt-as-a-service"): every G2 argument is one of the TRANSFORM KEY's
// (k_z, l_z, k_x per attribute) -- fixed for a batch served under one key, so all m + 2 Miller loops of an item replay prepared
// lines and no G2 arithmetic happens at all.  Item i owns pairs [pair_off[i], pair_off[i+1]) = m_i + 2:
//   s < m : P = -(w_e * D[ct row]),              lines of k_x[tk attr]   (block 2 + attr)        (e = sel_start[i] + s)
//   m     : P = C1[i],                           lines of k_z            (block 0)
//   m + 1 : P = sum_e (-w_e) * C[ct row]  (MSM), lines of l_z            (block 1)
//   t_i = FE( prod of the Miller values ) = e(c1, k_z) / ( prod_e e(w_e D_e, K_e) * e(sum_e w_e C_e, l_z) )
__global__ void __launch_bounds__(RB_PAIRS_BLOCK, 2) k_ghw11_pairs(size_t n_items, size_t total_pairs, const uint32_t* pair_off, uint32_t ppi, const uint32_t* tile_off, const uint32_t* sel_start,
                                                                  const uint32_t* sel_ct_row, const uint32_t* sel_tk_attr, const rhip_fr* sel_coeff,
                                                                  const rhip_g1* ct_c1, const rhip_g1* ct_c, const rhip_g1* ct_d, const uint32_t* ct_row_off,
                                                                  const uint8_t* line_inf, G1M* P, uint32_t* qref, G1M* terms) {
  __shared__ uint32_t lds[2 * 8 * RB_PAIRS_BLOCK];
  size_t t, item;
  bool active;
  pair_lane((size_t)blockIdx.x * blockDim.x + threadIdx.x, n_items, total_pairs, pair_off, ppi, tile_off, &t, &item, &active);
  const uint32_t j = (uint32_t)(t - pair_off[item]);
  const uint32_t m = pair_off[item + 1] - pair_off[item] - 2;
  G1Aff base = aff_inf<Fp>();
  uint32_t k[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  uint32_t row = 0, line = 1;
  if (j < m) {
    const uint32_t e = sel_start[item] + j;
    row = ct_row_off[item] + sel_ct_row[e];
    base = load_g1(ct_d[row].l);
    ld_scalar(k, sel_coeff + e);
    line = 2 + sel_tk_attr[e];
  } else if (j == m) {
    base = load_g1(ct_c1[item].l);
    line = 0;
  }
  bool p_inf;
  scale_and_store(lds, active && j <= m, base, k, j < m, P + t, &p_inf);
  if (!active) return;
  if (j > m) { qref[t] = line_inf[1] ? RHIP_Q_SKIP : 1u; return; }          // P comes from k_msm_finish_g1 (which may turn the pair into a skip)
  if (j < m) st_g1_q(terms + (t - 2 * item), load_g1(ct_c[row].l));
  qref[t] = (p_inf || line_inf[line]) ? RHIP_Q_SKIP : line;
}
extern "C" int32_t rhip_ghw11_transform_batch(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel, const uint32_t* pair_off,
                                              const uint32_t* sel_start, const uint32_t* sel_ct_row, const uint32_t* sel_tk_attr, const rhip_fr* sel_coeff,
                                              const rhip_g1* ct_c1, const rhip_g1* ct_c, const rhip_g1* ct_d, const uint32_t* ct_row_off,
                                              const rhip_g2_lines* tk_lines, rhip_gt* out) {
  NEED(ctx);
  if (!n_items) return RHIP_OK;
  if (!total_pairs || !pair_off || !tk_lines || max_pairs < 2 || total_pairs < 2 * n_items) return RHIP_ERR_ARG;
  PairLists pl;
  int32_t rc = alloc_pair_lists(ctx, total_pairs, &pl);
  if (rc) return rc;
  const size_t total_terms = total_pairs - 2 * n_items;
  void *w_terms = nullptr, *w_masks = nullptr, *w_part = nullptr, *w_off = nullptr;
  rc = rhip_ensure_work(ctx, 4, (total_terms ? total_terms : 1) * sizeof(G1M), &w_terms);
  if (!rc) rc = rhip_ensure_work(ctx, 5, (n_sel ? n_sel : 1) * 16 * sizeof(uint32_t), &w_masks);
  uint32_t L, C;
  choose_msm_chunks(ctx, n_items, max_pairs - 2, &L, &C);
  if (!rc) rc = rhip_ensure_work(ctx, 6, n_items * L * sizeof(G1JM), &w_part);
  if (!rc) rc = rhip_ensure_work(ctx, 7, (n_items + 1) * sizeof(uint32_t), &w_off);
  if (rc) return rc;
  if (n_sel) KLAUNCH(ctx, "k_naf_masks", k_naf_masks, dim3(blocks_for(n_sel, 256)), dim3(256), 0, ctx->stream, n_sel, sel_coeff, (uint32_t*)w_masks);
  KLAUNCH(ctx, "k_term_off", k_term_off, dim3(blocks_for(n_items + 1, 256)), dim3(256), 0, ctx->stream, n_items, pair_off, (uint32_t*)w_off, 2u);
  const uint32_t ppi = uniform_ppi(n_items, max_pairs, total_pairs);
  size_t g_lanes = 0;
  const uint32_t* tile_off = nullptr;
  if ((rc = gather_lanes(ctx, n_items, max_pairs, total_pairs, pair_off, ppi, &tile_off, &g_lanes)) != RHIP_OK) return rc;
  KLAUNCH(ctx, "k_ghw11_pairs", k_ghw11_pairs, dim3(blocks_for(g_lanes, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_items,
          total_pairs, pair_off, ppi, tile_off, sel_start, sel_ct_row, sel_tk_attr, sel_coeff, ct_c1, ct_c, ct_d, ct_row_off, (const uint8_t*)tk_lines->q_inf, pl.P,
          pl.qref, (G1M*)w_terms);
  KLAUNCH(ctx, "k_msm_partial_g1", (k_msm_partial<Fp, G1M, G1JM>), dim3(blocks_for(n_items * L, 64)), dim3(64), 0, ctx->stream, n_items, L, C,
          (const uint32_t*)w_off, sel_start, (const G1M*)w_terms, (const uint32_t*)w_masks, 1, (G1JM*)w_part);
  KLAUNCH(ctx, "k_msm_finish_g1", k_msm_finish_g1, dim3(blocks_for(n_items, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_items, L,
          (const G1JM*)w_part, pair_off, pl.P, pl.qref);
  return run_pair_lists(ctx, n_items, pair_off, max_pairs, total_pairs, pl, (const LineM*)tk_lines->lines, tk_lines->lines29, (const rhip_gt*)nullptr, out);
}

// ------------------------------------------------------------------------------------------------ AW11 multi-authority CP-ABE
// Public side of aw11::encrypt: gk (g1, g2), E = e(g1, g2) (the constant the reference recomputes per row, aw11/mod.rs:263,274)
// and, for every attribute of every authority in play, (egg_alpha_x, g2*y_x) (Aw11PublicKey, :56-61): fixed bases, so tables.
struct rhip_aw11_pk {
  rhip_ctx* ctx;
  rhip_g2_table* g2;
  rhip_gt_table* E;
  size_t n_attrs;
  GtM* attr_gt;       // [n_attrs][32][255]
  G2M* attr_g2;       // [n_attrs][32][255]
  // 16-bit windows of the same bases, [n_attrs][16][65535] (402 + 134 MB per attribute: 107 GB for 200 attributes -- memory traded
  // for work on a 288 GB part: 16 instead of 32 table entries per power).  Built when the device has the room (rhip_aw11_pk_create).
  GtM* attr_gt16;
  G2M* attr_g216;
  // signed w-bit windows per attribute (the default: RABE_AW11_ATTR_BITS, 10 = 13 312 entries = 6.8 MB per attribute for both bases):
  // sgn_windows(w) instead of 32 table entries per power, the sign of a digit is a conjugation (Gt) / a negated y (G2)
  GtM* attr_gt_s;
  G2M* attr_g2_s;
  int s_bits;
};
extern "C" void rhip_aw11_pk_destroy(rhip_aw11_pk* pk) {
  if (!pk) return;
  rhip_g2_table_destroy(pk->g2);
  rhip_gt_table_destroy(pk->E);
  if (pk->attr_gt) (void)hipFree(pk->attr_gt);
  if (pk->attr_g2) (void)hipFree(pk->attr_g2);
  if (pk->attr_gt16) (void)hipFree(pk->attr_gt16);
  if (pk->attr_g216) (void)hipFree(pk->attr_g216);
  if (pk->attr_gt_s) (void)hipFree(pk->attr_gt_s);
  if (pk->attr_g2_s) (void)hipFree(pk->attr_g2_s);
  delete pk;
}
// signed-window tables from the 8-bit ones: entry (a, i, d-1) = (d << (w i)) * base_a, at most three 8-bit digits
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_attr_tables_gt_signed(size_t n_attrs, int w, const GtM* t8, GtM* out) {
  const size_t per = sgn_entries(w);
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = t < n_attrs * per;
  if (!active) t = n_attrs * per - 1;
  const size_t a = t / per, e = t % per;
  const int i = (int)(e >> (w - 1));
  const uint64_t d = (e & (((size_t)1 << (w - 1)) - 1)) + 1;
  const int b = w * i, word = b >> 5, sh = b & 31;
  const uint64_t lo = d << sh;
  uint32_t m[8];
#pragma unroll
  for (int j = 0; j < 8; j++) m[j] = (j == word) ? (uint32_t)lo : (j == word + 1) ? (uint32_t)(lo >> 32) : 0u;
  bool started = false;
  home_table_pow_gt(started, t8 + a * TBL_WINDOWS * TBL_DIGITS, m);
  const Fp12 v = home_result(started);
  if (active) st_gt_m(out + t, v);
}
__global__ void __launch_bounds__(128, RB_MIN_WAVES) k_attr_tables_g2_signed(size_t n_attrs, int w, const G2M* t8, G2M* out) {
  const size_t per = sgn_entries(w);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_attrs * per) return;
  const size_t a = t / per, e = t % per;
  const int i = (int)(e >> (w - 1));
  const uint64_t d = (e & (((size_t)1 << (w - 1)) - 1)) + 1;
  const int b = w * i, word = b >> 5, sh = b & 31;
  const uint64_t lo = d << sh;
  uint32_t m[8];
#pragma unroll
  for (int j = 0; j < 8; j++) m[j] = (j == word) ? (uint32_t)lo : (j == word + 1) ? (uint32_t)(lo >> 32) : 0u;
  st_g2_m(out + t, jac_to_aff(table_mul_g2(t8 + a * TBL_WINDOWS * TBL_DIGITS, m)));
}
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_attr_tables_gt(size_t n_attrs, const rhip_gt* base, GtM* tbl) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_attrs * TBL_WINDOWS * TBL_DIGITS) return;
  const size_t a = t / (TBL_WINDOWS * TBL_DIGITS), e = t % (TBL_WINDOWS * TBL_DIGITS);
  const int w = (int)(e / TBL_DIGITS);
  const uint32_t d = (uint32_t)(e % TBL_DIGITS) + 1;
  uint32_t k[8];
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = (i == (w >> 2)) ? (d << (8 * (w & 3))) : 0u;
  st_gt_m(tbl + t, gt_pow_window(load_gt(base[a].l), k));
}
__global__ void __launch_bounds__(128, RB_MIN_WAVES) k_attr_tables_g2(size_t n_attrs, const rhip_g2* base, G2M* tbl) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_attrs * TBL_WINDOWS * TBL_DIGITS) return;
  const size_t a = t / (TBL_WINDOWS * TBL_DIGITS), e = t % (TBL_WINDOWS * TBL_DIGITS);
  const int w = (int)(e / TBL_DIGITS);
  const uint32_t d = (uint32_t)(e % TBL_DIGITS) + 1;
  uint32_t k[8];
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = (i == (w >> 2)) ? (d << (8 * (w & 3))) : 0u;
  st_g2_m(tbl + t, jac_to_aff(jac_mul_naf(load_g2(base[a].l), k)));
}
extern "C" int32_t rhip_aw11_pk_create(rhip_ctx* ctx, const rhip_g1* g1, const rhip_g2* g2, size_t n_attrs, const rhip_gt* host_egg_alpha,
                                       const rhip_g2* host_g2_y, rhip_aw11_pk** out) {
  if (!ctx || !g1 || !g2 || !n_attrs || !host_egg_alpha || !host_g2_y || !out) return RHIP_ERR_ARG;
  *out = nullptr;
  rhip_aw11_pk* pk = new rhip_aw11_pk{ctx, nullptr, nullptr, n_attrs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  rhip_gt e;
  int32_t rc = rhip_host_pairing(ctx, g1, g2, &e);
  if (!rc) rc = rhip_g2_table_create(ctx, g2, &pk->g2);
  if (!rc) rc = rhip_g2_table_add_w16(ctx, pk->g2);
  if (!rc) rc = rhip_gt_table_create(ctx, &e, &pk->E);
  if (!rc) rc = rhip_gt_table_add_w16(ctx, pk->E);
  if (rc) { rhip_aw11_pk_destroy(pk); return rc; }
  const size_t per = (size_t)TBL_WINDOWS * TBL_DIGITS;
  rhip_gt* dgt = nullptr;
  rhip_g2* dg2 = nullptr;
  hipError_t he = hipMalloc((void**)&pk->attr_gt, n_attrs * per * sizeof(GtM));
  if (he == hipSuccess) he = hipMalloc((void**)&pk->attr_g2, n_attrs * per * sizeof(G2M));
  if (he == hipSuccess) he = hipMalloc((void**)&dgt, n_attrs * sizeof(rhip_gt));
  if (he == hipSuccess) he = hipMalloc((void**)&dg2, n_attrs * sizeof(rhip_g2));
  if (he == hipSuccess) he = hipMemcpyAsync(dgt, host_egg_alpha, n_attrs * sizeof(rhip_gt), hipMemcpyHostToDevice, ctx->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(dg2, host_g2_y, n_attrs * sizeof(rhip_g2), hipMemcpyHostToDevice, ctx->stream);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(k_attr_tables_gt, dim3(blocks_for(n_attrs * per, 64)), dim3(64), 0, ctx->stream, n_attrs, (const rhip_gt*)dgt, pk->attr_gt);
    hipLaunchKernelGGL(k_attr_tables_g2, dim3(blocks_for(n_attrs * per, 128)), dim3(128), 0, ctx->stream, n_attrs, (const rhip_g2*)dg2, pk->attr_g2);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
  if (dgt) (void)hipFree(dgt);
  if (dg2) (void)hipFree(dg2);
  if (he != hipSuccess) { rhip_aw11_pk_destroy(pk); return fail(ctx, he, "rhip_aw11_pk_create"); }
  // 16-bit windows per attribute are OPT-IN (RABE_AW11_ATTR_W16=1): 536 MB per attribute -- 107 GB at 200 attributes -- is not a
  // side effect a constructor may have on a device other tenants share; without the variable the 8-bit tables (4 MB per
  // attribute) serve.  Even when asked for, they are only built while they leave a quarter of the device memory free.
  const size_t per16 = (size_t)TBL16_WINDOWS * TBL16_DIGITS;
  const size_t need = n_attrs * per16 * (sizeof(GtM) + sizeof(G2M));
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const char* env = getenv("RABE_AW11_ATTR_W16");
  const bool want = env && env[0] == '1' && free_b > need + total_b / 4;
  if (want) {
    he = hipMalloc((void**)&pk->attr_gt16, n_attrs * per16 * sizeof(GtM));
    if (he == hipSuccess) he = hipMalloc((void**)&pk->attr_g216, n_attrs * per16 * sizeof(G2M));
    if (he == hipSuccess) {
      for (size_t a = 0; a < n_attrs && !rc; a++) {
        rc = rhip_build_w16_gt(ctx, pk->attr_gt + a * per, pk->attr_gt16 + a * per16);
        if (!rc) rc = rhip_build_w16_g2(ctx, pk->attr_g2 + a * per, pk->attr_g216 + a * per16);
      }
      if (!rc) he = hipStreamSynchronize(ctx->stream);
    }
    if (he != hipSuccess || rc) {                    // no room after all: the 8-bit tables serve
      (void)hipGetLastError();
      if (pk->attr_gt16) (void)hipFree(pk->attr_gt16);
      if (pk->attr_g216) (void)hipFree(pk->attr_g216);
      pk->attr_gt16 = nullptr;
      pk->attr_g216 = nullptr;
    }
  }
  // signed windows per attribute (when the 16-bit tables were not asked for): RABE_AW11_ATTR_BITS = 8 keeps the plain 8-bit tables,
  // 9 ... 14 builds signed tables of that width if they leave half of the device memory free
  if (!pk->attr_gt16) {
    int w = 10;
    if (const char* eb = getenv("RABE_AW11_ATTR_BITS")) w = atoi(eb);
    const size_t per_s = (w >= 9 && w <= 14) ? sgn_entries(w) : 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (per_s && free_b > n_attrs * per_s * (sizeof(GtM) + sizeof(G2M)) + total_b / 2) {
      he = hipMalloc((void**)&pk->attr_gt_s, n_attrs * per_s * sizeof(GtM));
      if (he == hipSuccess) he = hipMalloc((void**)&pk->attr_g2_s, n_attrs * per_s * sizeof(G2M));
      if (he == hipSuccess) {
        hipLaunchKernelGGL(k_attr_tables_gt_signed, dim3(blocks_for(n_attrs * per_s, 64)), dim3(64), 0, ctx->stream, n_attrs, w, (const GtM*)pk->attr_gt,
                           pk->attr_gt_s);
        hipLaunchKernelGGL(k_attr_tables_g2_signed, dim3(blocks_for(n_attrs * per_s, 128)), dim3(128), 0, ctx->stream, n_attrs, w, (const G2M*)pk->attr_g2,
                           pk->attr_g2_s);
        he = hipGetLastError();
      }
      if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
      if (he != hipSuccess) {
        (void)hipGetLastError();
        if (pk->attr_gt_s) (void)hipFree(pk->attr_gt_s);
        if (pk->attr_g2_s) (void)hipFree(pk->attr_g2_s);
        pk->attr_gt_s = nullptr;
        pk->attr_g2_s = nullptr;
      } else {
        pk->s_bits = w;
      }
    }
  }
  *out = pk;
  return RHIP_OK;
}
// one lane per (item, leaf row): lambda_x = share of s, omega_x = share of 0 (second coefficient set follows the first in
// the item's draw list, aw11/mod.rs:259-260)
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_aw11_enc_scalars(size_t n_items, size_t total_rows, const uint32_t* item_row_off,
                                                                       const uint32_t* item_tree_leaf, const uint32_t* item_tree_gate,
                                                                       const uint32_t* item_n_coef, TreeTables tt, const rhip_fr* s, const rhip_fr* coef,
                                                                       const uint32_t* item_coef_off, rhip_fr* lam, rhip_fr* omg) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_rows) return;
  const size_t item = owner_of(item_row_off, n_items, t);
  const uint32_t leaf = item_tree_leaf[item] + (uint32_t)(t - item_row_off[item]);
  const rhip_fr* c0 = coef + item_coef_off[item];
  store_fr(lam[t].l, share_of_leaf(tt, leaf, item_tree_gate[item], c0, load_fr(s[item].l)));
  store_fr(omg[t].l, share_of_leaf(tt, leaf, item_tree_gate[item], c0 + item_n_coef[item], zero<FrParams>()));
}
// C1[row] = E^lambda * egg_alpha_x^r   (:272-274)
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_aw11_enc_c1(size_t total_rows, const uint32_t* item_row_off, size_t n_items, const uint32_t* item_tree_leaf,
                                                                 const uint32_t* leaf_attr, const GtM* e_tbl, int e_w16, const GtM* attr_tbl,
                                                                 int attr_w16 /* 0: 8-bit, 1: 16-bit, 9..14: signed windows of that width */,
                                                                 const rhip_fr* lam, const rhip_fr* rand, rhip_gt* c1) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_rows) return;
  const size_t item = owner_of(item_row_off, n_items, t);
  const uint32_t a = leaf_attr[item_tree_leaf[item] + (uint32_t)(t - item_row_off[item])];
  uint32_t kl[8], kr[8];
  ld_scalar(kl, lam + t);
  ld_scalar(kr, rand + t);
  bool started = false;       // E^lambda * egg_alpha_x^r as ONE running product on the lane's home value
  if (e_w16) home_table_pow_gt_w16(started, e_tbl, kl); else home_table_pow_gt(started, e_tbl, kl);
  if (attr_w16 > 1) home_table_pow_gt_signed(started, attr_tbl + (size_t)a * sgn_entries(attr_w16), kr, attr_w16);
  else if (attr_w16) home_table_pow_gt_w16(started, attr_tbl + (size_t)a * TBL16_WINDOWS * TBL16_DIGITS, kr);
  else home_table_pow_gt(started, attr_tbl + (size_t)a * TBL_WINDOWS * TBL_DIGITS, kr);
  store_gt(c1[t].l, home_result(started));
}
// C3[row] = (g2*y_x) * r + g2 * omega   (:275-277): two fixed-base sums on one accumulator
__global__ void __launch_bounds__(128, RB_G2_WAVES) k_aw11_enc_c3(size_t total_rows, const uint32_t* item_row_off, size_t n_items, const uint32_t* item_tree_leaf,
                                                                  const uint32_t* leaf_attr, const G2M* g2_tbl, const G2M* attr_tbl, int w16, int g2_is_w16,
                                                                  const rhip_fr* omg, const rhip_fr* rand, rhip_g2* c3) {
  __shared__ uint32_t lds[2 * 8 * 128];
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = t < total_rows;
  if (!active) t = total_rows - 1;
  const size_t item = owner_of(item_row_off, n_items, t);
  const uint32_t a = leaf_attr[item_tree_leaf[item] + (uint32_t)(t - item_row_off[item])];
  uint32_t kr[8], kw[8];
  ld_scalar(kr, rand + t);
  ld_scalar(kw, omg + t);
  G2Jac acc = jac_inf<Fp2>();
  if (w16 > 1) {         // attribute base: signed w16-bit windows; g2: its 16-bit table (g2_tbl) or, without one, its 8-bit table
    acc = table_mul_g2_signed(acc, attr_tbl + (size_t)a * sgn_entries(w16), kr, w16);
    if (g2_is_w16) {
#pragma unroll 1
      for (int w = 0; w < TBL16_WINDOWS; w++) {
        const uint32_t word = word_sel8(kw, w >> 1);
        const uint32_t d = (w & 1) ? (word >> 16) : (word & 0xffffu);
        if (d) acc = jac_add_aff(acc, ld_g2_m(g2_tbl + (size_t)w * TBL16_DIGITS + (d - 1)));
      }
    } else {
#pragma unroll 1
      for (int w = 0; w < TBL_WINDOWS; w++) {
        const uint32_t d = scalar_byte(kw, w);
        if (d) acc = jac_add_aff(acc, ld_g2_m(g2_tbl + w * TBL_DIGITS + (d - 1)));
      }
    }
  } else if (w16) {      // both tables with 16-bit windows: 2 x 16 entries
    const G2M* ta = attr_tbl + (size_t)a * TBL16_WINDOWS * TBL16_DIGITS;
#pragma unroll 1
    for (int w = 0; w < 2 * TBL16_WINDOWS; w++) {
      const int ww = w & (TBL16_WINDOWS - 1);
      const uint32_t word = word_sel8(w < TBL16_WINDOWS ? kr : kw, ww >> 1);
      const uint32_t d = (ww & 1) ? (word >> 16) : (word & 0xffffu);
      if (d) acc = jac_add_aff(acc, ld_g2_m((w < TBL16_WINDOWS ? ta : g2_tbl) + (size_t)ww * TBL16_DIGITS + (d - 1)));
    }
  } else {
    const G2M* ta = attr_tbl + (size_t)a * TBL_WINDOWS * TBL_DIGITS;
#pragma unroll 1
    for (int w = 0; w < 2 * TBL_WINDOWS; w++) {
      const int ww = w & (TBL_WINDOWS - 1);
      const uint32_t d = scalar_byte(w < TBL_WINDOWS ? kr : kw, ww);
      if (d) acc = jac_add_aff(acc, ld_g2_m((w < TBL_WINDOWS ? ta : g2_tbl) + ww * TBL_DIGITS + (d - 1)));
    }
  }
  store_g2_block128(lds, active, c3 + t, acc);
}
extern "C" int32_t rhip_aw11_encrypt_batch(rhip_ctx* ctx, const rhip_aw11_pk* pk, size_t n_items, size_t total_rows, const uint32_t* item_row_off,
                                           const uint32_t* item_tree_leaf, const uint32_t* item_tree_gate, const uint32_t* item_n_coef,
                                           const uint32_t* path_off, const uint32_t* path_gate, const uint32_t* path_x, const uint32_t* gate_k,
                                           const uint32_t* gate_coef_off, const uint32_t* leaf_attr, const rhip_fr* s, const rhip_fr* coef,
                                           const uint32_t* item_coef_off, const rhip_fr* rand, const rhip_gt* msg, rhip_gt* c0, rhip_gt* c1, rhip_g2* c2,
                                           rhip_g2* c3) {
  NEED(ctx);
  if (!pk) return RHIP_ERR_ARG;
  if (!n_items) return RHIP_OK;
  const rhip_gt_table* et = pk->E;
  KLAUNCH(ctx, "k_table_pow_gt_mul", k_table_pow_gt_mul, dim3(blocks_for(n_items, 64)), dim3(64), 0, ctx->stream,
          (const GtM*)(et->dev16 ? et->dev16 : et->dev), et->dev16 ? 1 : 0, n_items, s, msg, c0);
  if (!total_rows) return RHIP_OK;
  void* w = nullptr;
  int32_t rc = rhip_ensure_work(ctx, 4, total_rows * 2 * sizeof(rhip_fr), &w);
  if (rc) return rc;
  rhip_fr* lam = (rhip_fr*)w;
  rhip_fr* omg = lam + total_rows;
  const TreeTables tt{path_off, path_gate, path_x, gate_k, gate_coef_off};
  KLAUNCH(ctx, "k_aw11_enc_scalars", k_aw11_enc_scalars, dim3(blocks_for(total_rows, 256)), dim3(256), 0, ctx->stream, n_items, total_rows, item_row_off,
          item_tree_leaf, item_tree_gate, item_n_coef, tt, s, coef, item_coef_off, lam, omg);
  KLAUNCH(ctx, "k_aw11_enc_c1", k_aw11_enc_c1, dim3(blocks_for(total_rows, 64)), dim3(64), 0, ctx->stream, total_rows, item_row_off, n_items,
          item_tree_leaf, leaf_attr, (const GtM*)(et->dev16 ? et->dev16 : et->dev), et->dev16 ? 1 : 0,
          (const GtM*)(pk->attr_gt16 ? pk->attr_gt16 : pk->attr_gt_s ? pk->attr_gt_s : pk->attr_gt), pk->attr_gt16 ? 1 : pk->attr_gt_s ? pk->s_bits : 0,
          (const rhip_fr*)lam, rand, c1);
  rc = rhip_g2_table_mul(ctx, pk->g2, total_rows, rand, c2);
  if (rc) return rc;
  const bool both16 = pk->attr_g216 && pk->g2->dev16, sgn = !both16 && pk->attr_g2_s;
  KLAUNCH(ctx, "k_aw11_enc_c3", k_aw11_enc_c3, dim3(blocks_for(total_rows, 128)), dim3(128), 0, ctx->stream, total_rows, item_row_off, n_items,
          item_tree_leaf, leaf_attr, (const G2M*)((both16 || sgn) && pk->g2->dev16 ? pk->g2->dev16 : pk->g2->dev),
          (const G2M*)(both16 ? pk->attr_g216 : sgn ? pk->attr_g2_s : pk->attr_g2), both16 ? 1 : sgn ? pk->s_bits : 0, pk->g2->dev16 ? 1 : 0,
          (const rhip_fr*)omg, rand, c3);
  return RHIP_OK;
}
// decrypt (aw11/mod.rs:298-366 restated in SURVEY.md Appendix B.5): item i owns pairs [pair_off[i], pair_off[i+1]) = m_i + 1:
//   s < m : P = c_e * K[key attr],   Q = C2[ct row]
//   m     : P = -H(gid),             Q = sum_e c_e * C3[ct row]   (G2 MSM)
// and the leading factor c_0 * prod_e C1[ct row]^(-c_e)  (a Gt multi-exponentiation with shared squarings).
// This kernel does the scaled pairs and gathers the MSM's bases (C3) and the multi-exponentiation's bases (C1), Montgomery.
__global__ void __launch_bounds__(RB_PAIRS_BLOCK, 2) k_aw11_dec_pairs(size_t n_items, size_t total_pairs, const uint32_t* pair_off, uint32_t ppi, const uint32_t* tile_off, const uint32_t* sel_start,
                                                                     const uint32_t* sel_ct_row, const uint32_t* sel_sk_attr, const rhip_fr* sel_coeff,
                                                                     const rhip_g2* ct_c2, const uint32_t* ct_row_off, const rhip_g1* sk_hash,
                                                                     const rhip_g1* sk_k, const uint32_t* sk_attr_off, const uint32_t* sk_idx, G1M* P,
                                                                     G2M* Q, uint32_t* qref) {
  __shared__ uint32_t lds[2 * 8 * RB_PAIRS_BLOCK];
  size_t t, item;
  bool active;
  pair_lane((size_t)blockIdx.x * blockDim.x + threadIdx.x, n_items, total_pairs, pair_off, ppi, tile_off, &t, &item, &active);
  const uint32_t j = (uint32_t)(t - pair_off[item]);
  const uint32_t m = pair_off[item + 1] - pair_off[item] - 1;
  const uint32_t sk = sk_idx ? sk_idx[item] : (uint32_t)item;
  const bool last = (j == m);
  G1Aff base;
  uint32_t k[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  uint32_t row = 0;
  if (last) {
    base = load_g1(sk_hash[sk].l);
  } else {
    const uint32_t e = sel_start[item] + j;
    row = ct_row_off[item] + sel_ct_row[e];
    base = load_g1(sk_k[sk_attr_off[sk] + sel_sk_attr[e]].l);
    ld_scalar(k, sel_coeff + e);
  }
  bool p_inf;
  scale_and_store(lds, active, base, k, last, P + t, &p_inf);
  if (!active) return;
  if (last) { qref[t] = p_inf ? RHIP_Q_SKIP : RHIP_Q_WALK; return; }      // Q comes from k_msm_finish_g2 (which may turn the pair into a skip)
  const G2Aff q = load_g2(ct_c2[row].l);
  const bool skip = p_inf || aff_is_inf(q);
  if (!skip) st_g2_q(Q + t, q);
  qref[t] = skip ? RHIP_Q_SKIP : RHIP_Q_WALK;
}
// lane per term: C3 and C1 of the selected ciphertext row in Montgomery form (term index = pair index - item)
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_aw11_gather_terms(size_t n_items, size_t total_terms, const uint32_t* term_off, const uint32_t* sel_start,
                                                                       const uint32_t* sel_ct_row, const uint32_t* ct_row_off, const rhip_gt* ct_c1,
                                                                       const rhip_g2* ct_c3, G2M* t_c3, GtM* t_c1) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_terms) return;
  const size_t item = owner_of(term_off, n_items, t);
  const uint32_t e = sel_start[item] + (uint32_t)(t - term_off[item]);
  const uint32_t row = ct_row_off[item] + sel_ct_row[e];
  st_g2_q(t_c3 + t, load_g2(ct_c3[row].l));
  st_gt_m(t_c1 + t, load_gt(ct_c1[row].l));
}
// lane t = chunk * n_items + item: prod over the chunk's terms of base^(-c) with the squarings shared (Straus over the NAF
// masks; an inverse in Gt is a conjugation)
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_gt_multiexp_partial(size_t n_items, uint32_t L, uint32_t C, const uint32_t* term_off,
                                                                         const uint32_t* sel_start, const GtM* bases, const uint32_t* masks, int flip,
                                                                         GtM* part) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_items * L) return;
  const size_t c = t / n_items, item = t % n_items;
  const uint32_t lo = term_off[item], hi = term_off[item + 1];
  const uint64_t first = (uint64_t)lo + (uint64_t)c * C;
  int cnt = 0;
  if (first < hi) cnt = (int)((hi - first < C) ? (hi - first) : C);
  const GtM* b = bases + first;
  const uint32_t* mk = masks + 16 * ((size_t)sel_start[item] + c * C);
  Fp12 acc = fp12_one();
  bool started = false;
  for (int w = 7; w >= 0; w--) {
    for (int bit = 31; bit >= 0; bit--) {
      if (started) acc = fp12_cyclotomic_sqr(acc);
      for (int j = 0; j < cnt; j++) {
        const uint32_t pw = mk[16 * j + (flip ? 8 : 0) + w], nw = mk[16 * j + (flip ? 0 : 8) + w];
        const uint32_t pb = (pw >> bit) & 1u, nb = (nw >> bit) & 1u;
        if (pb | nb) {
          Fp12 x = ld_gt_m(b + j);
          if (nb) x = fp12_conj(x);
          acc = started ? fp12_mul(acc, x) : x;
          started = true;
        }
      }
    }
  }
  st_gt_m(part + item * L + c, acc);
}
// lead[i] = c_0[i] * prod_c part[i][c], canonical (the mul_in of the final-exponentiation kernel)
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_gt_lead(size_t n_items, uint32_t L, const rhip_gt* c0, const GtM* part, rhip_gt* lead) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  Fp12 acc = load_gt(c0[i].l);
  for (uint32_t c = 0; c < L; c++) acc = fp12_mul_fn(acc, ld_gt_m(part + i * L + c));
  store_gt(lead[i].l, acc);
}
extern "C" int32_t rhip_aw11_decrypt_batch(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t total_pairs, size_t n_sel, const uint32_t* pair_off,
                                           const uint32_t* sel_start, const uint32_t* sel_ct_row, const uint32_t* sel_sk_attr, const rhip_fr* sel_coeff,
                                           const rhip_gt* ct_c0, const rhip_gt* ct_c1, const rhip_g2* ct_c2, const rhip_g2* ct_c3,
                                           const uint32_t* ct_row_off, const rhip_g1* sk_hash, const rhip_g1* sk_k, const uint32_t* sk_attr_off,
                                           const uint32_t* sk_idx, rhip_gt* out) {
  NEED(ctx);
  if (!n_items) return RHIP_OK;
  if (!total_pairs || !pair_off || !n_sel) return RHIP_ERR_ARG;
  PairLists pl;
  int32_t rc = alloc_pair_lists(ctx, total_pairs, &pl);
  if (rc) return rc;
  const size_t total_terms = total_pairs - n_items;
  uint32_t L, C;
  choose_msm_chunks(ctx, n_items, max_pairs - 1, &L, &C);
  // arena 4: [C3 terms (G2M)][C1 terms (GtM)][lead (rhip_gt per item)];  5: masks;  6: [G2 partials][Gt partials];  7: term offsets
  void *w4 = nullptr, *w_masks = nullptr, *w6 = nullptr, *w_off = nullptr;
  const size_t nt = total_terms ? total_terms : 1;
  rc = rhip_ensure_work(ctx, 4, nt * (sizeof(G2M) + sizeof(GtM)) + n_items * sizeof(rhip_gt), &w4);
  if (!rc) rc = rhip_ensure_work(ctx, 5, n_sel * 16 * sizeof(uint32_t), &w_masks);
  if (!rc) rc = rhip_ensure_work(ctx, 6, n_items * L * (sizeof(G2JM) + sizeof(GtM)), &w6);
  if (!rc) rc = rhip_ensure_work(ctx, 7, (n_items + 1) * sizeof(uint32_t), &w_off);
  if (rc) return rc;
  G2M* t_c3 = (G2M*)w4;
  GtM* t_c1 = (GtM*)(t_c3 + nt);
  rhip_gt* lead = (rhip_gt*)(t_c1 + nt);
  G2JM* p_g2 = (G2JM*)w6;
  GtM* p_gt = (GtM*)(p_g2 + n_items * L);
  KLAUNCH(ctx, "k_naf_masks", k_naf_masks, dim3(blocks_for(n_sel, 256)), dim3(256), 0, ctx->stream, n_sel, sel_coeff, (uint32_t*)w_masks);
  KLAUNCH(ctx, "k_term_off", k_term_off, dim3(blocks_for(n_items + 1, 256)), dim3(256), 0, ctx->stream, n_items, pair_off, (uint32_t*)w_off);
  const uint32_t ppi = uniform_ppi(n_items, max_pairs, total_pairs);
  size_t g_lanes = 0;
  const uint32_t* tile_off = nullptr;
  if ((rc = gather_lanes(ctx, n_items, max_pairs, total_pairs, pair_off, ppi, &tile_off, &g_lanes)) != RHIP_OK) return rc;
  KLAUNCH(ctx, "k_aw11_dec_pairs", k_aw11_dec_pairs, dim3(blocks_for(g_lanes, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_items,
          total_pairs, pair_off, ppi, tile_off, sel_start, sel_ct_row, sel_sk_attr, sel_coeff, ct_c2, ct_row_off, sk_hash, sk_k, sk_attr_off, sk_idx, pl.P, pl.Q,
          pl.qref);
  if (total_terms)
    KLAUNCH(ctx, "k_aw11_gather_terms", k_aw11_gather_terms, dim3(blocks_for(total_terms, 64)), dim3(64), 0, ctx->stream, n_items, total_terms,
            (const uint32_t*)w_off, sel_start, sel_ct_row, ct_row_off, ct_c1, ct_c3, t_c3, t_c1);
  KLAUNCH(ctx, "k_msm_partial_g2", (k_msm_partial<Fp2, G2M, G2JM>), dim3(blocks_for(n_items * L, 64)), dim3(64), 0, ctx->stream, n_items, L, C,
          (const uint32_t*)w_off, sel_start, (const G2M*)t_c3, (const uint32_t*)w_masks, 0, p_g2);
  KLAUNCH(ctx, "k_msm_finish_g2", k_msm_finish_g2, dim3(blocks_for(n_items, 128)), dim3(128), 0, ctx->stream, n_items, L, (const G2JM*)p_g2, pair_off,
          pl.Q, pl.qref);
  KLAUNCH(ctx, "k_gt_multiexp_partial", k_gt_multiexp_partial, dim3(blocks_for(n_items * L, 64)), dim3(64), 0, ctx->stream, n_items, L, C,
          (const uint32_t*)w_off, sel_start, (const GtM*)t_c1, (const uint32_t*)w_masks, 1, p_gt);
  KLAUNCH(ctx, "k_gt_lead", k_gt_lead, dim3(blocks_for(n_items, 64)), dim3(64), 0, ctx->stream, n_items, L, ct_c0, (const GtM*)p_gt, lead);
  return run_pair_lists(ctx, n_items, pair_off, max_pairs, total_pairs, pl, (const LineM*)nullptr, (const void*)nullptr, (const rhip_gt*)lead, out);
}

// ------------------------------------------------------------------------------------------------ membership of decoded elements
// What a decoder has to establish before an untrusted key / ciphertext reaches the pairing kernels (rabe-bn's decoding
// raises FieldError::NotMember, src/error.rs:66): G1 has cofactor 1 (on-curve is enough: rhip_g1_on_curve); the twist has
// a large cofactor, so G2 needs r * P = O; a Gt value has to lie in the order-r subgroup -- the engine's Gt powers use
// cyclotomic squarings and conjugation-as-inverse, which are only right there.
// psi = twist^-1 o Frobenius o twist on Jacobian coordinates of the twist: (conj(X) gamma1_2, conj(Y) gamma1_3, conj(Z))
__device__ __forceinline__ G2Jac g2_psi_jac(const G2Jac& q) { return G2Jac{fp2_mul(fp2_conj(q.x), gamma1_2()), fp2_mul(fp2_conj(q.y), gamma1_3()), fp2_conj(q.z)}; }
// X1 Z2^2 == X2 Z1^2 and Y1 Z2^3 == Y2 Z1^3 (both finite), or both infinite
__device__ __noinline__ bool g2_jac_eq(const G2Jac& a, const G2Jac& b) {
  const bool ia = jac_is_inf(a), ib = jac_is_inf(b);
  if (ia || ib) return ia && ib;
  const Fp2 za2 = fp2_sqr(a.z), zb2 = fp2_sqr(b.z);
  if (!fp2_eq(fp2_mul(a.x, zb2), fp2_mul(b.x, za2))) return false;
  return fp2_eq(fp2_mul(a.y, fp2_mul(zb2, b.z)), fp2_mul(b.y, fp2_mul(za2, a.z)));
}
// Membership in G2 = the r-torsion of the twist.  mode 0: the BN test of El Housni / Guillevic / Piellard ("Co-factor clearing and
// subgroup membership testing on pairing-friendly curves", 2022): for a point Q of the twist,
//   Q in G2  <=>  [u+1]Q + psi([u]Q) + psi^2([u]Q) = psi^3([2u]Q)
// -- ONE multiplication by the 63-bit curve parameter u instead of the 254-bit order (7 k -> ~2.2 k field multiplications);
// mode 1: the definition, r * Q = O.  tests/test_gpu_validation.py runs both on members and on cofactor-torsion points.
__global__ void __launch_bounds__(128, RB_MIN_WAVES) k_g2_in_subgroup(size_t n, const rhip_g2* p_all, uint32_t* ok, int mode, const uint32_t* idx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const rhip_g2* pe = p_all + (idx ? (size_t)idx[i] : i);          // idx: the verdict of element idx[i] goes to ok[i]
  const G2Aff P = load_g2(pe->l);
  bool good = wire_words_canonical(pe->l, 4) && aff_on_curve(P);
  if (good && !aff_is_inf(P)) {                       // infinity is a member
    if (mode == 1) {
      uint32_t r[8];
#pragma unroll
      for (int w = 0; w < 8; w++) r[w] = FrParams::mod(w);
      good = jac_is_inf(jac_mul_naf(P, r));
    } else {
      const uint32_t u[8] = {0x4A6909F1u, 0x44E992B4u, 0, 0, 0, 0, 0, 0};          // u = 4965661367192848881
      const G2Jac uq = jac_mul_naf(P, u);
      const G2Jac p1 = g2_psi_jac(uq);
      const G2Jac p2 = g2_psi_jac(p1);
      const G2Jac lhs = jac_add(jac_add(jac_add_aff(uq, P), p1), p2);
      const G2Jac rhs = g2_psi_jac(g2_psi_jac(g2_psi_jac(jac_dbl(uq))));
      good = g2_jac_eq(lhs, rhs);
    }
  }
  ok[i] = good ? 1u : 0u;
}
// Membership in Gt = the order-r subgroup of Fq12*.  First the cyclotomic subgroup, f^(p^4 - p^2 + 1) = 1 <=> f^(p^4) f = f^(p^2)
// (Frobenius maps only; cyclotomic squarings and "conjugate = inverse" are valid from here on).  Then order r -- mode 0: the BN test
// f^p = f^(6u^2) (on Gt the Frobenius is the power p = t - 1 = 6u^2 mod r; El Housni / Guillevic / Piellard 2022): two powers by the
// 63-bit u and two squarings instead of a 254-bit power (8.6 k -> ~4.8 k field multiplications); mode 1: the definition f^r = 1.
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_gt_is_member(size_t n, const rhip_gt* a, uint32_t* ok, int mode) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fp12 f = load_gt(a[i].l);
  const Fp12 f2 = fp12_frob_fn(f, 2);
  const Fp12 f4 = fp12_frob_fn(f2, 2);
  // 0 satisfies both Frobenius identities (every map here sends 0 to 0) but is not a unit: rejected first, as f^r = 1 rejects it
  bool good = wire_words_canonical(a[i].l, 12) && !fp12_is_zero(f) && fp12_eq(fp12_mul_fn(f4, f), f2);
  if (good) {
    if (mode == 1) {
      uint32_t k[8];
#pragma unroll
      for (int w = 0; w < 8; w++) k[w] = FrParams::mod(w);
      k[0] -= 1u;                                       // r is odd
      good = fp12_eq(fp12_mul_fn(gt_pow_window(f, k), f), fp12_one());
    } else {
      const Fp12 h = fp12_cyclotomic_exp_u(fp12_cyclotomic_exp_u(f));          // f^(u^2)
      const Fp12 h2 = fp12_cyclotomic_sqr_fn(h);
      good = fp12_eq(fp12_mul_fn(h2, fp12_cyclotomic_sqr_fn(h2)), fp12_frob_fn(f, 1));          // f^(6 u^2) == f^p
    }
  }
  ok[i] = good ? 1u : 0u;
}
extern "C" int32_t rhip_ctx_collect_walk_verdicts(rhip_ctx* ctx, uint32_t* dev_fail, uint32_t* dev_count) {
  NEED(ctx);
  if ((dev_fail == nullptr) != (dev_count == nullptr)) return RHIP_ERR_ARG;
  ctx->walk_fail = dev_fail;
  ctx->walk_count = dev_count;
  return RHIP_OK;
}
extern "C" int32_t rhip_g2_in_subgroup_at(rhip_ctx* ctx, size_t n_idx, const uint32_t* idx, const rhip_g2* p, uint32_t* ok) {
  NEED(ctx);
  if (!n_idx) return RHIP_OK;
  if (!idx) return RHIP_ERR_ARG;
  KLAUNCH(ctx, "k_g2_in_subgroup", k_g2_in_subgroup, dim3(blocks_for(n_idx, 128)), dim3(128), 0, ctx->stream, n_idx, p, ok, 0, idx);
  return RHIP_OK;
}
extern "C" int32_t rhip_g2_in_subgroup(rhip_ctx* ctx, size_t n, const rhip_g2* p, uint32_t* ok) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  static const int mode = getenv("RABE_G2_CHECK_BY_ORDER") ? 1 : 0;          // the defining test r * Q = O, for A/B runs
  KLAUNCH(ctx, "k_g2_in_subgroup", k_g2_in_subgroup, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream, n, p, ok, mode, (const uint32_t*)nullptr);
  return RHIP_OK;
}
// the same verdicts by the definition (r * Q = O): the reference the fast test is checked against
extern "C" int32_t rhip_g2_in_subgroup_by_order(rhip_ctx* ctx, size_t n, const rhip_g2* p, uint32_t* ok) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_g2_in_subgroup", k_g2_in_subgroup, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream, n, p, ok, 1, (const uint32_t*)nullptr);
  return RHIP_OK;
}
extern "C" int32_t rhip_gt_is_member(rhip_ctx* ctx, size_t n, const rhip_gt* a, uint32_t* ok) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  static const int mode = getenv("RABE_GT_CHECK_BY_ORDER") ? 1 : 0;
  // up to ~30 k elements the six-lane form of the same test (engine_coop.hip: k_gt_is_member_c6) is ahead: 0.52 M instructions per wave
  // of ten elements against one lane's 4.8 k-multiplication chain
  if (mode == 0 && rhip_use_c6_gt_pow(ctx, n)) return rhip_launch_gt_is_member_c6(ctx, n, a, ok);
  KLAUNCH(ctx, "k_gt_is_member", k_gt_is_member, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, n, a, ok, mode);
  return RHIP_OK;
}
extern "C" int32_t rhip_gt_is_member_by_order(rhip_ctx* ctx, size_t n, const rhip_gt* a, uint32_t* ok) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_gt_is_member", k_gt_is_member, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, n, a, ok, 1);
  return RHIP_OK;
}
// the verdicts of a batch's elements folded per item on the device: out[s] = AND of flags[scale * seg_off[s] .. scale * seg_off[s + 1]).
// A 131 072-item AC17 batch has 19.7 M row elements: their flags are 79 MB that nobody needs on the host.
__global__ void __launch_bounds__(256) k_flags_all(size_t n_seg, const uint32_t* seg_off, uint32_t scale, const uint32_t* flags, uint32_t* out) {
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  uint32_t ok = 1;
  const size_t hi = (size_t)scale * seg_off[s + 1];
  for (size_t i = (size_t)scale * seg_off[s]; i < hi; i++) ok &= (flags[i] != 0) ? 1u : 0u;
  out[s] = ok;
}
extern "C" int32_t rhip_flags_all(rhip_ctx* ctx, size_t n_seg, const uint32_t* seg_off, uint32_t scale, const uint32_t* flags, uint32_t* out) {
  NEED(ctx);
  if (!n_seg) return RHIP_OK;
  if (!seg_off || !flags || !out || !scale) return RHIP_ERR_ARG;
  KLAUNCH(ctx, "k_flags_all", k_flags_all, dim3(blocks_for(n_seg, 256)), dim3(256), 0, ctx->stream, n_seg, seg_off, scale, flags, out);
  return RHIP_OK;
}

// ------------------------------------------------------------------------------------------------ generic pairing jobs
// The shape every decrypt of the host layer plans to (rabe_amd/csrc/host/schemes.cpp: PairingJob), device-resident:
//   out[i] = lead[i] * FE( prod_{j in pairs(i)} ML(scal_j * base_j, q_j)  *  ML( sum_{t in terms(i)} s_scal_t * s_base_t, s_q[i] ) )
// One launch set for the whole batch: NAF scaling of the G1 arguments, the shared-doubling sum, all pairs of an item on a few
// accumulators, one final exponentiation per item.  sum_off == NULL: no summed pair.  lead == NULL: no leading factor.
__global__ void __launch_bounds__(RB_PAIRS_BLOCK, 2) k_jobs_pairs(size_t n_items, size_t total, const uint32_t* cpair_off, const uint32_t* pair_off,
                                                                 const rhip_g1* base, const rhip_fr* scal, const rhip_g2* q, const rhip_g2* s_q,
                                                                 G1M* P, G2M* Q, uint32_t* qref) {
  __shared__ uint32_t lds[2 * 8 * RB_PAIRS_BLOCK];
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = t < total;
  if (!active) t = total - 1;
  const size_t item = owner_of(cpair_off, n_items, t);
  const uint32_t j = (uint32_t)(t - cpair_off[item]);
  const uint32_t np = pair_off[item + 1] - pair_off[item];
  const bool is_sum = (j >= np);
  G1Aff b = aff_inf<Fp>();
  uint32_t k[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  const rhip_g2* qs = s_q + item;
  if (!is_sum) {
    const size_t src = (size_t)pair_off[item] + j;
    b = load_g1(base[src].l);
    if (scal) ld_scalar(k, scal + src);
    qs = q + src;
  }
  bool p_inf;
  scale_and_store(lds, active && !is_sum, b, k, false, P + t, &p_inf);
  if (!active) return;
  const G2Aff qq = load_g2(qs->l);
  const bool skip = (!is_sum && p_inf) || aff_is_inf(qq);          // the sum pair's P comes from k_msm_finish_g1 (which may skip it)
  if (!skip) st_g2_q(Q + t, qq);
  qref[t] = skip ? RHIP_Q_SKIP : RHIP_Q_WALK;
}
// combined pair offsets: item i owns pair_off[i+1] - pair_off[i] plain pairs plus (with_sum) one summed pair
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_cpair_off(size_t n_items, const uint32_t* pair_off, int with_sum, uint32_t* cpair_off) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_items) cpair_off[i] = pair_off[i] + (with_sum ? (uint32_t)i : 0u);
}
// Montgomery copies of the sum's bases
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_g1_to_mont(size_t n, const rhip_g1* in, G1M* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st_g1_q(out + i, load_g1(in[i].l));
}
extern "C" int32_t rhip_pairing_jobs(rhip_ctx* ctx, size_t n_items, size_t max_pairs, size_t n_pairs, const uint32_t* pair_off, const rhip_g1* base,
                                     const rhip_fr* scal, const rhip_g2* q, size_t max_terms, size_t n_terms, const uint32_t* sum_off,
                                     const rhip_g1* s_base, const rhip_fr* s_scal, const rhip_g2* s_q, const rhip_gt* lead, rhip_gt* out) {
  NEED(ctx);
  if (!n_items) return RHIP_OK;
  if (!pair_off) return RHIP_ERR_ARG;
  const bool with_sum = sum_off != nullptr;
  const size_t total = n_pairs + (with_sum ? n_items : 0);
  if (!total) return RHIP_ERR_ARG;
  PairLists pl;
  int32_t rc = alloc_pair_lists(ctx, total, &pl);
  if (rc) return rc;
  void *w_terms = nullptr, *w_masks = nullptr, *w_part = nullptr, *w_off = nullptr;
  uint32_t L = 1, C = 1;
  if (with_sum) choose_msm_chunks(ctx, n_items, max_terms ? max_terms : 1, &L, &C);
  rc = rhip_ensure_work(ctx, 7, (n_items + 1) * sizeof(uint32_t), &w_off);
  if (!rc && with_sum) rc = rhip_ensure_work(ctx, 4, (n_terms ? n_terms : 1) * sizeof(G1M), &w_terms);
  if (!rc && with_sum) rc = rhip_ensure_work(ctx, 5, (n_terms ? n_terms : 1) * 16 * sizeof(uint32_t), &w_masks);
  if (!rc && with_sum) rc = rhip_ensure_work(ctx, 6, n_items * L * sizeof(G1JM), &w_part);
  if (rc) return rc;
  uint32_t* cpo = (uint32_t*)w_off;
  KLAUNCH(ctx, "k_cpair_off", k_cpair_off, dim3(blocks_for(n_items + 1, 256)), dim3(256), 0, ctx->stream, n_items, pair_off, with_sum ? 1 : 0, cpo);
  KLAUNCH(ctx, "k_jobs_pairs", k_jobs_pairs, dim3(blocks_for(total, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_items, total,
          (const uint32_t*)cpo, pair_off, base, scal, q, s_q, pl.P, pl.Q, pl.qref);
  if (with_sum) {
    if (n_terms) {
      KLAUNCH(ctx, "k_naf_masks", k_naf_masks, dim3(blocks_for(n_terms, 256)), dim3(256), 0, ctx->stream, n_terms, s_scal, (uint32_t*)w_masks);
      KLAUNCH(ctx, "k_g1_to_mont", k_g1_to_mont, dim3(blocks_for(n_terms, 256)), dim3(256), 0, ctx->stream, n_terms, s_base, (G1M*)w_terms);
    }
    // every item's terms start at sum_off[i] in the term arrays AND in the mask array (one mask per term)
    KLAUNCH(ctx, "k_msm_partial_g1", (k_msm_partial<Fp, G1M, G1JM>), dim3(blocks_for(n_items * L, 64)), dim3(64), 0, ctx->stream, n_items, L, C,
            sum_off, sum_off, (const G1M*)w_terms, (const uint32_t*)w_masks, 0, (G1JM*)w_part);
    KLAUNCH(ctx, "k_msm_finish_g1", k_msm_finish_g1, dim3(blocks_for(n_items, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_items, L,
            (const G1JM*)w_part, (const uint32_t*)cpo, pl.P, pl.qref);
  }
  return run_pair_lists(ctx, n_items, (const uint32_t*)cpo, max_pairs + (with_sum ? 1 : 0), total, pl, (const LineM*)nullptr, (const void*)nullptr, lead, out);
}


// ------------------------------------------------------------------------------------------------ AC17 decrypt on shared accumulators
// ac17::cp_decrypt (src/schemes/ac17/mod.rs:385-430) as a pair list: item i owns six pairs, couple j < 3 =
//   6i + 2j     : P =  sum_{x in ct_sel} C[x][j],                Q = k_0[j]   (the key: prepared lines when the key was prepared)
//   6i + 2j + 1 : P = -(k_p[j] + sum_{x in sk_sel} K[x][j]),     Q = c_0[j]
// so that ALL six pairings of an item can run on one Fq12 accumulator (one squaring per doubling step instead of three with a lane
// per couple, six with a lane per pairing) whenever the launch is large enough to fill the chip that way; small launches are
// chunked into couples by the same rounds x lane-time model as every other pair list.
__global__ void __launch_bounds__(RB_PAIRS_BLOCK, 2) k_ac17_dec_pairs(size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c, const uint32_t* ct_row_off,
                                                                     const rhip_g2* sk_k0, const uint8_t* sk_qinf, int prepared, const rhip_g1* sk_k,
                                                                     const uint32_t* sk_row_off, const rhip_g1* sk_kp, const uint32_t* sk_idx,
                                                                     const uint32_t* ct_sel, const uint32_t* ct_sel_off, const uint32_t* sk_sel,
                                                                     const uint32_t* sk_sel_off, G1M* P, G2M* Q, uint32_t* qref) {
  __shared__ uint32_t lds[2 * 8 * RB_PAIRS_BLOCK];
  // a WAVE holds one side of 64 couples (the two sides run different loops: interleaved in a wave they would execute one after the
  // other): launch lane v -> couple u = 64 (v / 128) + v % 64, side = (v / 64) & 1, pair t = 2 u + side = 6 item + 2 j + side
  static_assert(RB_PAIRS_BLOCK % 128 == 0, "a block holds whole (side 0, side 1) wave pairs");
  const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = n_items * 6;
  size_t t = 2 * ((v >> 7) * 64 + (v & 63)) + ((v >> 6) & 1);
  const bool active = t < total;
  if (!active) t = total - 2 + (t & 1);
  const size_t item = t / 6;
  const int j = (int)((t % 6) >> 1), side = (int)(t & 1);
  const uint32_t sk = sk_idx[item];
  G1Jac acc = jac_inf<Fp>();
  if (side == 0) {
    const uint32_t base = ct_row_off[item];
    for (uint32_t x = ct_sel_off[item]; x < ct_sel_off[item + 1]; x++)
      acc = jac_madd_fast(acc, load_g1(ct_c[(size_t)(base + ct_sel[x]) * 3 + j].l));
  } else {
    const uint32_t base = sk_row_off[sk];
    acc = aff_to_jac(load_g1(sk_kp[(size_t)sk * 3 + j].l));
    for (uint32_t x = sk_sel_off[item]; x < sk_sel_off[item + 1]; x++)
      acc = jac_madd_fast(acc, load_g1(sk_k[(size_t)(base + sk_sel[x]) * 3 + j].l));
    acc = jac_neg(acc);
  }
  const bool p_inf = !active || jac_is_inf(acc);
  const Fp zinv = block_batch_inverse_n<RB_PAIRS_BLOCK>(lds, p_inf ? one<FpParams>() : acc.z);
  if (!active) return;
  if (!p_inf) st_g1_q(P + t, jac_to_aff_with_zinv(acc, zinv));
  if (side == 0 && prepared) {
    const uint32_t line = sk * 3 + (uint32_t)j;
    qref[t] = (p_inf || sk_qinf[line]) ? RHIP_Q_SKIP : line;
    return;
  }
  const G2Aff q = load_g2((side == 0 ? sk_k0 + (size_t)sk * 3 + j : ct_c0 + item * 3 + j)->l);
  const bool skip = p_inf || aff_is_inf(q);
  if (!skip) st_g2_q(Q + t, q);
  qref[t] = skip ? RHIP_Q_SKIP : RHIP_Q_WALK;
}
static int32_t ac17_decrypt_shared(rhip_ctx* ctx, size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c, const uint32_t* ct_row_off,
                                   const rhip_gt* ct_cp, const rhip_g2* sk_k0, const rhip_ac17_sk_lines* sk_lines, const rhip_g1* sk_k,
                                   const uint32_t* sk_row_off, const rhip_g1* sk_kp, const uint32_t* sk_idx, const uint32_t* ct_sel,
                                   const uint32_t* ct_sel_off, const uint32_t* sk_sel, const uint32_t* sk_sel_off, rhip_gt* out) {
  PairLists pl;
  int32_t rc = alloc_pair_lists(ctx, n_items * 6, &pl);
  if (rc) return rc;
  // lanes: 128 per 64 couples (see the kernel), i.e. the couple count rounded up to whole waves, times two
  KLAUNCH(ctx, "k_ac17_dec_pairs", k_ac17_dec_pairs, dim3(blocks_for((n_items * 3 + 63) / 64 * 128, RB_PAIRS_BLOCK)), dim3(RB_PAIRS_BLOCK), 0, ctx->stream, n_items, ct_c0, ct_c,
          ct_row_off, sk_k0, (const uint8_t*)(sk_lines ? sk_lines->q_inf : nullptr), sk_lines ? 1 : 0, sk_k, sk_row_off, sk_kp, sk_idx, ct_sel, ct_sel_off,
          sk_sel, sk_sel_off, pl.P, pl.Q, pl.qref);
  return run_pair_lists(ctx, n_items, (const uint32_t*)nullptr, 6, 0, pl, sk_lines ? (const LineM*)sk_lines->lines : (const LineM*)nullptr, sk_lines ? sk_lines->lines29 : nullptr, ct_cp, out);
}
// which AC17 decrypt path a launch takes: 0 = shared accumulators (this file), 1 = one lane per pairing / couple (engine.hip).
// RABE_AC17_DEC_PATH overrides for A/B runs.  Small launches keep the pairwise kernels (their three-lane form is latency-optimised).
static int ac17_dec_path(const rhip_ctx* ctx, size_t n_items) {
  static const int forced = getenv("RABE_AC17_DEC_PATH") ? atoi(getenv("RABE_AC17_DEC_PATH")) : -1;
  if (forced >= 0) return forced;
  if (rhip_use_c6(ctx, n_items, 6)) return 0;            // small launches: six lanes per accumulator (engine_coop.hip) beat three lanes per pairing
  return rhip_use_c3(ctx, n_items * 6) ? 1 : 0;
}
extern "C" int32_t rhip_ac17_cp_decrypt_batch(rhip_ctx* ctx, size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c, const uint32_t* ct_row_off,
                                              const rhip_gt* ct_cp, const rhip_g2* sk_k0, const rhip_g1* sk_k, const uint32_t* sk_row_off,
                                              const rhip_g1* sk_kp, const uint32_t* sk_idx, const uint32_t* ct_sel, const uint32_t* ct_sel_off,
                                              const uint32_t* sk_sel, const uint32_t* sk_sel_off, rhip_gt* out) {
  NEED(ctx);
  if (!n_items) return RHIP_OK;
  if (ac17_dec_path(ctx, n_items) == 1)
    return rhip_ac17_cp_decrypt_batch_lanes6(ctx, n_items, ct_c0, ct_c, ct_row_off, ct_cp, sk_k0, sk_k, sk_row_off, sk_kp, sk_idx, ct_sel, ct_sel_off, sk_sel,
                                             sk_sel_off, out);
  return ac17_decrypt_shared(ctx, n_items, ct_c0, ct_c, ct_row_off, ct_cp, sk_k0, nullptr, sk_k, sk_row_off, sk_kp, sk_idx, ct_sel, ct_sel_off, sk_sel,
                             sk_sel_off, out);
}
extern "C" int32_t rhip_ac17_cp_decrypt_batch_prepared(rhip_ctx* ctx, size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c,
                                                       const uint32_t* ct_row_off, const rhip_gt* ct_cp, const rhip_ac17_sk_lines* sk_lines,
                                                       const rhip_g1* sk_k, const uint32_t* sk_row_off, const rhip_g1* sk_kp, const uint32_t* sk_idx,
                                                       const uint32_t* ct_sel, const uint32_t* ct_sel_off, const uint32_t* sk_sel,
                                                       const uint32_t* sk_sel_off, rhip_gt* out) {
  NEED(ctx);
  if (!sk_lines) return RHIP_ERR_ARG;
  if (!n_items) return RHIP_OK;
  static const int forced = getenv("RABE_AC17_DEC_PATH") ? atoi(getenv("RABE_AC17_DEC_PATH")) : -1;
  if (forced == 1)
    return rhip_ac17_cp_decrypt_batch_prepared_lanes3(ctx, n_items, ct_c0, ct_c, ct_row_off, ct_cp, sk_lines, sk_k, sk_row_off, sk_kp, sk_idx, ct_sel,
                                                      ct_sel_off, sk_sel, sk_sel_off, out);
  return ac17_decrypt_shared(ctx, n_items, ct_c0, ct_c, ct_row_off, ct_cp, nullptr, sk_lines, sk_k, sk_row_off, sk_kp, sk_idx, ct_sel, ct_sel_off, sk_sel,
                             sk_sel_off, out);
}
