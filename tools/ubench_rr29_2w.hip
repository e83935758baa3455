// Micro-benchmark for VERDICT round 5, item 2 ("one item on TWO lanes, two waves per SIMD"): the dot-product routine of the Miller loop's line
// products -- rr_dot3_core of engine_rr.hip: three accumulator coefficients and two right-hand operands read from LDS, twelve schoolbook
// products on two column sets, two reductions -- called in a dependent loop
//   (a) at the Miller kernel's own occupancy: ONE wave per SIMD, the 610 B-per-lane LDS footprint (a four-wave block owns a CU), and
//   (b) at TWO waves per SIMD with the halved footprint a two-lane split would have (216 B half-home per lane, the partner's coefficients read
//       from its rows, ONE shared right-hand slot pair per lane pair: 296 B per lane, eight waves per CU), <= 256 registers,
// and the same for a register-only Fq2 multiplication (mul2_core's shape).  SIMD cycles per call = elapsed cycles / (calls per wave x waves per SIMD).
// Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Irabe_amd/csrc tools/ubench_rr29_2w.hip -o /tmp/ubench_rr29_2w && /tmp/ubench_rr29_2w
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define RB29_INLINE_ALL 1          // this file brings its own out-of-line routines (one per occupancy: each gets its kernel's register budget)
#include "bn254/fp29.h"
using namespace rabe::bn254;
using rr::i32x9;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int32_t i32x8v __attribute__((ext_vector_type(8)));
typedef int32_t i32x4v __attribute__((ext_vector_type(4)));
struct Out16 { i32x8v lo0, lo1; };

extern __shared__ uint4 lds_q[];          // dynamic: [home quads][y quads][dwords...]; sized by the launch

__device__ __forceinline__ i32x9 lds_elem(const uint4* q, const uint32_t* d, int qstride) {
  const uint4 a = q[0], b = q[qstride];
  i32x9 r;
  r[0] = (int32_t)a.x; r[1] = (int32_t)a.y; r[2] = (int32_t)a.z; r[3] = (int32_t)a.w;
  r[4] = (int32_t)b.x; r[5] = (int32_t)b.y; r[6] = (int32_t)b.z; r[7] = (int32_t)b.w;
  r[8] = (int32_t)d[0];
  return r;
}
// layout per wave: home quads [HQ][64], y quads [YQ][64], home dwords [HD][64], y dwords [YD][64], side [2][64]
template <int HALF> struct Lay {
  static constexpr int HQ = HALF ? 12 : 24, YQ = HALF ? 4 : 8, HD = HALF ? 6 : 12, YD = HALF ? 2 : 4;
  static constexpr int WAVE_BYTES = (HQ + YQ) * 64 * 16 + (HD + YD + 2) * 64 * 4;
};
// HALF = 0: the one-lane layout (coefficient i of the lane's own six).  HALF = 1: a lane owns coefficients 3 h .. 3 h + 2 of its item (h = lane & 1);
// coefficient i lives in lane (pair base + i / 3), local index i % 3; the pair's two right-hand operands: one per lane (slot s in lane base + s).
template <int HALF>
__device__ __attribute__((noinline)) Out16 dot3_core(i32x9 y0a, i32x9 y0b, int ia, int ib, int ic) {
  typedef Lay<HALF> L;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const uint4* wq = lds_q + (size_t)wv * (L::WAVE_BYTES / 16);
  const uint32_t* wd = (const uint32_t*)(wq + (L::HQ + L::YQ) * 64);
  const uint4* yq = wq + L::HQ * 64;
  const uint32_t* yd = wd + L::HD * 64;
  i32x9 x0a, x0b, x1a, x1b, x2a, x2b, y1a, y1b, y2a, y2b;
  if (HALF) {
    const int base = ln & ~1;
    const int la = base + ia / 3, lb = base + ib / 3, lc = base + ic / 3, ja = ia % 3, jb = ib % 3, jc = ic % 3;
    x0a = lds_elem(wq + (4 * ja) * 64 + la, wd + (2 * ja) * 64 + la, 64); x0b = lds_elem(wq + (4 * ja + 2) * 64 + la, wd + (2 * ja + 1) * 64 + la, 64);
    x1a = lds_elem(wq + (4 * jb) * 64 + lb, wd + (2 * jb) * 64 + lb, 64); x1b = lds_elem(wq + (4 * jb + 2) * 64 + lb, wd + (2 * jb + 1) * 64 + lb, 64);
    x2a = lds_elem(wq + (4 * jc) * 64 + lc, wd + (2 * jc) * 64 + lc, 64); x2b = lds_elem(wq + (4 * jc + 2) * 64 + lc, wd + (2 * jc + 1) * 64 + lc, 64);
    y1a = lds_elem(yq + base, yd + base, 64); y1b = lds_elem(yq + 2 * 64 + base, yd + 64 + base, 64);
    y2a = lds_elem(yq + base + 1, yd + base + 1, 64); y2b = lds_elem(yq + 2 * 64 + base + 1, yd + 64 + base + 1, 64);
  } else {
    x0a = lds_elem(wq + (4 * ia) * 64 + ln, wd + (2 * ia) * 64 + ln, 64); x0b = lds_elem(wq + (4 * ia + 2) * 64 + ln, wd + (2 * ia + 1) * 64 + ln, 64);
    x1a = lds_elem(wq + (4 * ib) * 64 + ln, wd + (2 * ib) * 64 + ln, 64); x1b = lds_elem(wq + (4 * ib + 2) * 64 + ln, wd + (2 * ib + 1) * 64 + ln, 64);
    x2a = lds_elem(wq + (4 * ic) * 64 + ln, wd + (2 * ic) * 64 + ln, 64); x2b = lds_elem(wq + (4 * ic + 2) * 64 + ln, wd + (2 * ic + 1) * 64 + ln, 64);
    y1a = lds_elem(yq + ln, yd + ln, 64); y1b = lds_elem(yq + 2 * 64 + ln, yd + 64 + ln, 64);
    y2a = lds_elem(yq + 4 * 64 + ln, yd + 2 * 64 + ln, 64); y2b = lds_elem(yq + 6 * 64 + ln, yd + 3 * 64 + ln, 64);
  }
  i32x9 c0, c1;
  rr::dot3_raw(c0, c1, x0a, x0b, y0a, y0b, x1a, x1b, y1a, y1b, x2a, x2b, y2a, y2b);
  uint32_t* side = (uint32_t*)(wd + (L::HD + L::YD) * 64) + ln;
  side[0] = (uint32_t)c0[8]; side[64] = (uint32_t)c1[8];
  Out16 o;
#pragma unroll
  for (int i = 0; i < 8; i++) { o.lo0[i] = c0[i]; o.lo1[i] = c1[i]; }
  return o;
}
template <int HALF>
__device__ __attribute__((noinline)) Out16 mul2_core(i32x9 a0, i32x9 a1, i32x9 b0, i32x4v b1lo) {
  typedef Lay<HALF> L;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  uint32_t* side = (uint32_t*)(lds_q + (size_t)wv * (L::WAVE_BYTES / 16) + (L::HQ + L::YQ) * 64) + (L::HD + L::YD) * 64 + ln;
  i32x9 b1;
  b1[0] = b1lo[0]; b1[1] = b1lo[1]; b1[2] = b1lo[2]; b1[3] = b1lo[3];
  b1[4] = (int32_t)side[0]; b1[5] = (int32_t)side[64]; b1[6] = b1lo[0] ^ 5; b1[7] = b1lo[1] ^ 9; b1[8] = 77;
  const i32x9 c0 = rr::mac2_raw(a0, b0, -a1, b1), c1 = rr::mac2_raw(a0, b1, a1, b0);
  side[0] = (uint32_t)c0[8]; side[64] = (uint32_t)c1[8];
  Out16 o;
#pragma unroll
  for (int i = 0; i < 8; i++) { o.lo0[i] = c0[i]; o.lo1[i] = c1[i]; }
  return o;
}

template <int HALF>
__device__ __forceinline__ void body(uint32_t iters, int which, uint64_t* out) {
  typedef Lay<HALF> L;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  uint4* wq = lds_q + (size_t)wv * (L::WAVE_BYTES / 16);
  uint32_t* wd = (uint32_t*)(wq + (L::HQ + L::YQ) * 64);
  for (int i = 0; i < L::HQ + L::YQ; i++) wq[i * 64 + ln] = make_uint4(0x0123456u + i + ln, 0x0abcdefu ^ (i * 77), 0x0fedcbau - ln, 0x07777777u + i);
  for (int i = 0; i < L::HD + L::YD + 2; i++) wd[i * 64 + ln] = 1000u + i;
  i32x9 a, b;
  for (int i = 0; i < 9; i++) { a[i] = 0x0111111 + i + ln; b[i] = 0x0222222 - i; }
  a[8] = 1234; b[8] = 2345;
  const uint32_t* side = wd + (L::HD + L::YD) * 64 + ln;
  __syncthreads();
  const uint64_t t0 = clock64();
  for (uint32_t it = 0; it < iters; it++) {
    // the index pattern of a line product; in the two-lane split the two lanes of a pair take different coefficients (a per-lane index)
    int ia = (int)(it % 6), ib = (int)((it + 1) % 6), ic = (int)((it + 3) % 6);
    if (HALF && (ln & 1)) { ia = (ia + 3) % 6; ib = (ib + 3) % 6; ic = (ic + 3) % 6; }
    Out16 o;
    if (which == 0) o = dot3_core<HALF>(a, b, ia, ib, ic);
    else { i32x4v lo; lo[0] = b[0]; lo[1] = b[1]; lo[2] = b[2]; lo[3] = b[3]; o = mul2_core<HALF>(a, b, a, lo); }
    a[0] = (a[0] ^ (o.lo0[0] & 1)) & 0x0fffffff;
    b[1] = (b[1] ^ ((int32_t)side[0] & 1)) & 0x0fffffff;
  }
  const uint64_t t1 = clock64();
  if (ln == 0) out[(size_t)blockIdx.x * 4 + wv] = (t1 - t0) + (uint64_t)((a[0] ^ b[1]) & 1);
}
extern "C" __global__ void __launch_bounds__(256, 1) k_one_wave(uint32_t iters, int which, uint64_t* out) { body<0>(iters, which, out); }
extern "C" __global__ void __launch_bounds__(256, 2) k_two_waves(uint32_t iters, int which, uint64_t* out) { body<1>(iters, which, out); }
// the one-lane layout at two waves per SIMD does not fit the LDS (2 x 152.5 KB): only its register-only routine can be run there
extern "C" __global__ void __launch_bounds__(256, 2) k_two_waves_fullregs(uint32_t iters, int which, uint64_t* out) { body<1>(iters, 1, out); }
// the same with a scratch frame of PAD dwords per lane (touched once): does a kernel's private segment cost it its second wave?
template <int PAD> __global__ void __launch_bounds__(256, 2) k_two_waves_scratch(uint32_t iters, int which, uint64_t* out) {
  volatile uint32_t pad[PAD];
  for (int i = 0; i < PAD; i += 16) pad[i] = iters + i;
  body<1>(iters, which, out);
  if (pad[(iters * 16) % PAD] == 0xdeadbeefu) out[0] = 1;
}

template <class K>
static double run(const char* name, K kern, int blocks, size_t lds_bytes, int which, uint32_t iters, uint64_t* d_out, int waves_per_simd) {
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  int occ = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds_bytes));
  hipFuncAttributes fa;
  CHECK(hipFuncGetAttributes(&fa, (const void*)kern));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, 0, 16u, which, d_out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, 0, iters, which, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  uint64_t* h = (uint64_t*)malloc((size_t)blocks * 4 * 8);
  CHECK(hipMemcpy(h, d_out, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int i = 0; i < blocks * 4; i++) sum += (double)h[i];
  free(h);
  const double per_call_wave = sum / (blocks * 4.0) / iters;          // shader cycles a WAVE spends per call (s_memtime)
  const double simd = per_call_wave / waves_per_simd;                  // ... and a SIMD per call of one of its waves
  // the shader clock this launch ran at: s_memtime cycles of a wave over the launch's event time (launch overhead makes it a lower bound)
  const double ghz = per_call_wave * iters / (ms * 1e-3) * 1e-9;
  const double calls_per_s = (double)blocks * 4.0 * iters / (ms * 1e-3);
  printf("%-58s blocks/CU %d regs %3d lds %6zu  %8.3f ms  wave %7.1f  SIMD %7.1f cycles per call  clock >= %.2f GHz  %.0f M calls/s\n", name, occ, fa.numRegs, lds_bytes, ms,
         per_call_wave, simd, ghz, calls_per_s * 1e-6);
  return simd;
}

int main() {
  uint64_t* d_out;
  CHECK(hipMalloc(&d_out, 4096 * 4 * 8));
  const uint32_t iters = 8000;          // ~25-45 ms per launch: long enough for the clock to settle
  const size_t lds1 = 4 * (size_t)Lay<0>::WAVE_BYTES, lds2 = 4 * (size_t)Lay<1>::WAVE_BYTES;
  printf("LDS per four-wave block: one-lane layout %zu B, two-lane layout %zu B\n", lds1, lds2);
  // how much LDS may a four-wave block take before a CU no longer holds two of them?  (the Miller kernel's block: 78 848 B)
  for (size_t l : {(size_t)75776, (size_t)78848, (size_t)79872, (size_t)80896, (size_t)81920}) run("dot3, two-lane layout, 512 blocks, LDS sweep", k_two_waves, 512, l, 0, iters, d_out, 2);
  for (int rep = 0; rep < 2; rep++) {
    const double a1 = run("dot3 (LDS operands), one-lane layout, 1 wave/SIMD", k_one_wave, 256, lds1, 0, iters, d_out, 1);
    const double a2 = run("dot3 (LDS operands), two-lane layout, 2 waves/SIMD", k_two_waves, 512, lds2, 0, iters, d_out, 2);
    const double a3 = run("dot3 (LDS operands), two-lane layout, 1 wave/SIMD", k_two_waves, 256, lds2, 0, iters, d_out, 1);
    const double b1 = run("Fq2 mul (registers), 1 wave/SIMD", k_one_wave, 256, lds1, 1, iters, d_out, 1);
    const double b2 = run("Fq2 mul (registers), 2 waves/SIMD", k_two_waves, 512, lds2, 1, iters, d_out, 2);
    run("dot3, two-lane layout, 2 waves/SIMD, 448 B scratch", k_two_waves_scratch<112>, 512, lds2, 0, iters, d_out, 2);
    run("dot3, two-lane layout, 2 waves/SIMD, 768 B scratch", k_two_waves_scratch<192>, 512, lds2, 0, iters, d_out, 2);
    run("dot3, two-lane layout, 2 waves/SIMD, 1280 B scratch", k_two_waves_scratch<320>, 512, lds2, 0, iters, d_out, 2);
    run("dot3, two-lane layout, 2 waves/SIMD, 2048 B scratch", k_two_waves_scratch<512>, 512, lds2, 0, iters, d_out, 2);
    printf("  -> dot3: %.3f x at two waves per SIMD (two-lane layout alone: %.3f x); Fq2 mul: %.3f x\n", a1 / a2, a1 / a3, b1 / b2);
  }
  return 0;
}
