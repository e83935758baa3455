// Issue-rate / latency micro-benchmark for the multiply-accumulate step the field arithmetic is made of
// (bn254/fp.h: one v_mad_u64_u32 into a 64-bit column accumulator + one v_addc_co_u32 banking the carry).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench_mac.hip -o /tmp/ubench_mac && /tmp/ubench_mac
//
// Each kernel runs `iters` iterations of REP steps; a step is K independent MACs issued as K mads followed by K
// addcs (K = interleave depth).  Launched with 1 wave per SIMD (256 blocks x 256 threads, the pairing kernels'
// occupancy) and with 4 waves per SIMD.  Output: cycles per MAC per wave at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define MAD(i) "v_mad_u64_u32 %" #i ", %[c" #i "], %[x], %[y], %" #i "\n\t"
#define ADC(i, o) "v_addc_co_u32_e64 %" #o ", %[c" #i "], 0, %" #o ", %[c" #i "]\n\t"

template <int K, int PAD>
__global__ void __launch_bounds__(256) k_mac(uint32_t iters, uint32_t seed, uint32_t* sink) {
  uint32_t x = seed + threadIdx.x * 2654435761u, y = seed ^ (blockIdx.x * 40503u + 77u);
  uint64_t a0 = x, a1 = y, a2 = x ^ y, a3 = x + y, a4 = x * 3u, a5 = y * 5u;
  uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0;
  uint64_t c0 = 0, c1 = 0, c2, c3, c4, c5;
  uint32_t p0 = x, p1 = y, p2 = x, p3 = y;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
      if (K == 1)
        asm volatile("v_mad_u64_u32 %0, %[c0], %[x], %[y], %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %[c0], 0, %1, %[c0]"
                     : "+v"(a0), "+v"(o0), [c0] "=&s"(c0) : [x] "v"(x), [y] "v"(y));
      if (K == 2)
        asm volatile("v_mad_u64_u32 %0, %[c0], %[x], %[y], %0\n\tv_mad_u64_u32 %2, %[c1], %[x], %[y], %2\n\ts_nop 0\n\t"
                     "v_addc_co_u32_e64 %1, %[c0], 0, %1, %[c0]\n\tv_addc_co_u32_e64 %3, %[c1], 0, %3, %[c1]"
                     : "+v"(a0), "+v"(o0), "+v"(a1), "+v"(o1), [c0] "=&s"(c0), [c1] "=&s"(c1) : [x] "v"(x), [y] "v"(y));
      if (K == 3)
        asm volatile("v_mad_u64_u32 %0, %[c0], %[x], %[y], %0\n\tv_mad_u64_u32 %2, %[c1], %[x], %[y], %2\n\t"
                     "v_mad_u64_u32 %4, %[c2], %[x], %[y], %4\n\t"
                     "v_addc_co_u32_e64 %1, %[c0], 0, %1, %[c0]\n\tv_addc_co_u32_e64 %3, %[c1], 0, %3, %[c1]\n\t"
                     "v_addc_co_u32_e64 %5, %[c2], 0, %5, %[c2]"
                     : "+v"(a0), "+v"(o0), "+v"(a1), "+v"(o1), "+v"(a2), "+v"(o2), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
                     : [x] "v"(x), [y] "v"(y));
      if (K == 4)
        asm volatile("v_mad_u64_u32 %0, %[c0], %[x], %[y], %0\n\tv_mad_u64_u32 %2, %[c1], %[x], %[y], %2\n\t"
                     "v_mad_u64_u32 %4, %[c2], %[x], %[y], %4\n\tv_mad_u64_u32 %6, %[c3], %[x], %[y], %6\n\t"
                     "v_addc_co_u32_e64 %1, %[c0], 0, %1, %[c0]\n\tv_addc_co_u32_e64 %3, %[c1], 0, %3, %[c1]\n\t"
                     "v_addc_co_u32_e64 %5, %[c2], 0, %5, %[c2]\n\tv_addc_co_u32_e64 %7, %[c3], 0, %7, %[c3]"
                     : "+v"(a0), "+v"(o0), "+v"(a1), "+v"(o1), "+v"(a2), "+v"(o2), "+v"(a3), "+v"(o3), [c0] "=&s"(c0),
                       [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
                     : [x] "v"(x), [y] "v"(y));
      if (K == 6)
        asm volatile("v_mad_u64_u32 %0, %[c0], %[x], %[y], %0\n\tv_mad_u64_u32 %2, %[c1], %[x], %[y], %2\n\t"
                     "v_mad_u64_u32 %4, %[c2], %[x], %[y], %4\n\tv_mad_u64_u32 %6, %[c3], %[x], %[y], %6\n\t"
                     "v_mad_u64_u32 %8, %[c4], %[x], %[y], %8\n\tv_mad_u64_u32 %10, %[c5], %[x], %[y], %10\n\t"
                     "v_addc_co_u32_e64 %1, %[c0], 0, %1, %[c0]\n\tv_addc_co_u32_e64 %3, %[c1], 0, %3, %[c1]\n\t"
                     "v_addc_co_u32_e64 %5, %[c2], 0, %5, %[c2]\n\tv_addc_co_u32_e64 %7, %[c3], 0, %7, %[c3]\n\t"
                     "v_addc_co_u32_e64 %9, %[c4], 0, %9, %[c4]\n\tv_addc_co_u32_e64 %11, %[c5], 0, %11, %[c5]"
                     : "+v"(a0), "+v"(o0), "+v"(a1), "+v"(o1), "+v"(a2), "+v"(o2), "+v"(a3), "+v"(o3), "+v"(a4), "+v"(o4),
                       "+v"(a5), "+v"(o5), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [c4] "=&s"(c4),
                       [c5] "=&s"(c5)
                     : [x] "v"(x), [y] "v"(y));
      if (K == 8)   // the same six MACs, addcs interleaved between the mads (each addc two instructions behind its mad)
        asm volatile("v_mad_u64_u32 %0, %[c0], %[x], %[y], %0\n\tv_mad_u64_u32 %2, %[c1], %[x], %[y], %2\n\t"
                     "v_addc_co_u32_e64 %1, %[c0], 0, %1, %[c0]\n\tv_mad_u64_u32 %4, %[c2], %[x], %[y], %4\n\t"
                     "v_addc_co_u32_e64 %3, %[c1], 0, %3, %[c1]\n\tv_mad_u64_u32 %6, %[c3], %[x], %[y], %6\n\t"
                     "v_addc_co_u32_e64 %5, %[c2], 0, %5, %[c2]\n\tv_mad_u64_u32 %8, %[c4], %[x], %[y], %8\n\t"
                     "v_addc_co_u32_e64 %7, %[c3], 0, %7, %[c3]\n\tv_mad_u64_u32 %10, %[c5], %[x], %[y], %10\n\t"
                     "v_addc_co_u32_e64 %9, %[c4], 0, %9, %[c4]\n\tv_addc_co_u32_e64 %11, %[c5], 0, %11, %[c5]"
                     : "+v"(a0), "+v"(o0), "+v"(a1), "+v"(o1), "+v"(a2), "+v"(o2), "+v"(a3), "+v"(o3), "+v"(a4), "+v"(o4),
                       "+v"(a5), "+v"(o5), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [c4] "=&s"(c4),
                       [c5] "=&s"(c5)
                     : [x] "v"(x), [y] "v"(y));
      if (K == 9)   // mads only (no carry capture): the multiplier's own rate for one wave
        asm volatile("v_mad_u64_u32 %0, %[c0], %[x], %[y], %0\n\tv_mad_u64_u32 %1, %[c0], %[x], %[y], %1\n\t"
                     "v_mad_u64_u32 %2, %[c0], %[x], %[y], %2\n\tv_mad_u64_u32 %3, %[c0], %[x], %[y], %3\n\t"
                     "v_mad_u64_u32 %4, %[c0], %[x], %[y], %4\n\tv_mad_u64_u32 %5, %[c0], %[x], %[y], %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), [c0] "=&s"(c0) : [x] "v"(x), [y] "v"(y));
      if (K == 10)  // addcs only
        asm volatile("v_addc_co_u32_e64 %0, %[c0], 0, %0, %[c0]\n\tv_addc_co_u32_e64 %1, %[c1], 0, %1, %[c1]\n\t"
                     "v_addc_co_u32_e64 %2, %[c0], 0, %2, %[c0]\n\tv_addc_co_u32_e64 %3, %[c1], 0, %3, %[c1]\n\t"
                     "v_addc_co_u32_e64 %4, %[c0], 0, %4, %[c0]\n\tv_addc_co_u32_e64 %5, %[c1], 0, %5, %[c1]"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), [c0] "+s"(c0), [c1] "+s"(c1));
      // PAD independent filler instructions per step (what the glue code looks like)
      if (PAD == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(p0) : "v"(p1));
      if (PAD == 2) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %1, a0" : "=v"(p2) : "v"(p3) : "a0");
      if (PAD == 3) asm volatile("s_nop 1");
      if (PAD == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(p0) : "v"(p1));
    }
  }
  uint64_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5;
  uint32_t o = o0 ^ o1 ^ o2 ^ o3 ^ o4 ^ o5 ^ p0 ^ p2;
  if ((uint32_t)s + o == 0x12345678u) sink[0] = o;
}

template <int K, int PAD>
static void run(const char* what, int blocks, uint32_t* sink) {
  const uint32_t iters = 2000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mac<K, PAD>), dim3(blocks), dim3(256), 0, 0, 10u, 1u, sink);
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_mac<K, PAD>), dim3(blocks), dim3(256), 0, 0, iters, 1u, sink);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double macs_per_wave = (double)iters * 32 * (K >= 8 ? 6 : K);
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  const double cyc = ms * 1e-3 * 2.4e9 / (macs_per_wave * (waves_per_simd < 1 ? 1 : waves_per_simd));
  printf("%-34s K=%d pad=%d blocks=%5d  %.3f ms  %.2f cycles/MAC/SIMD  (%.1f T MAC/s)\n", what, K, PAD, blocks, ms, cyc,
         macs_per_wave * blocks * 256 / (ms * 1e-3) / 1e12);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
  uint32_t* sink;
  CHECK(hipMalloc(&sink, 64));
  for (int blocks : {256, 512, 1024, 2048}) {
    run<1, 0>("mad+nop+addc", blocks, sink);
    run<2, 0>("2 mads, nop, 2 addcs", blocks, sink);
    run<3, 0>("3 mads, 3 addcs", blocks, sink);
    run<4, 0>("4 mads, 4 addcs", blocks, sink);
    run<6, 0>("6 mads, 6 addcs", blocks, sink);
    run<8, 0>("6 mads / 6 addcs alternating", blocks, sink);
    run<9, 0>("6 mads, no addc", blocks, sink);
    run<10, 0>("6 addcs, no mad", blocks, sink);
  }
  for (int blocks : {256, 1024}) {
    run<3, 1>("3+3 + v_mov", blocks, sink);
    run<3, 2>("3+3 + accvgpr write/read", blocks, sink);
    run<3, 3>("3+3 + s_nop 1", blocks, sink);
    run<3, 4>("3+3 + v_add_u32", blocks, sink);
  }
  CHECK(hipFree(sink));
  return 0;
}
