// What a v_mad_u64_u32 costs depending on (a) whether consecutive mads write the SAME carry-out SGPR pair and (b) whether they
// accumulate into the same 64-bit register (a dependent chain).  The carry-free reduced-radix field core (bn254/fp29.h) never
// reads the carry-out, so the compiler gives every mad the same sdst.
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/ubench_mad_sdst.hip -o /tmp/ubench_mad_sdst && /tmp/ubench_mad_sdst
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define M(acc, sd) "v_mad_u64_u32 %" #acc ", " sd ", %[x], %[y], %" #acc "\n\t"
// MODE 0: 8 accumulators, one sdst.  1: 8 accumulators, 4 sdst pairs in rotation.  2: ONE accumulator, one sdst.  3: ONE accumulator, 4 sdst.
// 4: 2 accumulators alternating, one sdst.   5: 8 accumulators, sdst = vcc
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t iters, uint32_t seed, uint32_t* sink) {
  extern __shared__ uint32_t lds[];
  uint32_t x = seed + threadIdx.x * 2654435761u, y = seed ^ (blockIdx.x * 40503u + 77u);
  uint64_t a0 = x, a1 = y, a2 = x ^ y, a3 = x + y, a4 = x * 3u, a5 = y * 5u, a6 = x * 7u, a7 = y * 9u;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (MODE == 0)
        asm volatile(M(0, "s[20:21]") M(1, "s[20:21]") M(2, "s[20:21]") M(3, "s[20:21]") M(4, "s[20:21]") M(5, "s[20:21]") M(6, "s[20:21]") M(7, "s[20:21]")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : [x] "v"(x), [y] "v"(y) : "s20", "s21");
      if (MODE == 1)
        asm volatile(M(0, "s[20:21]") M(1, "s[22:23]") M(2, "s[24:25]") M(3, "s[26:27]") M(4, "s[20:21]") M(5, "s[22:23]") M(6, "s[24:25]") M(7, "s[26:27]")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : [x] "v"(x), [y] "v"(y)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      if (MODE == 2)
        asm volatile(M(0, "s[20:21]") M(0, "s[20:21]") M(0, "s[20:21]") M(0, "s[20:21]") M(0, "s[20:21]") M(0, "s[20:21]") M(0, "s[20:21]") M(0, "s[20:21]")
                     : "+v"(a0) : [x] "v"(x), [y] "v"(y) : "s20", "s21");
      if (MODE == 3)
        asm volatile(M(0, "s[20:21]") M(0, "s[22:23]") M(0, "s[24:25]") M(0, "s[26:27]") M(0, "s[20:21]") M(0, "s[22:23]") M(0, "s[24:25]") M(0, "s[26:27]")
                     : "+v"(a0) : [x] "v"(x), [y] "v"(y) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      if (MODE == 4)
        asm volatile(M(0, "s[20:21]") M(1, "s[20:21]") M(0, "s[20:21]") M(1, "s[20:21]") M(0, "s[20:21]") M(1, "s[20:21]") M(0, "s[20:21]") M(1, "s[20:21]")
                     : "+v"(a0), "+v"(a1) : [x] "v"(x), [y] "v"(y) : "s20", "s21");
      if (MODE == 5)
        asm volatile(M(0, "vcc") M(1, "vcc") M(2, "vcc") M(3, "vcc") M(4, "vcc") M(5, "vcc") M(6, "vcc") M(7, "vcc")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : [x] "v"(x), [y] "v"(y) : "vcc");
      if (MODE == 6)   // 3 accumulators in rotation, one sdst
        asm volatile(M(0, "s[20:21]") M(1, "s[20:21]") M(2, "s[20:21]") M(0, "s[20:21]") M(1, "s[20:21]") M(2, "s[20:21]") M(0, "s[20:21]") M(1, "s[20:21]")
                     : "+v"(a0), "+v"(a1), "+v"(a2) : [x] "v"(x), [y] "v"(y) : "s20", "s21");
      if (MODE == 7)   // 4 accumulators in rotation, one sdst
        asm volatile(M(0, "s[20:21]") M(1, "s[20:21]") M(2, "s[20:21]") M(3, "s[20:21]") M(0, "s[20:21]") M(1, "s[20:21]") M(2, "s[20:21]") M(3, "s[20:21]")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : [x] "v"(x), [y] "v"(y) : "s20", "s21");
    }
  }
  sink[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (uint32_t)((a0 ^ a1) >> 32);
  if (iters == 0xffffffffu) lds[threadIdx.x] = x;
}

template <int MODE>
static void run(const char* name, uint32_t* d_out) {
  const uint32_t iters = 4000;
  for (int w = 1; w <= 4; w *= 2) {
    const int blocks = 256 * w;
    const size_t lds = (144 * 1024) / w;
    CHECK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, 16u, 1u, d_out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, iters, 1u, d_out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s %d wave(s)/SIMD  %7.2f SIMD-cycles per mad\n", name, w, ms * 1e-3 * 2.4e9 / (iters * 16.0 * 8.0 * w));
  }
}
int main() {
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, 1024 * 256 * 4));
  run<0>("8 accumulators, ONE sdst", d_out);
  run<1>("8 accumulators, 4 sdst in rotation", d_out);
  run<5>("8 accumulators, sdst = vcc", d_out);
  run<2>("1 accumulator (dependent), ONE sdst", d_out);
  run<3>("1 accumulator (dependent), 4 sdst", d_out);
  run<4>("2 accumulators alternating, ONE sdst", d_out);
  run<6>("3 accumulators in rotation, ONE sdst", d_out);
  run<7>("4 accumulators in rotation, ONE sdst", d_out);
  return 0;
}
