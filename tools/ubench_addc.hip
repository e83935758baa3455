// Carry chains with and without the wait states hipcc puts between dependent v_addc_co_u32 (LLVM's gfx940+ rule "VALU writes
// SGPR/VCC -> VALU reads it: 2 wait states"): speed at one wave per SIMD, and whether the results differ.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench_addc.hip -o /tmp/ubench_addc && /tmp/ubench_addc
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_add(uint32_t iters, uint32_t seed, uint32_t* out) {
  uint32_t a[8], b[8];
  uint32_t x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
  for (int i = 0; i < 8; i++) { x = x * 1664525u + 1013904223u; a[i] = x; x = x * 1664525u + 1013904223u; b[i] = x | 0x80000000u; }
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (MODE == 0) {          // compiler chain (hipcc pads it)
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = __builtin_addc(a[i], b[i], c, &c);
        b[0] += c;
      } else if (MODE == 1) {   // the same chain in one asm statement, no wait states
        uint32_t c;
        asm volatile("v_add_co_u32 %0, vcc, %0, %9\n\tv_addc_co_u32 %1, vcc, %1, %10, vcc\n\tv_addc_co_u32 %2, vcc, %2, %11, vcc\n\t"
                     "v_addc_co_u32 %3, vcc, %3, %12, vcc\n\tv_addc_co_u32 %4, vcc, %4, %13, vcc\n\tv_addc_co_u32 %5, vcc, %5, %14, vcc\n\t"
                     "v_addc_co_u32 %6, vcc, %6, %15, vcc\n\tv_addc_co_u32 %7, vcc, %7, %16, vcc\n\tv_addc_co_u32 %8, vcc, 0, 0, vcc"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "=v"(c)
                     : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
        b[0] += c;
      } else if (MODE == 3) {   // VOP3 forms with the carry in a compiler-allocated SGPR pair, no wait states
        uint32_t c;
        uint64_t cy;
        asm volatile("v_add_co_u32_e64 %0, %9, %0, %10\n\tv_addc_co_u32_e64 %1, %9, %1, %11, %9\n\tv_addc_co_u32_e64 %2, %9, %2, %12, %9\n\t"
                     "v_addc_co_u32_e64 %3, %9, %3, %13, %9\n\tv_addc_co_u32_e64 %4, %9, %4, %14, %9\n\tv_addc_co_u32_e64 %5, %9, %5, %15, %9\n\t"
                     "v_addc_co_u32_e64 %6, %9, %6, %16, %9\n\tv_addc_co_u32_e64 %7, %9, %7, %17, %9\n\tv_addc_co_u32_e64 %8, %9, 0, 0, %9"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "=v"(c), "=&s"(cy)
                     : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
        b[0] += c;
      } else {                  // asm with the two wait states after every carry producer
        uint32_t c;
        asm volatile("v_add_co_u32 %0, vcc, %0, %9\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %10, vcc\n\ts_nop 1\n\tv_addc_co_u32 %2, vcc, %2, %11, vcc\n\ts_nop 1\n\t"
                     "v_addc_co_u32 %3, vcc, %3, %12, vcc\n\ts_nop 1\n\tv_addc_co_u32 %4, vcc, %4, %13, vcc\n\ts_nop 1\n\tv_addc_co_u32 %5, vcc, %5, %14, vcc\n\ts_nop 1\n\t"
                     "v_addc_co_u32 %6, vcc, %6, %15, vcc\n\ts_nop 1\n\tv_addc_co_u32 %7, vcc, %7, %16, vcc\n\ts_nop 1\n\tv_addc_co_u32 %8, vcc, 0, 0, vcc"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "=v"(c)
                     : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
        b[0] += c;
      }
    }
  }
  uint32_t h = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) h = h * 31u + a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}

template <int MODE>
static void run(int blocks, uint32_t iters, uint32_t* dev, uint32_t* host) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_add<MODE>), dim3(blocks), dim3(256), 0, 0, 8u, 7u, dev);
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_add<MODE>), dim3(blocks), dim3(256), 0, 0, iters, 7u, dev);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(host, dev, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost));
  const double chains = (double)iters * 8;
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  printf("mode %d blocks %5d: %.3f ms, %.1f cycles per 9-instruction chain per SIMD\n", MODE, blocks, ms,
         ms * 1e-3 * 2.4e9 / (chains * (waves_per_simd < 1 ? 1 : waves_per_simd)));
}

int main() {
  const int maxb = 2048;
  size_t bad_total = 0;
  uint32_t* dev;
  CHECK(hipMalloc(&dev, (size_t)maxb * 256 * 4));
  uint32_t* h0 = (uint32_t*)malloc((size_t)maxb * 256 * 4);
  uint32_t* h1 = (uint32_t*)malloc((size_t)maxb * 256 * 4);
  uint32_t* h2 = (uint32_t*)malloc((size_t)maxb * 256 * 4);
  uint32_t* h3 = (uint32_t*)malloc((size_t)maxb * 256 * 4);
  for (int blocks : {256, 1024, 2048}) {
    const uint32_t iters = 20000;
    run<0>(blocks, iters, dev, h0);
    run<1>(blocks, iters, dev, h1);
    run<2>(blocks, iters, dev, h2);
    run<3>(blocks, iters, dev, h3);
    size_t bad1 = 0, bad2 = 0, bad3 = 0;
    for (size_t i = 0; i < (size_t)blocks * 256; i++) { bad1 += h0[i] != h1[i]; bad2 += h0[i] != h2[i]; bad3 += h0[i] != h3[i]; }
    printf("   lanes whose result differs from the compiler chain: without wait states %zu (vcc) / %zu (SGPR pair), with %zu (of %d; %.2e chained additions each)\n", bad1, bad3, bad2,
           blocks * 256, 20000.0 * 8 * 8);
    bad_total += bad1 + bad2 + bad3;
  }
  // exit status for tests/test_gpu_carry_interlock.py: non-zero when any lane of any unpadded chain differs from the padded one
  return bad_total ? 3 : 0;
}
