// What a lone wave per SIMD pays per VALU instruction, by instruction form (the pairing kernels run one wave per SIMD).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o build/ubench_issue && build/ubench_issue
// Six independent instructions of one form per step, 32 steps per loop iteration; 256 blocks x 256 threads = one wave per
// SIMD, 1024 blocks = four.  Output: cycles per instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int F>
__global__ void __launch_bounds__(256) k_form(uint32_t iters, uint32_t seed, uint32_t* sink) {
  uint32_t x = seed + threadIdx.x * 2654435761u, y = seed ^ (blockIdx.x * 40503u + 77u);
  uint64_t a0 = x, a1 = y, a2 = x ^ y, a3 = x + y, a4 = x * 3u, a5 = y * 5u;
  uint32_t o0 = 1, o1 = 2, o2 = 3, o3 = 4, o4 = 5, o5 = 6;
  uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
      if (F == 0)   // v_mad_u64_u32, six distinct carry-out pairs
        asm volatile("v_mad_u64_u32 %0, %6, %12, %13, %0\n\tv_mad_u64_u32 %1, %7, %12, %13, %1\n\tv_mad_u64_u32 %2, %8, %12, %13, %2\n\t"
                     "v_mad_u64_u32 %3, %9, %12, %13, %3\n\tv_mad_u64_u32 %4, %10, %12, %13, %4\n\tv_mad_u64_u32 %5, %11, %12, %13, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3), "=&s"(c4), "=&s"(c5)
                     : "v"(x), "v"(y));
      if (F == 1)   // v_mad_u64_u32, carry-out to vcc every time
        asm volatile("v_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_mad_u64_u32 %1, vcc, %6, %7, %1\n\tv_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
                     "v_mad_u64_u32 %3, vcc, %6, %7, %3\n\tv_mad_u64_u32 %4, vcc, %6, %7, %4\n\tv_mad_u64_u32 %5, vcc, %6, %7, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(x), "v"(y) : "vcc");
      if (F == 2)   // v_mul_hi_u32 (VOP3, no scalar destination)
        asm volatile("v_mul_hi_u32 %0, %6, %0\n\tv_mul_hi_u32 %1, %6, %1\n\tv_mul_hi_u32 %2, %6, %2\n\t"
                     "v_mul_hi_u32 %3, %6, %3\n\tv_mul_hi_u32 %4, %6, %4\n\tv_mul_hi_u32 %5, %6, %5"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5) : "v"(x));
      if (F == 3)   // v_add_u32 (VOP2, 4-byte encoding)
        asm volatile("v_add_u32 %0, %6, %0\n\tv_add_u32 %1, %6, %1\n\tv_add_u32 %2, %6, %2\n\t"
                     "v_add_u32 %3, %6, %3\n\tv_add_u32 %4, %6, %4\n\tv_add_u32 %5, %6, %5"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5) : "v"(x));
      if (F == 4)   // v_add_co_u32 e64, six distinct carry-out pairs, no carry in
        asm volatile("v_add_co_u32_e64 %0, %6, %12, %0\n\tv_add_co_u32_e64 %1, %7, %12, %1\n\tv_add_co_u32_e64 %2, %8, %12, %2\n\t"
                     "v_add_co_u32_e64 %3, %9, %12, %3\n\tv_add_co_u32_e64 %4, %10, %12, %4\n\tv_add_co_u32_e64 %5, %11, %12, %5"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3), "=&s"(c4), "=&s"(c5)
                     : "v"(x));
      if (F == 5)   // v_addc_co_u32 e64: carry in AND out through six distinct pairs
        asm volatile("v_addc_co_u32_e64 %0, %6, 0, %0, %6\n\tv_addc_co_u32_e64 %1, %7, 0, %1, %7\n\tv_addc_co_u32_e64 %2, %8, 0, %2, %8\n\t"
                     "v_addc_co_u32_e64 %3, %9, 0, %3, %9\n\tv_addc_co_u32_e64 %4, %10, 0, %4, %10\n\tv_addc_co_u32_e64 %5, %11, 0, %5, %11"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), "+s"(c0), "+s"(c1), "+s"(c2), "+s"(c3), "+s"(c4), "+s"(c5));
      if (F == 6)   // v_lshl_add_u32 (VOP3, three sources, no scalar destination)
        asm volatile("v_lshl_add_u32 %0, %6, 1, %0\n\tv_lshl_add_u32 %1, %6, 1, %1\n\tv_lshl_add_u32 %2, %6, 1, %2\n\t"
                     "v_lshl_add_u32 %3, %6, 1, %3\n\tv_lshl_add_u32 %4, %6, 1, %4\n\tv_lshl_add_u32 %5, %6, 1, %5"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5) : "v"(x));
      if (F == 7)   // v_mad_u32_u24 (VOP3, three VGPR sources, no scalar destination)
        asm volatile("v_mad_u32_u24 %0, %6, %7, %0\n\tv_mad_u32_u24 %1, %6, %7, %1\n\tv_mad_u32_u24 %2, %6, %7, %2\n\t"
                     "v_mad_u32_u24 %3, %6, %7, %3\n\tv_mad_u32_u24 %4, %6, %7, %4\n\tv_mad_u32_u24 %5, %6, %7, %5"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5) : "v"(x), "v"(y));
      if (F == 8)   // v_mov_b32
        asm volatile("v_mov_b32 %0, %6\n\tv_mov_b32 %1, %6\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %6\n\tv_mov_b32 %4, %6\n\tv_mov_b32 %5, %6"
                     : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3), "=v"(o4), "=v"(o5) : "v"(x));
      if (F == 9)   // v_mul_lo_u32
        asm volatile("v_mul_lo_u32 %0, %6, %0\n\tv_mul_lo_u32 %1, %6, %1\n\tv_mul_lo_u32 %2, %6, %2\n\t"
                     "v_mul_lo_u32 %3, %6, %3\n\tv_mul_lo_u32 %4, %6, %4\n\tv_mul_lo_u32 %5, %6, %5"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5) : "v"(x));
      if (F == 10)  // v_cndmask_b32 e64 with an SGPR-pair mask
        asm volatile("v_cndmask_b32_e64 %0, %0, %6, %7\n\tv_cndmask_b32_e64 %1, %1, %6, %7\n\tv_cndmask_b32_e64 %2, %2, %6, %7\n\t"
                     "v_cndmask_b32_e64 %3, %3, %6, %7\n\tv_cndmask_b32_e64 %4, %4, %6, %7\n\tv_cndmask_b32_e64 %5, %5, %6, %7"
                     : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5) : "v"(x), "s"(c0));
      if (F == 11)  // v_fma_f64 (the double-precision-limb alternative to v_mad_u64_u32)
        asm volatile("v_fma_f64 %0, %6, %7, %0\n\tv_fma_f64 %1, %6, %7, %1\n\tv_fma_f64 %2, %6, %7, %2\n\t"
                     "v_fma_f64 %3, %6, %7, %3\n\tv_fma_f64 %4, %6, %7, %4\n\tv_fma_f64 %5, %6, %7, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(c0), "v"(c1));
      if (F == 12)  // v_add_f64
        asm volatile("v_add_f64 %0, %6, %0\n\tv_add_f64 %1, %6, %1\n\tv_add_f64 %2, %6, %2\n\t"
                     "v_add_f64 %3, %6, %3\n\tv_add_f64 %4, %6, %4\n\tv_add_f64 %5, %6, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(c0));
      if (F == 13)  // v_lshl_add_u64 (64-bit integer add in one instruction)
        asm volatile("v_lshl_add_u64 %0, %6, 0, %0\n\tv_lshl_add_u64 %1, %6, 0, %1\n\tv_lshl_add_u64 %2, %6, 0, %2\n\t"
                     "v_lshl_add_u64 %3, %6, 0, %3\n\tv_lshl_add_u64 %4, %6, 0, %4\n\tv_lshl_add_u64 %5, %6, 0, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(c0));
    }
  }
  uint64_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5;
  uint32_t o = o0 ^ o1 ^ o2 ^ o3 ^ o4 ^ o5;
  if ((uint32_t)s + o == 0x12345678u) sink[0] = o;
}

template <int F>
static void run(const char* what, int blocks, uint32_t* sink) {
  const uint32_t iters = 2000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_form<F>), dim3(blocks), dim3(256), 0, 0, 10u, 1u, sink);
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_form<F>), dim3(blocks), dim3(256), 0, 0, iters, 1u, sink);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double per_wave = (double)iters * 32 * 6;
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  printf("%-58s blocks=%5d  %.3f ms  %.2f cycles/instruction/SIMD\n", what, blocks, ms, ms * 1e-3 * 2.4e9 / (per_wave * waves_per_simd));
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
  uint32_t* sink;
  CHECK(hipMalloc(&sink, 64));
  for (int blocks : {256, 512, 1024}) {
    run<0>("v_mad_u64_u32, distinct carry-out pairs", blocks, sink);
    run<1>("v_mad_u64_u32, carry-out to vcc", blocks, sink);
    run<2>("v_mul_hi_u32", blocks, sink);
    run<9>("v_mul_lo_u32", blocks, sink);
    run<7>("v_mad_u32_u24", blocks, sink);
    run<6>("v_lshl_add_u32", blocks, sink);
    run<3>("v_add_u32 (VOP2)", blocks, sink);
    run<4>("v_add_co_u32 e64, distinct carry-out pairs", blocks, sink);
    run<5>("v_addc_co_u32 e64, carry in+out, distinct pairs", blocks, sink);
    run<10>("v_cndmask_b32 e64, SGPR mask", blocks, sink);
    run<8>("v_mov_b32", blocks, sink);
    run<11>("v_fma_f64", blocks, sink);
    run<12>("v_add_f64", blocks, sink);
    run<13>("v_lshl_add_u64", blocks, sink);
  }
  CHECK(hipFree(sink));
  return 0;
}
