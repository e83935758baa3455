// Micro-benchmark: one dependent chain of Fq2 multiplications per lane, in fp.h's 8 x 32-bit representation (the generated
// lazy routine the pairing kernels call) and in fp29.h's 9 x 29-bit reduced radix.  One wave per SIMD (the Fq12 kernels'
// occupancy, forced by a 144 KB LDS allocation per 4-wave block) and as many as the registers allow.
// Build + run on the GPU box:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Irabe_amd/csrc tools/ubench_rr29.hip -o /tmp/ubench_rr29 && /tmp/ubench_rr29
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "bn254/tower.h"
#include "bn254/fp29.h"
using namespace rabe::bn254;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ F29x2 rr2_mul(const F29x2& a, const F29x2& b) { return rr2_mul_inl(a, b); }
__device__ __attribute__((noinline)) F29x2 rr2_sqr(F29x2 a) { return rr2_sqr_inl(a); }

// (a b + c d) / R: the ONE out-of-line routine of the call-based form -- four 9-dword operands (31 dwords in VGPRs, 5 on the stack), 9 back
__device__ __attribute__((noinline)) F29 rr_mac2(F29 a, F29 b, F29 c, F29 d) {
  int64_t t[18];
  rr_cols_init(t);
  rr_cols_mac(t, a, b);
  rr_cols_mac(t, c, d);
  return rr_redc(t);
}
__device__ __forceinline__ F29x2 rr2_mul_calls(const F29x2& a, const F29x2& b) {
  F29x2 r;
  r.c0 = rr_mac2(a.c0, b.c0, rr_neg(a.c1), b.c1);
  r.c1 = rr_mac2(a.c0, b.c1, a.c1, b.c0);
  return r;
}
extern "C" __global__ void __launch_bounds__(256) k_chain29c(uint32_t iters, const uint32_t* in, uint32_t* out) {
  extern __shared__ uint32_t lds[];
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  F29x2 x, y;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    x.c0.l[i] = (int32_t)((in[i] ^ (t & 0xff)) & 0x0fffffff); x.c1.l[i] = (int32_t)(in[9 + i] & 0x0fffffff);
    y.c0.l[i] = (int32_t)(in[18 + i] & 0x0fffffff); y.c1.l[i] = (int32_t)(in[27 + i] & 0x0fffffff);
  }
  x.c0.l[8] &= 0xffff; x.c1.l[8] &= 0xffff; y.c0.l[8] &= 0xffff; y.c1.l[8] &= 0xffff;
  for (uint32_t it = 0; it < iters; it++) {
    x = rr2_mul_calls(x, y);
    y = rr2_mul_calls(y, x);
    x = rr2_mul_calls(x, y);
    y = rr2_mul_calls(y, x);
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) acc ^= (uint32_t)(x.c0.l[i] ^ x.c1.l[i] ^ y.c0.l[i] ^ y.c1.l[i]);
  out[t] = acc;
  if (iters == 0xffffffffu) lds[threadIdx.x] = acc;
}
extern "C" __global__ void __launch_bounds__(256) k_chain32(uint32_t iters, const uint32_t* in, uint32_t* out) {
  extern __shared__ uint32_t lds[];
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  Fp2 x, y;
#pragma unroll
  for (int i = 0; i < 8; i++) { x.c0.v[i] = in[i] ^ (t & 0xff); x.c1.v[i] = in[8 + i]; y.c0.v[i] = in[16 + i]; y.c1.v[i] = in[24 + i]; }
  x.c0.v[7] &= 0x0fffffff; x.c1.v[7] &= 0x0fffffff; y.c0.v[7] &= 0x0fffffff; y.c1.v[7] &= 0x0fffffff;
  for (uint32_t it = 0; it < iters; it++) {
    x = fp2_mul(x, y);
    y = fp2_mul(y, x);
    x = fp2_mul(x, y);
    y = fp2_mul(y, x);
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) acc ^= x.c0.v[i] ^ x.c1.v[i] ^ y.c0.v[i] ^ y.c1.v[i];
  out[t] = acc;
  if (iters == 0xffffffffu) lds[threadIdx.x] = acc;
}
extern "C" __global__ void __launch_bounds__(256) k_chain29(uint32_t iters, const uint32_t* in, uint32_t* out) {
  extern __shared__ uint32_t lds[];
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  F29x2 x, y;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    x.c0.l[i] = (int32_t)((in[i] ^ (t & 0xff)) & 0x0fffffff); x.c1.l[i] = (int32_t)(in[9 + i] & 0x0fffffff);
    y.c0.l[i] = (int32_t)(in[18 + i] & 0x0fffffff); y.c1.l[i] = (int32_t)(in[27 + i] & 0x0fffffff);
  }
  x.c0.l[8] &= 0xffff; x.c1.l[8] &= 0xffff; y.c0.l[8] &= 0xffff; y.c1.l[8] &= 0xffff;
  for (uint32_t it = 0; it < iters; it++) {
    x = rr2_mul(x, y);
    y = rr2_mul(y, x);
    x = rr2_mul(x, y);
    y = rr2_mul(y, x);
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) acc ^= (uint32_t)(x.c0.l[i] ^ x.c1.l[i] ^ y.c0.l[i] ^ y.c1.l[i]);
  out[t] = acc;
  if (iters == 0xffffffffu) lds[threadIdx.x] = acc;
}

template <class K>
static void run(const char* name, K kern, int blocks, size_t lds_bytes, uint32_t iters, const uint32_t* d_in, uint32_t* d_out) {
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, 0, 64u, d_in, d_out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, 0, iters, d_in, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double muls = 4.0 * iters;                                   // per lane
  const double waves = blocks * 4.0;
  const double waves_per_simd = waves / 1024.0;
  // cycles a SIMD spends per Fq2 multiplication of one wave
  const double cyc = ms * 1e-3 * 2.4e9 / (muls * (waves_per_simd < 1 ? 1 : waves_per_simd));
  printf("%-34s blocks %5d lds %6zu  %8.3f ms  %7.1f SIMD-cycles per Fq2 mul  (%.2f G Fq2-mul/s)\n", name, blocks, lds_bytes, ms, cyc,
         muls * blocks * 256 / ms * 1e-6);
}

int main() {
  uint32_t h_in[64];
  for (int i = 0; i < 64; i++) h_in[i] = 0x9e3779b9u * (i + 1) ^ (0x85ebca6bu >> (i & 7));
  uint32_t *d_in, *d_out;
  CHECK(hipMalloc(&d_in, sizeof(h_in)));
  CHECK(hipMalloc(&d_out, 4096 * 256 * 4));
  CHECK(hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice));
  const uint32_t iters = 2000;
  const size_t big = 144 * 1024;
  run("8x32 lazy Fq2, 1 wave/SIMD", k_chain32, 256, big, iters, d_in, d_out);
  run("9x29 Fq2,      1 wave/SIMD", k_chain29, 256, big, iters, d_in, d_out);
  run("9x29 Fq2 = 2 calls, 1 wave/SIMD", k_chain29c, 256, big, iters, d_in, d_out);
  run("8x32 lazy Fq2, 2 waves/SIMD", k_chain32, 512, 72 * 1024, iters, d_in, d_out);
  run("9x29 Fq2,      2 waves/SIMD", k_chain29, 512, 72 * 1024, iters, d_in, d_out);
  run("8x32 lazy Fq2, 4 waves/SIMD", k_chain32, 1024, 36 * 1024, iters, d_in, d_out);
  run("9x29 Fq2,      4 waves/SIMD", k_chain29, 1024, 36 * 1024, iters, d_in, d_out);
  return 0;
}
