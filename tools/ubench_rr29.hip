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
using rr::i32x9;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int32_t i32x4v __attribute__((ext_vector_type(4)));
typedef int32_t i32x8v __attribute__((ext_vector_type(8)));
struct F29x2 { i32x9 c0, c1; };
__device__ __forceinline__ F29x2 rr2_mul(const F29x2& a, const F29x2& b) {
  return F29x2{rr::mac2_raw(a.c0, b.c0, -a.c1, b.c1), rr::mac2_raw(a.c0, b.c1, a.c1, b.c0)};
}
// call-based form: 31 argument dwords in VGPRs, the other 5 and the 2 result dwords beyond 16 through a per-lane LDS slot
static __shared__ uint32_t rr_side[8 * 256];
struct Out16 { i32x8v lo0, lo1; };
__device__ __attribute__((noinline)) Out16 rr2_mul_core(i32x9 a0, i32x9 a1, i32x9 b0, i32x4v b1lo) {
  uint32_t* side = rr_side + threadIdx.x;
  i32x9 b1;
  b1[0] = b1lo[0]; b1[1] = b1lo[1]; b1[2] = b1lo[2]; b1[3] = b1lo[3];
  b1[4] = (int32_t)side[0]; b1[5] = (int32_t)side[256]; b1[6] = (int32_t)side[512]; b1[7] = (int32_t)side[768]; b1[8] = (int32_t)side[1024];
  const i32x9 c0 = rr::mac2_raw(a0, b0, -a1, b1), c1 = rr::mac2_raw(a0, b1, a1, b0);
  side[0] = (uint32_t)c0[8]; side[256] = (uint32_t)c1[8];
  Out16 o;
  for (int i = 0; i < 8; i++) { o.lo0[i] = c0[i]; o.lo1[i] = c1[i]; }
  return o;
}
__device__ __forceinline__ F29x2 rr2_mul_calls(const F29x2& a, const F29x2& b) {
  uint32_t* side = rr_side + threadIdx.x;
  side[0] = (uint32_t)b.c1[4]; side[256] = (uint32_t)b.c1[5]; side[512] = (uint32_t)b.c1[6]; side[768] = (uint32_t)b.c1[7]; side[1024] = (uint32_t)b.c1[8];
  i32x4v lo; lo[0] = b.c1[0]; lo[1] = b.c1[1]; lo[2] = b.c1[2]; lo[3] = b.c1[3];
  const Out16 o = rr2_mul_core(a.c0, a.c1, b.c0, lo);
  F29x2 r;
  for (int i = 0; i < 8; i++) { r.c0[i] = o.lo0[i]; r.c1[i] = o.lo1[i]; }
  r.c0[8] = (int32_t)side[0]; r.c1[8] = (int32_t)side[256];
  return r;
}
#define LOAD29(x, y) \
  F29x2 x, y; \
  _Pragma("unroll") for (int i = 0; i < 9; i++) { \
    x.c0[i] = (int32_t)((in[i] ^ (t & 0xff)) & 0x0fffffff); x.c1[i] = (int32_t)(in[9 + i] & 0x0fffffff); \
    y.c0[i] = (int32_t)(in[18 + i] & 0x0fffffff); y.c1[i] = (int32_t)(in[27 + i] & 0x0fffffff); \
  } \
  x.c0[8] &= 0xffff; x.c1[8] &= 0xffff; y.c0[8] &= 0xffff; y.c1[8] &= 0xffff;
#define SINK29(x, y) \
  uint32_t acc = 0; \
  _Pragma("unroll") for (int i = 0; i < 9; i++) acc ^= (uint32_t)(x.c0[i] ^ x.c1[i] ^ y.c0[i] ^ y.c1[i]); \
  out[t] = acc; \
  if (iters == 0xffffffffu) lds[threadIdx.x] = acc;
extern "C" __global__ void __launch_bounds__(256) k_chain29c(uint32_t iters, const uint32_t* in, uint32_t* out) {
  extern __shared__ uint32_t lds[];
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  LOAD29(x, y)
  for (uint32_t it = 0; it < iters; it++) {
    x = rr2_mul_calls(x, y);
    y = rr2_mul_calls(y, x);
    x = rr2_mul_calls(x, y);
    y = rr2_mul_calls(y, x);
  }
  SINK29(x, y)
}
extern "C" __global__ void __launch_bounds__(256) k_chain29(uint32_t iters, const uint32_t* in, uint32_t* out) {
  extern __shared__ uint32_t lds[];
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  LOAD29(x, y)
  for (uint32_t it = 0; it < iters; it++) {
    x = rr2_mul(x, y);
    y = rr2_mul(y, x);
    x = rr2_mul(x, y);
    y = rr2_mul(y, x);
  }
  SINK29(x, y)
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
template <class K>
static void run(const char* name, K kern, int blocks, size_t lds_bytes, uint32_t iters, const uint32_t* d_in, uint32_t* d_out) {
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
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
  const size_t big = 136 * 1024;
  run("8x32 lazy Fq2, 1 wave/SIMD", k_chain32, 256, big, iters, d_in, d_out);
  run("9x29 Fq2,      1 wave/SIMD", k_chain29, 256, big, iters, d_in, d_out);
  run("9x29 Fq2, one call + LDS side slot, 1 wave/SIMD", k_chain29c, 256, big, iters, d_in, d_out);
  run("8x32 lazy Fq2, 2 waves/SIMD", k_chain32, 512, 72 * 1024, iters, d_in, d_out);
  run("9x29 Fq2,      2 waves/SIMD", k_chain29, 512, 72 * 1024, iters, d_in, d_out);
  run("8x32 lazy Fq2, 4 waves/SIMD", k_chain32, 1024, 36 * 1024, iters, d_in, d_out);
  run("9x29 Fq2,      4 waves/SIMD", k_chain29, 1024, 36 * 1024, iters, d_in, d_out);
  return 0;
}
