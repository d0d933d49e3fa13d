// Calibration (developer tool): does v_mfma_f32_32x32x16_bf16 throughput depend on the
// operand DATA (power/clock management)?  Same loop, operands constant vs random bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe(float* out, const u32x4* __restrict__ tab, int iters) {
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n)
    for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
  bf16x8 a[4], b[4];
  for (int j = 0; j < 4; ++j) {
    a[j] = __builtin_bit_cast(bf16x8, tab[(threadIdx.x * 4 + j) & 4095]);
    b[j] = __builtin_bit_cast(bf16x8, tab[(threadIdx.x * 4 + j + 1777) & 4095]);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + n) & 3], b[u], acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < 4; ++n)
    for (int q = 0; q < 16; ++q) s += acc[n][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  u32x4* tab;
  float* out;
  (void)hipMalloc(&tab, 4096 * 16);
  (void)hipMalloc(&out, 1024 * 256 * 4);
  static unsigned h[4096 * 4];
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    unsigned st = 12345;
    for (int i = 0; i < 4096 * 4; ++i) {
      st = st * 1664525u + 1013904223u;
      unsigned lo = 0x3f80, hi = 0x3f80;
      if (mode == 1) {  // random bf16 in [-2, 2): random sign, exponent 125..127, random mantissa
        lo = ((st >> 3) & 0x807f) | ((125 + ((st >> 20) % 3)) << 7);
        hi = ((st >> 9) & 0x807f) | ((125 + ((st >> 24) % 3)) << 7);
      } else if (mode == 2) {
        lo = hi = 0;
      }
      h[i] = lo | (hi << 16);
    }
    (void)hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
    for (int iters : {500, 20000, 200000}) {
      for (int blocks : {256, 512}) {
        probe<<<blocks, 256>>>(out, tab, iters);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        probe<<<blocks, 256>>>(out, tab, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s iters=%d waves/SIMD=%d: %.3f ms  %.2f ns per MFMA per SIMD\n",
               mode == 0 ? "const-1.0" : mode == 1 ? "random   " : "zeros    ", iters, blocks / 256, ms,
               ms * 1e6 / ((double)iters * 16 * (blocks / 256.0)));
      }
    }
  }
  return 0;
}
