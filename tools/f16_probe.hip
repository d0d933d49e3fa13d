// Calibration (developer tool): v_mfma_f32_32x32x16_f16 -- are fp16 subnormal inputs kept,
// what does v_cvt_pkrtz do on overflow, and the issue rate on random data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void sub_probe(float* out, float a, float b) {
  // every A element = a (fp16), every B element = b: C[i][j] = 16 * a * b
  _Float16 ha = (_Float16)a, hb = (_Float16)b;
  f16x8 A, B;
  for (int j = 0; j < 8; ++j) { A[j] = ha; B[j] = hb; }
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)ha; out[2] = (float)hb; }
  auto pk = __builtin_amdgcn_cvt_pkrtz(a * 1e9f, -b * 1e9f);
  if (threadIdx.x == 0) { out[3] = (float)pk[0]; out[4] = (float)pk[1]; }
}

__global__ __launch_bounds__(256) void rate_probe(float* out, const u32x4* tab, int iters) {
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n)
    for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
  f16x8 a[4], b[4];
  for (int j = 0; j < 4; ++j) {
    a[j] = __builtin_bit_cast(f16x8, tab[(threadIdx.x * 4 + j) & 4095]);
    b[j] = __builtin_bit_cast(f16x8, tab[(threadIdx.x * 4 + j + 1777) & 4095]);
  }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + n) & 3], b[u], acc[n], 0, 0, 0);
  float s = 0;
  for (int n = 0; n < 4; ++n)
    for (int q = 0; q < 16; ++q) s += acc[n][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *out, h[8];
  (void)hipMalloc(&out, 1024 * 256 * 4);
  const float tests[][2] = {{1.0f, 1.0f}, {5.9604645e-8f, 1024.0f}, {3.0e-6f, 1024.0f}, {6.2e-5f, 1.0f}};
  for (auto& t : tests) {
    sub_probe<<<1, 64>>>(out, t[0], t[1]);
    (void)hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("a=%g b=%g: fp16(a)=%g fp16(b)=%g  mfma C=%g (expect %g)   pkrtz(a*1e9,-b*1e9)=(%g,%g)\n", t[0], t[1],
           h[1], h[2], h[0], 16.0 * h[1] * h[2], h[3], h[4]);
  }
  u32x4* tab;
  (void)hipMalloc(&tab, 4096 * 16);
  static unsigned hh[4096 * 4];
  unsigned st = 777;
  for (int i = 0; i < 4096 * 4; ++i) {
    st = st * 1664525u + 1013904223u;
    // random fp16 in [-2,2): sign, exponent 13..15, random 10-bit mantissa
    unsigned lo = ((st >> 3) & 0x83ff) | ((13 + ((st >> 20) % 3)) << 10);
    unsigned hi = ((st >> 9) & 0x83ff) | ((13 + ((st >> 24) % 3)) << 10);
    hh[i] = lo | (hi << 16);
  }
  (void)hipMemcpy(tab, hh, sizeof(hh), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int blocks : {256, 512}) {
    rate_probe<<<blocks, 256>>>(out, tab, 20000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    rate_probe<<<blocks, 256>>>(out, tab, 20000);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("f16 32x32x16 random data, waves/SIMD=%d: %.2f ns per MFMA per SIMD\n", blocks / 256,
           ms * 1e6 / (20000.0 * 16 * (blocks / 256.0)));
  }
  return 0;
}
