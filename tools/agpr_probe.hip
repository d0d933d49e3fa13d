// Calibration (developer tool): does v_mfma_f32_32x32x16_f16 issue at the same rate with its
// accumulators in AGPRs as in VGPRs, and with the B operand in AGPRs?  Random fp16 data,
// NACC independent accumulators in round robin, one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>  // MODE 0: C in VGPR; 1: C in AGPR; 2: C and B in AGPR
__global__ __launch_bounds__(256) void rate_probe(float* out, const u32x4* tab, int iters) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
  f16x8 a[2], b[4];
  for (int j = 0; j < 2; ++j) a[j] = __builtin_bit_cast(f16x8, tab[(threadIdx.x * 4 + j) & 4095]);
  for (int j = 0; j < 4; ++j) b[j] = __builtin_bit_cast(f16x8, tab[(threadIdx.x * 4 + j + 1777) & 4095]);
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        if constexpr (MODE == 0)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a[n & 1]), "v"(b[u]));
        else if constexpr (MODE == 1)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(a[n & 1]), "v"(b[u]));
        else
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(a[n & 1]), "a"(b[u]));
      }
  float s = 0;
  for (int n = 0; n < NACC; ++n)
    for (int q = 0; q < 16; ++q) s += acc[n][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* what, float* out, const u32x4* tab) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  for (int blocks : {256, 512}) {
    if (blocks == 512 && NACC * 16 > 100) continue;  // two waves per SIMD need <= 256 registers
    rate_probe<MODE, NACC><<<blocks, 256>>>(out, tab, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    rate_probe<MODE, NACC><<<blocks, 256>>>(out, tab, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s NACC=%2d waves/SIMD=%d: %.2f ns per MFMA per SIMD\n", what, NACC, blocks / 256,
           ms * 1e6 / ((double)iters * 4 * NACC * (blocks / 256.0)));
  }
}

// the backward-input kernel's operand pattern: K*PL = 10 B fragments, RM = 2 A fragments per
// plane, three cross terms (A plane pa, B plane pb), 10 accumulators
template <int MODE>
__global__ __launch_bounds__(256) void pattern_probe(float* out, const u32x4* tab, int iters) {
  f32x16 acc[5][2];
  for (int k = 0; k < 5; ++k)
    for (int r = 0; r < 2; ++r)
      for (int q = 0; q < 16; ++q) acc[k][r][q] = 0.f;
  f16x8 a[2][2], b[5][2];
  for (int r = 0; r < 2; ++r)
    for (int p = 0; p < 2; ++p) a[r][p] = __builtin_bit_cast(f16x8, tab[(threadIdx.x * 4 + r * 2 + p) & 4095]);
  for (int k = 0; k < 5; ++k)
    for (int p = 0; p < 2; ++p) b[k][p] = __builtin_bit_cast(f16x8, tab[(threadIdx.x * 4 + k * 2 + p + 1777) & 4095]);
  const int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if constexpr (MODE == 0)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[k][r]) : "v"(a[r][pa[c]]), "v"(b[k][pb[c]]));
          else
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[k][r]) : "v"(a[r][pa[c]]), "a"(b[k][pb[c]]));
        }
  float s = 0;
  for (int k = 0; k < 5; ++k)
    for (int r = 0; r < 2; ++r)
      for (int q = 0; q < 16; ++q) s += acc[k][r][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run_pattern(const char* what, float* out, const u32x4* tab) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  pattern_probe<MODE><<<256, 256>>>(out, tab, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  pattern_probe<MODE><<<256, 256>>>(out, tab, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s: %.2f ns per MFMA per SIMD\n", what, ms * 1e6 / ((double)iters * 30));
}

int main() {
  float* out;
  (void)hipMalloc(&out, 1024 * 256 * 4);
  u32x4* tab;
  (void)hipMalloc(&tab, 4096 * 16);
  static unsigned hh[4096 * 4];
  unsigned st = 777;
  for (int i = 0; i < 4096 * 4; ++i) {
    st = st * 1664525u + 1013904223u;
    unsigned lo = ((st >> 3) & 0x83ff) | ((13 + ((st >> 20) % 3)) << 10);
    unsigned hi = ((st >> 9) & 0x83ff) | ((13 + ((st >> 24) % 3)) << 10);
    hh[i] = lo | (hi << 16);
  }
  (void)hipMemcpy(tab, hh, sizeof(hh), hipMemcpyHostToDevice);
  run<0, 4>("C in VGPR", out, tab);
  run<1, 4>("C in AGPR", out, tab);
  run<2, 4>("C and B in AGPR", out, tab);
  run<0, 10>("C in VGPR", out, tab);
  run<1, 10>("C in AGPR", out, tab);
  run<2, 10>("C and B in AGPR", out, tab);
  run_pattern<0>("kernel pattern, B in VGPR", out, tab);
  run_pattern<1>("kernel pattern, B in AGPR", out, tab);
  return 0;
}
