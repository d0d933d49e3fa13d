// Calibration (developer tool): cost of v_mfma_f32_32x32x2_f32 back-to-back from
// registers, alone and with NV independent VALU instructions per MFMA in the same
// wave (does vector work hide under the f32 MFMA?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NV>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = a0 * j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], b, a);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      }
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n)
    for (int q = 0; q < 16; ++q) s += acc[n][q];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// random operands (8 values per lane, rotated) instead of constants: DVFS check
template <int NACC>
__global__ __launch_bounds__(256) void probe_rand(float* out, int iters, const float* tab) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) {
    a[j] = tab[(threadIdx.x * 8 + j) & 4095];
    b[j] = tab[(threadIdx.x * 8 + j + 2048) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * NACC + n) & 7], b[(u + n) & 7], acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n)
    for (int q = 0; q < 16; ++q) s += acc[n][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

void run_rand(int blocks, int iters) {
  float *out, *tab;
  (void)hipMalloc(&out, blocks * 256 * 4);
  (void)hipMalloc(&tab, 4096 * 4);
  float h[4096];
  unsigned st = 12345;
  for (int i = 0; i < 4096; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
  (void)hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    probe_rand<4><<<blocks, 256>>>(out, iters, tab);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double ns = ms * 1e6 / ((double)iters * 16 * (blocks / 256.0));
    printf("random operands blocks=%d: %.3f ms, %.2f ns per MFMA per SIMD\n", blocks, ms, ns);
  }
}

template <int NACC, int NV>
void run(int blocks, int iters) {
  float* out;
  (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<NACC, NV><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<NACC, NV><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double mfma_per_wave = (double)iters * 4 * NACC;
  double waves_per_simd = blocks / 256.0;
  double ns = ms * 1e6 / (mfma_per_wave * waves_per_simd);
  printf("NACC=%d NV=%d blocks=%d (%.0f waves/SIMD): %.2f ns per MFMA per SIMD\n", NACC, NV, blocks,
         waves_per_simd, ns);
  (void)hipFree(out);
}

int main() {
  run_rand(256, 20000);
  run_rand(512, 20000);
  run<4, 0>(256, 5000);
  run<4, 0>(256, 5000);
  run<4, 1>(256, 5000);
  run<4, 2>(256, 5000);
  run<4, 4>(256, 5000);
  run<4, 8>(256, 5000);
  run<4, 12>(256, 5000);
  run<4, 4>(512, 5000);
  run<4, 8>(512, 5000);
  return 0;
}
