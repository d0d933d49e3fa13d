// Calibration: how many ns does a v_mfma_f32_32x32x2_f32 take on this chip,
// back-to-back from registers, at 1, 2 and 4 waves per SIMD?  (developer tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n)
    for (int q = 0; q < 16; ++q) s += acc[n][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters) {
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double mfma_per_wave = (double)iters * 4 * NACC;
  double waves_per_simd = blocks / 256.0;
  double ns = ms * 1e6 / (mfma_per_wave * waves_per_simd);
  double tf = (double)blocks * 4 * mfma_per_wave * 32 * 32 * 2 * 2 / (ms * 1e-3) * 1e-12;
  printf("NACC=%d blocks=%d (%.0f waves/SIMD): %.3f ms, %.2f ns per MFMA per SIMD, %.1f TFLOP/s\n",
         NACC, blocks, waves_per_simd, ms, ns, tf);
  hipFree(out);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<1>(256, 20000);
    run<2>(256, 10000);
    run<4>(256, 5000);
    run<4>(512, 5000);
    run<4>(1024, 5000);
    run<1>(1024, 20000);
  }
  return 0;
}
