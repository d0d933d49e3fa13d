// Calibration (developer tool): what issues for free under v_mfma_f32_32x32x16_bf16?
// One wave per SIMD (256 blocks) or two (512); per MFMA the wave also issues
//   NV VALU ops (kind 0: v_fma_f32, 1: v_and_b32/v_sub_f32 pairs), ND ds_read_b128,
//   NG buffer/global b128 loads (L2-resident table), forced to interleave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND, int ND, int NG>
__global__ __launch_bounds__(256) void probe(float* out, const u32x4* __restrict__ tab, int iters,
                                             float a0) {
  __shared__ u32x4 L[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) L[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n)
    for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
  u32x4 ua = {threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = a;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = a0 * j + threadIdx.x;
  u32x4 d[4] = {}, g[4] = {};
  const u32x4* tp = tab + (blockIdx.x & 7) * 4096 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    u32x4 dn[4], gn[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if (n < ND * 4 / 4 && ND > 0)
        for (int j = 0; j < ND; ++j) dn[(n * ND + j) & 3] = L[(threadIdx.x + 64 * n + 256 * j + it * 8) & 2047];
      if (NG > 0)
        for (int j = 0; j < NG; ++j) gn[(n * NG + j) & 3] = tp[((it * 4 + n) * NG + j) * 256 & 3840];
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (KIND == 0) {
          v[j & 7] = __builtin_fmaf(v[j & 7], a0, 1.0f);
        } else {
          v[j & 7] = (j & 1) ? __uint_as_float(__float_as_uint(v[j & 7]) & 0xffff0000u)
                             : v[j & 7] - a0;
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (ND) __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
      if (NG) __builtin_amdgcn_sched_group_barrier(0x020, NG, 0);
      if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
    for (int j = 0; j < 4; ++j) {
      if (ND) d[j] = d[j] ^ dn[j];
      if (NG) g[j] = g[j] ^ gn[j];
    }
  }
  float s = 0;
  for (int n = 0; n < 4; ++n)
    for (int q = 0; q < 16; ++q) s += acc[n][q];
  for (int j = 0; j < 8; ++j) s += v[j];
  for (int j = 0; j < 4; ++j) s += (float)(d[j][0] + g[j][1]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static u32x4* g_tab;

template <int NV, int KIND, int ND, int NG>
void run(int blocks, int iters) {
  float* out;
  (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<NV, KIND, ND, NG><<<blocks, 256>>>(out, g_tab, iters, 1.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<NV, KIND, ND, NG><<<blocks, 256>>>(out, g_tab, iters, 1.f);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double ns = ms * 1e6 / ((double)iters * 4 * (blocks / 256.0));
  printf("NV=%d kind=%d ND=%d NG=%d waves/SIMD=%d: %.2f ns per MFMA per SIMD\n", NV, KIND, ND, NG,
         blocks / 256, ns);
  (void)hipFree(out);
}

int main() {
  (void)hipMalloc(&g_tab, 8 * 4096 * 16);
  (void)hipMemset(g_tab, 1, 8 * 4096 * 16);
  const int it = 20000;
  run<0, 0, 0, 0>(256, it);
  run<0, 0, 0, 0>(256, it);
  run<0, 0, 0, 0>(512, it);
  run<2, 0, 0, 0>(256, it);
  run<4, 0, 0, 0>(256, it);
  run<8, 0, 0, 0>(256, it);
  run<8, 0, 0, 0>(512, it);
  run<2, 1, 0, 0>(256, it);
  run<4, 1, 0, 0>(256, it);
  run<8, 1, 0, 0>(256, it);
  run<8, 1, 0, 0>(512, it);
  run<0, 0, 1, 0>(256, it);
  run<0, 0, 1, 0>(512, it);
  run<0, 0, 2, 0>(256, it);
  run<0, 0, 0, 1>(256, it);
  run<0, 0, 0, 1>(512, it);
  run<0, 0, 0, 2>(256, it);
  run<4, 1, 1, 1>(256, it);
  run<4, 1, 1, 1>(512, it);
  return 0;
}
