// Developer probe: what a wave-wide LDS read costs when all 64 lanes read the SAME address
// (broadcast of a per-row record) against one address per lane, b32 / b64 / b128, 8 waves a CU.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_bcast_probe.hip -o tools/lds_bcast_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <int WIDTH, bool BCAST>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  typedef __attribute__((address_space(3))) void lv;
  const unsigned base = (unsigned)(__UINTPTR_TYPE__)(lv*)lds;
  const unsigned addr = base + (BCAST ? 0u : (unsigned)lane * (WIDTH * 4));
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (WIDTH == 1) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    } else if constexpr (WIDTH == 2) {
      float2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
    } else {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].w;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int WIDTH, bool BCAST>
int run(float* out, const char* what) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<WIDTH, BCAST>), dim3(blocks), dim3(512), 0, 0, out, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((probe<WIDTH, BCAST>), dim3(blocks), dim3(512), 0, 0, out, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // per CU: 8 waves x iters x 8 reads
  const double reads = 8.0 * iters * 8;
  printf("%-28s %7.1f us  -> %.2f ns per wave-read per CU (%.1f cycles at 2.4 GHz)\n", what, ms * 1e3,
         ms * 1e6 / reads, ms * 1e6 / reads * 2.4);
  return 0;
}

int main() {
  float* out;
  CK(hipMalloc(&out, 256 * 512 * 4));
  run<1, false>(out, "b32, one address a lane");
  run<1, true>(out, "b32, broadcast");
  run<2, false>(out, "b64, one address a lane");
  run<2, true>(out, "b64, broadcast");
  run<4, false>(out, "b128, one address a lane");
  run<4, true>(out, "b128, broadcast");
  return 0;
}
