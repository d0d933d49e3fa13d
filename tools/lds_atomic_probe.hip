// Hardware probe (gfx950): throughput of LDS float atomics (ds_add_f32, no return) in the access
// shape of an LDS-resident accumulator table: every wave-instruction adds 64 consecutive floats
// (or two 32-float halves) into a randomly chosen row of the table.
//   mode 0: ds_add_f32, one random row of 64 floats per instruction
//   mode 1: ds_add_f32, two random rows of 32 floats per instruction (lanes 0-31 / 32-63)
//   mode 2: ds_read_b32 + ds_add_f32 pairs (source tile read + accumulate), rows of 64
//   mode 3: plain read-modify-write (ds_read_b32, v_add, ds_write_b32), rows of 64, no atomics
//   mode 4: ds_add_f32 where all waves hit the SAME row (worst-case contention)
// Build: hipcc -O2 --offload-arch=gfx950 tools/lds_atomic_probe.hip -o tools/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kRows = 512;  // rows of 64 floats = 128 KiB of accumulators
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* __restrict__ out, int iters,
                                             unsigned long long* __restrict__ cycles) {
  extern __shared__ float acc[];  // kRows * 64 (+ 4096 source floats for mode 2)
  for (int i = threadIdx.x; i < kRows * 64 + 4096; i += 256) acc[i] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned st = blockIdx.x * 7919u + (threadIdx.x >> 6) * 104729u + 17u;
  const float* src = acc + kRows * 64;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      st = st * 1664525u + 1013904223u;
      const int r0 = (st >> 10) % kRows;
      const int r1 = (st >> 20) % kRows;
      if (MODE == 0) {
        atomicAdd(&acc[r0 * 64 + lane], 1.0f);
      } else if (MODE == 1) {
        const int r = lane < 32 ? r0 : r1;
        atomicAdd(&acc[r * 64 + (lane & 31)], 1.0f);
      } else if (MODE == 2) {
        const float g = src[((st >> 5) & 63) * 64 + lane];
        atomicAdd(&acc[r0 * 64 + lane], g + 1.0f);
      } else if (MODE == 3) {
        acc[r0 * 64 + lane] += 1.0f;
      } else {
        atomicAdd(&acc[7 * 64 + lane], 1.0f);
      }
    }
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  float s = 0.0f;
  for (int i = threadIdx.x; i < kRows * 64; i += 256) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(float* out, unsigned long long* cyc, int nblk, int iters) {
  const size_t lds = (size_t)(kRows * 64 + 4096) * 4;
  (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((probe<MODE>), dim3(nblk), dim3(256), lds, nullptr, out, iters, cyc);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((probe<MODE>), dim3(nblk), dim3(256), lds, nullptr, out, iters, cyc);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  unsigned long long h[256];
  (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  float hs[256];
  (void)hipMemcpy(hs, out, sizeof hs, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < 256; ++i) sum += hs[i];
  const double instr = (double)iters * 8 * 4;  // wave-instructions per block (4 waves)
  printf("mode %d: %.1f us, %.2f shader cycles per wave-instruction per CU (4 waves/CU), check %.0f (expect %.0f)\n",
         MODE, ms * 1e3, (double)h[0] / instr, sum, instr * 64);
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&cyc, 256 * 8);
  run<0>(out, cyc, 256, 2000);
  run<1>(out, cyc, 256, 2000);
  run<2>(out, cyc, 256, 2000);
  run<3>(out, cyc, 256, 2000);
  run<4>(out, cyc, 256, 2000);
  return 0;
}
