// Hardware probe (gfx950): two dependent packed-fp32 multiplies back to back, the first with
// op_sel:[0,1] op_sel_hi:[0,1] (hipcc puts no wait state between them in that form), as found in
// pp_fwd_bf3_kernel<SchemeH2,7,7,1,2,2>:
//   v_pk_mul_f32 v[48:49], v[92:93], v[98:99] op_sel:[0,1] op_sel_hi:[0,1]
//   v_pk_mul_f32 v[48:49], v[94:95], v[48:49]
//   s_nop 0
//   v_pk_mul_f32 v[48:49], s[62:63], v[48:49]
// Each lane compares the packed result with the same product from plain v_mul_f32.
// Build: hipcc -O2 --offload-arch=gfx950 tools/pk_dep_probe.hip -o tools/pk_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int NOP>
__global__ __launch_bounds__(512) void probe(unsigned* __restrict__ bad, int iters, float s_lo,
                                             float s_hi) {
  const int lane = threadIdx.x & 63;
  unsigned st = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
  unsigned nbad_lo = 0, nbad_hi = 0;
  f2 sv = {s_lo, s_hi};
  for (int it = 0; it < iters; ++it) {
    st = st * 1664525u + 1013904223u;
    const float a = 0.5f + (float)(st >> 8) * (1.0f / 16777216.0f);
    st = st * 1664525u + 1013904223u;
    const float b = 0.5f + (float)(st >> 8) * (1.0f / 16777216.0f);
    st = st * 1664525u + 1013904223u;
    const float c0 = 0.5f + (float)(st >> 8) * (1.0f / 16777216.0f);
    const float c1 = c0 + 0.25f;
    f2 va = {a, -a}, vb = {-b, b}, vc = {c0, c1};
    f2 r = {123.0f + a, 456.0f + b};  // "stale" contents of the destination pair
    // a dependent use of r first, as in the kernel (the pair held the running maximum)
    float pre;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(pre) : "v"(r.x), "v"(r.y));
    if (NOP == 0)
      asm volatile(
          "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]\n"
          "v_pk_mul_f32 %0, %3, %0\n"
          "s_nop 0\n"
          "v_pk_mul_f32 %0, %4, %0\n"
          "s_nop 1\n"
          : "+v"(r)
          : "v"(va), "v"(vb), "v"(vc), "s"(sv));
    else
      asm volatile(
          "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]\n"
          "s_nop 0\n"
          "v_pk_mul_f32 %0, %3, %0\n"
          "s_nop 0\n"
          "v_pk_mul_f32 %0, %4, %0\n"
          "s_nop 1\n"
          : "+v"(r)
          : "v"(va), "v"(vb), "v"(vc), "s"(sv));
    const float p = a * b;  // lo(src0) * hi(src1) for both halves
    const float e0 = s_lo * (c0 * p), e1 = s_hi * (c1 * p);
    nbad_lo += (r.x != e0);
    nbad_hi += (r.y != e1);
    if (pre == -1.0f) nbad_lo += 1000;
  }
  if (nbad_lo) atomicAdd(&bad[(lane >> 4)], nbad_lo);
  if (nbad_hi) atomicAdd(&bad[4 + (lane >> 4)], nbad_hi);
}

template <int NOP>
void run(unsigned* d, int nblk, int iters) {
  (void)hipMemset(d, 0, 64);
  hipLaunchKernelGGL((probe<NOP>), dim3(nblk), dim3(512), 0, nullptr, d, iters, 1.5f, 0.75f);
  (void)hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("nop_between=%d blocks=%d iters=%d: wrong lo results by lane quarter %u %u %u %u | hi %u %u %u %u\n",
         NOP, nblk, iters, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}

int main() {
  unsigned* d;
  (void)hipMalloc(&d, 64);
  for (int rep = 0; rep < 3; ++rep) {
    run<0>(d, 1024, 20000);
    run<1>(d, 1024, 20000);
  }
  run<0>(d, 16, 20000);
  run<0>(d, 257, 20000);
  return 0;
}
