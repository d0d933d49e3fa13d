// Hardware probe (gfx950): does a packed-fp32 multiply give a wrong result when other waves of
// the same SIMD are issuing MFMAs?  Found in pp_fwd_bf3_kernel<SchemeH2,7,7,1,2,2>, where
//   v_pk_mul_f32 v[48:49], v[92:93], v[98:99] op_sel:[0,1] op_sel_hi:[0,1]
// returned 0.0 in its LOW result for lanes 48..63 (sources non-zero) in ~1/3 of the workgroups
// whenever two workgroups shared a CU (4 waves per SIMD), and never with one workgroup per CU.
//
// Block = 512 threads.  Waves 0-3 ("checkers") run the packed multiply FORM on random operands
// and compare both halves with plain v_mul_f32; waves 4-7 ("matrix") run v_mfma_f32_32x32x16_f16
// back to back (MODE 1) or idle (MODE 0).  Wrong results are counted per lane quarter.
// Build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off tools/pk_mfma_probe.hip -o tools/pk_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// FORM 0: op_sel:[0,1] op_sel_hi:[0,1]  (lo = a.lo*b.hi, hi = a.lo*b.hi)
// FORM 1: plain                          (lo = a.lo*b.lo, hi = a.hi*b.hi)
// FORM 2: op_sel:[1,0] op_sel_hi:[0,1]  (lo = a.hi*b.lo, hi = a.lo*b.hi)
template <int FORM, int MODE>
__global__ __launch_bounds__(512, 2) void probe(unsigned* __restrict__ bad, float* __restrict__ sink,
                                               int iters) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  if (wv >= 4) {
    if (MODE == 0) return;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (_Float16)(0.001f * (lane + j));
      b[j] = (_Float16)(0.002f * (lane - j));
    }
    f32x16 c0, c1;
    for (int q = 0; q < 16; ++q) c0[q] = c1[q] = 0.0f;
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
    if (c0[0] + c1[3] == 12345.678f) sink[threadIdx.x] = c0[1];
    return;
  }
  unsigned st = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
  unsigned nlo = 0, nhi = 0, nzero = 0;
  for (int it = 0; it < iters; ++it) {
    st = st * 1664525u + 1013904223u;
    const float a0 = 0.5f + (float)(st >> 8) * (1.0f / 16777216.0f);
    st = st * 1664525u + 1013904223u;
    const float b0 = 0.5f + (float)(st >> 8) * (1.0f / 16777216.0f);
    const float a1 = a0 + 0.125f, b1 = b0 + 0.375f;
    f2 va = {a0, a1}, vb = {b0, b1};
    f2 r = {123.0f, 456.0f};
    float e0, e1;
    if (FORM == 0) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]\ns_nop 1" : "+v"(r) : "v"(va), "v"(vb));
      e0 = a0 * b1; e1 = a0 * b1;
    } else if (FORM == 1) {
      asm volatile("v_pk_mul_f32 %0, %1, %2\ns_nop 1" : "+v"(r) : "v"(va), "v"(vb));
      e0 = a0 * b0; e1 = a1 * b1;
    } else {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]\ns_nop 1" : "+v"(r) : "v"(va), "v"(vb));
      e0 = a1 * b0; e1 = a0 * b1;
    }
    nlo += (r.x != e0);
    nhi += (r.y != e1);
    nzero += (r.x == 0.0f) + (r.y == 0.0f);
  }
  if (nlo) atomicAdd(&bad[lane >> 4], nlo);
  if (nhi) atomicAdd(&bad[4 + (lane >> 4)], nhi);
  if (nzero) atomicAdd(&bad[8], nzero);
}

template <int FORM, int MODE>
void run(unsigned* d, float* sink, int nblk, int iters) {
  (void)hipMemset(d, 0, 64);
  hipLaunchKernelGGL((probe<FORM, MODE>), dim3(nblk), dim3(512), 0, nullptr, d, sink, iters);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[9];
  (void)hipMemcpy(h, d, 36, hipMemcpyDeviceToHost);
  printf("form=%d matrix_partner=%d blocks=%d iters=%d (%s): wrong lo by lane quarter %u %u %u %u | hi %u %u %u %u | zeros %u\n",
         FORM, MODE, nblk, iters, hipGetErrorString(e), h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
}

int main() {
  unsigned* d;
  float* sink;
  (void)hipMalloc(&d, 64);
  (void)hipMalloc(&sink, 4096);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 1>(d, sink, 512, 20000);
    run<1, 1>(d, sink, 512, 20000);
    run<2, 1>(d, sink, 512, 20000);
    run<0, 0>(d, sink, 512, 20000);
    run<0, 1>(d, sink, 256, 20000);
    run<0, 1>(d, sink, 2048, 20000);
  }
  return 0;
}
