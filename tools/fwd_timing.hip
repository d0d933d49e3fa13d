// Developer tool: phase timestamps of the split-operand forward kernel (built with
// -DPP_TIMING from the kernel source itself).
#define PP_TIMING 1
#include "../language_interpolation_amd/csrc/pp_bf3_fwd.hip"
#include <cstdio>
#include <vector>
bool pp_dense_supported(const PPShape&) { return true; }
hipError_t pp_launch_segment_ids(const float*, int32_t*, const PPShape&, hipStream_t) { return hipSuccess; }

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 16384;
  int scheme = argc > 2 ? atoi(argv[2]) : 4;
  int ln = argc > 3 ? atoi(argv[3]) : 0;
  PPShape sh{};
  sh.B = B; sh.I = 250; sh.O = 250; sh.n = 3; sh.S = 2; sh.K = 5; sh.stride = 2; sh.disc = 0;
  sh.length = 2.0f; sh.half = 1.0f; sh.Sf = 2.0f; sh.out_scale = 1.0f;
  BasisTab tab{};
  float X[3] = {-1.f, 0.f, 1.f};
  for (int j = 0; j < 3; ++j) {
    tab.X[j] = X[j];
    float d = 1.f;
    for (int m = 0; m < 3; ++m) if (m != j) d *= X[j] - X[m];
    tab.invden[j] = 1.f / d;
  }
  std::vector<float> hx((size_t)B * sh.I), hw((size_t)sh.I * sh.K * sh.O);
  unsigned st = 1;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = 1.3f * rnd();
  for (auto& v : hw) v = rnd() / 250;
  float *x, *w, *y; void* ws;
  (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&w, hw.size() * 4);
  (void)hipMalloc(&y, (size_t)B * sh.O * 4);
  (void)hipMalloc(&ws, pp_bf3_fwd_workspace_bytes(sh));
  float* rstd; (void)hipMalloc(&rstd, (size_t)B * 4);
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  for (int it = 0; it < 3; ++it) {
    hipError_t e = scheme == 4 ? pp_launch_fwd_h2(x, w, y, nullptr, ws, sh, tab, false, ln ? rstd : nullptr, 1e-5f, nullptr)
                               : pp_launch_fwd_bf3(x, w, y, nullptr, ws, sh, tab, nullptr);
    if (e != hipSuccess) { printf("launch failed %d\n", (int)e); return 1; }
  }
  (void)hipDeviceSynchronize();
  int nb = (B + 63) / 64; if (nb > 8192) nb = 8192;
  std::vector<unsigned long long> t(8 * 8192);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(pp_dbg_times), t.size() * 8);
  unsigned long long t00 = ~0ull;
  for (int b = 0; b < nb; ++b) if (t[b * 8] < t00) t00 = t[b * 8];
  printf("ln=%d ", ln);
  printf("scheme %d  block: start | prologue | main loop | exchange+scale | LayerNorm | store  (us)\n", scheme);
  const int order[6] = {0, 1, 2, 4, 5, 3};  // tick slots in program order
  double sum[6] = {0};
  for (int b = 0; b < nb; ++b) {
    double d[6];
    d[0] = (t[b * 8] - t00) * 0.01;
    for (int j = 1; j < 6; ++j)
      d[j] = (double)(long long)(t[b * 8 + order[j]] - t[b * 8 + order[j - 1]]) * 0.01;
    for (int j = 0; j < 6; ++j) sum[j] += d[j];
    if (b < 3 || b % (nb / 4) == 0)
      printf("block %4d: %7.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f\n", b, d[0], d[1], d[2], d[3], d[4], d[5]);
  }
  printf("mean      : %7.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f\n", sum[0] / nb, sum[1] / nb, sum[2] / nb,
         sum[3] / nb, sum[4] / nb, sum[5] / nb);
  return 0;
}
