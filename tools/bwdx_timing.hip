// Developer tool: phase timestamps of pp_bwd_x_bf3_kernel (built with -DPP_TIMING from
// the kernel source itself).  Prints per-phase durations for a few blocks.
#define PP_TIMING 1
#include "../language_interpolation_amd/csrc/pp_bf3_bwdx.hip"
#include <cstdio>
#include <vector>
#include <cmath>
bool pp_dense_supported(const PPShape&) { return true; }
hipError_t pp_launch_segment_ids(const float*, int32_t*, const PPShape&, hipStream_t) { return hipSuccess; }

int main(int argc, char** argv) {
  const float dscale = getenv("PP_DATA_SCALE") ? atof(getenv("PP_DATA_SCALE")) : 1.0f;
  int B = argc > 1 ? atoi(argv[1]) : 16384;
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
  std::vector<float> hx((size_t)B * sh.I), hw((size_t)sh.I * sh.K * sh.O), hg((size_t)B * sh.O);
  unsigned st = 1;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() / 250;
  for (auto& v : hg) v = rnd();
  float *x, *w, *g, *gx; void* ws;
  (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&w, hw.size() * 4);
  (void)hipMalloc(&g, hg.size() * 4); (void)hipMalloc(&gx, hx.size() * 4);
  (void)hipMalloc(&ws, pp_bf3_bwd_x_workspace_bytes(sh));
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
  for (int it = 0; it < 3; ++it) {
    hipError_t e = (getenv("H2") ? pp_launch_bwd_x_h2(x, g, w, gx, ws, sh, tab, false, 256, nullptr) : pp_launch_bwd_x_bf3(x, g, w, gx, ws, sh, tab, nullptr));
    if (e != hipSuccess) { printf("launch failed %d\n", (int)e); return 1; }
  }
  (void)hipDeviceSynchronize();
  int nb = (B + 63) / 64; if (nb > 8192) nb = 8192;
  std::vector<unsigned long long> t(8 * 8192);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(pp_dbg_times), t.size() * 8);
  unsigned long long t00 = ~0ull;
  for (int b = 0; b < nb; ++b) if (t[b * 8] < t00) t00 = t[b * 8];
  printf("wall_clock64 ticks are 10 ns.  block: start | x+gy issue | publish+barrier | chunk0 | pub1 | chunk1 | (bar) | epilogue  (us)\n");
  double sum[8] = {0};
  for (int b = 0; b < nb; ++b) {
    double d[8];
    d[0] = (t[b * 8] - t00) * 0.01;
    for (int j = 1; j < 8; ++j) d[j] = (double)(long long)(t[b * 8 + j] - t[b * 8 + j - 1]) * 0.01;
    for (int j = 0; j < 8; ++j) sum[j] += d[j];
    if (b < 4 || b % (nb / 8) == 0)
      printf("block %4d: %7.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f\n", b, d[0], d[1],
             d[2], d[3], d[4], d[5], d[6], d[7]);
  }
  {
    std::vector<unsigned long long> st(32 * 8192);
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(pp_dbg_steps), st.size() * 8);
    printf("step starts relative to the first step of the block (us), mean over blocks:\n");
    double wall = 0, shader = 0;
    for (int j = 0; j < 16; ++j) {
      double a = 0;
      for (int b = 0; b < nb; ++b) a += (double)(long long)(st[b * 32 + 2 * j] - st[b * 32]) * 0.01;
      printf(" %5.2f", a / nb);
    }
    for (int b = 0; b < nb; ++b) {
      wall += (double)(long long)(st[b * 32 + 14] - st[b * 32]) * 0.01;
      shader += (double)(long long)(st[b * 32 + 15] - st[b * 32 + 1]);
    }
    printf("\nshader clock over steps 0..7: %.0f MHz", shader / wall);
    printf("\n");
  }
  printf("mean      : %7.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f\n", sum[0] / nb,
         sum[1] / nb, sum[2] / nb, sum[3] / nb, sum[4] / nb, sum[5] / nb, sum[6] / nb, sum[7] / nb);
  return 0;
}
