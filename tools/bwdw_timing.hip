// Developer tool: run time of pp_bwd_w_bf3_kernel (fp16x2, cfg2 hidden layer) with parts of its
// step removed at compile time (-DPP_NO_XLOAD / _BLOAD / _ALOAD / _PRODUCE / _MFMA / _BARRIER):
// what the step waits for.  Results are wrong in every ablated build; only the time is read.
#include "../language_interpolation_amd/csrc/pp_bf3_bwdw.hip"
#include <cstdio>
#include <vector>
bool pp_dense_supported(const PPShape&) { return true; }
void pp_launch_chunk_reduce(const float*, float*, long, int, float, hipStream_t) {}

int main(int argc, char** argv) {
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
  std::vector<float> hx((size_t)B * sh.I), hg((size_t)B * sh.O);
  unsigned st = 1;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = 1.3f * rnd();
  for (auto& v : hg) v = rnd();
  float *x, *g, *gw; void* ws;
  (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&g, hg.size() * 4);
  (void)hipMalloc(&gw, (size_t)sh.I * sh.K * sh.O * 4);
  (void)hipMalloc(&ws, pp_bf3_bwd_w_workspace_bytes(sh));
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it)
    if (pp_launch_bwd_w_h2(x, g, gw, ws, sh, tab, nullptr) != hipSuccess) { printf("launch failed\n"); return 1; }
  (void)hipDeviceSynchronize();
  const int reps = 20;
  (void)hipEventRecord(e0, nullptr);
  for (int it = 0; it < reps; ++it) (void)pp_launch_bwd_w_h2(x, g, gw, ws, sh, tab, nullptr);
  (void)hipEventRecord(e1, nullptr);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.1f us per call (gy pack + main kernel; the partial-sum kernel is stubbed out)\n",
         argc > 2 ? argv[2] : "full", ms * 1000 / reps);
  return 0;
}
