// Developer tool: time pp_backward_input_ws_f32 through the C ABI without Python.
//   bwdx_bench [B] [method] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/pp_hip.h"
int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 16384, method = argc > 2 ? atoi(argv[2]) : 4;
  int iters = argc > 3 ? atoi(argv[3]) : 100;
  const int I = 250, O = 250, n = 3, S = 2, K = 5;
  std::vector<float> hx((size_t)B * I), hg((size_t)B * O), hw((size_t)I * K * O);
  unsigned st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = 1.3f * rnd();
  for (auto& v : hg) v = rnd() / O;
  for (auto& v : hw) v = rnd() / I;
  float *x, *g, *w, *gx; void* ws = nullptr;
  (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&g, hg.size() * 4);
  (void)hipMalloc(&w, hw.size() * 4); (void)hipMalloc(&gx, hx.size() * 4);
  size_t wsb = pp_backward_input_workspace_bytes(B, I, O, n, S, 0, 2.0f, method);
  if (wsb) (void)hipMalloc(&ws, wsb);
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto call = [&]() { return pp_backward_input_ws_f32(x, g, w, gx, ws, wsb, B, I, O, n, S, 0, 2.0f, nullptr, 1.0f, method, nullptr); };
  for (int i = 0; i < 5; ++i) { int rc = call(); if (rc) { printf("rc=%d %s\n", rc, pp_error_string(rc)); return 1; } }
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) call();
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("B=%d method=%d: %.1f us/call\n", B, method, ms * 1e3 / iters);
  return 0;
}
