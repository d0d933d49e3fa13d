// Developer tool: time pp_forward_ws_f32 through the C ABI without Python in the way.
//   fwd_bench [B] [I] [O] [n] [S] [disc] [method] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/pp_hip.h"

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 16384, I = argc > 2 ? atoi(argv[2]) : 250;
  int O = argc > 3 ? atoi(argv[3]) : 250, n = argc > 4 ? atoi(argv[4]) : 3;
  int S = argc > 5 ? atoi(argv[5]) : 2, disc = argc > 6 ? atoi(argv[6]) : 0;
  int method = argc > 7 ? atoi(argv[7]) : 3, iters = argc > 8 ? atoi(argv[8]) : 50;
  int K = disc ? n * S : (n - 1) * S + 1;
  std::vector<float> hx((size_t)B * I), hw((size_t)I * K * O);
  unsigned st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = 1.3f * rnd();
  for (auto& v : hw) v = rnd() / I;
  float *x, *w, *y;
  void* ws = nullptr;
  (void)hipMalloc(&x, hx.size() * 4);
  (void)hipMalloc(&w, hw.size() * 4);
  (void)hipMalloc(&y, (size_t)B * O * 4);
  size_t wsb = pp_forward_workspace_bytes(B, I, O, n, S, disc, 2.0f, method);
  if (wsb) (void)hipMalloc(&ws, wsb);
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto call = [&]() {
    return pp_forward_ws_f32(x, w, y, nullptr, ws, wsb, B, I, O, n, S, disc, 2.0f, nullptr, 1.0f,
                             method, nullptr);
  };
  for (int i = 0; i < 5; ++i) {
    int rc = call();
    if (rc) { printf("rc=%d %s\n", rc, pp_error_string(rc)); return 1; }
  }
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) call();
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / iters;
  printf("B=%d I=%d O=%d n=%d S=%d disc=%d method=%d: %.1f us/call  %.1f TFLOP/s (algorithmic)\n", B,
         I, O, n, S, disc, method, us, 2.0 * B * I * O * n / us * 1e-6);
  return 0;
}
