// Batches of character windows straight from the text held on the device.
//
// The reference builds every window of `features` characters + the next one as Python lists of
// ord() values (generate_dataset, /root/reference/language_interpolation/single_text_dataset.py:
// 102-123), normalises them with (c + 0.5 - 64) / 64 (:15-16) and lets DataLoader workers collate
// them; for the Mamba model the windows stay indices and are shifted by one for the targets
// (MambaMixin.collate_fn, lightning_datamodule.py:272-291).  Here the text is one uint8 buffer
// in HBM and a batch is one kernel: thread (b, j) reads text[idx[b] + j] and writes whichever
// of the three outputs the caller asked for.
#include "pp_device.h"

namespace {

__global__ __launch_bounds__(256) void pp_char_windows_kernel(const uint8_t* __restrict__ text,
                                                              long last_start,
                                                              const long* __restrict__ idx, int B,
                                                              int F, float* __restrict__ x,
                                                              long* __restrict__ tokens,
                                                              long* __restrict__ y) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int W = F + 1;
  if (t >= (long)B * W) return;
  const int b = (int)(t / W), j = (int)(t - (long)b * W);
  long start = idx[b];
  start = start < 0 ? 0 : (start > last_start ? last_start : start);  // never outside the text
  const int c = text[start + j];
  if (tokens) tokens[t] = c;
  if (j < F) {
    if (x) x[(size_t)b * F + j] = ((float)c - 64.0f + 0.5f) / 64.0f;
  } else if (y) {
    y[b] = c;
  }
}

}  // namespace

hipError_t pp_launch_char_windows(const uint8_t* text, long text_len, const long* idx, int B, int F,
                                  float* x, long* tokens, long* y, hipStream_t st) {
  const long total = (long)B * (F + 1);
  hipLaunchKernelGGL(pp_char_windows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                     text, text_len - (F + 1), idx, B, F, x, tokens, y);
  return hipGetLastError();
}
