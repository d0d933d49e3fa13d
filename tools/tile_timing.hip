// Developer tool: the row-tile forward (csrc/pp_tile.hip) with parts of the kernel switched off,
// timed with HIP events.  Build (from the repo root) and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize \
//     -fhip-fp32-correctly-rounded-divide-sqrt -DPP_TILE_DBG -Iinclude -Ilanguage_interpolation_amd/csrc \
//     tools/tile_timing.hip -o tools/tile_timing
//   tools/tile_timing [cfg3|cfg4|cfg1] [rows]
// dbg bits: 1 no slab DMA in the loop, 2 no gathers, 4 no wait+barrier, 16 no record loads.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../language_interpolation_amd/csrc/pp_tile.hip"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "cfg3";
  const int B = argc > 2 ? atoi(argv[2]) : 16384;
  int I = 200, O = 200, n = 1, S = 150, disc = 1;
  if (!strcmp(which, "cfg4")) { I = 16; O = 1000; n = 2; S = 128; }
  if (!strcmp(which, "cfg1")) { I = 16; O = 200; n = 2; S = 128; disc = 0; }
  if (!strcmp(which, "cfg2")) { I = 16; O = 250; n = 2; S = 128; disc = 0; }
  PPShape sh;
  sh.B = B; sh.I = I; sh.O = O; sh.n = n; sh.S = S;
  sh.K = disc ? n * S : (n - 1) * S + 1;
  sh.stride = disc ? n : n - 1; sh.disc = disc;
  sh.length = 2.0f; sh.half = 1.0f; sh.Sf = (float)S; sh.out_scale = 1.0f;
  BasisTab tab;
  for (int j = 0; j < PP_MAX_N; ++j) { tab.X[j] = 0; tab.invden[j] = 1; }
  if (n == 2) { tab.X[0] = -1; tab.X[1] = 1; tab.invden[0] = -0.5f; tab.invden[1] = 0.5f; }
  std::vector<float> hx((size_t)B * I), hw((size_t)I * sh.K * O);
  srand(1);
  for (auto& v : hx) {  // ~N(0,1) by a sum of uniforms (hidden layers), character grid for I = 16
    if (I == 16) v = ((rand() % 95 + 32) - 63.5f) / 64.0f;
    else { float s = 0; for (int k = 0; k < 12; ++k) s += rand() / (float)RAND_MAX; v = s - 6.0f; }
  }
  for (auto& v : hw) v = (rand() / (float)RAND_MAX * 2 - 1) / I;
  float *x, *w, *y, *ws;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4));
  CK(hipMalloc(&y, (size_t)B * O * 4));
  CK(hipMalloc(&ws, (size_t)64 * B * O * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct V { int nw, uw, nsplit, dbg; const char* what; int nbuf = 0; };
  std::vector<V> vs = {
      {0, 0, 0, 0, "plan as shipped"},
      {0, 0, 0, 1, "no slab DMA in the loop"},
      {0, 0, 0, 2, "no gathers"},
      {0, 0, 0, 4, "no wait + barrier"},
      {0, 0, 0, 16, "no record loads"},
      {0, 0, 0, 1 | 4, "no DMA, no barrier"},
      {0, 0, 0, 2 | 1 | 4, "no DMA, no barrier, no gathers"},
      {0, 0, 0, 1 | 2 | 4 | 16, "empty loop"},
      {16, 1, 0, 0, "16 waves x 64 rows"},
      {16, 1, 0, 0, "16 waves, 2 slab buffers", 2},
      {16, 1, 0, 0, "16 waves, 3 slab buffers", 3},
      {8, 2, 0, 0, "8 waves, 3 slab buffers", 3},
      {16, 1, 2, 0, "16 waves, 2 splits"},
      {16, 1, 3, 0, "16 waves, 3 splits"},
      {16, 1, 5, 0, "16 waves, 5 splits"},
      {16, 1, 8, 0, "16 waves, 8 splits"},
      {16, 1, 0, 1 | 2 | 4 | 16, "16 waves: empty loop"},
      {16, 1, 0, 1 | 4, "16 waves: no DMA, no barrier"},
      {16, 1, 0, 2, "16 waves: no gathers"},
      {8, 2, 1, 0, "R=1024, no input split"},
      {8, 2, 2, 0, "R=1024, 2 splits"},
      {8, 2, 8, 0, "R=1024, 8 splits"},
      {4, 2, 1, 0, "R=512, 1 split"},
      {4, 2, 2, 0, "R=512, 2 splits"},
      {4, 2, 4, 0, "R=512, 4 splits"},
      {4, 1, 1, 0, "R=256, 1 split"},
      {4, 1, 2, 0, "R=256, 2 splits"},
  };
  const int only = argc > 3 ? atoi(argv[3]) : -1;  // run one variant (for rocprofv3)
  int vi = -1;
  for (const V& v : vs) {
    if (++vi != only && only >= 0) continue;
    g_tile_nw = v.nw; g_tile_uw = v.uw; g_tile_nsplit = v.nsplit; g_tile_dbg = v.dbg; g_tile_nbuf = v.nbuf;
    const TilePlan p = tile_plan(sh);
    for (int it = 0; it < 3; ++it) CK(pp_launch_fwd_tile(x, w, y, nullptr, ws, nullptr, 0, sh, tab, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int iters = 20;
    for (int it = 0; it < iters; ++it) CK(pp_launch_fwd_tile(x, w, y, nullptr, ws, nullptr, 0, sh, tab, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s NW=%d UW=%d nbuf=%d nsplit=%2d blocks=%4d lds=%6zu : %8.1f us per call (main kernel%s)\n",
           v.what, p.NW, p.UW, p.nbuf, p.nsplit, p.nchunk * p.nsplit * p.ntile, p.lds, ms * 1e3 / iters,
           p.nsplit > 1 ? " + sum of partials" : "");
  }
  return 0;
}
