// Stand-alone comparison of the shared-MLP GEMM kernels on the ScoreNet layer shapes (batch of 8 scenes):
//   v1  = mlp.hip:mlp_gemm_kernel (register-staged double buffer, 128x128 tile, 4 workgroups / CU)
//   g2* = gemm2.h:gemm2_kernel variants (LDS-DMA ring, counted waits, optional tail split)
// Interleaved rounds in one process (variants x rounds), median of the rounds; every variant is checked against v1.
// With -DMLP_TRACE=1 also dumps per-workgroup timelines of v1 (gpurun_out/g2/trace_*.bin) and times the start-up
// stagger.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMLP_TRACE=1 g2_bench.cpp -o g2_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "../../regnet_for_3d_grasping_amd/csrc/mlp.hip"
#include "../../regnet_for_3d_grasping_amd/csrc/gemm2.h"

struct Shape { long long P; int K, N; int pool; };

template <int TBM, int TBN, int WM, int WN, int STAGES, int OCC, bool POOL>
static void launch_g2(const float* A, const float* W, const float* sc, const float* sh, float* C, const Shape& s, bool tail) {
  G2Args a = {};
  a.A = A; a.lda = s.K; a.Ka = s.K; a.W = W; a.Kpad = s.K; a.scale = sc; a.shift = sh;
  a.C = C; a.ldc = s.N; a.P = s.P; a.N = s.N; a.relu = 1;
  a.tiles_m = (int)((s.P + TBM - 1) / TBM); a.tiles_n = (s.N + TBN - 1) / TBN;
  const long long tiles = (long long)a.tiles_m * a.tiles_n;
  const long long slots = 256ll * OCC;
  long long main_blocks = tiles, tail_tiles = 0;
  constexpr bool can_split = !POOL && (TBM / WM / 32 == 2) && ((WM * 32 + TBN) / 16) % (WM * WN) == 0;
  if (tail && can_split) {
    const long long rem = tiles % slots;
    if (rem > 0 && rem * 2 <= slots && tiles > slots) { main_blocks = tiles - rem; tail_tiles = rem; }
  }
  a.main_blocks = (int)main_blocks; a.tail_tiles = (int)tail_tiles; a.tail_split = 2;
  const unsigned grid = (unsigned)(main_blocks + tail_tiles * 2);
  hipLaunchKernelGGL((gemm2_kernel<TBM, TBN, WM, WN, STAGES, OCC, POOL>), dim3(grid), dim3(WM * WN * 64), 0, 0, a);
}

int main(int argc, char** argv) {
  std::vector<Shape> shapes = {{204800, 256, 256, 0}, {204800, 256, 512, 0}, {204800, 512, 256, 0}, {204800, 256, 128, 0},
                               {524288, 256, 512, 64}, {524288, 256, 256, 0}, {131072, 512, 1024, 64}, {131072, 512, 512, 0},
                               {40960, 512, 512, 0}, {40960, 256, 512, 0}, {8192, 1024, 1024, 0}, {8192, 1024, 512, 0}};
  if (argc > 1 && std::string(argv[1]) == "medium")   // the small / medium layers of a step (FP1, FP2, U / Ys producers)
    shapes = {{40960, 512, 512, 0}, {40960, 256, 512, 0}, {40960, 512, 256, 0}, {40960, 272, 256, 0}, {8192, 1024, 1024, 0},
              {8192, 1024, 512, 0}, {8192, 512, 1024, 0}, {8192, 528, 512, 0}, {2048, 1024, 1024, 0}};
  size_t maxA = 0, maxC = 0, maxW = 0;
  for (auto& s : shapes) {
    maxA = std::max(maxA, (size_t)s.P * s.K); maxC = std::max(maxC, (size_t)s.P * s.N); maxW = std::max(maxW, (size_t)s.N * s.K);
  }
  float *A, *W, *C, *C2, *sc, *sh;
  hipMalloc(&A, maxA * 4); hipMalloc(&C, maxC * 4); hipMalloc(&C2, maxC * 4); hipMalloc(&W, maxW * 4 + 1024 * 2048 * 4);
  hipMalloc(&sc, 4096 * 4); hipMalloc(&sh, 4096 * 4);
  std::vector<float> h(1 << 22);
  srand(3);
  for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  for (size_t off = 0; off < maxA; off += h.size()) hipMemcpy(A + off, h.data(), std::min(h.size(), maxA - off) * 4, hipMemcpyHostToDevice);
  for (size_t off = 0; off < maxW; off += h.size() / 2) hipMemcpy(W + off, h.data() + 77, std::min(h.size() / 2, maxW - off) * 4, hipMemcpyHostToDevice);
  hipMemcpy(sc, h.data(), 4096 * 4, hipMemcpyHostToDevice); hipMemcpy(sh, h.data() + 5000, 4096 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);

  struct Variant { std::string name; std::function<void(const Shape&, float*)> run; bool pool_ok; };
  std::vector<Variant> vars;
  vars.push_back({"v1", [&](const Shape& s, float* out) { regnet_mlp_layer_f32(A, s.K, s.K, W, s.K, sc, sh, out, s.N, s.P, s.N, 1, s.pool, nullptr); }, true});
#define G2V(NAME, M_, N_, WM, WN, ST, OCC, TAIL)                                                                    \
  vars.push_back({NAME, [&](const Shape& s, float* out) {                                                           \
                    if (s.pool) launch_g2<M_, N_, WM, WN, ST, OCC, (M_ / WM == 64)>(A, W, sc, sh, out, s, TAIL);     \
                    else launch_g2<M_, N_, WM, WN, ST, OCC, false>(A, W, sc, sh, out, s, TAIL);                       \
                  }, (M_ / WM == 64)});
  G2V("g2_256x128_s3", 256, 128, 4, 2, 3, 2, false)
  G2V("g2_256x128_s3_tail", 256, 128, 4, 2, 3, 2, true)
  G2V("g2_128x128_s3", 128, 128, 2, 2, 3, 3, false)
  G2V("g2_128x128_s2", 128, 128, 2, 2, 2, 4, false)
  G2V("g2_128x128_s2_tail", 128, 128, 2, 2, 2, 4, true)
  G2V("g2_256x256_s4", 256, 256, 2, 4, 4, 1, false)
  G2V("g2_128x256_s3", 128, 256, 2, 4, 3, 2, false)
  G2V("g2_64x128_s3", 64, 128, 1, 4, 3, 4, false)
  G2V("g2_128x64_s3", 128, 64, 2, 2, 3, 4, false)

  const int rounds = 5, reps = 4;
  printf("%-22s", "shape");
  for (auto& v : vars) printf(" %18s", v.name.c_str());
  printf("\n");
  std::vector<double> tot_ms(vars.size(), 0.0);
  double tot_flop = 0;
  for (auto& s : shapes) {
    const double flop = 2.0 * s.P * s.K * s.N;
    const size_t outn = (size_t)(s.pool ? s.P / 64 : s.P) * s.N;
    // reference = v1
    hipMemset(C, 0, outn * 4);
    vars[0].run(s, C);
    hipDeviceSynchronize();
    std::vector<float> ref(outn), got(outn);
    hipMemcpy(ref.data(), C, outn * 4, hipMemcpyDeviceToHost);
    std::vector<std::vector<float>> ms(vars.size());
    std::vector<double> err(vars.size(), 0.0);
    for (size_t vi = 1; vi < vars.size(); ++vi) {
      if (s.pool && !vars[vi].pool_ok) { err[vi] = -1; continue; }
      hipMemset(C2, 0xff, outn * 4);
      vars[vi].run(s, C2);
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { printf("variant %s failed: %s\n", vars[vi].name.c_str(), hipGetErrorString(e)); return 1; }
      hipMemcpy(got.data(), C2, outn * 4, hipMemcpyDeviceToHost);
      double m = 0;
      for (size_t i = 0; i < outn; ++i) { double d = std::fabs((double)got[i] - ref[i]); if (!(d <= m)) m = d; }
      err[vi] = m;
    }
    for (int r = 0; r < rounds; ++r)
      for (size_t vi = 0; vi < vars.size(); ++vi) {
        if (err[vi] < 0) continue;
        hipEventRecord(e0);
        for (int k = 0; k < reps; ++k) vars[vi].run(s, C2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        ms[vi].push_back(t / reps);
      }
    char nm[64]; snprintf(nm, sizeof nm, "P%lld K%d N%d%s", s.P, s.K, s.N, s.pool ? " pool" : "");
    printf("%-22s", nm);
    for (size_t vi = 0; vi < vars.size(); ++vi) {
      if (err[vi] < 0) { printf(" %18s", "-"); continue; }
      std::sort(ms[vi].begin(), ms[vi].end());
      const double med = ms[vi][ms[vi].size() / 2];
      tot_ms[vi] += med;
      printf(" %6.1fTF %.0e%s", flop / med / 1e9, err[vi], err[vi] > 2e-3 ? "!!" : "  ");
    }
    printf("\n");
    tot_flop += flop;
  }
  printf("%-22s", "TOTAL (TF, pool-less skip)");
  for (size_t vi = 0; vi < vars.size(); ++vi) printf(" %8.3fms        ", tot_ms[vi]);
  printf("\n");

#if MLP_TRACE
  // per-workgroup timelines of v1 on one shape, plain and staggered
  {
    Shape s = shapes[0];
    const long long blocks = ((s.P + 127) / 128) * ((s.N + 127) / 128);
    unsigned long long* tr; hipMalloc(&tr, blocks * 6 * 8);
    std::vector<unsigned long long> ht(blocks * 6);
    for (int mode = 0; mode < 3; ++mode) {
      const int KT = s.K / 16;
      g_mlp_trace = tr; g_mlp_first_round = 1024;
      g_mlp_stagger = mode == 0 ? 0 : (mode == 1 ? KT * 2048 : KT * 1024);
      hipMemset(tr, 0, blocks * 6 * 8);
      for (int k = 0; k < 3; ++k) vars[0].run(s, C2);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int k = 0; k < reps; ++k) vars[0].run(s, C2);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float t; hipEventElapsedTime(&t, e0, e1);
      printf("v1 trace mode %d (stagger %d cycles/slot): %.4f ms  %.1f TF\n", mode, g_mlp_stagger, t / reps,
             2.0 * s.P * s.K * s.N / (t / reps) / 1e9);
      hipMemcpy(ht.data(), tr, blocks * 6 * 8, hipMemcpyDeviceToHost);
      char fn[128]; snprintf(fn, sizeof fn, "gpurun_out/g2/trace_%d.bin", mode);
      FILE* f = fopen(fn, "wb");
      if (f) { fwrite(ht.data(), 8, ht.size(), f); fclose(f); }
    }
    g_mlp_trace = nullptr; g_mlp_stagger = 0;
  }
#endif
  return 0;
}
