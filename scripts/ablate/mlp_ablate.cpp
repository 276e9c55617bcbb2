// Micro-benchmark of the shared-MLP GEMM kernel on the ScoreNet layer shapes (B = 8 scenes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMLP_VARIANT=n] mlp_ablate.cpp -o mlp_ablate_n
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../regnet_for_3d_grasping_amd/csrc/mlp.hip"

struct Shape { long long P; int K, N; int pool; };

int main(int argc, char** argv) {
  std::vector<Shape> shapes = {{2621440, 128, 128, 0}, {2621440, 128, 256, 64}, {524288, 256, 256, 0},
                               {524288, 256, 512, 64}, {131072, 512, 512, 0}, {131072, 512, 1024, 64},
                               {204800, 256, 256, 0}, {204800, 256, 512, 0}, {8192, 1536, 1024, 0}, {40960, 1280, 512, 0}};
  int only = argc > 1 ? atoi(argv[1]) : -1;
  size_t maxA = 0, maxC = 0, maxW = 0;
  for (auto& s : shapes) {
    maxA = std::max(maxA, (size_t)s.P * s.K); maxC = std::max(maxC, (size_t)s.P * s.N); maxW = std::max(maxW, (size_t)s.N * s.K);
  }
  float *A, *W, *C, *sc, *sh;
  hipMalloc(&A, maxA * 4); hipMalloc(&C, maxC * 4); hipMalloc(&W, maxW * 4 + 1024 * 2048 * 4); hipMalloc(&sc, 4096 * 4); hipMalloc(&sh, 4096 * 4);
  std::vector<float> h(1 << 22);
  srand(3);
  for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  for (size_t off = 0; off < maxA; off += h.size()) hipMemcpy(A + off, h.data(), std::min(h.size(), maxA - off) * 4, hipMemcpyHostToDevice);
  for (size_t off = 0; off < maxW; off += h.size()) hipMemcpy(W + off, h.data(), std::min(h.size(), maxW - off) * 4, hipMemcpyHostToDevice);
  hipMemcpy(sc, h.data(), 4096 * 4, hipMemcpyHostToDevice); hipMemcpy(sh, h.data() + 5000, 4096 * 4, hipMemcpyHostToDevice);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  double tot_flop = 0, tot_ms = 0;
  for (size_t i = 0; i < shapes.size(); ++i) {
    if (only >= 0 && (int)i != only) continue;
    auto& sp = shapes[i];
    int rc = regnet_mlp_layer_f32(A, sp.K, sp.K, W, sp.K, sc, sh, C, sp.N, sp.P, sp.N, 1, sp.pool, nullptr);
    hipDeviceSynchronize();
    if (rc) { printf("rc=%d\n", rc); return 1; }
    const int reps = 5;
    hipEventRecord(s);
    for (int r = 0; r < reps; ++r) regnet_mlp_layer_f32(A, sp.K, sp.K, W, sp.K, sc, sh, C, sp.N, sp.P, sp.N, 1, sp.pool, nullptr);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= reps;
    double flop = 2.0 * sp.P * sp.K * sp.N;
    printf("P=%8lld K=%4d N=%4d pool=%2d : %7.3f ms  %6.1f TFLOP/s\n", sp.P, sp.K, sp.N, sp.pool, ms, flop / ms / 1e9);
    tot_flop += flop; tot_ms += ms;
  }
  printf("TOTAL %.3f ms  %.1f TFLOP/s\n", tot_ms, tot_flop / tot_ms / 1e9);
  return 0;
}
