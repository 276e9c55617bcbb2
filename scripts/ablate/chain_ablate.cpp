// Micro-benchmark of the register-chained level-1 set-abstraction kernel (csrc/sa_chain.hip) at the ScoreNet
// level-1 shape of a batch of 8 scenes: 8 x 5120 neighbourhoods of 64 points, 6 -> 128 -> 128 -> 256.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DCH_VARIANT=n] chain_ablate.cpp -o chain_ablate_n
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../regnet_for_3d_grasping_amd/csrc/sa_chain.hip"

int main() {
  const long long B = 8, N = 25600, M = 5120, G = 64;
  std::vector<float> h(1 << 22);
  srand(5);
  for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  std::vector<long long> nbr(B * M * G), ctr(B * M);
  for (auto& v : nbr) v = rand() % N;
  for (auto& v : ctr) v = rand() % N;
  float *pc, *W1, *W2, *W3, *sc, *sh, *out;
  long long *dn, *dc;
  hipMalloc(&pc, B * N * 6 * 4); hipMalloc(&W1, 128 * 8 * 4); hipMalloc(&W2, 128 * 128 * 4); hipMalloc(&W3, 256 * 128 * 4);
  hipMalloc(&sc, 4096 * 4); hipMalloc(&sh, 4096 * 4); hipMalloc(&out, B * M * 256 * 4);
  hipMalloc(&dn, nbr.size() * 8); hipMalloc(&dc, ctr.size() * 8);
  hipMemcpy(pc, h.data(), B * N * 6 * 4, hipMemcpyHostToDevice);
  hipMemcpy(W1, h.data() + 100, 128 * 8 * 4, hipMemcpyHostToDevice);
  hipMemcpy(W2, h.data() + 5000, 128 * 128 * 4, hipMemcpyHostToDevice);
  hipMemcpy(W3, h.data() + 50000, 256 * 128 * 4, hipMemcpyHostToDevice);
  hipMemcpy(sc, h.data() + 90000, 4096 * 4, hipMemcpyHostToDevice);
  hipMemcpy(sh, h.data() + 95000, 4096 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dn, nbr.data(), nbr.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dc, ctr.data(), ctr.size() * 8, hipMemcpyHostToDevice);
  // pc is (B,N,6): xyz = columns 0..2 (strides N*6, 1, 6), rgb = columns 3..5
  auto run = [&]() {
    return regnet_sa_chain3_f32(pc + 3, N * 6, 6, 1, 3, pc, N * 6, 1, 6, (const int64_t*)dn, (const int64_t*)dc, nullptr, nullptr, B, M, G, W1, sc,
                                sh, 128, W2, 128, sc + 128, sh + 128, 128, W3, 128, sc + 256, sh + 256, 256, 1, out, 256,
                                nullptr);
  };
  int rc = run();
  hipDeviceSynchronize();
  if (rc) { printf("rc=%d\n", rc); return 1; }
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  const int reps = 10;
  hipEventRecord(s);
  for (int r = 0; r < reps; ++r) run();
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= reps;
  const double flop = 2.0 * B * M * G * (8.0 * 128 + 128.0 * 128 + 128.0 * 256);
  printf("sa_chain3 B=%lld M=%lld: %.3f ms  %.1f TFLOP/s\n", B, M, ms, flop / ms / 1e9);
  return 0;
}
