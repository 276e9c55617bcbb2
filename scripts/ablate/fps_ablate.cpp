// Ablation harness for the FPS kernels: hipcc -DFPS_ABLATE=n ... ; times B scenes of N points.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "geometry_measure.hip"   // the measurement twin GENERATED from csrc/geometry.hip: python -c "from regnet_for_3d_grasping_amd.csrc import build; build.measurement_twin('geometry.hip', 'scripts/ablate')" first (build with -I regnet_for_3d_grasping_amd/csrc)

int main(int argc, char** argv) {
  int B = 8, N = argc > 1 ? atoi(argv[1]) : 25600, M = argc > 2 ? atoi(argv[2]) : 5120;
  std::vector<float> h((size_t)B * N * 3);
  srand(1);
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < N; ++j) {
      h[((size_t)b * N + j) * 3 + 0] = -0.4f + 0.8f * rand() / RAND_MAX;
      h[((size_t)b * N + j) * 3 + 1] = -0.35f + 0.7f * rand() / RAND_MAX;
      h[((size_t)b * N + j) * 3 + 2] = 0.75f + ((rand() % 10) < 6 ? 0.001f * rand() / RAND_MAX : 0.1f * rand() / RAND_MAX);
    }
  float* d; int64_t* idx;
  hipMalloc(&d, h.size() * 4); hipMalloc(&idx, (size_t)B * M * 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  float* ws = nullptr;
  const int64_t wsb = regnet_fps_workspace_bytes(B, N, M);
  if (wsb) hipMalloc(&ws, wsb);
  regnet_fps_f32(d, (int64_t)N * 3, 1, 3, B, N, M, idx, ws, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(s);
  for (int r = 0; r < 3; ++r) regnet_fps_f32(d, (int64_t)N * 3, 1, 3, B, N, M, idx, ws, nullptr);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  std::vector<int64_t> out((size_t)B * M);
  hipMemcpy(out.data(), idx, out.size() * 8, hipMemcpyDeviceToHost);
  long long cs = 0; for (auto v : out) cs += v;
#if FPS_ABLATE == 11
  {
    unsigned long long dbg[8];
    hipMemcpyFromSymbol(dbg, HIP_SYMBOL(fps_dbg), sizeof(dbg));
    const int marks[5] = {32, 64, 128, 256, 1024};
    for (int k = 0; k < 5; ++k) printf("  pick %4d reached after %10.0f counter ticks of the round loop\n", marks[k], (double)dbg[k]);
  }
#endif
#if FPS_ABLATE == 9
  unsigned long long dbg[8];
  hipMemcpyFromSymbol(dbg, HIP_SYMBOL(fps_dbg), sizeof(dbg));
  const char* names[8] = {"loop-top(after unkey/store)", "centroid load+test", "scan", "wave reduce+lds write", "barrier1", "read partials+stage2", "barrier2", ""};
  double rounds = 4.0 * (M - 1);
#if FPS_PICKS > 1 || FPS_CLUSTERS
  const char* names2[8] = {"after barrier 2 -> loop top", "centroid tests + scans", "top-2 / slot / row records", "barrier 1", "selection + acceptance (wave 0)", "barrier 2", "", ""};
  for (int k = 0; k < 8; ++k) names[k] = names2[k];
  rounds = (double)dbg[7];
  printf("  %.0f rounds for %d picks x 4 launches: %.2f picks per round\n", rounds, M - 1, 4.0 * (M - 1) / rounds);
#if FPS_CLUSTERS
  printf("  clusters flagged per centroid: %.1f of %d\n", (double)dbg[6] / (4.0 * (M - 1)), (N + 63) / 64);
  dbg[6] = 0;
#endif
#endif
  for (int k = 0; k < 7; ++k) printf("  phase %d %-28s %8.1f cycles/round (thread 0 of block 0)\n", k, names[k], dbg[k] / rounds);
#endif
  printf("ABLATE=%d N=%d M=%d: %.3f ms  %.3f us/round  checksum %lld\n", FPS_ABLATE, N, M, ms / 3, ms / 3 * 1e3 / (M - 1), cs);
  return 0;
}
