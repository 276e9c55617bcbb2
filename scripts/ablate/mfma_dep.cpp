// How many independent accumulators / resident waves does v_mfma_f32_32x32x2_f32 need to saturate the matrix pipe?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_dep.cpp -o mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  extern __shared__ float pad[];
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const float x = (float)(threadIdx.x & 7) * 0.125f, y = 0.5f + pad[threadIdx.x & 3] * 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, int iters) {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  // one block = 4 waves = 1 wave per SIMD; LDS padding limits the blocks per CU
  const int lds = waves_per_simd == 1 ? 100 * 1024 : waves_per_simd == 2 ? 70 * 1024 : waves_per_simd == 4 ? 36 * 1024 : 16 * 1024;
  hipFuncSetAttribute((const void*)probe<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int blocks = 256 * waves_per_simd * 4;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  probe<NACC><<<blocks, 256, lds>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(s);
  probe<NACC><<<blocks, 256, lds>>>(out, iters);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double flop = (double)blocks * 4 * iters * 8.0 * NACC * 4096.0;
  printf("accumulators/wave %d  waves/SIMD %d : %7.3f ms  %6.1f TFLOP/s\n", NACC, waves_per_simd, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) { run<1>(w, 4000); run<2>(w, 2000); run<4>(w, 1000); run<8>(w, 500); }
  return 0;
}
