// Ceiling experiment, fenced off from the product: fp32-faithful products on the bf16 matrix pipe of gfx950.
//   x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (24 mantissa bits in three 8-bit pieces; the
//   residuals are exact in fp32), and  x w ~= x1 w1 + x1 w2 + x2 w1 + x2 w2 + x1 w3 + x3 w1  (the dropped terms are <= 2^-24
//   relative); every bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32.
// Part 1: issue rate of the bf16 instruction against v_mfma_f32_32x32x2_f32 (bare register loops, whole chip).
// Part 2: C = A B^T (M = N = 256, K given) by fp32 MFMA, by 6 / 3 / 1 split products, against a float64 host evaluation --
//         operands N(0,1) (weights scaled 1/sqrt(K)) and "activation-like" (ReLU of N(0.3,1)).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 split_products.cpp -o split_products && ./split_products
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }   // round to nearest even

template <int BF>
__global__ __launch_bounds__(256, 2) void rate(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int tid = threadIdx.x;
  if (BF) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = to_bf16(0.001f * (float)((tid * 7 + i) & 63)); b[i] = to_bf16(0.002f * (float)((tid * 3 + i) & 31)); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
  } else {
    const float a = 0.001f * (float)(tid & 63), b = 0.002f * (float)(tid & 31);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + tid] = s;
}

// One wave per 32 x 32 tile of C; operands straight from global memory (a correctness harness, not a fast kernel).
// A (M x K), B (N x K) row-major fp32.  MODE 0: fp32 MFMA.  MODE 6 / 3 / 1: that many split products.
template <int MODE>
__global__ __launch_bounds__(64) void gemm(const float* A, const float* B, float* C, int M, int N, int K) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int tm = blockIdx.x * 32, tn = blockIdx.y * 32;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (MODE == 0) {
    for (int k = 0; k < K; k += 2) {
      const float a = A[(long long)(tm + r) * K + k + h], b = B[(long long)(tn + r) * K + k + h];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  } else {
    for (int k = 0; k < K; k += 16) {
      bf16x8 a1, a2, a3, b1, b2, b3;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float a = A[(long long)(tm + r) * K + k + 8 * h + i], b = B[(long long)(tn + r) * K + k + 8 * h + i];
        a1[i] = to_bf16(a); const float ra = a - (float)a1[i]; a2[i] = to_bf16(ra); a3[i] = to_bf16(ra - (float)a2[i]);
        b1[i] = to_bf16(b); const float rb = b - (float)b1[i]; b2[i] = to_bf16(rb); b3[i] = to_bf16(rb - (float)b2[i]);
      }
      // small terms first, so that they meet an accumulator of their own size before the large term is added
      if (MODE >= 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
      }
      if (MODE >= 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i / 4) * 8 + h * 4 + (i % 4);
    C[(long long)(tm + row) * N + tn + r] = acc[i];
  }
}

static double nrand(unsigned long long& s) {   // Box-Muller on a 64-bit LCG
  auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 11) + 0.5) / 9007199254740992.0; };
  return std::sqrt(-2.0 * std::log(u())) * std::cos(6.283185307179586 * u());
}

int main(int argc, char** argv) {
  float* out; CK(hipMalloc(&out, 1024 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int bf = 0; bf < 2; ++bf) {
    const int iters = bf ? 40000 : 5000, grid = 1024;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (bf) hipLaunchKernelGGL(rate<1>, dim3(grid), dim3(256), 0, 0, out, iters); else hipLaunchKernelGGL(rate<0>, dim3(grid), dim3(256), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flop = (double)grid * 4 /*waves*/ * iters * 4 /*tiles*/ * 2.0 * 32 * 32 * (bf ? 16 : 2);
      if (rep) printf("%s: %.1f TFLOP/s (%.3f ms)%s\n", bf ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32  ", flop / ms / 1e9, ms,
                      bf ? "  -> / 6 products = fp32-equivalent ceiling" : "");
    }
  }
  const int M = 256, N = 256;
  for (int K : {256, 1024}) for (int kind = 0; kind < 2; ++kind) {
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    unsigned long long s = 12345 + K + kind;
    for (auto& v : A) { double x = kind ? std::fmax(0.0, 0.3 + nrand(s)) : nrand(s); v = (float)x; }
    for (auto& v : B) v = (float)(nrand(s) / std::sqrt((double)K));
    std::vector<double> ref((size_t)M * N);
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
      double acc = 0; for (int k = 0; k < K; ++k) acc += (double)A[(size_t)i * K + k] * (double)B[(size_t)j * K + k];
      ref[(size_t)i * N + j] = acc;
    }
    float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> C((size_t)M * N);
    printf("K = %4d, %s activations:", K, kind ? "relu(N(0.3,1))" : "N(0,1)        ");
    for (int mode : {0, 6, 3, 1}) {
      const dim3 g(M / 32, N / 32);
      if (mode == 0) hipLaunchKernelGGL(gemm<0>, g, dim3(64), 0, 0, dA, dB, dC, M, N, K);
      if (mode == 6) hipLaunchKernelGGL(gemm<6>, g, dim3(64), 0, 0, dA, dB, dC, M, N, K);
      if (mode == 3) hipLaunchKernelGGL(gemm<3>, g, dim3(64), 0, 0, dA, dB, dC, M, N, K);
      if (mode == 1) hipLaunchKernelGGL(gemm<1>, g, dim3(64), 0, 0, dA, dB, dC, M, N, K);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0, sq = 0;
      for (size_t i = 0; i < C.size(); ++i) { const double d = std::fabs((double)C[i] - ref[i]); worst = std::fmax(worst, d); sq += d * d; }
      printf("  %s max %.2e rms %.2e", mode == 0 ? "fp32-mfma" : mode == 6 ? "split-6" : mode == 3 ? "split-3" : "bf16", worst, std::sqrt(sq / C.size()));
    }
    printf("\n");
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
