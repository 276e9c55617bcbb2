// Ceiling probes for the shared-MLP kernel on gfx950: (0) a pure v_mfma_f32_32x32x2_f32 loop, (1) the
// same with the kernel's LDS fragment reads (4 ds_read_b128 per 16 MFMAs), (2) plus a barrier per 32
// MFMAs.  Prints TFLOP/s and the shader clock derived from s_memtime against wall time.
// Modes 3 / 4 (round 5) = modes 0 / 2 on PSEUDO-RANDOM operands (every lane, every k-step another value in [-1, 1)): the
// low-entropy constants of modes 0-2 barely toggle the multipliers; what the matrix pipe sustains on real activations
// is what the power manager lets it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_peak.cpp -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(float* out, unsigned long long* clk, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 20 * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 128 * 20 * 2; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    lds[i] = MODE >= 3 ? (float)(int)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f : (float)(i & 7) * 0.125f;
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  float4 a[2] = {make_float4(1.f, 2.f, 3.f, 4.f), make_float4(1.f, 2.f, 3.f, 4.f)};
  float4 b[2] = {make_float4(.1f, .2f, .3f, .4f), make_float4(.1f, .2f, .3f, .4f)};
  if (MODE == 3) {   // random per-lane operands, held in registers
    a[0] = *reinterpret_cast<const float4*>(&lds[4 * tid]); a[1] = *reinterpret_cast<const float4*>(&lds[4 * tid + 1024]);
    b[0] = *reinterpret_cast<const float4*>(&lds[4 * tid + 2048]); b[1] = *reinterpret_cast<const float4*>(&lds[4 * tid + 3072]);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (MODE == 1 || MODE == 2 || MODE == 4) {
        const int buf = it & 1;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          a[mi] = *reinterpret_cast<const float4*>(&lds[(buf * 128 + wr * 64 + mi * 32 + fr) * 20 + kk * 8 + 4 * fh]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          b[ni] = *reinterpret_cast<const float4*>(&lds[((2 + buf) * 128 + wc * 64 + ni * 32 + fr) * 20 + kk * 8 + 4 * fh]);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    if (MODE == 2 || MODE == 4) __syncthreads();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int r = 0; r < 16; ++r) s += acc[x][y][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
void run(int blocks, int iters) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 8);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  probe<MODE><<<blocks, 256>>>(out, clk, iters);
  hipDeviceSynchronize();
  hipEventRecord(s);
  probe<MODE><<<blocks, 256>>>(out, clk, iters);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double flop = (double)blocks * 4 /*waves*/ * iters * 32.0 * (32 * 32 * 2 * 2);
  printf("mode %d blocks %5d iters %6d : %8.3f ms  %6.1f TFLOP/s   block0 cycles %llu -> %.0f MHz (s_memtime ticks / wall)\n", MODE,
         blocks, iters, ms, flop / ms / 1e9, c, c / (ms * 1e3));
  hipFree(out); hipFree(clk);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(1024, 4000); run<1>(1024, 4000); run<2>(1024, 4000);
  }
  run<0>(1024, 40000);   // ~100 ms: long enough for the power manager to settle
  run<2>(1024, 40000);
  run<2>(2048, 4000);
  for (int rep = 0; rep < 2; ++rep) { run<3>(1024, 4000); run<4>(1024, 4000); }
  run<3>(1024, 40000);
  run<4>(1024, 40000);
  run<0>(1024, 40000);
  return 0;
}
