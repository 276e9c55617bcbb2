// clock_probe: one wave runs a dependent chain of N v_fma_f32 and reports how long it took on the constant 100 MHz
// counter (s_memrealtime) and on the shader counter (s_memtime).  A dependent fp32 FMA chain costs a fixed number of
// shader cycles per link, so (links / realtime) tracks the shader clock: launched beside other work it shows whether that
// work lowers the clock for everybody.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libclock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

// The workgroup asks for (almost) the whole LDS so that no workgroup of an LDS-using kernel shares its CU: the chain then
// has its SIMD to itself and the rate depends on the clock only.
__global__ void clock_probe_kernel(unsigned long long* out, int links, float seed) {
  extern __shared__ float hog[];
  if (seed < 0.f) hog[threadIdx.x] = seed;
  float a = seed, b = 1.0000001f, c = 1e-9f;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int i = 0; i < links; i += 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a = __builtin_fmaf(a, b, c);
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
    out[2] = (unsigned long long)__float_as_uint(a);
  }
}

extern "C" int clock_probe(unsigned long long* out, int links, void* stream) {
  static bool once = false;
  const int lds = 156 * 1024;
  if (!once) {
    hipFuncSetAttribute((const void*)clock_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    once = true;
  }
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), lds, (hipStream_t)stream, out, links, 0.5f);
  return (int)hipGetLastError();
}

// side_load: `blocks` workgroups of `threads` threads that stay resident for `ticks` x 10 ns: mode 0 sleeps (s_sleep loop),
// mode 1 runs dependent FMAs in every wave, mode 2 hammers an LDS atomic + barrier per iteration (what an FPS round does).
__global__ void side_load_kernel(unsigned long long ticks, int mode, float* sink) {
  extern __shared__ float side_hog[];   // optional: nothing that needs LDS then shares the CU
  __shared__ unsigned slot[64];
  if (ticks == 0) side_hog[threadIdx.x] = 0.f;
  if (mode >= 16) {   // only the workgroups whose index is congruent to (mode - 16) mod 8 stay: all on one XCD
    if ((int)(blockIdx.x & 7) != ((mode - 16) & 7)) return;
    mode = 6;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float a = 0.5f + threadIdx.x;
  if (threadIdx.x < 64) slot[threadIdx.x] = 0;
  __syncthreads();
  if (mode == 6) {   // no memory instruction at all: a counted loop of sleeps (~3.4 us each at 2.4 GHz), `ticks` = iterations
    for (unsigned long long i = 0; i < ticks; ++i) __builtin_amdgcn_s_sleep(127);
    return;
  }
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    if (mode == 0) {
      __builtin_amdgcn_s_sleep(32);
    } else if (mode == 1) {
#pragma unroll
      for (int j = 0; j < 64; ++j) a = __builtin_fmaf(a, 1.0000001f, 1e-9f);
    } else if (mode == 3) {         // FMAs a quarter of the time
#pragma unroll
      for (int j = 0; j < 64; ++j) a = __builtin_fmaf(a, 1.0000001f, 1e-9f);
      __builtin_amdgcn_s_sleep(12);   // 64 FMAs ~ 256+ issue cycles per wave; sleep 12 x 64 cycles
    } else if (mode == 4) {         // integer VALU work
      unsigned u = __float_as_uint(a);
#pragma unroll
      for (int j = 0; j < 64; ++j) u = u * 1664525u + 1013904223u;
      a = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    } else if (mode == 5) {         // scalar-only loop (no vector issue)
      asm volatile("s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15");
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) a = __builtin_fmaf(a, 1.0000001f, 1e-9f);
      atomicMax(&slot[threadIdx.x & 63], __float_as_uint(a));
      __syncthreads();
      a += __uint_as_float(slot[0]) * 1e-30f;
      __syncthreads();
    }
  }
  if (a == 12345.678f) sink[0] = a;
}

extern "C" int side_load_lds(int blocks, int threads, double ms, int mode, int lds_bytes, float* sink, void* stream) {
  (void)hipFuncSetAttribute((const void*)side_load_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
  hipLaunchKernelGGL(side_load_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream,
                     (mode == 6 || mode >= 16) ? (unsigned long long)(ms * 1e3 / 3.4) : (unsigned long long)(ms * 1e5), mode, sink);
  return (int)hipGetLastError();
}

extern "C" int side_load(int blocks, int threads, double ms, int mode, float* sink, void* stream) {
  hipLaunchKernelGGL(side_load_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream,
                     (unsigned long long)(ms * 1e5), mode, sink);
  return (int)hipGetLastError();
}

// mfma_burn: every wave issues `iters` x 8 independent v_mfma_f32_32x32x2f32 from registers (no memory at all).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_burn_kernel(int iters, float* sink) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) sink[0] = s;
}
extern "C" int mfma_burn(int blocks, int iters, float* sink, void* stream) {
  hipLaunchKernelGGL(mfma_burn_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
  return (int)hipGetLastError();
}

// hbm_stream: grid-stride float4 copy of n4 elements (read + write).
__global__ __launch_bounds__(256) void hbm_stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int hbm_stream(const void* src, void* dst, long long n4, int blocks, void* stream) {
  hipLaunchKernelGGL(hbm_stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, n4);
  return (int)hipGetLastError();
}
