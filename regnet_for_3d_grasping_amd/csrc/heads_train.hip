// heads_train.hip -- the grasp heads of the region stage in TRAINING mode: one launch per layer forward, two backward.
//
// Reference behaviour restated (paths relative to /root/reference/multi_model/utils): every layer of PointNet2TwoStage
// (pointnet2.py:174-188) and PointNet2Refine (pointnet2.py:240-253) is nn.Conv1d(K, N, 1) WITH bias on (n, K, 1) rows, then
// nn.BatchNorm1d(N) in training mode (batch statistics over the n rows, biased variance for the normalisation, unbiased for
// running_var, momentum 0.1, num_batches_tracked += 1), then ReLU except on the branches' last layers.  n is the number of
// labelled centres / valid crops of the iteration: a few hundred rows, widths 2 .. 1024.
// Through torch that is six launches per layer forward (GEMM, counter, statistics, running update, transform, clamp) and
// seven backward, ~160 of them per iteration on a stream the HOST paces (profiles/r04k_train_region_hole.txt): the GPU waits.
//
// Here a workgroup owns 16 output channels and ALL rows, so the batch statistics are local to it:
//   forward   z = X . W^T by v_mfma_f32_16x16x4_f32 (exact fp32 products; a lane's float4 of K feeds four MFMAs) into LDS,
//             mean / variance over the rows (two passes, fixed order: deterministic), running statistics, xhat and
//             y = [relu](gamma * xhat + beta) written once.  The bias only moves the batch mean: it enters running_mean.
//   backward  (a) dyh = dY * [y > 0];  dbeta = sum dyh;  dgamma = sum dyh * xhat;
//                 dZ = gamma * invstd * (dyh - dbeta / n - xhat * dgamma / n)  -> LDS + global;  dbias = sum dZ;
//                 dW[16 channels, :] = dZ^T . X by MFMA (contraction over the rows, dZ from LDS, X from L2);
//             (b) dX (+)= dZ . W by MFMA, a wave per 16 rows x 64 columns (accumulating for the second branch of a fork).
#include "common.h"

typedef float ht_f32x4 __attribute__((ext_vector_type(4)));

#define HT_THREADS 512           // 8 waves
#define HT_WAVES (HT_THREADS / 64)
#define HT_MAX_ROWS 1024         // 16 x 1024 floats of LDS per workgroup

__device__ __forceinline__ ht_f32x4 ht_mfma(float a, float b, ht_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct HtFwd {
  const float* X; long long ldx;
  const float* W; const float* bias; const float* gamma; const float* beta;
  float* run_mean; float* run_var; long long* nbt;
  float momentum, eps;
  int R, K, N, relu;
  float* XH; float* Y; float* invstd;
};

// sum over the rows of f(row, j) for the 16 channels of the tile: thread -> (channel j = tid & 15, row group q = tid >> 4),
// partials through `red`, the first 16 threads add them in a fixed order.  Result in out[0..15] (shared), synchronised.
template <typename F>
__device__ __forceinline__ void ht_column_sums(int R, float (*red)[17], float* out, F f) {
  const int tid = threadIdx.x, j = tid & 15, q = tid >> 4;
  float s = 0.f;
  for (int r = q; r < R; r += HT_THREADS / 16) s += f(r, j);
  red[q][j] = s;
  __syncthreads();
  if (tid < 16) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < HT_THREADS / 16; ++k) t += red[k][tid];
    out[tid] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(HT_THREADS) void ht_fwd_kernel(const HtFwd p) {
  extern __shared__ __attribute__((aligned(16))) float zbuf[];     // [16 * tiles][16]
  __shared__ float red[HT_THREADS / 16][17];
  __shared__ float s_sum[16], s_mean[16], s_inv[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int tiles = (p.R + 15) >> 4;
  const bool wvalid = n0 + i < p.N;
  const float* wp = p.W + (long long)(wvalid ? n0 + i : 0) * p.K + 4 * g;
  // ---- z = X . W^T for the tile's 16 channels, two row tiles per pass (one weight fragment feeds both)
  for (int t = wave; t < tiles; t += 2 * HT_WAVES) {
    const int t1 = t + HT_WAVES;
    const bool two = t1 < tiles;
    const float* xa = p.X + (long long)min(16 * t + i, p.R - 1) * p.ldx + 4 * g;
    const float* xb = p.X + (long long)min(16 * (two ? t1 : t) + i, p.R - 1) * p.ldx + 4 * g;
    ht_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int kc = 0; kc < p.K; kc += 16) {
      ht_f32x4 w = {0.f, 0.f, 0.f, 0.f};
      if (wvalid) w = *reinterpret_cast<const ht_f32x4*>(wp + kc);
      const ht_f32x4 a = *reinterpret_cast<const ht_f32x4*>(xa + kc);
      const ht_f32x4 b = *reinterpret_cast<const ht_f32x4*>(xb + kc);
      acc0 = ht_mfma(a.x, w.x, acc0); acc1 = ht_mfma(b.x, w.x, acc1);
      acc0 = ht_mfma(a.y, w.y, acc0); acc1 = ht_mfma(b.y, w.y, acc1);
      acc0 = ht_mfma(a.z, w.z, acc0); acc1 = ht_mfma(b.z, w.z, acc1);
      acc0 = ht_mfma(a.w, w.w, acc0); acc1 = ht_mfma(b.w, w.w, acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      zbuf[(16 * t + 4 * g + r) * 16 + i] = acc0[r];
      if (two) zbuf[(16 * t1 + 4 * g + r) * 16 + i] = acc1[r];
    }
  }
  __syncthreads();
  // ---- batch statistics of the R real rows
  const float inv_n = 1.f / (float)p.R;
  ht_column_sums(p.R, red, s_sum, [&](int r, int j) { return zbuf[r * 16 + j]; });
  if (tid < 16) s_mean[tid] = s_sum[tid] * inv_n;
  __syncthreads();
  ht_column_sums(p.R, red, s_sum, [&](int r, int j) { const float d = zbuf[r * 16 + j] - s_mean[j]; return d * d; });
  if (tid < 16) {
    const float var = s_sum[tid] * inv_n;
    const float inv = 1.f / sqrtf(var + p.eps);
    s_inv[tid] = inv;
    const int n = n0 + tid;
    if (n < p.N) {
      p.invstd[n] = inv;
      if (p.run_mean) {
        const float m = s_mean[tid] + (p.bias ? p.bias[n] : 0.f);
        const float unbiased = p.R > 1 ? var * ((float)p.R / (float)(p.R - 1)) : var;
        p.run_mean[n] = (1.f - p.momentum) * p.run_mean[n] + p.momentum * m;
        p.run_var[n] = (1.f - p.momentum) * p.run_var[n] + p.momentum * unbiased;
      }
    }
    if (blockIdx.x == 0 && tid == 0 && p.nbt) *p.nbt += 1;
  }
  __syncthreads();
  // ---- xhat and y, written once
  for (int idx = tid; idx < p.R * 16; idx += HT_THREADS) {
    const int r = idx >> 4, j = idx & 15, n = n0 + j;
    if (n >= p.N) continue;
    const float xh = (zbuf[idx] - s_mean[j]) * s_inv[j];
    float y = p.gamma[n] * xh + p.beta[n];
    if (p.relu) y = fmaxf(y, 0.f);
    p.XH[(long long)r * p.N + n] = xh;
    p.Y[(long long)r * p.N + n] = y;
  }
}

struct HtBwd {
  const float* dY; long long lddy;
  const float* Y; const float* XH; const float* gamma; const float* invstd;
  const float* X; long long ldx;
  int R, K, N, relu;
  float* dZ; float* dW; float* dbias; float* dgamma; float* dbeta;
};

__global__ __launch_bounds__(HT_THREADS) void ht_bwd_kernel(const HtBwd p) {
  extern __shared__ __attribute__((aligned(16))) float buf[];      // [16 * tiles][16]: dyh, then dZ (rows >= R: zero)
  __shared__ float red[HT_THREADS / 16][17];
  __shared__ float s_a[16], s_b[16], s_c[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int tiles = (p.R + 15) >> 4;
  for (int idx = tid; idx < tiles * 256; idx += HT_THREADS) {
    const int r = idx >> 4, n = n0 + (idx & 15);
    float d = 0.f;
    if (r < p.R && n < p.N) {
      d = p.dY[(long long)r * p.lddy + n];
      if (p.relu && !(p.Y[(long long)r * p.N + n] > 0.f)) d = 0.f;
    }
    buf[idx] = d;
  }
  __syncthreads();
  const auto xh = [&](int r, int j) { return n0 + j < p.N ? p.XH[(long long)r * p.N + n0 + j] : 0.f; };
  ht_column_sums(p.R, red, s_a, [&](int r, int j) { return buf[r * 16 + j]; });                // dbeta
  ht_column_sums(p.R, red, s_b, [&](int r, int j) { return buf[r * 16 + j] * xh(r, j); });     // dgamma
  const float inv_n = 1.f / (float)p.R;
  for (int idx = tid; idx < p.R * 16; idx += HT_THREADS) {
    const int r = idx >> 4, j = idx & 15, n = n0 + j;
    float dz = 0.f;
    if (n < p.N) {
      dz = p.gamma[n] * p.invstd[n] * (buf[idx] - s_a[j] * inv_n - xh(r, j) * s_b[j] * inv_n);
      p.dZ[(long long)r * p.N + n] = dz;
    }
    buf[idx] = dz;
  }
  __syncthreads();
  ht_column_sums(p.R, red, s_c, [&](int r, int j) { return buf[r * 16 + j]; });                // dbias
  if (tid < 16 && n0 + tid < p.N) {
    p.dbeta[n0 + tid] = s_a[tid];
    p.dgamma[n0 + tid] = s_b[tid];
    if (p.dbias) p.dbias[n0 + tid] = s_c[tid];
  }
  // ---- dW[n0 + c][k] = sum_r dZ[r][c] X[r][k]: a wave per 64 columns of K, contraction over the rows
  for (int kg = wave; kg * 64 < p.K; kg += HT_WAVES) {
    ht_f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = ht_f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xcol = p.X + 64 * kg + i;
    for (int rc = 0; rc < tiles; ++rc) {
      float a[4], x[4][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = 16 * rc + 4 * g + s;
        a[s] = buf[r * 16 + i];
        const float* xr = xcol + (long long)min(r, p.R - 1) * p.ldx;       // rows >= R: a[s] is zero
#pragma unroll
        for (int b = 0; b < 4; ++b) x[b][s] = xr[16 * b];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = ht_mfma(a[s], x[b][s], acc[b]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * g + r;
      if (n < p.N) {
#pragma unroll
        for (int b = 0; b < 4; ++b) p.dW[(long long)n * p.K + 64 * kg + 16 * b + i] = acc[b][r];
      }
    }
  }
}

struct HtDgrad {
  const float* dZ; const float* W; float* dX; long long lddx;
  int R, K, N, accumulate;
};

// dX[r][k] (+)= sum_n dZ[r][n] W[n][k]: grid (K / 64, row tiles / 4 waves), a wave per 16 rows x 64 columns
__global__ __launch_bounds__(256) void ht_dgrad_kernel(const HtDgrad p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int t = blockIdx.y * 4 + wave;
  if (16 * t >= p.R) return;
  const int k0 = blockIdx.x * 64;
  const float* zrow = p.dZ + (long long)min(16 * t + i, p.R - 1) * p.N;
  ht_f32x4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = ht_f32x4{0.f, 0.f, 0.f, 0.f};
  for (int nc = 0; nc < p.N; nc += 16) {
    float a[4], w[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = nc + 4 * g + s;
      const bool ok = n < p.N;
      a[s] = ok ? zrow[n] : 0.f;
      const float* wr = p.W + (long long)(ok ? n : 0) * p.K + k0 + i;
#pragma unroll
      for (int b = 0; b < 4; ++b) w[b][s] = ok ? wr[16 * b] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = ht_mfma(a[s], w[b][s], acc[b]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * t + 4 * g + r;
    if (row < p.R) {
      float* o = p.dX + (long long)row * p.lddx + k0 + i;
#pragma unroll
      for (int b = 0; b < 4; ++b) o[16 * b] = p.accumulate ? o[16 * b] + acc[b][r] : acc[b][r];
    }
  }
}

static bool ht_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// dynamic + static LDS beyond 64 KiB (more than ~960 rows) needs the function attribute; rare, so simply set per launch
static int ht_allow_lds(const void* kernel, size_t bytes) {
  if (bytes <= 60000) return REGNET_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? REGNET_OK : (int)e;
}

extern "C" int regnet_head_layer_train_supported(int64_t R, int64_t K, int64_t N) {
  return R >= 2 && R <= HT_MAX_ROWS && K >= 64 && (K % 64) == 0 && N >= 1;
}

extern "C" int regnet_head_layer_train_fwd_f32(const float* X, int64_t ldx, const float* W, const float* bias, const float* gamma,
                                               const float* beta, float* running_mean, float* running_var,
                                               int64_t* num_batches_tracked, float momentum, float eps, int64_t R, int64_t K,
                                               int64_t N, int relu, float* xhat, float* Y, float* save_invstd, void* stream) {
  if (!regnet_head_layer_train_supported(R, K, N) || ldx < K || (ldx % 4)) return REGNET_ERR_SHAPE;
  if (!X || !W || !gamma || !beta || !xhat || !Y || !save_invstd) return REGNET_ERR_NULL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return REGNET_ERR_NULL;
  if (!ht_aligned16(X) || !ht_aligned16(W)) return REGNET_ERR_SHAPE;
  HtFwd a = {X, ldx, W, bias, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps,
             (int)R, (int)K, (int)N, relu, xhat, Y, save_invstd};
  const size_t lds = (size_t)((R + 15) / 16) * 256 * sizeof(float);
  if (int rc = ht_allow_lds(reinterpret_cast<const void*>(ht_fwd_kernel), lds)) return rc;
  hipLaunchKernelGGL(ht_fwd_kernel, dim3((unsigned)((N + 15) / 16)), dim3(HT_THREADS), lds, as_stream(stream), a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_head_layer_train_bwd_f32(const float* dY, int64_t lddy, const float* Y, const float* xhat, const float* gamma,
                                               const float* save_invstd, const float* X, int64_t ldx, const float* W, int64_t R,
                                               int64_t K, int64_t N, int relu, float* dZ, float* dW, float* dbias, float* dgamma,
                                               float* dbeta, float* dX, int64_t lddx, int accumulate_dx, void* stream) {
  if (!regnet_head_layer_train_supported(R, K, N) || ldx < K || lddy < N || (dX && lddx < K)) return REGNET_ERR_SHAPE;
  if (!dY || !xhat || !gamma || !save_invstd || !X || !W || !dZ || !dW || !dgamma || !dbeta || (relu && !Y)) return REGNET_ERR_NULL;
  HtBwd a = {dY, lddy, Y, xhat, gamma, save_invstd, X, ldx, (int)R, (int)K, (int)N, relu, dZ, dW, dbias, dgamma, dbeta};
  const size_t lds = (size_t)((R + 15) / 16) * 256 * sizeof(float);
  if (int rc = ht_allow_lds(reinterpret_cast<const void*>(ht_bwd_kernel), lds)) return rc;
  hipLaunchKernelGGL(ht_bwd_kernel, dim3((unsigned)((N + 15) / 16)), dim3(HT_THREADS), lds, as_stream(stream), a);
  REGNET_LAUNCH_CHECK();
  if (dX) {
    HtDgrad d = {dZ, W, dX, lddx, (int)R, (int)K, (int)N, accumulate_dx};
    const unsigned tiles = (unsigned)((R + 15) / 16);
    hipLaunchKernelGGL(ht_dgrad_kernel, dim3((unsigned)(K / 64), (tiles + 3) / 4), dim3(256), 0, as_stream(stream), d);
    REGNET_LAUNCH_CHECK();
  }
  return REGNET_OK;
}
