// bn_train.hip -- training-mode BatchNorm + ReLU (+ max over the K neighbours) of the shared-MLP blocks, forward and
// backward, as fused HBM passes (gfx950).
//
// Reference behaviour restated: a shared-MLP block is conv(1x1, no bias) -> BatchNorm(eps 1e-5, momentum 0.1, batch
// statistics in training) -> ReLU (multi_model/utils/pn2_utils/nn/modules/conv.py:30-36, :70-76), and a set-abstraction
// block ends with torch.max over the K grouped neighbours (modules.py:245).  Through torch these are separate passes over
// activations of up to 1.3 GB: BN reads x twice and writes y, ReLU reads and writes y, max reads y; backward: the max
// scatter zero-fills and writes dy, ReLU backward reads y, dy and writes, BN backward reads x, dy twice and writes dx --
// 13 to 18 touches per element, HBM-bound.  Here:
//   forward   stats pass (1 read)  +  apply pass (1 read, 1 write; with pooling only the (B,C,M) maxima are written)
//   backward  reduce pass (2 reads; with pooling M gathered elements)  +  apply pass (2 reads / 1 read, 1 write)
// The ReLU mask is recomputed from x with the same expression as the forward, so y is not needed by the backward.
//
// Layout: x (B, C, L) contiguous, channel statistics over (B, L); L = M * K for grouped tensors with the K neighbours of
// a centroid innermost (what conv2d over (B,C,M,K) produces).  Per-channel sums are accumulated in fp64 (thread partials
// of <= 32 fp32 values, then fp64 wave/workgroup reduction and one fp64 atomic per workgroup).
#include "common.h"

#define BN_T 256
#define BN_CHUNK 8192   // elements of one (b, c) row per workgroup: 256 threads x 8 x float4
#define BN_STATS_CHUNK 32768   // the statistics pass: a workgroup's reduction + two atomics cost as much as reading 32 KB -- measured
                               // 2.6 TB/s with 8 192-element chunks where the apply pass streams at 5.6

__device__ __forceinline__ float bn_value(float x, float mean, float invstd, float gamma, float beta) {
  return gamma * (x - mean) * invstd + beta;   // association of torch's batch_norm_transform_input
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// adds the workgroup's (a, b) to sums[0], sums[1]
__device__ __forceinline__ void block_accumulate(double a, double b, double* sums) {
  __shared__ double red[2][BN_T / 64];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = a; red[1][w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < BN_T / 64; ++i) { sa += red[0][i]; sb += red[1][i]; }
    atomicAdd(sums, sa);
    atomicAdd(sums + 1, sb);
  }
}

// ---- forward -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BN_T) void bn_stats_kernel(const float* __restrict__ x, int C, int64_t L,
                                                        double* __restrict__ sums) {
  const int c = blockIdx.y;
  const float* row = x + ((int64_t)blockIdx.z * C + c) * L;
  const int64_t beg = (int64_t)blockIdx.x * BN_STATS_CHUNK, end = min(L, beg + BN_STATS_CHUNK);
  double s = 0.0, q = 0.0;          // fp32 over 8 float4 (32 values) at a time, fp64 across them
  if ((L & 3) == 0) {
    for (int64_t i0 = beg + threadIdx.x * 4; i0 < end; i0 += BN_T * 4 * 8) {
      float s32 = 0.f, q32 = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = i0 + (int64_t)u * BN_T * 4;
        if (i < end) {
          const float4 v = *reinterpret_cast<const float4*>(row + i);
          s32 += (v.x + v.y) + (v.z + v.w);
          q32 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
      }
      s += (double)s32;
      q += (double)q32;
    }
  } else {
    float s32 = 0.f, q32 = 0.f;
    int n = 0;
    for (int64_t i = beg + threadIdx.x; i < end; i += BN_T) {
      const float v = row[i];
      s32 += v; q32 += v * v;
      if (++n == 32) { s += (double)s32; q += (double)q32; s32 = q32 = 0.f; n = 0; }
    }
    s += (double)s32; q += (double)q32;
  }
  block_accumulate(s, q, sums + 2 * c);
}

// one thread per channel: batch mean / inverse std (saved for the backward), running statistics (momentum update with
// the unbiased variance, as torch)
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double n, float eps, float momentum, int C,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                   const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr,
                                   float* __restrict__ scale = nullptr, float* __restrict__ shift = nullptr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = sums[2 * c] / n;
  double var = sums[2 * c + 1] / n - mean * mean;
  if (var < 0) var = 0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (scale) {      // y = scale * x + shift: the normalisation as the affine a consuming contraction applies to its operand
    const float sc = gamma[c] * save_invstd[c];
    scale[c] = sc;
    shift[c] = beta[c] - save_mean[c] * sc;
  }
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unbiased = n > 1 ? var * n / (n - 1) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(BN_T) void bn_apply_kernel(const float* __restrict__ x, int C, int64_t L,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ save_mean,
                                                        const float* __restrict__ save_invstd, int relu,
                                                        float* __restrict__ y) {
  const int c = blockIdx.y;
  const int64_t off = ((int64_t)blockIdx.z * C + c) * L;
  const float mean = save_mean[c], invstd = save_invstd[c], g = gamma[c], bt = beta[c];
  const float lo = relu ? 0.f : -INFINITY;
  const int64_t beg = (int64_t)blockIdx.x * BN_CHUNK, end = min(L, beg + BN_CHUNK);
  if ((L & 3) == 0) {
    for (int64_t i = beg + threadIdx.x * 4; i < end; i += BN_T * 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + off + i);
      float4 o;
      o.x = fmaxf(bn_value(v.x, mean, invstd, g, bt), lo);
      o.y = fmaxf(bn_value(v.y, mean, invstd, g, bt), lo);
      o.z = fmaxf(bn_value(v.z, mean, invstd, g, bt), lo);
      o.w = fmaxf(bn_value(v.w, mean, invstd, g, bt), lo);
      *reinterpret_cast<float4*>(y + off + i) = o;
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += BN_T) y[off + i] = fmaxf(bn_value(x[off + i], mean, invstd, g, bt), lo);
  }
}

// BN + ReLU + max over each run of `group` consecutive elements (group = 4 * lanes, lanes a power of two <= 64): a lane
// holds 4 elements, `lanes` adjacent lanes one group.  Writes the maxima and the position (0..group-1, smallest on
// ties) of the element that produced them.
__global__ __launch_bounds__(BN_T) void bn_pool_kernel(const float* __restrict__ x, int C, int64_t L, int group,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_invstd, int relu,
                                                       float* __restrict__ y, int* __restrict__ index) {
  const int c = blockIdx.y;
  const int64_t row = (int64_t)blockIdx.z * C + c;
  const float mean = save_mean[c], invstd = save_invstd[c], g = gamma[c], bt = beta[c];
  const int lanes = group >> 2;
  const int64_t beg = (int64_t)blockIdx.x * BN_CHUNK, end = min(L, beg + BN_CHUNK);
  const int64_t M = L / group;
  for (int64_t i = beg + threadIdx.x * 4; i < end; i += BN_T * 4) {   // uniform trip count per wave: L % 256 == 0 not needed,
    const float4 v = *reinterpret_cast<const float4*>(x + row * L + i);  // a group never straddles `end` (BN_CHUNK % group == 0)
    const int k0 = (int)(i % group);
    float best = bn_value(v.x, mean, invstd, g, bt);
    int arg = k0;
    float t = bn_value(v.y, mean, invstd, g, bt);
    if (t > best) { best = t; arg = k0 + 1; }
    t = bn_value(v.z, mean, invstd, g, bt);
    if (t > best) { best = t; arg = k0 + 2; }
    t = bn_value(v.w, mean, invstd, g, bt);
    if (t > best) { best = t; arg = k0 + 3; }
    for (int o = 1; o < lanes; o <<= 1) {
      const float ob = __shfl_xor(best, o);
      const int oa = __shfl_xor(arg, o);
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (k0 == 0) {
      const int64_t m = i / group;
      y[row * M + m] = relu ? fmaxf(best, 0.f) : best;
      index[row * M + m] = arg;
    }
  }
}

// ---- backward ------------------------------------------------------------------------------------------------------
// sums[c] = (sum g, sum g * xhat) with g = dy masked by the ReLU
__global__ __launch_bounds__(BN_T) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             int C, int64_t L, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ save_mean,
                                                             const float* __restrict__ save_invstd, int relu,
                                                             double* __restrict__ sums) {
  const int c = blockIdx.y;
  const int64_t off = ((int64_t)blockIdx.z * C + c) * L;
  const float mean = save_mean[c], invstd = save_invstd[c], g = gamma[c], bt = beta[c];
  const int64_t beg = (int64_t)blockIdx.x * BN_CHUNK, end = min(L, beg + BN_CHUNK);
  float s = 0.f, q = 0.f;
  auto one = [&](float xv, float d) {
    if (relu && !(bn_value(xv, mean, invstd, g, bt) > 0.f)) d = 0.f;
    s += d;
    q += d * ((xv - mean) * invstd);
  };
  if ((L & 3) == 0) {
    for (int64_t i = beg + threadIdx.x * 4; i < end; i += BN_T * 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + off + i);
      const float4 d = *reinterpret_cast<const float4*>(dy + off + i);
      one(v.x, d.x); one(v.y, d.y); one(v.z, d.z); one(v.w, d.w);
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += BN_T) one(x[off + i], dy[off + i]);
  }
  block_accumulate((double)s, (double)q, sums + 2 * c);
}

// pooled variant: the gradient of a (b, c, m) maximum goes to the single element that produced it
__global__ __launch_bounds__(BN_T) void bn_pool_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  const float* __restrict__ dy,
                                                                  const int* __restrict__ index, int C, int64_t M,
                                                                  int group, const float* __restrict__ save_mean,
                                                                  const float* __restrict__ save_invstd, int relu,
                                                                  double* __restrict__ sums) {
  const int c = blockIdx.y;
  const int64_t row = (int64_t)blockIdx.z * C + c;
  const float mean = save_mean[c], invstd = save_invstd[c];
  float s = 0.f, q = 0.f;
  const int64_t beg = (int64_t)blockIdx.x * BN_CHUNK, end = min(M, beg + BN_CHUNK);
  for (int64_t m = beg + threadIdx.x; m < end; m += BN_T) {
    float d = dy[row * M + m];
    if (relu && !(y[row * M + m] > 0.f)) d = 0.f;
    const float xv = x[(row * M + m) * group + index[row * M + m]];
    s += d;
    q += d * ((xv - mean) * invstd);
  }
  block_accumulate((double)s, (double)q, sums + 2 * c);
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, int C, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)sums[2 * c];
  dgamma[c] = (float)sums[2 * c + 1];
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat))        (torch batch_norm_backward, training)
__global__ __launch_bounds__(BN_T) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            int C, int64_t L, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_invstd, int relu,
                                                            const double* __restrict__ sums, double n,
                                                            float* __restrict__ dx) {
  const int c = blockIdx.y;
  const int64_t off = ((int64_t)blockIdx.z * C + c) * L;
  const float mean = save_mean[c], invstd = save_invstd[c], g = gamma[c], bt = beta[c];
  const float k1 = (float)(sums[2 * c] / n), k2 = (float)(sums[2 * c + 1] / n), f = g * invstd;
  const int64_t beg = (int64_t)blockIdx.x * BN_CHUNK, end = min(L, beg + BN_CHUNK);
  auto one = [&](float xv, float d) {
    if (relu && !(bn_value(xv, mean, invstd, g, bt) > 0.f)) d = 0.f;
    return (d - k1 - (xv - mean) * invstd * k2) * f;
  };
  if ((L & 3) == 0) {
    for (int64_t i = beg + threadIdx.x * 4; i < end; i += BN_T * 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + off + i);
      const float4 d = *reinterpret_cast<const float4*>(dy + off + i);
      float4 o;
      o.x = one(v.x, d.x); o.y = one(v.y, d.y); o.z = one(v.z, d.z); o.w = one(v.w, d.w);
      *reinterpret_cast<float4*>(dx + off + i) = o;
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += BN_T) dx[off + i] = one(x[off + i], dy[off + i]);
  }
}

__global__ __launch_bounds__(BN_T) void bn_pool_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                 const float* __restrict__ dy,
                                                                 const int* __restrict__ index, int C, int64_t L,
                                                                 int group, const float* __restrict__ gamma,
                                                                 const float* __restrict__ save_mean,
                                                                 const float* __restrict__ save_invstd, int relu,
                                                                 const double* __restrict__ sums, double n,
                                                                 float* __restrict__ dx) {
  const int c = blockIdx.y;
  const int64_t row = (int64_t)blockIdx.z * C + c;
  const float mean = save_mean[c], invstd = save_invstd[c];
  const float k1 = (float)(sums[2 * c] / n), k2 = (float)(sums[2 * c + 1] / n), f = gamma[c] * invstd;
  const int64_t M = L / group;
  const int64_t beg = (int64_t)blockIdx.x * BN_CHUNK, end = min(L, beg + BN_CHUNK);
  for (int64_t i = beg + threadIdx.x * 4; i < end; i += BN_T * 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + row * L + i);
    const int64_t m = i / group;
    const int k0 = (int)(i - m * group);
    float d = dy[row * M + m];
    if (relu && !(y[row * M + m] > 0.f)) d = 0.f;
    const int a = index[row * M + m] - k0;   // 0..3 when the selected element is one of this lane's four
    float4 o;
    o.x = ((a == 0 ? d : 0.f) - k1 - (v.x - mean) * invstd * k2) * f;
    o.y = ((a == 1 ? d : 0.f) - k1 - (v.y - mean) * invstd * k2) * f;
    o.z = ((a == 2 ? d : 0.f) - k1 - (v.z - mean) * invstd * k2) * f;
    o.w = ((a == 3 ? d : 0.f) - k1 - (v.w - mean) * invstd * k2) * f;
    *reinterpret_cast<float4*>(dx + row * L + i) = o;
  }
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------
static inline bool pool_ok(int64_t L, int64_t group) {
  if (group < 4 || group > 256 || (group & (group - 1)) || L % group) return false;
  return BN_CHUNK % group == 0;
}

static inline bool bn_dims_ok(int64_t B, int64_t C, int64_t L) {
  return B <= 65535 && C <= 65535 && (L + BN_CHUNK - 1) / BN_CHUNK < ((int64_t)1 << 31);
}

extern "C" int64_t regnet_bn_workspace_bytes(int64_t C) { return C > 0 ? C * 2 * (int64_t)sizeof(double) : 0; }

static int bn_relu_train_fwd(const float* x, int64_t B, int64_t C, int64_t L, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, int relu, int64_t pool_group, float* y, int32_t* pool_index,
                             float* save_mean, float* save_invstd, void* workspace, void* stream, bool sums_ready) {
  if (B < 0 || C < 0 || L < 0 || pool_group < 0) return REGNET_ERR_SHAPE;
  if (B == 0 || C == 0 || L == 0) return REGNET_OK;
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace || (pool_group && !pool_index))
    return REGNET_ERR_NULL;
  if (!bn_dims_ok(B, C, L) || (pool_group && !pool_ok(L, pool_group))) return REGNET_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  double* sums = static_cast<double*>(workspace);
  dim3 grid((unsigned)((L + BN_CHUNK - 1) / BN_CHUNK), (unsigned)C, (unsigned)B);
  if (!sums_ready) {
    hipError_t e = hipMemsetAsync(sums, 0, regnet_bn_workspace_bytes(C), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)((L + BN_STATS_CHUNK - 1) / BN_STATS_CHUNK), (unsigned)C, (unsigned)B), dim3(BN_T), 0, st,
                       x, (int)C, L, sums);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, st, sums, (double)B * (double)L,
                     eps, momentum, (int)C, running_mean, running_var, save_mean, save_invstd);
  if (pool_group)
    hipLaunchKernelGGL(bn_pool_kernel, grid, dim3(BN_T), 0, st, x, (int)C, L, (int)pool_group, gamma, beta, save_mean,
                       save_invstd, relu, y, pool_index);
  else
    hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(BN_T), 0, st, x, (int)C, L, gamma, beta, save_mean, save_invstd, relu, y);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_bn_relu_train_fwd_f32(const float* x, int64_t B, int64_t C, int64_t L, const float* gamma,
                                            const float* beta, float eps, float momentum, float* running_mean,
                                            float* running_var, int relu, int64_t pool_group, float* y, int32_t* pool_index,
                                            float* save_mean, float* save_invstd, void* workspace, void* stream) {
  return bn_relu_train_fwd(x, B, C, L, gamma, beta, eps, momentum, running_mean, running_var, relu, pool_group, y, pool_index,
                           save_mean, save_invstd, workspace, stream, false);
}

// ... with the statistics pass already done: `workspace` holds the per-channel (sum, sum of squares) of x over all B L elements as
// regnet_conv1x1_fwd_stats_stream_f32, the convolution that produced x, left them
extern "C" int regnet_bn_relu_train_fwd_from_sums_f32(const float* x, int64_t B, int64_t C, int64_t L, const float* gamma,
                                                      const float* beta, float eps, float momentum, float* running_mean,
                                                      float* running_var, int relu, int64_t pool_group, float* y,
                                                      int32_t* pool_index, float* save_mean, float* save_invstd, void* workspace,
                                                      void* stream) {
  return bn_relu_train_fwd(x, B, C, L, gamma, beta, eps, momentum, running_mean, running_var, relu, pool_group, y, pool_index,
                           save_mean, save_invstd, workspace, stream, true);
}

extern "C" int regnet_bn_relu_train_bwd_f32(const float* x, const float* y, const float* dy, const int32_t* pool_index,
                                            int64_t B, int64_t C, int64_t L, const float* gamma, const float* beta,
                                            const float* save_mean, const float* save_invstd, int relu, int64_t pool_group,
                                            float* dx, float* dgamma, float* dbeta, void* workspace, void* stream) {
  if (B < 0 || C < 0 || L < 0 || pool_group < 0) return REGNET_ERR_SHAPE;
  if (B == 0 || C == 0 || L == 0) return REGNET_OK;
  if (!x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace ||
      (pool_group && (!pool_index || !y)))
    return REGNET_ERR_NULL;
  if (!bn_dims_ok(B, C, L) || (pool_group && !pool_ok(L, pool_group))) return REGNET_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  double* sums = static_cast<double*>(workspace);
  hipError_t e = hipMemsetAsync(sums, 0, regnet_bn_workspace_bytes(C), st);
  if (e != hipSuccess) return (int)e;
  const double n = (double)B * (double)L;
  dim3 grid((unsigned)((L + BN_CHUNK - 1) / BN_CHUNK), (unsigned)C, (unsigned)B);
  if (pool_group) {
    const int64_t M = L / pool_group;
    dim3 rgrid((unsigned)((M + BN_CHUNK - 1) / BN_CHUNK), (unsigned)C, (unsigned)B);
    hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, rgrid, dim3(BN_T), 0, st, x, y, dy, pool_index, (int)C, M, (int)pool_group,
                       save_mean, save_invstd, relu, sums);
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, grid, dim3(BN_T), 0, st, x, y, dy, pool_index, (int)C, L, (int)pool_group,
                       gamma, save_mean, save_invstd, relu, sums, n, dx);
  } else {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(BN_T), 0, st, x, dy, (int)C, L, gamma, beta, save_mean, save_invstd,
                       relu, sums);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(BN_T), 0, st, x, dy, (int)C, L, gamma, beta, save_mean, save_invstd,
                       relu, sums, n, dx);
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, st, sums, (int)C, dgamma, dbeta);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// The statistics half of the forward alone: batch mean / inverse std (+ running statistics), and the normalisation as a
// per-channel affine (scale = gamma * invstd, shift = beta - mean * scale) for a consumer that applies it itself
// (regnet_conv1x1_fwd_bnrelu_stream_f32 / regnet_conv1x1_wgrad_bnrelu_f32): the normalised activation is never written.
static int bn_train_stats(const float* x, int64_t B, int64_t C, int64_t L, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                          float* save_invstd, float* scale, float* shift, void* workspace, void* stream) {
  if (B < 0 || C < 0 || L < 0) return REGNET_ERR_SHAPE;
  if (B == 0 || C == 0 || L == 0) return REGNET_OK;
  if (!gamma || !beta || !save_mean || !save_invstd || !scale || !shift || !workspace) return REGNET_ERR_NULL;
  if (!bn_dims_ok(B, C, L)) return REGNET_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  double* sums = static_cast<double*>(workspace);
  if (x) {      // x == NULL: `workspace` already holds the sums
    hipError_t e = hipMemsetAsync(sums, 0, regnet_bn_workspace_bytes(C), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)((L + BN_STATS_CHUNK - 1) / BN_STATS_CHUNK), (unsigned)C, (unsigned)B), dim3(BN_T), 0, st,
                       x, (int)C, L, sums);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, st, sums, (double)B * (double)L,
                     eps, momentum, (int)C, running_mean, running_var, save_mean, save_invstd, gamma, beta, scale, shift);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_bn_train_stats_f32(const float* x, int64_t B, int64_t C, int64_t L, const float* gamma, const float* beta,
                                         float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                                         float* save_invstd, float* scale, float* shift, void* workspace, void* stream) {
  if (!x && B > 0 && C > 0 && L > 0) return REGNET_ERR_NULL;
  return bn_train_stats(x, B, C, L, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift,
                        workspace, stream);
}

// ... from per-channel (sum, sum of squares) already in `workspace` (regnet_conv1x1_fwd_stats_stream_f32): the finalize step alone
extern "C" int regnet_bn_train_stats_from_sums_f32(int64_t B, int64_t C, int64_t L, const float* gamma, const float* beta, float eps,
                                                   float momentum, float* running_mean, float* running_var, float* save_mean,
                                                   float* save_invstd, float* scale, float* shift, void* workspace, void* stream) {
  return bn_train_stats(nullptr, B, C, L, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale,
                        shift, workspace, stream);
}
