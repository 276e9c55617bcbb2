// gather.hip -- group_points / interpolate forward+backward, gather_knn (gfx950).
//
// These are HBM-bound gathers / scatter-adds at the reference's operator granularity (the fused
// set-abstraction path in fused_mlp.hip never materialises them).  Layout: consecutive threads
// walk the innermost (n,k) axis of the contiguous output so stores coalesce; each thread reads
// its int64 index once and loops over a slab of channels.
//
// Reference behaviour restated (relative to multi_model/utils/pn2_utils/):
//   group fwd   csrc/grouping_kernel.cu:29-51     group bwd   csrc/grouping_kernel.cu:54-93
//   interp fwd  csrc/interpolate_kernel.cu:134-177 interp bwd csrc/interpolate_kernel.cu:239-282
//   gather_knn  functions/csrc/gather_knn_kernel.cu:27-92
#include "common.h"

#define GT 256
#define CH_PER_BLOCK 16

__global__ __launch_bounds__(GT) void group_fwd_kernel(const float* __restrict__ in, int64_t sb, int64_t sc,
                                                       int64_t sn, const int64_t* __restrict__ index, int C, int N1,
                                                       int64_t NK, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int64_t e = (int64_t)blockIdx.x * GT + threadIdx.x;
  if (e >= NK) return;
  const int64_t j = index[(int64_t)b * NK + e];
  const bool ok = j >= 0 && j < N1;  // the reference asserts; out-of-range rows read as 0
  const int cbeg = blockIdx.y * CH_PER_BLOCK, cend = min(C, cbeg + CH_PER_BLOCK);
  const float* src = in + (int64_t)b * sb + (ok ? j : 0) * sn;
  float* dst = out + ((int64_t)b * C) * NK + e;
  for (int c = cbeg; c < cend; ++c) dst[(int64_t)c * NK] = ok ? src[(int64_t)c * sc] : 0.f;
}

__global__ __launch_bounds__(GT) void group_bwd_kernel(const float* __restrict__ go, int64_t sb, int64_t sc,
                                                       int64_t sn2, int64_t sk, const int64_t* __restrict__ index,
                                                       int C, int N1, int N2, int K, float* __restrict__ gi) {
  const int b = blockIdx.z;
  const int64_t NK = (int64_t)N2 * K;
  const int64_t e = (int64_t)blockIdx.x * GT + threadIdx.x;
  if (e >= NK) return;
  const int64_t j = index[(int64_t)b * NK + e];
  if (j < 0 || j >= N1) return;
  const int64_t n = e / K, k = e - n * K;
  const int cbeg = blockIdx.y * CH_PER_BLOCK, cend = min(C, cbeg + CH_PER_BLOCK);
  const float* src = go + (int64_t)b * sb + n * sn2 + k * sk;
  float* dst = gi + ((int64_t)b * C) * N1 + j;
  for (int c = cbeg; c < cend; ++c) atomicAdd(dst + (int64_t)c * N1, src[(int64_t)c * sc]);
}

__global__ __launch_bounds__(GT) void interp_fwd_kernel(const float* __restrict__ in, int64_t sb, int64_t sc,
                                                        int64_t sm, const int64_t* __restrict__ index,
                                                        const float* __restrict__ weight, int C, int M, int N,
                                                        float* __restrict__ out) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * GT + threadIdx.x;
  if (n >= N) return;
  const int64_t o = ((int64_t)b * N + n) * 3;
  const int64_t j0 = index[o], j1 = index[o + 1], j2 = index[o + 2];
  const float w0 = weight[o], w1 = weight[o + 1], w2 = weight[o + 2];
  const int cbeg = blockIdx.y * CH_PER_BLOCK, cend = min(C, cbeg + CH_PER_BLOCK);
  const float* src = in + (int64_t)b * sb;
  float* dst = out + ((int64_t)b * C) * N + n;
  for (int c = cbeg; c < cend; ++c) {
    const float* s = src + (int64_t)c * sc;
    float acc = 0.f;  // accumulate from 0 in k order (interpolate_kernel.cu:165-170)
    acc += s[j0 * sm] * w0;
    acc += s[j1 * sm] * w1;
    acc += s[j2 * sm] * w2;
    dst[(int64_t)c * N] = acc;
  }
}

__global__ __launch_bounds__(GT) void interp_bwd_kernel(const float* __restrict__ go, int64_t sb, int64_t sc,
                                                        int64_t sn, const int64_t* __restrict__ index,
                                                        const float* __restrict__ weight, int C, int M, int N,
                                                        float* __restrict__ gi) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * GT + threadIdx.x;
  if (n >= N) return;
  const int64_t o = ((int64_t)b * N + n) * 3;
  const int64_t j0 = index[o], j1 = index[o + 1], j2 = index[o + 2];
  const float w0 = weight[o], w1 = weight[o + 1], w2 = weight[o + 2];
  const int cbeg = blockIdx.y * CH_PER_BLOCK, cend = min(C, cbeg + CH_PER_BLOCK);
  const float* src = go + (int64_t)b * sb + (int64_t)n * sn;
  float* dst = gi + ((int64_t)b * C) * M;
  for (int c = cbeg; c < cend; ++c) {
    const float g = src[(int64_t)c * sc];
    float* d = dst + (int64_t)c * M;
    atomicAdd(d + j0, g * w0);
    atomicAdd(d + j1, g * w1);
    atomicAdd(d + j2, g * w2);
  }
}

// ---------------------------------------------------------------------------------------
// Scatter-add backward of group_points / interpolate with the accumulators in LDS.
// gi[b, c, idx[b, l, k]] += w[b, l, k] * go[b, c, l]  (k < KK; group: KK = 1, w = 1; interpolate: KK = 3).
// The global-atomic kernels above issue one 4-byte atomic per (channel, source element, k) to scattered addresses
// (157 M of them, 3.1 ms, for the level-1 feature propagation of 4 scenes).  Here a workgroup owns `cpb` channels of
// one scene, keeps their whole output rows (cpb x R floats) in LDS, walks ALL source elements with coalesced reads and
// ds_add_f32, and writes the rows out with plain stores: no global atomics and no zero fill.  Used when a row fits
// (R <= 36 864 floats); summation order still varies from run to run, as with any atomic scatter-add.
#define SCAT_T 1024
#define SCAT_LDS_FLOATS 36864   // 144 KB of accumulators

template <int KK>
__global__ __launch_bounds__(SCAT_T) void scatter_add_lds_kernel(const float* __restrict__ go, int64_t sb, int64_t sc,
                                                                 int64_t s_hi, int64_t s_lo, int inner,
                                                                 const int64_t* __restrict__ index,
                                                                 const float* __restrict__ weight, int C, int R, int64_t L,
                                                                 int cpb, float* __restrict__ gi) {
  extern __shared__ float acc[];   // [cpb][R]
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * cpb, nc = min(cpb, C - c0);
  for (int i = threadIdx.x; i < nc * R; i += SCAT_T) acc[i] = 0.f;
  __syncthreads();
  const float* src = go + (int64_t)b * sb + (int64_t)c0 * sc;
  for (int64_t l = threadIdx.x; l < L; l += SCAT_T) {
    // source element l = (hi, lo) with lo < inner: address hi * s_hi + lo * s_lo (interpolate: inner = 1)
    const int64_t hi = inner > 1 ? l / inner : l, lo = inner > 1 ? l - hi * inner : 0;
    const float* e = src + hi * s_hi + lo * s_lo;
    int64_t j[KK];
    float w[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      j[k] = index[((int64_t)b * L + l) * KK + k];
      w[k] = weight ? weight[((int64_t)b * L + l) * KK + k] : 1.f;
      if (j[k] < 0 || j[k] >= R) { j[k] = 0; w[k] = 0.f; }
    }
    for (int c = 0; c < nc; ++c) {
      const float g = e[(int64_t)c * sc];
#pragma unroll
      for (int k = 0; k < KK; ++k) atomicAdd(&acc[c * R + (int)j[k]], g * w[k]);
    }
  }
  __syncthreads();
  float* dst = gi + ((int64_t)b * C + c0) * R;
  for (int i = threadIdx.x; i < nc * R; i += SCAT_T) dst[i] = acc[i];
}

template <int KK>
static int launch_scatter_lds(const float* go, int64_t sb, int64_t sc, int64_t s_hi, int64_t s_lo, int64_t inner,
                              const int64_t* index, const float* weight, int64_t B, int64_t C, int64_t R, int64_t L,
                              float* gi, hipStream_t st) {
  int cpb = (int)(SCAT_LDS_FLOATS / R);
  if (cpb > 32) cpb = 32;
  // enough workgroups to fill the chip when there are many channels
  while (cpb > 1 && B * ((C + cpb - 1) / cpb) < 512) cpb = (cpb + 1) / 2;
  const size_t lds = (size_t)cpb * R * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)scatter_add_lds_kernel<KK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SCAT_LDS_FLOATS * (int)sizeof(float));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid((unsigned)((C + cpb - 1) / cpb), (unsigned)B);
  hipLaunchKernelGGL((scatter_add_lds_kernel<KK>), grid, dim3(SCAT_T), lds, st, go, sb, sc, s_hi, s_lo, (int)inner, index,
                     weight, (int)C, (int)R, L, cpb, gi);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

static inline bool dims_ok(int64_t B, int64_t C) { return B <= 65535 && (C + CH_PER_BLOCK - 1) / CH_PER_BLOCK <= 65535; }

extern "C" int regnet_group_points_fwd_f32(const float* input, int64_t sb, int64_t sc, int64_t sn,
                                           const int64_t* index, int64_t B, int64_t C, int64_t N1, int64_t N2,
                                           int64_t K, float* out, void* stream) {
  if (B < 0 || C < 0 || N1 < 0 || N2 < 0 || K < 0) return REGNET_ERR_SHAPE;
  const int64_t NK = N2 * K;
  if (B == 0 || C == 0 || NK == 0) return REGNET_OK;
  if (!dims_ok(B, C) || N1 >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (!input || !index || !out) return REGNET_ERR_NULL;
  dim3 grid((unsigned)((NK + GT - 1) / GT), (unsigned)((C + CH_PER_BLOCK - 1) / CH_PER_BLOCK), (unsigned)B);
  hipLaunchKernelGGL(group_fwd_kernel, grid, dim3(GT), 0, as_stream(stream), input, sb, sc, sn, index, (int)C,
                     (int)N1, NK, out);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_group_points_bwd_f32(const float* grad_out, int64_t sb, int64_t sc, int64_t sn2, int64_t sk,
                                           const int64_t* index, int64_t B, int64_t C, int64_t N1, int64_t N2,
                                           int64_t K, float* grad_in, void* stream) {
  if (B < 0 || C < 0 || N1 < 0 || N2 < 0 || K < 0) return REGNET_ERR_SHAPE;
  if (B == 0 || C == 0 || N1 == 0) return REGNET_OK;
  if (!grad_in) return REGNET_ERR_NULL;
  const int64_t NK = N2 * K;
  if (NK > 0 && N1 <= SCAT_LDS_FLOATS && B <= 65535 && C < (int64_t)1 << 31 && K < (int64_t)1 << 31) {
    if (!grad_out || !index) return REGNET_ERR_NULL;
    return launch_scatter_lds<1>(grad_out, sb, sc, sn2, sk, K, index, nullptr, B, C, N1, NK, grad_in, as_stream(stream));
  }
  hipError_t e = hipMemsetAsync(grad_in, 0, sizeof(float) * (size_t)(B * C * N1), as_stream(stream));
  if (e != hipSuccess) return (int)e;
  if (NK == 0) return REGNET_OK;
  if (!dims_ok(B, C) || N1 >= (int64_t)1 << 31 || N2 >= (int64_t)1 << 31 || K >= (int64_t)1 << 31)
    return REGNET_ERR_UNSUPPORTED;
  if (!grad_out || !index) return REGNET_ERR_NULL;
  dim3 grid((unsigned)((NK + GT - 1) / GT), (unsigned)((C + CH_PER_BLOCK - 1) / CH_PER_BLOCK), (unsigned)B);
  hipLaunchKernelGGL(group_bwd_kernel, grid, dim3(GT), 0, as_stream(stream), grad_out, sb, sc, sn2, sk, index, (int)C,
                     (int)N1, (int)N2, (int)K, grad_in);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_interpolate_fwd_f32(const float* input, int64_t sb, int64_t sc, int64_t sm,
                                          const int64_t* index, const float* weight, int64_t B, int64_t C, int64_t M,
                                          int64_t N, float* out, void* stream) {
  if (B < 0 || C < 0 || M < 0 || N < 0) return REGNET_ERR_SHAPE;
  if (B == 0 || C == 0 || N == 0) return REGNET_OK;
  if (M == 0) return REGNET_ERR_SHAPE;
  if (!dims_ok(B, C) || N >= (int64_t)1 << 31 || M >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (!input || !index || !weight || !out) return REGNET_ERR_NULL;
  dim3 grid((unsigned)((N + GT - 1) / GT), (unsigned)((C + CH_PER_BLOCK - 1) / CH_PER_BLOCK), (unsigned)B);
  hipLaunchKernelGGL(interp_fwd_kernel, grid, dim3(GT), 0, as_stream(stream), input, sb, sc, sm, index, weight,
                     (int)C, (int)M, (int)N, out);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_interpolate_bwd_f32(const float* grad_out, int64_t sb, int64_t sc, int64_t sn,
                                          const int64_t* index, const float* weight, int64_t B, int64_t C, int64_t M,
                                          int64_t N, float* grad_in, void* stream) {
  if (B < 0 || C < 0 || M < 0 || N < 0) return REGNET_ERR_SHAPE;
  if (B == 0 || C == 0 || M == 0) return REGNET_OK;
  if (!grad_in) return REGNET_ERR_NULL;
  if (N > 0 && M <= SCAT_LDS_FLOATS && B <= 65535 && C < (int64_t)1 << 31) {
    if (!grad_out || !index || !weight) return REGNET_ERR_NULL;
    return launch_scatter_lds<3>(grad_out, sb, sc, sn, 0, 1, index, weight, B, C, M, N, grad_in, as_stream(stream));
  }
  hipError_t e = hipMemsetAsync(grad_in, 0, sizeof(float) * (size_t)(B * C * M), as_stream(stream));
  if (e != hipSuccess) return (int)e;
  if (N == 0) return REGNET_OK;
  if (!dims_ok(B, C) || N >= (int64_t)1 << 31 || M >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (!grad_out || !index || !weight) return REGNET_ERR_NULL;
  dim3 grid((unsigned)((N + GT - 1) / GT), (unsigned)((C + CH_PER_BLOCK - 1) / CH_PER_BLOCK), (unsigned)B);
  hipLaunchKernelGGL(interp_bwd_kernel, grid, dim3(GT), 0, as_stream(stream), grad_out, sb, sc, sn, index, weight,
                     (int)C, (int)M, (int)N, grad_in);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_gather_knn_fwd_f32(const float* input, int64_t sb, int64_t sc, int64_t sn, const int64_t* index,
                                         int64_t B, int64_t C, int64_t N, int64_t NI, int64_t K, float* out,
                                         void* stream) {
  return regnet_group_points_fwd_f32(input, sb, sc, sn, index, B, C, N, NI, K, out, stream);
}

extern "C" int regnet_gather_knn_bwd_f32(const float* grad_out, int64_t sb, int64_t sc, int64_t sn2, int64_t sk,
                                         const int64_t* index, int64_t B, int64_t C, int64_t N, int64_t NI,
                                         int64_t K, float* grad_in, void* stream) {
  return regnet_group_points_bwd_f32(grad_out, sb, sc, sn2, sk, index, B, C, N, NI, K, grad_in, stream);
}

// ---------------------------------------------------------------------------------------
// Row packer for the per-source-point evaluation of a set-abstraction block's first layer (fused.sa_features):
//   out[b*N + n][0:Cf] = feat[b, 0:Cf, n]   out[..][Cf:Cf+3] = xyz[b, 0:3, n]   out[..][Cf+3:W] = 0
// one launch instead of a zero fill and two strided copies.  feat element (b,c,n) at feat[b*fb + c*fc + n*fn].
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ feat, long long fb, long long fc,
                                                        long long fn, int Cf, const float* __restrict__ xyz, long long xb,
                                                        long long xc, long long xn, const float* __restrict__ mu, long long N,
                                                        int W, float* __restrict__ out, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one output element per thread, channel fastest
  if (i >= total) return;
  const long long row = i / W;
  const int c = (int)(i - row * W);
  const long long b = row / N, n = row - b * N;
  float v = 0.f;
  if (c < Cf) v = feat[b * fb + (long long)c * fc + n * fn];
  else if (c < Cf + 3) {
    v = xyz[b * xb + (long long)(c - Cf) * xc + n * xn];
    if (mu) v = v - mu[3 * b + (c - Cf)];          // one fp32 subtraction: what `xyz - mu` computed as a tensor first
  }
  out[i] = v;
}

extern "C" int regnet_pack_rows_centred_f32(const float* feat, int64_t fb, int64_t fc, int64_t fn, int64_t Cf, const float* xyz,
                                            int64_t xb, int64_t xc, int64_t xn, const float* mu, int64_t B, int64_t N,
                                            int64_t W, float* out, void* stream) {
  if (B < 0 || N < 0 || Cf < 0 || W < Cf + 3) return REGNET_ERR_SHAPE;
  const long long total = B * N * W;
  if (total == 0) return REGNET_OK;
  if (!xyz || !out || (Cf > 0 && !feat)) return REGNET_ERR_NULL;
  const long long blocks = (total + 255) / 256;
  if (blocks >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), feat, (long long)fb,
                     (long long)fc, (long long)fn, (int)Cf, xyz, (long long)xb, (long long)xc, (long long)xn, mu,
                     (long long)N, (int)W, out, total);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_pack_rows_f32(const float* feat, int64_t fb, int64_t fc, int64_t fn, int64_t Cf, const float* xyz,
                                    int64_t xb, int64_t xc, int64_t xn, int64_t B, int64_t N, int64_t W, float* out,
                                    void* stream) {
  return regnet_pack_rows_centred_f32(feat, fb, fc, fn, Cf, xyz, xb, xc, xn, nullptr, B, N, W, out, stream);
}

// ---- gather_points (pn2_utils/function.py:11-26): out[b][c][m] = points[b][c][index[b][m]], any strides on both sides.
// One thread per output element, m fastest (the centroid gather of every set-abstraction level: (B,3,N) -> (B,3,M); with the
// strides swapped also rows of a (B,N,C) cloud: the grasp centres, get_regiondataset.py:288).  An index outside [0, N) flags
// the launch (status word) and writes 0 -- torch.gather raises for it.
__global__ __launch_bounds__(256) void gather_points_kernel(const float* __restrict__ pts, long long pb, long long pc,
                                                            long long pn, long long N, const long long* __restrict__ idx,
                                                            long long ib, long long im, int C, long long M,
                                                            float* __restrict__ out, long long ob, long long oc, long long om,
                                                            long long total, int* __restrict__ status) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long m = i % M;
  const long long t = i / M;
  const int c = (int)(t % C);
  const long long b = t / C;
  const long long n = idx[b * ib + m * im];
  float v = 0.f;
  if (n >= 0 && n < N) v = pts[b * pb + (long long)c * pc + n * pn];
  else if (status) atomicOr(status, 1);
  out[b * ob + (long long)c * oc + m * om] = v;
}

extern "C" int regnet_gather_points_f32(const float* points, int64_t pb, int64_t pc, int64_t pn, int64_t B, int64_t C, int64_t N,
                                        const int64_t* index, int64_t ib, int64_t im, int64_t M, float* out, int64_t ob,
                                        int64_t oc, int64_t om, int32_t* status, void* stream) {
  if (B < 0 || C < 0 || N < 0 || M < 0) return REGNET_ERR_SHAPE;
  const long long total = B * C * M;
  if (total == 0) return REGNET_OK;
  if (!points || !index || !out) return REGNET_ERR_NULL;
  const long long blocks = (total + 255) / 256;
  if (blocks >= (1ll << 31) || C >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gather_points_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), points, (long long)pb,
                     (long long)pc, (long long)pn, (long long)N, (const long long*)index, (long long)ib, (long long)im, (int)C,
                     (long long)M, out, (long long)ob, (long long)oc, (long long)om, total, status);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// ---- processing order of the level-1 neighbourhoods by cost class (fused.chain3_order): class = (count > 32) + (count > 48),
// order = the STABLE sort permutation by class (what torch.argsort(stable=True) of the class ids returns): a three-bin
// counting sort by one workgroup.  n <= 2^24 elements (8 x 5 120 in a step).
#define CO_WAVES 8
#define CO_THREADS (CO_WAVES * 64)
// Every wave owns a contiguous segment and walks it in coalesced 64-element steps: per class a ballot + popcount (pass 1: the
// segment's three totals; pass 2: position = class base + hits of the class in earlier segments + so far in this one + in lower
// lanes).  Two barriers, 2 x n / 1024 dependent-free 512-byte reads per wave (a thread-per-chunk version read 320-byte strides
// lane by lane: 80 us for 40 960 elements).
__global__ __launch_bounds__(CO_THREADS) void class_order_kernel(const long long* __restrict__ count, long long n,
                                                                 long long* __restrict__ order) {
  __shared__ int tot[3][CO_WAVES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const long long seg = ((n + CO_WAVES - 1) / CO_WAVES + 63) / 64 * 64;
  const long long lo = (long long)wave * seg, hi = lo + seg < n ? lo + seg : n;
  int c0 = 0, c1 = 0, c2 = 0;
  for (long long i0 = lo; i0 < hi; i0 += 64) {
    const long long i = i0 + lane;
    const long long c = i < hi ? count[i] : 0;
    const int k = i < hi ? (c > 32) + (c > 48) : 3;
    c0 += (int)__popcll(__ballot(k == 0)); c1 += (int)__popcll(__ballot(k == 1)); c2 += (int)__popcll(__ballot(k == 2));
  }
  if (lane == 0) { tot[0][wave] = c0; tot[1][wave] = c1; tot[2][wave] = c2; }
  __syncthreads();
  int all0 = 0, all1 = 0, p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
  for (int w = 0; w < CO_WAVES; ++w) {
    const int t0 = tot[0][w], t1 = tot[1][w], t2 = tot[2][w];
    all0 += t0; all1 += t1;
    if (w < wave) { p0 += t0; p1 += t1; p2 += t2; }
  }
  p1 += all0; p2 += all0 + all1;            // class bases: [class 0 | class 1 | class 2]
  for (long long i0 = lo; i0 < hi; i0 += 64) {
    const long long i = i0 + lane;
    const long long c = i < hi ? count[i] : 0;
    const int k = i < hi ? (c > 32) + (c > 48) : 3;
    const unsigned long long m0 = __ballot(k == 0), m1 = __ballot(k == 1), m2 = __ballot(k == 2);
    if (k == 0) order[p0 + (int)__popcll(m0 & lt_mask)] = i;
    else if (k == 1) order[p1 + (int)__popcll(m1 & lt_mask)] = i;
    else if (k == 2) order[p2 + (int)__popcll(m2 & lt_mask)] = i;
    p0 += (int)__popcll(m0); p1 += (int)__popcll(m1); p2 += (int)__popcll(m2);
  }
}

extern "C" int regnet_class_order_i64(const int64_t* count, int64_t n, int64_t* order, void* stream) {
  if (n < 0) return REGNET_ERR_SHAPE;
  if (n == 0) return REGNET_OK;
  if (n > (1ll << 24)) return REGNET_ERR_UNSUPPORTED;
  if (!count || !order) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(class_order_kernel, dim3(1), dim3(CO_THREADS), 0, as_stream(stream), (const long long*)count, (long long)n,
                     (long long*)order);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
