// tsplit.hip -- EXPERIMENT, off by default (conv1x1_train.SPLIT_PRODUCTS): the forward and the input gradient of the training
// path's 1x1 convolutions (tgemm.hip's tgemm_stream_kernel; pn2_utils/nn/modules/conv.py:20-36, :60-76 under autograd,
// train.py:376-384) with fp32-FAITHFUL products on the bf16 matrix pipe of gfx950.
//
// fp32 MFMA on this chip runs at the vector rate (134 TFLOP/s measured in a bare loop), v_mfma_f32_32x32x16_bf16 at 1 864
// (scripts/ablate/split_products.cpp).  An fp32 value is the exact sum of three bf16 pieces, x = x1 + x2 + x3 with x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (the residuals are exact in fp32); every bf16 x bf16 product is exact in fp32 and
// the instruction accumulates in fp32, so
//     x w  ~=  x3 w1 + x1 w3 + x2 w2 + x2 w1 + x1 w2 + x1 w1          (dropped: x2 w3, x3 w2, x3 w3 <= 2^-24 relative)
// is as close to the exact product sum as the fp32 instruction's own result (measured against float64, K = 256 / 1024: max 2.2e-6 /
// 4.5e-6 against 2.5e-6 / 5.3e-6 for v_mfma_f32_32x32x2_f32).  Six instructions of 16 k instead of eight of 2: 2.3x the
// fp32 instruction's ceiling.
//
// C[M x N] = A[M x K] . B[K x N] per scene, A = the weights (forward: W, M = Co, K = Ci; input gradient: W^T, M = Ci, K = Co)
// pre-split into three bf16 planes [plane][M padded to 128][K] by ts_planes_kernel, B = the activations / output gradients in
// the tensors' own channel-first layout (K rows of N = L contiguous points).  That layout is the B operand's own: lane (n, h) of
// v_mfma_f32_32x32x16_bf16 holds column n, k = 8 h .. 8 h + 7 -- eight coalesced dword loads straight from global memory, split in
// registers (optionally after the pending BatchNorm + ReLU of the layer below: tgemm.hip's B_AFFINE), no LDS, no transpose.  The
// A planes of a k-chunk go through LDS (12 KB per 16 k, double buffered, 16-byte chunks XOR-swizzled by row).  Workgroup = 4
// waves = a 128 x 256 tile, a wave 128 x 64 (8 accumulator tiles: every B fragment feeds four row tiles, every A fragment two
// column tiles).
#include "common.h"

typedef float ts_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ts_bf16x8 __attribute__((ext_vector_type(8)));

#define TS_BM 128
#define TS_BN 256
#define TS_THREADS 256
#define TS_AFF_MAXK 1024

struct TsArgs {
  const __bf16* Ap;            // [3][Mpad][K]
  const float* B; float* C;
  int M, Mpad, K;
  long long N;                 // points per scene (row length of B and C)
  long long b_batch, c_batch;  // scene strides (floats)
  const float* bscale; const float* bshift; int brelu;   // AFFINE: B is used as max(bscale[k] x + bshift[k], brelu ? 0 : -inf)
  int tiles_m;
};

// W (Co x Ci) row-major -> planes [3][Mpad][K]: transposed == 0: M = Co, K = Ci (plane[m][k] = W[m][k]); 1: M = Ci, K = Co
// (plane[m][k] = W[k][m]).  Rows M .. Mpad - 1 are zero.
__global__ __launch_bounds__(256) void ts_planes_kernel(const float* __restrict__ W, int Co, int Ci, int transposed, __bf16* __restrict__ Ap,
                                                         int M, int Mpad, int K) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)Mpad * K) return;
  const int m = (int)(i / K), k = (int)(i % K);
  float x = 0.f;
  if (m < M) x = transposed ? W[(long long)k * Ci + m] : W[(long long)m * Ci + k];
  const __bf16 a = (__bf16)x;
  const float r = x - (float)a;
  const __bf16 b = (__bf16)r;
  const __bf16 c = (__bf16)(r - (float)b);
  Ap[i] = a;
  Ap[(long long)Mpad * K + i] = b;
  Ap[2ll * Mpad * K + i] = c;
}

template <bool AFFINE>
__global__ __launch_bounds__(TS_THREADS, 2) void tsplit_kernel(const TsArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_a[2][3][TS_BM][32];     // [buffer][plane][row][16 k of bf16], 24 KB
  __shared__ __attribute__((aligned(16))) float btab[AFFINE ? 2 * TS_AFF_MAXK : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int tm = blockIdx.x % p.tiles_m;
  const long long tn = blockIdx.x / p.tiles_m;
  const int m0 = tm * TS_BM;
  const long long n0 = tn * TS_BN + 64 * wave;
  const float* Bb = p.B + (long long)blockIdx.y * p.b_batch;
  float* Cb = p.C + (long long)blockIdx.y * p.c_batch;
  if (AFFINE) {
    for (int i = tid; i < p.K; i += TS_THREADS) { btab[i] = p.bscale[i]; btab[TS_AFF_MAXK + i] = p.bshift[i]; }
  }
  const float blo = (AFFINE && !p.brelu) ? -INFINITY : 0.f;
  ts_f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // A chunk (3 planes x 128 rows x 16 k = 768 16-byte pieces): thread t moves pieces t, t + 256, t + 512 -> plane = piece / 256
  const int a_row = (tid & 255) >> 1, a_h = tid & 1;
  const __bf16* a_src[3];
  unsigned char* a_dst[2][3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    a_src[q] = p.Ap + ((long long)q * p.Mpad + m0 + a_row) * p.K + 8 * a_h;
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) a_dst[bf][q] = &lds_a[bf][q][a_row][16 * (a_h ^ ((a_row >> 3) & 1))];
  }
  const bool col_ok[2] = {n0 + n < p.N, n0 + 32 + n < p.N};
  const float* b_src[2] = {Bb + (col_ok[0] ? n0 + n : 0), Bb + (col_ok[1] ? n0 + 32 + n : 0)};

  const int chunks = p.K / 16;
  float4 a_next[3];
  float b_next[2][8];
  // ---- prologue: chunk 0
#pragma unroll
  for (int q = 0; q < 3; ++q) a_next[q] = *reinterpret_cast<const float4*>(a_src[q]);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int i = 0; i < 8; ++i) b_next[ni][i] = b_src[ni][(long long)(8 * h + i) * p.N];
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<float4*>(a_dst[0][q]) = a_next[q];
  __syncthreads();

  for (int kc = 0; kc < chunks; ++kc) {
    const int buf = kc & 1;
    float b_cur[2][8];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 8; ++i) b_cur[ni][i] = b_next[ni][i];
    if (kc + 1 < chunks) {      // the next chunk's operands leave now, land behind this chunk's MFMAs
      const int k1 = 16 * (kc + 1);
#pragma unroll
      for (int q = 0; q < 3; ++q) a_next[q] = *reinterpret_cast<const float4*>(a_src[q] + k1);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 8; ++i) b_next[ni][i] = b_src[ni][(long long)(k1 + 8 * h + i) * p.N];
    }
    // ---- B fragments: (pending BatchNorm + ReLU,) split into three bf16 pieces
    ts_bf16x8 b1[2], b2[2], b3[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float x = b_cur[ni][i];
        if (AFFINE) {
          const int k = 16 * kc + 8 * h + i;
          x = fmaxf(btab[k] * x + btab[TS_AFF_MAXK + k], blo);
        }
        if (!col_ok[ni]) x = 0.f;
        const __bf16 u = (__bf16)x;
        const float r = x - (float)u;
        const __bf16 v = (__bf16)r;
        const __bf16 w = (__bf16)(r - (float)v);
        b1[ni][i] = u; b2[ni][i] = v; b3[ni][i] = w;
      }
    // ---- A fragments of the four row tiles from LDS
    ts_bf16x8 a1[4], a2[4], a3[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int row = 32 * mi + n;
      const int off = 16 * (h ^ ((row >> 3) & 1));
      a1[mi] = *reinterpret_cast<const ts_bf16x8*>(&lds_a[buf][0][row][off]);
      a2[mi] = *reinterpret_cast<const ts_bf16x8*>(&lds_a[buf][1][row][off]);
      a3[mi] = *reinterpret_cast<const ts_bf16x8*>(&lds_a[buf][2][row][off]);
    }
    // ---- six products, small terms first; eight independent accumulators between two uses of the same one
#define TS_PRODUCT(AP, BP)                                                                                  \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                        \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                      \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AP[mi], BP[ni], acc[mi][ni], 0, 0, 0);
    TS_PRODUCT(a3, b1)
    TS_PRODUCT(a1, b3)
    TS_PRODUCT(a2, b2)
    TS_PRODUCT(a2, b1)
    TS_PRODUCT(a1, b2)
    TS_PRODUCT(a1, b1)
#undef TS_PRODUCT
    if (kc + 1 < chunks) {
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<float4*>(a_dst[buf ^ 1][q]) = a_next[q];
    }
    __syncthreads();
  }
  // ---- store: register r of tile (mi, ni) is row 32 mi + 8 (r / 4) + 4 h + r % 4, column 32 ni + n
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      if (!col_ok[ni]) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * mi + 8 * (r / 4) + 4 * h + (r % 4);
        if (row < p.M) Cb[(long long)row * p.N + n0 + 32 * ni + n] = acc[mi][ni][r];
      }
    }
}

static bool ts_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int regnet_conv1x1_split_supported(int64_t Co, int64_t Ci, int64_t L) {
  return Co > 0 && Ci > 0 && L > 0 && Co % 16 == 0 && Ci % 16 == 0 && Co <= 4096 && Ci <= 4096;
}

// bytes of the plane workspace one call needs (three bf16 planes of the weight, rows padded to 128)
extern "C" int64_t regnet_conv1x1_split_workspace_bytes(int64_t Co, int64_t Ci, int64_t transposed) {
  const int64_t M = transposed ? Ci : Co, K = transposed ? Co : Ci;
  return 3 * ((M + TS_BM - 1) / TS_BM * TS_BM) * K * 2;
}

// transposed == 0: Y (B, Co, L) = W (Co, Ci) . [affine](X (B, Ci, L));  1: dX (B, Ci, L) = W^T . dY (B, Co, L).
// bscale / bshift (per INPUT channel of the product, i.e. Ci entries; forward only) may be NULL: no affine.  workspace: see above.
extern "C" int regnet_conv1x1_split_f32(int transposed, const float* W, const float* in, float* out, int64_t B, int64_t Co, int64_t Ci,
                                        int64_t L, const float* bscale, const float* bshift, int brelu, void* workspace, void* stream) {
  if (B <= 0 || !regnet_conv1x1_split_supported(Co, Ci, L)) return REGNET_ERR_SHAPE;
  if (!W || !in || !out || !workspace) return REGNET_ERR_NULL;
  if (!ts_aligned16(workspace)) return REGNET_ERR_SHAPE;
  const bool affine = bscale != nullptr;
  if (affine && (transposed || !bshift || Ci > TS_AFF_MAXK)) return REGNET_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  TsArgs a = {};
  a.M = (int)(transposed ? Ci : Co); a.K = (int)(transposed ? Co : Ci);
  a.Mpad = (a.M + TS_BM - 1) / TS_BM * TS_BM;
  a.Ap = reinterpret_cast<const __bf16*>(workspace);
  a.B = in; a.C = out; a.N = L;
  a.b_batch = (long long)a.K * L; a.c_batch = (long long)a.M * L;
  a.bscale = bscale; a.bshift = bshift; a.brelu = brelu;
  a.tiles_m = a.Mpad / TS_BM;
  const long long elems = (long long)a.Mpad * a.K;
  hipLaunchKernelGGL(ts_planes_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, W, (int)Co, (int)Ci, transposed,
                     reinterpret_cast<__bf16*>(workspace), a.M, a.Mpad, a.K);
  const long long tiles_n = (L + TS_BN - 1) / TS_BN;
  const dim3 grid((unsigned)(tiles_n * a.tiles_m), (unsigned)B);
  if (affine) hipLaunchKernelGGL(tsplit_kernel<true>, grid, dim3(TS_THREADS), 0, st, a);
  else hipLaunchKernelGGL(tsplit_kernel<false>, grid, dim3(TS_THREADS), 0, st, a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
