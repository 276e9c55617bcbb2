// heads.hip -- the grasp heads of the region stage as ONE launch each (inference).
//
// Reference behaviour restated (paths relative to /root/reference/multi_model/utils): PointNet2TwoStage.forward after its
// max-pool (pointnet2.py:174-188) is conv(256 -> 1024) + BN + ReLU, then a class branch 1024 -> 256 -> 128 -> k_cls and a
// regression branch 1024 -> 256 -> 128 -> k_reg (BN everywhere, ReLU except the last layers); PointNet2Refine.forward
// (pointnet2.py:240-253) is conv(384 -> 1024) + BN + ReLU, then 1024 -> 128 -> k_cls and 1024 -> 128 -> k_reg.  The "points"
// are the B * 64 centres (or the valid crops): at most a few hundred rows, seven / five tiny layers.  Layer by layer that was
// 14 / 10 launches (a skinny split-K GEMM + its reduction each) of a few microseconds of work, on a host-paced stream.
//
// Here a workgroup takes 16 rows through ALL layers of the tree: activations live in LDS (input, trunk, two branch
// buffers: 16 x (384 + 1024 + 256 + 128) floats), weights stream from L2 / HBM straight into MFMA operands
// (v_mfma_f32_16x16x4_f32: exact fp32 products; a lane's float4 of K feeds four MFMAs), folded BatchNorm + ReLU in the
// epilogue.  16 waves split a layer's 16-column blocks.  Eval-mode folded affine: y = relu(scale[n] * acc + shift[n]).
#include "common.h"

typedef float hd_f32x4 __attribute__((ext_vector_type(4)));

#define HD_ROWS 16
#define HD_THREADS 1024   // 16 waves: a workgroup streams ~3.4 MB of weights through ONE CU -- what counts is loads in flight
#define HD_MAX_LAYERS 8
#define HD_PAD 4   // row padding (floats) of the LDS activation buffers: 16 rows x (K + 4) -> rows start 4 banks apart

struct HdLayer {
  const float* W;          // packed [>= ceil16(N)][Kpad] row-major, zero padded (fused._pack)
  const float* scale;      // [N]
  const float* shift;      // [N]
  int K, Kpad, N, relu;
  int src, dst;            // buffer ids: 0 input, 1 trunk, 2 / 3 branch buffers; dst 4 / 5 = global outputs a / b
};

struct HdArgs {
  const float* x; long long ldx; int Kx;      // input rows (n, Kx), row stride ldx
  float* out_a; int lda;                     // dst 4: (n, lda)
  float* out_b; int ldb;                     // dst 5: (n, ldb)
  int n, layers;
  int width[4];                              // row width (floats, multiple of 16) of buffers 0..3
  HdLayer layer[HD_MAX_LAYERS];
};

__global__ __launch_bounds__(HD_THREADS) void heads_chain_kernel(const HdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* buf[4];
  {
    int off = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) { buf[b] = lds + off; off += HD_ROWS * (p.width[b] + HD_PAD); }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * HD_ROWS;
  // ---- input rows -> buffer 0 (rows past n: zeros, never stored)
  {
    const int w = p.width[0], ld = w + HD_PAD;
    for (int i = tid; i < HD_ROWS * (w / 4); i += HD_THREADS) {
      const int r = i / (w / 4), c4 = i % (w / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < p.n && 4 * c4 < p.Kx) {
        const float* src = p.x + (long long)(row0 + r) * p.ldx + 4 * c4;
        if (4 * c4 + 3 < p.Kx) v = *reinterpret_cast<const float4*>(src);
        else { v.x = src[0]; if (4 * c4 + 1 < p.Kx) v.y = src[1]; if (4 * c4 + 2 < p.Kx) v.z = src[2]; }
      }
      *reinterpret_cast<float4*>(&buf[0][r * ld + 4 * c4]) = v;
    }
  }
  __syncthreads();
  const int ar = lane & 15, ag = lane >> 4;          // A: row ar, K group ag;  B: column ar, K group ag;  D: rows 4 ag .. + 3, column ar
  for (int li = 0; li < p.layers; ++li) {
    const HdLayer L = p.layer[li];
    const float* act = buf[L.src];
    const int lds_ld = p.width[L.src] + HD_PAD;
    const int blocks = (L.N + 15) / 16;
    for (int cb = wave; cb < blocks; cb += HD_THREADS / 64) {
      const int col = cb * 16 + ar;                                    // (rows of W beyond N are zero padding: safe to read)
      const float* wrow = L.W + (long long)col * L.Kpad + 4 * ag;
      const float* arow = act + ar * lds_ld + 4 * ag;
      hd_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // K in steps of 16: lane group g supplies k = k0 + 4 g + j to MFMA j (both operands agree, so the order is free).
      // Eight steps' weight loads are issued before the first MFMA (8 KiB in flight per wave, 128 KiB per workgroup).
      int k0 = 0;
      for (; k0 + 128 <= L.Kpad; k0 += 128) {
        float4 b[8], a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(wrow + k0 + 16 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4*>(arow + k0 + 16 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
        }
      }
      for (; k0 < L.Kpad; k0 += 16) {
        const float4 a = *reinterpret_cast<const float4*>(arow + k0);
        const float4 b = *reinterpret_cast<const float4*>(wrow + k0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
      }
      if (col < L.N) {
        const float s = L.scale[col], t = L.shift[col];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float y = acc[j] * s + t;
          if (L.relu) y = fmaxf(y, 0.f);
          const int r = 4 * ag + j;
          if (L.dst < 4) buf[L.dst][r * (p.width[L.dst] + HD_PAD) + col] = y;
          else if (row0 + r < p.n) {
            if (L.dst == 4) p.out_a[(long long)(row0 + r) * p.lda + col] = y;
            else p.out_b[(long long)(row0 + r) * p.ldb + col] = y;
          }
        }
      } else if (L.dst < 4 && col < p.width[L.dst]) {
        // padding columns of an LDS destination (its width is N rounded up to 16): zeros, the next layer reads them as K padding
#pragma unroll
        for (int j = 0; j < 4; ++j) buf[L.dst][(4 * ag + j) * (p.width[L.dst] + HD_PAD) + col] = 0.f;
      }
    }
    __syncthreads();
  }
}

static bool hd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// descr: `layers` records of 9 int64 each: [W, scale, shift (device addresses), K, Kpad, N, relu, src, dst].
extern "C" int regnet_heads_chain_f32(const float* x, int64_t ldx, int64_t Kx, int64_t n, const int64_t* descr, int64_t layers,
                                      float* out_a, int64_t lda, float* out_b, int64_t ldb, void* stream) {
  if (n < 0 || layers <= 0 || layers > HD_MAX_LAYERS || Kx <= 0 || ldx < Kx || (ldx & 3)) return REGNET_ERR_SHAPE;
  if (n == 0) return REGNET_OK;
  if (!x || !descr || !out_a) return REGNET_ERR_NULL;
  if (!hd_aligned16(x)) return REGNET_ERR_SHAPE;
  HdArgs a = {};
  a.x = x; a.ldx = ldx; a.Kx = (int)Kx; a.out_a = out_a; a.lda = (int)lda; a.out_b = out_b; a.ldb = (int)ldb;
  a.n = (int)n; a.layers = (int)layers;
  int width[4] = {(int)((Kx + 15) / 16 * 16), 0, 0, 0};
  for (int i = 0; i < layers; ++i) {
    const int64_t* d = descr + 9 * i;
    HdLayer& L = a.layer[i];
    L.W = reinterpret_cast<const float*>(d[0]);
    L.scale = reinterpret_cast<const float*>(d[1]);
    L.shift = reinterpret_cast<const float*>(d[2]);
    L.K = (int)d[3]; L.Kpad = (int)d[4]; L.N = (int)d[5]; L.relu = (int)d[6]; L.src = (int)d[7]; L.dst = (int)d[8];
    if (!L.W || !L.scale || !L.shift || !hd_aligned16(L.W)) return REGNET_ERR_NULL;
    if (L.K <= 0 || L.N <= 0 || L.Kpad % 16 || L.Kpad < L.K || L.src < 0 || L.src > 3 || L.dst < 1 || L.dst > 5 || L.dst == L.src)
      return REGNET_ERR_SHAPE;
    if (L.dst == 5 && !out_b) return REGNET_ERR_NULL;
    if ((L.dst == 4 && lda < L.N) || (L.dst == 5 && ldb < L.N)) return REGNET_ERR_SHAPE;
    if (L.dst < 4) width[L.dst] = width[L.dst] > (L.N + 15) / 16 * 16 ? width[L.dst] : (L.N + 15) / 16 * 16;
  }
  // a layer reads Kpad columns of its source: the buffer must be that wide (and hold zeros beyond what its producer wrote)
  size_t floats = 0;
  for (int i = 0; i < layers; ++i) {
    const HdLayer& L = a.layer[i];
    if (width[L.src] != L.Kpad) return REGNET_ERR_SHAPE;     // input width / producer's N rounded to 16 == consumer's Kpad
  }
  for (int b = 0; b < 4; ++b) { a.width[b] = width[b]; floats += (size_t)HD_ROWS * (width[b] + HD_PAD); }
  const size_t bytes = floats * sizeof(float);
  if (bytes > 160 * 1024) return REGNET_ERR_UNSUPPORTED;
  // more than 64 KiB of dynamic LDS is an opt-in, per device (a bit per device ordinal: idempotent, so a race only repeats it)
  static unsigned long long opted_in = 0ull;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  if (dev < 0 || dev >= 64 || !((opted_in >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64) opted_in |= 1ull << dev;
  }
  hipLaunchKernelGGL(heads_chain_kernel, dim3((unsigned)((n + HD_ROWS - 1) / HD_ROWS)), dim3(HD_THREADS), bytes, as_stream(stream), a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
