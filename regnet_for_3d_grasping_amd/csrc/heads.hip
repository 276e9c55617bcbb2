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


// ---------------------------------------------------------------------------------------------------------------------------
// The same trees for MANY rows (B * 64 = 512 centres at 8 scenes, 4000 at test.py:68's scale) -- heads_tree_kernel.
//
// heads_chain_kernel streams every weight through a CU once per 16 rows: 2 MFMAs per 16 bytes of weights, load-bound, and its
// LDS cannot hold more rows (the 1024-wide trunk activation alone is 64 KiB per 16 rows).  Here a workgroup takes 32 rows and
// the trunk activation is never whole: it is produced 256 columns at a time (16 waves x one 16-column block x two row blocks)
// into a 32 x 256 LDS chunk and consumed at once by BOTH branches' first layers, whose 32 x N2 outputs (N2 = 512 / 256: the
// two branches side by side, weights concatenated by the host) accumulate in registers across the four chunks (a wave owns N2 /
// 256 column blocks x two row blocks).  Every weight fragment now feeds two row blocks: 4 MFMAs per 16 bytes, the matrix pipe
// bounds a workgroup (~0.85 M MACs x 32 rows), and 512 rows are 16 workgroups, 4000 rows 125.  Operand mapping and the order
// of the K sum are heads_chain_kernel's, so both kernels give the same bits.
#define HT_ROWS 32
#define HT_CHUNK 256
#define HT_MAX_TAILS 4

struct HtTail {
  const float* W; const float* scale; const float* shift;
  int K, Kpad, N, relu;
  int src, src_off;        // 0: the stage-2 buffer, 1: the tail buffer; first column read
  int dst, dst_off;        // 1: the tail buffer (column offset), 4 / 5: out_a / out_b
};

struct HtArgs {
  const float* x; long long ldx; int Kx, K0pad;
  float* out_a; int lda; float* out_b; int ldb;
  int n, tails;
  const float* Wt; const float* st; const float* tt; int Nt, relu_t;            // trunk: K0pad -> Nt
  const float* W2; const float* s2; const float* t2; int N2, relu_2;            // both branches' first layers: Nt -> N2
  int w3;                                                                      // width of the tail buffer
  HtTail tail[HT_MAX_TAILS];
};

#define HT_MFMA4(acc, a, b)                                             \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).x, (b).x, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).y, (b).y, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).z, (b).z, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a).w, (b).w, acc, 0, 0, 0)

template <int NB2>   // 16-column blocks of stage 2 per wave: N2 = 256 * NB2
__global__ __launch_bounds__(HD_THREADS) void heads_tree_kernel(const HtArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ldx_ = p.K0pad + HD_PAD, ldc = HT_CHUNK + HD_PAD, ld2 = p.N2 + HD_PAD, ld3 = p.w3 + HD_PAD;
  float* xin = lds;                                  // [32][K0pad + 4]
  float* chunk = lds + HT_ROWS * ldx_;               // [32][256 + 4]
  float* sbuf = lds;                                 // [32][N2 + 4]: over xin + chunk once the last chunk has been consumed
  const int regionA = HT_ROWS * (ldx_ + ldc) > HT_ROWS * ld2 ? HT_ROWS * (ldx_ + ldc) : HT_ROWS * ld2;
  float* tbuf = lds + regionA;                       // [32][w3 + 4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ar = lane & 15, ag = lane >> 4;
  const int row0 = blockIdx.x * HT_ROWS;
  {
    const int w4 = p.K0pad / 4;
    for (int i = tid; i < HT_ROWS * w4; i += HD_THREADS) {
      const int r = i / w4, c4 = i % w4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < p.n && 4 * c4 < p.Kx) {
        const float* src = p.x + (long long)(row0 + r) * p.ldx + 4 * c4;
        if (4 * c4 + 3 < p.Kx) v = *reinterpret_cast<const float4*>(src);
        else { v.x = src[0]; if (4 * c4 + 1 < p.Kx) v.y = src[1]; if (4 * c4 + 2 < p.Kx) v.z = src[2]; }
      }
      *reinterpret_cast<float4*>(&xin[r * ldx_ + 4 * c4]) = v;
    }
  }
  __syncthreads();
  hd_f32x4 acc2[NB2][2];
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb) { acc2[nb][0] = hd_f32x4{0.f, 0.f, 0.f, 0.f}; acc2[nb][1] = acc2[nb][0]; }
  for (int c0 = 0; c0 < p.Nt; c0 += HT_CHUNK) {
    // ---- trunk columns c0 + 16 wave .. + 15, both row blocks
    {
      const int col = c0 + 16 * wave + ar;
      const float* wrow = p.Wt + (long long)col * p.K0pad + 4 * ag;
      const float* a0 = xin + ar * ldx_ + 4 * ag;
      const float* a1 = a0 + 16 * ldx_;
      hd_f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
      int k0 = 0;
      for (; k0 + 128 <= p.K0pad; k0 += 128) {
        float4 b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(wrow + k0 + 16 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 x0 = *reinterpret_cast<const float4*>(a0 + k0 + 16 * u);
          const float4 x1 = *reinterpret_cast<const float4*>(a1 + k0 + 16 * u);
          HT_MFMA4(t0, x0, b[u]);
          HT_MFMA4(t1, x1, b[u]);
        }
      }
      for (; k0 < p.K0pad; k0 += 16) {
        const float4 b = *reinterpret_cast<const float4*>(wrow + k0);
        const float4 x0 = *reinterpret_cast<const float4*>(a0 + k0);
        const float4 x1 = *reinterpret_cast<const float4*>(a1 + k0);
        HT_MFMA4(t0, x0, b);
        HT_MFMA4(t1, x1, b);
      }
      const float s = p.st[col], t = p.tt[col];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float y0 = t0[j] * s + t, y1 = t1[j] * s + t;
        if (p.relu_t) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
        chunk[(4 * ag + j) * ldc + 16 * wave + ar] = y0;
        chunk[(16 + 4 * ag + j) * ldc + 16 * wave + ar] = y1;
      }
    }
    __syncthreads();
    // ---- both branches' first layers: k = c0 .. c0 + 255 of their K = Nt sum
    {
      const float* a0 = chunk + ar * ldc + 4 * ag;
      const float* a1 = a0 + 16 * ldc;
#pragma unroll
      for (int kk = 0; kk < HT_CHUNK; kk += 64) {
        float4 b[NB2][4];
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
          const float* wrow = p.W2 + (long long)((wave * NB2 + nb) * 16 + ar) * p.Nt + c0 + kk + 4 * ag;
#pragma unroll
          for (int u = 0; u < 4; ++u) b[nb][u] = *reinterpret_cast<const float4*>(wrow + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 x0 = *reinterpret_cast<const float4*>(a0 + kk + 16 * u);
          const float4 x1 = *reinterpret_cast<const float4*>(a1 + kk + 16 * u);
#pragma unroll
          for (int nb = 0; nb < NB2; ++nb) {
            HT_MFMA4(acc2[nb][0], x0, b[nb][u]);
            HT_MFMA4(acc2[nb][1], x1, b[nb][u]);
          }
        }
      }
    }
    __syncthreads();      // the chunk (and, after the last one, xin) may be overwritten
  }
  // ---- stage-2 epilogue -> sbuf (over xin / chunk)
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb) {
    const int col = (wave * NB2 + nb) * 16 + ar;
    const float s = p.s2[col], t = p.t2[col];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float y0 = acc2[nb][0][j] * s + t, y1 = acc2[nb][1][j] * s + t;
      if (p.relu_2) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
      sbuf[(4 * ag + j) * ld2 + col] = y0;
      sbuf[(16 + 4 * ag + j) * ld2 + col] = y1;
    }
  }
  __syncthreads();
  // ---- tails (a few hundred kMACs per row block): 16-column blocks over the waves, both row blocks per weight fragment
  for (int li = 0; li < p.tails; ++li) {
    const HtTail L = p.tail[li];
    const float* act = (L.src == 0 ? sbuf : tbuf) + L.src_off;
    const int lds_ld = L.src == 0 ? ld2 : ld3;
    const int blocks = (L.N + 15) / 16;
    for (int cb = wave; cb < blocks; cb += HD_THREADS / 64) {
      const int col = cb * 16 + ar;                      // (rows of W beyond N are zero padding: safe to read)
      const float* wrow = L.W + (long long)col * L.Kpad + 4 * ag;
      const float* a0 = act + ar * lds_ld + 4 * ag;
      const float* a1 = a0 + 16 * lds_ld;
      hd_f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
      int k0 = 0;
      for (; k0 + 128 <= L.Kpad; k0 += 128) {
        float4 b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(wrow + k0 + 16 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 x0 = *reinterpret_cast<const float4*>(a0 + k0 + 16 * u);
          const float4 x1 = *reinterpret_cast<const float4*>(a1 + k0 + 16 * u);
          HT_MFMA4(t0, x0, b[u]);
          HT_MFMA4(t1, x1, b[u]);
        }
      }
      for (; k0 < L.Kpad; k0 += 16) {
        const float4 b = *reinterpret_cast<const float4*>(wrow + k0);
        const float4 x0 = *reinterpret_cast<const float4*>(a0 + k0);
        const float4 x1 = *reinterpret_cast<const float4*>(a1 + k0);
        HT_MFMA4(t0, x0, b);
        HT_MFMA4(t1, x1, b);
      }
      if (col < L.N) {
        const float s = L.scale[col], t = L.shift[col];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float y0 = t0[j] * s + t, y1 = t1[j] * s + t;
          if (L.relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
          const int r0 = 4 * ag + j, r1 = 16 + r0;
          if (L.dst == 1) {
            tbuf[r0 * ld3 + L.dst_off + col] = y0;
            tbuf[r1 * ld3 + L.dst_off + col] = y1;
          } else {
            float* out = L.dst == 4 ? p.out_a : p.out_b;
            const int ldo = L.dst == 4 ? p.lda : p.ldb;
            if (row0 + r0 < p.n) out[(long long)(row0 + r0) * ldo + col] = y0;
            if (row0 + r1 < p.n) out[(long long)(row0 + r1) * ldo + col] = y1;
          }
        }
      } else if (L.dst == 1 && L.dst_off + col < p.w3) {
        // padding columns of the tail buffer (a producer's N rounded up to 16): zeros, the consumer reads them as K padding
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tbuf[(4 * ag + j) * ld3 + L.dst_off + col] = 0.f;
          tbuf[(16 + 4 * ag + j) * ld3 + L.dst_off + col] = 0.f;
        }
      }
    }
    __syncthreads();
  }
}

// descr: (2 + tails) records of 11 int64 [W, scale, shift (device addresses), K, Kpad, N, relu, src, src_off, dst, dst_off]:
// record 0 the trunk (x -> Nt columns; src / dst fields unused), record 1 the branches' first layers side by side (Nt -> N2),
// then the tails in execution order.
extern "C" int regnet_heads_tree_f32(const float* x, int64_t ldx, int64_t Kx, int64_t n, const int64_t* descr, int64_t tails,
                                     float* out_a, int64_t lda, float* out_b, int64_t ldb, void* stream) {
  if (n < 0 || tails < 0 || tails > HT_MAX_TAILS || Kx <= 0 || ldx < Kx || (ldx & 3)) return REGNET_ERR_SHAPE;
  if (n == 0) return REGNET_OK;
  if (!x || !descr || !out_a) return REGNET_ERR_NULL;
  if (!hd_aligned16(x)) return REGNET_ERR_SHAPE;
  HtArgs a = {};
  a.x = x; a.ldx = ldx; a.Kx = (int)Kx; a.K0pad = (int)((Kx + 15) / 16 * 16);
  a.out_a = out_a; a.lda = (int)lda; a.out_b = out_b; a.ldb = (int)ldb; a.n = (int)n; a.tails = (int)tails;
  const int64_t* d = descr;
  a.Wt = reinterpret_cast<const float*>(d[0]); a.st = reinterpret_cast<const float*>(d[1]); a.tt = reinterpret_cast<const float*>(d[2]);
  a.Nt = (int)d[5]; a.relu_t = (int)d[6];
  if (!a.Wt || !a.st || !a.tt || !hd_aligned16(a.Wt)) return REGNET_ERR_NULL;
  if (d[3] != Kx || d[4] != a.K0pad || a.Nt <= 0 || a.Nt % HT_CHUNK) return REGNET_ERR_SHAPE;
  d = descr + 11;
  a.W2 = reinterpret_cast<const float*>(d[0]); a.s2 = reinterpret_cast<const float*>(d[1]); a.t2 = reinterpret_cast<const float*>(d[2]);
  a.N2 = (int)d[5]; a.relu_2 = (int)d[6];
  if (!a.W2 || !a.s2 || !a.t2 || !hd_aligned16(a.W2)) return REGNET_ERR_NULL;
  if (d[3] != a.Nt || d[4] != a.Nt || (a.N2 != 256 && a.N2 != 512)) return REGNET_ERR_SHAPE;
  int w3 = 0;
  for (int i = 0; i < tails; ++i) {
    d = descr + 11 * (2 + i);
    HtTail& L = a.tail[i];
    L.W = reinterpret_cast<const float*>(d[0]); L.scale = reinterpret_cast<const float*>(d[1]); L.shift = reinterpret_cast<const float*>(d[2]);
    L.K = (int)d[3]; L.Kpad = (int)d[4]; L.N = (int)d[5]; L.relu = (int)d[6];
    L.src = (int)d[7]; L.src_off = (int)d[8]; L.dst = (int)d[9]; L.dst_off = (int)d[10];
    if (!L.W || !L.scale || !L.shift || !hd_aligned16(L.W)) return REGNET_ERR_NULL;
    if (L.K <= 0 || L.N <= 0 || L.Kpad % 16 || L.Kpad < L.K || (L.src != 0 && L.src != 1) || L.src_off < 0 || (L.src_off & 3) ||
        (L.dst != 1 && L.dst != 4 && L.dst != 5) || L.dst_off < 0 || (L.dst == 1 && L.src == 1))
      return REGNET_ERR_SHAPE;
    if (L.dst == 5 && !out_b) return REGNET_ERR_NULL;
    if ((L.dst == 4 && lda < L.N) || (L.dst == 5 && ldb < L.N) || (L.dst != 1 && L.dst_off != 0)) return REGNET_ERR_SHAPE;
    if (L.dst == 1) { const int end = L.dst_off + (L.N + 15) / 16 * 16; w3 = end > w3 ? end : w3; }
  }
  for (int i = 0; i < tails; ++i) {       // a tail reads Kpad columns of its source from src_off on: they must exist
    const HtTail& L = a.tail[i];
    if (L.src_off + L.Kpad > (L.src == 0 ? a.N2 : w3)) return REGNET_ERR_SHAPE;
  }
  a.w3 = w3;
  const size_t ph1 = (size_t)HT_ROWS * (a.K0pad + HD_PAD + HT_CHUNK + HD_PAD), ph2 = (size_t)HT_ROWS * (a.N2 + HD_PAD);
  const size_t bytes = ((ph1 > ph2 ? ph1 : ph2) + (size_t)HT_ROWS * (w3 + HD_PAD)) * sizeof(float);
  if (bytes > 160 * 1024) return REGNET_ERR_UNSUPPORTED;
  static unsigned long long opted_in = 0ull;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  if (dev < 0 || dev >= 64 || !((opted_in >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_tree_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_tree_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64) opted_in |= 1ull << dev;
  }
  const dim3 grid((unsigned)((n + HT_ROWS - 1) / HT_ROWS));
  if (a.N2 == 512) hipLaunchKernelGGL(heads_tree_kernel<2>, grid, dim3(HD_THREADS), bytes, as_stream(stream), a);
  else hipLaunchKernelGGL(heads_tree_kernel<1>, grid, dim3(HD_THREADS), bytes, as_stream(stream), a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
