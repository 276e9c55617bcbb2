// tgemm.hip -- the 1x1 convolutions of the shared-MLP blocks in TRAINING (forward, input gradient, weight gradient) on
// gfx950 matrix cores, in the tensors' own channel-first layout (B, C, L) -- no transposes, no library GEMM.
//
// Reference behaviour restated (paths relative to /root/reference/multi_model/utils): SharedMLP blocks are
// nn.Conv1d / nn.Conv2d(kernel_size=1, bias=False) (pn2_utils/nn/modules/conv.py:20-36, :60-76) under torch autograd
// (train.py:376-384); per scene b with X[b] (Ci x L), W (Co x Ci):
//     forward           Y[b]  = W   . X[b]                       (Co x L)
//     input gradient    dX[b] = W^T . dY[b]                      (Ci x L)
//     weight gradient   dW    = sum_b dY[b] . X[b]^T             (Co x Ci), reduction over all B*L points
//
// One kernel, C[M x N] = A[M x K] . B[K x N] per batch / slice, with each operand in one of two memory layouts:
//   "row"   operand element (r, k) at base + r * ld + k   -- K contiguous  (W in the forward; dY and X in the weight gradient)
//   "kmaj"  operand element (r, k) at base + k * ld + r   -- rows contiguous (X, dY with r = point; W^T in the input gradient)
// Both are fetched by LDS-DMA in 16-byte chunks into a 3-stage ring with counted waits (gemm2.h's scheme).  A "row"
// tile is stored [row][16 k] with the chunk swizzle of gemm2.h and read with ds_read_b128 (4 k-steps per read); a "kmaj"
// tile is stored [16 k][rows] exactly as it lies in memory (a k-row of the tile is one contiguous run: ideal global
// reads) and read with one conflict-free ds_read_b32 per MFMA operand (lanes = consecutive rows = consecutive banks).
// Tile 128 (M) x 256 (N) x 16 (K), 8 waves as 2 x 4, 64 x 64 per wave, v_mfma_f32_32x32x2_f32: M is the channel axis
// (128 .. 1024), N the long axis (points, or Ci in the weight gradient); 2 workgroups per CU.
// The weight gradient cuts the point axis into slices (split-K): slice partial sums go to a workspace and a second
// kernel adds them in slice order (deterministic).
//
// Forward and input gradient (K = a channel count, 128 .. 1024: 8 .. 64 k-tiles per output tile) run as a STREAM
// (tgemm_stream_kernel): two workgroups per CU walk the output tiles with one LDS ring that never drains -- the loads of
// the next tile's first k-tiles are issued during the last two k-tiles of the current one, and its stores leave while
// those loads are in flight.  Measured with one workgroup per tile: time per tile = 20 us + 0.25 us per unit of K on an
// idle chip (workgroup dispatch, address set-up, two exposed load latencies, the store tail) -- 40 % of a K = 128 tile.
#include "common.h"

typedef float tg_f32x16 __attribute__((ext_vector_type(16)));

#define TG_BM 128
#define TG_BK 16
#define TG_STAGES 3
#define TG_THREADS 512
#define TG_AFF_MAXK 512       // B_AFFINE forward: channels whose (scale, shift) fit the workgroup's LDS table (2 workgroups per CU)
#ifndef TG_WGRAD_BLOCKS
#define TG_WGRAD_BLOCKS 256    // weight gradient: (slice x tile x scene) workgroups to aim for -- one per CU: every slice costs a partial matrix and a
                               // shorter K loop; measured over a training iteration (8 x 25 600): 1024 -> 29.7 ms of tgemm, 512 -> 28.6, 256 -> 27.4, 128 -> 33.3
#endif

struct TgArgs {
  const float* A; long long lda, a_batch, a_slice;   // per block: A + batch * a_batch + slice * a_slice
  const float* B; long long ldb, b_batch, b_slice;
  float* C;       long long ldc, c_batch, c_slice;
  int M, N, K;                                        // K = per-slice depth, multiple of 16
  int tiles_m, tiles_n, slices;                       // grid = tiles_m * tiles_n * slices * batches
  int batches;                                        // tgemm_stream_kernel only (its grid does not encode them)
  int* ticket;                                        // tgemm_stream_kernel: next tile to hand out - gridDim.x (zeroed by the caller)
  // B_AFFINE: the B operand is used as max(bscale[c] * x + bshift[c], brelu ? 0 : -inf), c = its channel -- the BatchNorm (+ ReLU)
  // between two convolutions of a shared-MLP block applied on the fragment, so that its output is never written (see the ABI)
  const float* bscale; const float* bshift; int brelu;
  // STATS (tgemm_stream_kernel, forward): per-channel sums of the OUTPUT over all points, sum at stat[2 c], sum of squares at
  // stat[2 c + 1] (fp64, zeroed by the caller) -- the statistics pass of the training BatchNorm that follows the convolution
  double* stat;
};

__device__ __forceinline__ void tg_glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void tg_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(N) : "memory");
}

// TG_BN = 256 (wave tile 64 x 64) or 128 (64 x 32: the weight gradient of layers with <= 128 input channels, whose N
// axis is only that wide).
template <bool A_KMAJ, bool B_KMAJ, int TG_BN, bool B_AFFINE = false>
__global__ __launch_bounds__(TG_THREADS, 4) void tgemm_kernel(const TgArgs p) {
  static_assert(!(B_AFFINE && B_KMAJ), "tgemm_kernel applies the affine to row-layout B operands (the weight gradient)");
  constexpr int TG_STAGE_FLOATS = (TG_BM + TG_BN) * TG_BK;
  constexpr int TNI = TG_BN / 128;                      // 32-column accumulators per wave along N
  constexpr int WN_COLS = TG_BN / 4;                    // columns per wave
  constexpr int NPIECE = (TG_BM + TG_BN) / 16 / 8;      // LDS-DMA pieces per wave per k-tile: 3 or 2
  static_assert(!(B_KMAJ && TG_BN != 256), "k-major B tiles are 256 wide (one piece = one k row)");
  __shared__ __attribute__((aligned(1024))) float smem[TG_STAGES * TG_STAGE_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;            // 2 x 4 waves
  const int fr = lane & 31, fh = lane >> 5;
  // block -> (batch, slice, tile): tiles of one (batch, slice) are consecutive so they share the operand panels in L2
  const int tiles = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int unit = bid / tiles, t = bid - unit * tiles;
  const int batch = unit / p.slices, slice = unit - batch * p.slices;
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * TG_BM, n0 = tn * TG_BN;
  const float* Ab = p.A + batch * p.a_batch + slice * p.a_slice;
  const float* Bb = p.B + batch * p.b_batch + slice * p.b_slice;
  float* Cb = p.C + batch * p.c_batch + slice * p.c_slice;

  // ---- this wave's LDS-DMA pieces: 8 pieces for A, 16 for B, 24 / 8 waves = 3 per wave (piece = 1 KiB)
  const float* gp[NPIECE];       // per-lane source at k = 0
  long long kstep[NPIECE];       // floats to advance per k-tile
  unsigned dst[NPIECE];          // byte offset inside a stage
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) {
    const int piece = wave + 8 * j;                  // wave-uniform; pieces 0..7 = A, 8.. = B (16 or 8 of them)
    const bool isA = piece < 8;
    const int q = isA ? piece : piece - 8;
    dst[j] = (unsigned)((isA ? 0 : TG_BM * TG_BK * 4) + q * 1024);
    if (isA) {
      if (A_KMAJ) {          // piece q = k rows 2q, 2q + 1 of [16 k][128 m]: lane -> (k = 2q + l/32, m chunk l % 32)
        int m = m0 + 4 * (lane & 31);
        if (m >= p.M) m = 0;                          // past the edge: any valid chunk (never stored)
        gp[j] = Ab + (long long)(2 * q + (lane >> 5)) * p.lda + m;
        kstep[j] = (long long)TG_BK * p.lda;
      } else {               // piece q = rows 16q .. 16q + 15 of [128 rows][16 k], chunk swizzle (gemm2.h)
        int r = m0 + 16 * q + (lane >> 2);
        if (r >= p.M) r = 0;
        gp[j] = Ab + (long long)r * p.lda + 4 * ((lane & 3) ^ ((lane >> 4) & 3));
        kstep[j] = TG_BK;
      }
    } else {
      if (B_KMAJ) {          // piece q = k row q of [16 k][256 n]: lane -> n chunk l
        int n = n0 + 4 * lane;
        if (n >= p.N) n = 0;
        gp[j] = Bb + (long long)q * p.ldb + n;
        kstep[j] = (long long)TG_BK * p.ldb;
      } else {               // piece q = rows 16q .. of [256 rows][16 k]
        int r = n0 + 16 * q + (lane >> 2);
        if (r >= p.N) r = 0;
        gp[j] = Bb + (long long)r * p.ldb + 4 * ((lane & 3) ^ ((lane >> 4) & 3));
        kstep[j] = TG_BK;
      }
    }
  }
  // B_AFFINE: a lane's B fragments are rows n0 + wn * WN_COLS + 32 ni + fr of the operand: one (scale, shift) per ni
  float bsc[TNI], bsh[TNI];
  const float blo = (B_AFFINE && !p.brelu) ? -INFINITY : 0.f;
  if (B_AFFINE) {
#pragma unroll
    for (int ni = 0; ni < TNI; ++ni) {
      int r = n0 + wn * WN_COLS + 32 * ni + fr;
      if (r >= p.N) r = 0;
      bsc[ni] = p.bscale[r]; bsh[ni] = p.bshift[r];
    }
    // the values must have arrived HERE: left to the compiler, its wait for them (vmcnt(0)) lands at their first use inside the
    // k loop and drains the ring's loads on every k-tile
#pragma unroll
    for (int ni = 0; ni < TNI; ++ni) asm volatile("" : "+v"(bsc[ni]), "+v"(bsh[ni]));
  }
  const unsigned smem_base = (unsigned)(uintptr_t)smem;
#define TG_ISSUE(KT_, STAGE_)                                                                         \
  _Pragma("unroll") for (int j = 0; j < NPIECE; ++j)                                                   \
    tg_glds16(gp[j] + (long long)(KT_) * kstep[j], smem_base + (unsigned)((STAGE_) * TG_STAGE_FLOATS * 4) + dst[j]);

  tg_f32x16 acc[2][TNI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < TNI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // fragment addressing (floats inside a stage)
  const int sw = (fr >> 2) & 3;
  int offA[2], offB[2];      // "row" layout: [kk] -> float4 of 4 k-steps
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    offA[kk] = (wm * 64 + fr) * TG_BK + 4 * ((2 * kk + fh) ^ sw);
    offB[kk] = TG_BM * TG_BK + (wn * WN_COLS + fr) * TG_BK + 4 * ((2 * kk + fh) ^ sw);
  }
  const int kmA = 4 * fh * TG_BM + wm * 64 + fr;                     // "kmaj": + (8 kk + t) * TG_BM + 32 mi
  const int kmB = TG_BM * TG_BK + 4 * fh * TG_BN + wn * WN_COLS + fr;     //         + (8 kk + t) * TG_BN + 32 ni

#define TG_COMPUTE(STAGE_)                                                                                         \
  {                                                                                                                \
    const float* st = smem + (STAGE_) * TG_STAGE_FLOATS;                                                           \
    float a[2][2][4], b[2][TNI][4];   /* all fragments of the k-tile first, order pinned (see gemm2.h) */           \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                             \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                                           \
        if (A_KMAJ) {                                                                                              \
          _Pragma("unroll") for (int t = 0; t < 4; ++t) a[kk][mi][t] = st[kmA + (8 * kk + t) * TG_BM + 32 * mi];   \
        } else {                                                                                                   \
          const float4 v = *reinterpret_cast<const float4*>(st + offA[kk] + mi * 32 * TG_BK);                      \
          a[kk][mi][0] = v.x; a[kk][mi][1] = v.y; a[kk][mi][2] = v.z; a[kk][mi][3] = v.w;                          \
        }                                                                                                          \
      }                                                                                                            \
      _Pragma("unroll") for (int ni = 0; ni < TNI; ++ni) {                                                         \
        if (B_KMAJ) {                                                                                              \
          _Pragma("unroll") for (int t = 0; t < 4; ++t) b[kk][ni][t] = st[kmB + (8 * kk + t) * TG_BN + 32 * ni];   \
        } else {                                                                                                   \
          const float4 v = *reinterpret_cast<const float4*>(st + offB[kk] + ni * 32 * TG_BK);                      \
          b[kk][ni][0] = v.x; b[kk][ni][1] = v.y; b[kk][ni][2] = v.z; b[kk][ni][3] = v.w;                          \
          if (B_AFFINE) {                                                                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                          \
              b[kk][ni][t] = fmaxf(fmaf(b[kk][ni][t], bsc[ni], bsh[ni]), blo);                                     \
          }                                                                                                        \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                               \
      _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                           \
          _Pragma("unroll") for (int ni = 0; ni < TNI; ++ni)                                                       \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][mi][t], b[kk][ni][t], acc[mi][ni], 0, 0, 0);  \
  }

  const int KT = p.K / TG_BK;           // >= 2 (launcher)
  TG_ISSUE(0, 0)
  TG_ISSUE(1, 1)
  int stage = 0;
  for (int kt = 0; kt < KT - 2; ++kt) {
    tg_wait_barrier<NPIECE>();
    int ns = stage + 2; if (ns >= TG_STAGES) ns -= TG_STAGES;
    TG_ISSUE(kt + 2, ns)
    TG_COMPUTE(stage)
    if (++stage == TG_STAGES) stage = 0;
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    tg_wait_barrier<0>();
    TG_COMPUTE(stage)
    if (++stage == TG_STAGES) stage = 0;
  }
#undef TG_ISSUE
#undef TG_COMPUTE

  // ---- store: element r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
#pragma unroll
  for (int ni = 0; ni < TNI; ++ni) {
    const int col = n0 + wn * WN_COLS + ni * 32 + fr;
    const bool col_ok = col < p.N;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int row0 = m0 + wm * 64 + mi * 32 + 4 * fh;
      float* cp = Cb + (long long)row0 * p.ldc + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        if (col_ok && row < p.M) *cp = acc[mi][ni][r];
        cp += ((r & 3) == 3) ? 5 * p.ldc : p.ldc;
      }
    }
  }
}

// the same with a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: half the address registers
__device__ __forceinline__ void tg_glds16_s(const float* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void tg_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }
// *p += 1, old value -> the returned register ONCE A LATER COUNTED WAIT HAS PASSED IT (no wait here: the compiler's own
// vmcnt(0) in front of an atomic's result would also drain the k-tile loads in flight)
__device__ __forceinline__ int tg_ticket_nowait(int* p) {
  int r;
  const int one = 1;
  asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(r) : "v"(p), "v"(one) : "memory");
  return r;
}

// the value lane (i ^ MASK) holds, MASK = 1, 2, 4, 8 (DPP moves inside the row of 16), 16 (one ds_bpermute)
template <int MASK> __device__ __forceinline__ float tg_xor_lane(float v) {
  const int x = __builtin_bit_cast(int, v);
  if (MASK == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
  if (MASK == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
  if (MASK == 4) {      // banks 0, 2 (lanes with bit 2 clear) read lane i + 4, banks 1, 3 lane i - 4
    const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);                                     // row_shl:4
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false));                 // row_shr:4
  }
  if (MASK == 8) {
    const int t = __builtin_amdgcn_update_dpp(0, x, 0x108, 0xf, 0x3, false);                                     // row_shl:8
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, x, 0x118, 0xf, 0xc, false));                 // row_shr:8
  }
  return __shfl_xor(v, 16);
}
// One step of a HALVING reduction over lanes: the lane pair (i, i ^ MASK) holds two values each and wants their two sums; the lane
// with `side` clear ends up with lo + partner's lo, the other with hi + partner's hi -- one exchange and one add for two sums.
template <int MASK> __device__ __forceinline__ float tg_halve(float lo, float hi, bool side) {
  const float keep = side ? hi : lo, give = side ? lo : hi;
  return keep + tg_xor_lane<MASK>(give);
}
#define TG_STAT_MAXM 256        // STATS: output channels whose fp64 sums fit the workgroup's LDS table beside the ring
#define TG_STAT_AFF_MAXK 256    // ... and input channels of a B_AFFINE operand then

// C[M x N] = A[M x K] . B[K x N] per batch with B k-major (the activations / output gradients, points contiguous) and A the
// weights, row (forward) or k-major (input gradient): tile 128 x 256 x 16 as tgemm_kernel, but persistent -- workgroup w
// starts with tile w and draws further tiles from a ticket counter (channel tile fastest: the tiles sharing an activation
// panel run at the same time), treating their k-tiles as ONE sequence through the 3-stage ring.  Tickets, not a fixed
// stride: in the training iteration the next batch's sampling holds eight CUs for 4 ms at a time; a workgroup that cannot
// start there must not own tiles (measured with a fixed stride: 5 % faster on an idle chip, 3 % slower in the iteration).
//
// STATS (forward): the per-channel sum and sum of squares of the output -- the statistics pass of the BatchNorm that follows -- are
// taken from the accumulators (a lane holds 32 channels of two points: two in-lane adds, a halving reduction over the 32 lanes of a
// lane half that leaves every lane ONE channel's number, one LDS atomic per lane into an fp64 table, one fp64 global atomic per
// channel and workgroup at the end), so that the output is not read again (bn_stats_kernel: 1 read of up to 2.7 GB per layer).
template <bool A_KMAJ, bool B_AFFINE = false, bool STATS = false>
__global__ __launch_bounds__(TG_THREADS, 4) void tgemm_stream_kernel(const TgArgs p) {
  constexpr int TG_BN = 256, TNI = 2, WN_COLS = 64, NPIECE = 3;
  constexpr int TG_STAGE_FLOATS = (TG_BM + TG_BN) * TG_BK;
  __shared__ __attribute__((aligned(1024))) float smem[TG_STAGES * TG_STAGE_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 31, fh = lane >> 5;
  const int tiles = p.tiles_m * p.tiles_n;
  const long long total = (long long)tiles * p.batches;
  const int G = gridDim.x, bid = blockIdx.x;
  if (bid >= total) return;
  __shared__ long long s_next[2];                                 // tiles drawn from the counter, by parity of their number
  __shared__ __attribute__((aligned(16))) float btab[B_AFFINE ? 2 * (STATS ? TG_STAT_AFF_MAXK : TG_AFF_MAXK) : 4];   // B_AFFINE: (scale, shift) per channel of K
  __shared__ double stab[STATS ? 2 * TG_STAT_MAXM : 1];
  if (B_AFFINE) {
    for (int i = tid; i < p.K; i += TG_THREADS) { btab[2 * i] = p.bscale[i]; btab[2 * i + 1] = p.bshift[i]; }
  }
  if (STATS) {
    for (int i = tid; i < 2 * p.M; i += TG_THREADS) stab[i] = 0.0;
  }
  if (B_AFFINE || STATS) __syncthreads();
  const float blo = (B_AFFINE && !p.brelu) ? -INFINITY : 0.f;
  const int KT = p.K / TG_BK;                                     // >= 2 (launcher)

  // ---- LDS-DMA pieces of this wave (as tgemm_kernel: pieces 0..7 = A, 8..23 = B; piece = 1 KiB): tile-independent parts
  unsigned dst[NPIECE];
  unsigned kstep[NPIECE];           // bytes per k-tile (a batch's operand is < 4 GiB: launcher)
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) {
    const int piece = wave + 8 * j;
    const bool isA = piece < 8;
    const int q = isA ? piece : piece - 8;
    dst[j] = (unsigned)((isA ? 0 : TG_BM * TG_BK * 4) + q * 1024);
    kstep[j] = 4u * (unsigned)(isA ? (A_KMAJ ? TG_BK * p.lda : (long long)TG_BK) : TG_BK * p.ldb);
  }
  unsigned gp[NPIECE];              // issue cursor: this lane's byte offset of the cursor's k-tile from its operand's batch base
  const float* baseA = p.A;         // wave-uniform bases of the cursor's batch
  const float* baseB = p.B;
  int issue_kt = 0;                 // cursor = k-tile issue_kt of tile issue_t (-1: no tile left)
  long long issue_t = bid;
  int drawn = 0;                    // tiles drawn so far; a draw is pending while issue_t == -2
  int ticket_reg = 0;               // thread 0: the pending draw's ticket (valid after the next counted wait)
  auto set_issue_tile = [&](long long t) {
    const int batch = (int)(t / tiles), r = (int)(t - (long long)batch * tiles);
    const int tn = r / p.tiles_m, tm = r - tn * p.tiles_m;
    const int m0 = tm * TG_BM, n0 = tn * TG_BN;
    baseA = p.A + batch * p.a_batch;
    baseB = p.B + batch * p.b_batch;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int piece = wave + 8 * j;
      const int q = piece < 8 ? piece : piece - 8;
      if (piece < 8) {
        if (A_KMAJ) {
          int m = m0 + 4 * (lane & 31);
          if (m >= p.M) m = 0;
          gp[j] = 4u * (unsigned)((2 * q + (lane >> 5)) * (int)p.lda + m);
        } else {
          int r2 = m0 + 16 * q + (lane >> 2);
          if (r2 >= p.M) r2 = 0;
          gp[j] = 4u * (unsigned)(r2 * (int)p.lda + 4 * ((lane & 3) ^ ((lane >> 4) & 3)));
        }
      } else {
        int n2 = n0 + 4 * lane;
        if (n2 >= p.N) n2 = 0;
        gp[j] = 4u * ((unsigned)q * (unsigned)p.ldb + (unsigned)n2);
      }
    }
  };
  const unsigned smem_base = (unsigned)(uintptr_t)smem;
  auto issue = [&](int stage_) {     // the cursor's k-tile -> stage_, cursor + 1
    const bool last = issue_kt + 1 == KT;
    // the cursor's tile ends with this group: draw the next tile IN FRONT of the group, so that the counted wait that
    // lets this group be the only one outstanding (next iteration, or the wait before a tile's stores) covers the draw
    if (last && tid == 0) ticket_reg = tg_ticket_nowait(p.ticket);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      tg_glds16_s(wave + 8 * j < 8 ? baseA : baseB, gp[j], smem_base + (unsigned)(stage_ * TG_STAGE_FLOATS * 4) + dst[j]);
      gp[j] += kstep[j];
    }
    if (last) { issue_kt = 0; issue_t = -2; } else ++issue_kt;
  };
  auto publish_drawn = [&]() {       // thread 0, after that counted wait and in front of a barrier
    if (tid == 0) {
      s_next[drawn & 1] = (long long)G + ticket_reg;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  auto take_drawn = [&]() {          // after that barrier
    const volatile int* word = reinterpret_cast<volatile int*>(&s_next[drawn & 1]);
    const long long t = ((long long)__builtin_amdgcn_readfirstlane(word[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(word[0]);
    ++drawn;
    issue_t = t < total ? t : -1;
    if (issue_t >= 0) set_issue_tile(issue_t);
  };
  const int sw = (fr >> 2) & 3;
  int offA[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) offA[kk] = (wm * 64 + fr) * TG_BK + 4 * ((2 * kk + fh) ^ sw);
  const int kmA = 4 * fh * TG_BM + wm * 64 + fr;
  const int kmB = TG_BM * TG_BK + 4 * fh * TG_BN + wn * WN_COLS + fr;

  set_issue_tile(issue_t);
  issue(0);
  issue(1);                          // (KT >= 2: still the first tile; a draw is pending afterwards when KT == 2)
  int stage = 0;
  long long cur_t = bid, next_t = -1;          // tile being computed; the one after it, once known
  while (cur_t >= 0) {
    tg_f32x16 acc[2][TNI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < TNI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const bool first = cur_t == bid;
    for (int kt = 0; kt < KT; ++kt) {
      // this k-tile's loads are complete when at most the NEWER group (if one was issued) is outstanding; right after a
      // tile's stores the wait was made before them (below), so that the stores are not waited for here.
      // Groups are issued two k-tiles ahead: a newer one exists unless the cursor ran out of tiles before reaching it.
      const bool newer = kt + 1 < KT || next_t >= 0;
      if (kt != 0 || first) {
        if (newer) tg_wait<NPIECE>(); else tg_wait<0>();
      }
      if (issue_t == -2) publish_drawn();
      asm volatile("s_barrier" ::: "memory");
      if (issue_t == -2) {
        take_drawn();
        if (next_t < 0) next_t = issue_t;        // (the cursor is at most one tile ahead of the computation)
      }
      if (issue_t >= 0) { int ns = stage + 2; if (ns >= TG_STAGES) ns -= TG_STAGES; issue(ns); }
      {
        const float* st = smem + stage * TG_STAGE_FLOATS;
        float a[2][2][4], b[2][TNI][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            if (A_KMAJ) {
#pragma unroll
              for (int t = 0; t < 4; ++t) a[kk][mi][t] = st[kmA + (8 * kk + t) * TG_BM + 32 * mi];
            } else {
              const float4 v = *reinterpret_cast<const float4*>(st + offA[kk] + mi * 32 * TG_BK);
              a[kk][mi][0] = v.x; a[kk][mi][1] = v.y; a[kk][mi][2] = v.z; a[kk][mi][3] = v.w;
            }
          }
#pragma unroll
          for (int ni = 0; ni < TNI; ++ni) {
#pragma unroll
            for (int t = 0; t < 4; ++t) b[kk][ni][t] = st[kmB + (8 * kk + t) * TG_BN + 32 * ni];
          }
          if (B_AFFINE) {      // fragment element t is channel 16 kt + 8 kk + 4 fh + t of the operand
            int fhb = fh;
            if (STATS) {       // (this instantiation has no register left for the table address: made here, from the hardware's lane number)
              asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshrrev_b32 %0, 5, %0" : "=v"(fhb));
            }
            const float4 c01 = *reinterpret_cast<const float4*>(&btab[2 * (TG_BK * kt + 8 * kk + 4 * fhb)]);
            const float4 c23 = *reinterpret_cast<const float4*>(&btab[2 * (TG_BK * kt + 8 * kk + 4 * fhb) + 4]);
#pragma unroll
            for (int ni = 0; ni < TNI; ++ni) {
              b[kk][ni][0] = fmaxf(fmaf(b[kk][ni][0], c01.x, c01.y), blo);
              b[kk][ni][1] = fmaxf(fmaf(b[kk][ni][1], c01.z, c01.w), blo);
              b[kk][ni][2] = fmaxf(fmaf(b[kk][ni][2], c23.x, c23.y), blo);
              b[kk][ni][3] = fmaxf(fmaf(b[kk][ni][3], c23.z, c23.w), blo);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int ni = 0; ni < TNI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][mi][t], b[kk][ni][t], acc[mi][ni], 0, 0, 0);
      }
      if (++stage == TG_STAGES) stage = 0;
    }
    // ---- the next tile's first k-tile (issued two k-tiles ago) before this tile's stores enter the queue behind it
    if (next_t >= 0) tg_wait<NPIECE>();        // (its second k-tile is the one group that may still be outstanding: KT >= 2)
    {
      const long long t = cur_t;
      const int batch = (int)(t / tiles), r0 = (int)(t - (long long)batch * tiles);
      const int tn = r0 / p.tiles_m, tm = r0 - tn * p.tiles_m;
      const int m0 = tm * TG_BM, n0 = tn * TG_BN;
      float* Cb = p.C + batch * p.c_batch;
      if (STATS) {      // (in FRONT of the stores: behind them the allocator spills 48 registers, some inside the k loop)
        int ln;      // the lane number again, from the hardware: no address of this block is to be kept in a register across the k loop
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int fr_ = ln & 31, fh_ = ln >> 5;
        const float ok0 = n0 + wn * WN_COLS + fr_ < p.N ? 1.f : 0.f, ok1 = n0 + wn * WN_COLS + 32 + fr_ < p.N ? 1.f : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          // A lane holds 16 channels (registers r) x 2 points of this row tile: 32 numbers (sum, sum of squares per channel) to be
          // added over the 32 lanes of its half.  Five halving steps (lane bits 0 .. 4 against: which quantity, register bits
          // 3 .. 0) leave every lane with ONE finished number -- 16 + 8 + 4 + 2 + 1 exchanges instead of 32 x 5, and one LDS
          // atomic per lane instead of 32 from one.
          float w[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float a0 = acc[mi][0][r] * ok0, a1 = acc[mi][1][r] * ok1;
            w[r] = tg_halve<1>(a0 + a1, a0 * a0 + a1 * a1, (ln & 1) != 0);
          }
          float x8[8], x4[4], x2[2];
#pragma unroll
          for (int j = 0; j < 8; ++j) x8[j] = tg_halve<2>(w[j], w[j + 8], (ln & 2) != 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) x4[j] = tg_halve<4>(x8[j], x8[j + 4], (ln & 4) != 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) x2[j] = tg_halve<8>(x4[j], x4[j + 2], (ln & 8) != 0);
          const float total = tg_halve<16>(x2[0], x2[1], (ln & 16) != 0);
          // this lane's number: quantity = lane bit 0, register r = lane bits 1 .. 4 (most significant first)
          const int r = ((ln >> 1) & 1) * 8 + ((ln >> 2) & 1) * 4 + ((ln >> 3) & 1) * 2 + ((ln >> 4) & 1);
          const int row = m0 + wm * 64 + mi * 32 + 4 * fh_ + (r & 3) + 8 * (r >> 2);
          if (row < p.M) unsafeAtomicAdd(&stab[2 * row + (ln & 1)], (double)total);
        }
      }
#pragma unroll
      for (int ni = 0; ni < TNI; ++ni) {
        const int col = n0 + wn * WN_COLS + ni * 32 + fr;
        const bool col_ok = col < p.N;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row0 = m0 + wm * 64 + mi * 32 + 4 * fh;
          float* cp = Cb + (long long)row0 * p.ldc + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            if (col_ok && row < p.M) *cp = acc[mi][ni][r];
            cp += ((r & 3) == 3) ? 5 * p.ldc : p.ldc;
          }
        }
      }
    }
    cur_t = next_t;
    next_t = -1;
  }
  if (STATS) {
    __syncthreads();
    for (int i = tid; i < 2 * p.M; i += TG_THREADS) unsafeAtomicAdd(&p.stat[i], stab[i]);
  }
}

// out[i] = sum_s part[s][i], slices added in index order
__global__ __launch_bounds__(256) void tg_reduce_kernel(const float* __restrict__ part, int S, long long n,
                                                        float* __restrict__ out) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 acc = *reinterpret_cast<const float4*>(part + i);
  for (int s = 1; s < S; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(part + (long long)s * n + i);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(out + i) = acc;
}

static bool tg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool A_KMAJ, bool B_KMAJ, int TG_BN, bool B_AFFINE = false>
static int tg_launch(TgArgs a, long long batches, hipStream_t st) {
  a.tiles_m = (a.M + TG_BM - 1) / TG_BM;
  a.tiles_n = (a.N + TG_BN - 1) / TG_BN;
  const long long blocks = (long long)a.tiles_m * a.tiles_n * a.slices * batches;
  if (blocks <= 0) return REGNET_OK;
  if (blocks >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((tgemm_kernel<A_KMAJ, B_KMAJ, TG_BN, B_AFFINE>), dim3((unsigned)blocks), dim3(TG_THREADS), 0, st, a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Workgroup slots the persistent kernel leaves EMPTY (of its two per CU): its workgroups fill the register file and 147 of the
// 160 KB of LDS of a CU for the whole launch, so a small kernel of another stream can only start where a slot was never taken.
static int tg_reserved_slots = 0;
extern "C" int regnet_conv1x1_stream_reserve_slots(int slots) {
  const int before = tg_reserved_slots;
  if (slots >= 0) tg_reserved_slots = slots;
  return before;
}

template <bool A_KMAJ, bool B_AFFINE = false, bool STATS = false>
static int tg_launch_stream(TgArgs a, long long batches, int32_t* ticket, hipStream_t st) {
  a.tiles_m = (a.M + TG_BM - 1) / TG_BM;
  a.tiles_n = (a.N + 255) / 256;
  a.batches = (int)batches;
  a.ticket = ticket;
  const long long total = (long long)a.tiles_m * a.tiles_n * batches;
  if (total <= 0) return REGNET_OK;
  if (total >= (1ll << 31) - 4096 || batches >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  if ((long long)a.K * a.ldb * 4 >= (1ll << 32) || (long long)(a.M > a.K ? a.M : a.K) * a.lda * 4 >= (1ll << 32))
    return REGNET_ERR_UNSUPPORTED;       // 32-bit byte offsets inside one batch's operands
  static int slots = 0;                  // two workgroups per CU (73.7 KB of LDS each)
  if (slots == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    slots = 2 * cus;
  }
  const long long open_slots = slots - tg_reserved_slots > slots / 2 ? slots - tg_reserved_slots : slots / 2;
  const unsigned grid = (unsigned)(total < open_slots ? total : open_slots);
  hipLaunchKernelGGL((tgemm_stream_kernel<A_KMAJ, B_AFFINE, STATS>), dim3(grid), dim3(TG_THREADS), 0, st, a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Forward of a convolution with a HANDFUL of input channels (the first layer of the level-1 block on its grouped rows: 6 -> 128
// over 8 x 327 680 points, nn/modules/conv.py:60-76): 4 GFLOP against a 1.3 GB output -- a store stream, not a contraction.  A
// thread keeps four consecutive points of every input channel in registers and writes one float4 per output channel (a wave: 1 KB
// runs); the weights are wave-uniform scalar loads.  rocBLAS took 0.64 ms for it (2.1 TB/s).
//
// STATS: the statistics of the BatchNorm that follows, WITHOUT touching the output: y = W x is linear in CI <= 8 inputs, so
//   sum_p y_c = w_c . (sum_p x_p),     sum_p y_c^2 = w_c^T (sum_p x_p x_p^T) w_c
// -- the CI sums and CI (CI + 1) / 2 second moments of the INPUT (27 numbers for CI = 6, from the registers the kernel holds anyway:
// fp32 over a wave's 256 points, fp64 from there on, one partial vector per workgroup, no atomics) give every output channel's
// pair in fp64 (smallci_stats_finalize_kernel).
#define SMALLCI_NG(CI) ((CI) + (CI) * ((CI) + 1) / 2)
template <int CI, bool STATS>
__global__ __launch_bounds__(256) void conv_smallci_kernel(const float* __restrict__ W, const float* __restrict__ X,
                                                          float* __restrict__ Y, int Co, long long L, double* __restrict__ partials) {
  const long long l = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const bool live = l < L;
  if (!STATS && !live) return;
  const float* x = X + (long long)blockIdx.y * CI * L + l;
  float* y = Y + (long long)blockIdx.y * Co * L + l;
  float4 v[CI];
#pragma unroll
  for (int i = 0; i < CI; ++i) v[i] = live ? *reinterpret_cast<const float4*>(x + (long long)i * L) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (STATS) {
    constexpr int NG = SMALLCI_NG(CI);
    __shared__ double red[4][NG];
    float g[NG];
    int n = 0;
#pragma unroll
    for (int i = 0; i < CI; ++i) g[n++] = (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
      for (int j = i; j < CI; ++j) g[n++] = (v[i].x * v[j].x + v[i].y * v[j].y) + (v[i].z * v[j].z + v[i].w * v[j].w);
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      float t = g[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = (double)t;
    }
    __syncthreads();
    if (threadIdx.x < NG)
      partials[((long long)blockIdx.y * gridDim.x + blockIdx.x) * NG + threadIdx.x] =
          (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (!live) return;
  }
  for (int o = 0; o < Co; ++o) {
    const float* w = W + o * CI;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < CI; ++i) {
      const float wi = w[i];
      acc.x = fmaf(wi, v[i].x, acc.x); acc.y = fmaf(wi, v[i].y, acc.y);
      acc.z = fmaf(wi, v[i].z, acc.z); acc.w = fmaf(wi, v[i].w, acc.w);
    }
    *reinterpret_cast<float4*>(y + (long long)o * L) = acc;
  }
}

// partials (nblocks x NG) -> sums[2 c] = w_c . S, sums[2 c + 1] = w_c^T G w_c (fp64), one workgroup
__global__ __launch_bounds__(256) void smallci_stats_finalize_kernel(const double* __restrict__ partials, long long nblocks, int CI,
                                                                     const float* __restrict__ W, int Co, double* __restrict__ sums) {
  __shared__ double tot[SMALLCI_NG(8)];
  __shared__ double red[4];
  const int NG = SMALLCI_NG(CI);
  for (int k = 0; k < NG; ++k) {
    double t = 0.0;
    for (long long b = threadIdx.x; b < nblocks; b += 256) t += partials[b * NG + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) tot[k] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
  for (int c = threadIdx.x; c < Co; c += 256) {
    const float* w = W + c * CI;
    double s = 0.0, q = 0.0;
    int n = CI;
    for (int i = 0; i < CI; ++i) s += (double)w[i] * tot[i];
    for (int i = 0; i < CI; ++i)
      for (int j = i; j < CI; ++j, ++n) q += (i == j ? 1.0 : 2.0) * (double)w[i] * (double)w[j] * tot[n];
    sums[2 * c] = s;
    sums[2 * c + 1] = q;
  }
}

static int tg_fwd_smallci(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci, int64_t L, void* workspace,
                          void* sums, void* stream) {
  if (B < 0 || Co < 1 || Ci < 1 || Ci > 8 || L < 4 || (L % 4) || B >= 65536 || Co >= (1ll << 31)) return REGNET_ERR_SHAPE;
  if (B == 0) return REGNET_OK;
  if (!W || !X || !Y || (sums && !workspace)) return REGNET_ERR_NULL;
  if (!tg_aligned16(X) || !tg_aligned16(Y)) return REGNET_ERR_SHAPE;
  if (sums && ((reinterpret_cast<uintptr_t>(sums) & 7) || (reinterpret_cast<uintptr_t>(workspace) & 7))) return REGNET_ERR_SHAPE;
  const dim3 grid((unsigned)((L / 4 + 255) / 256), (unsigned)B);
  hipStream_t st = as_stream(stream);
  double* part = static_cast<double*>(workspace);
  switch (Ci) {
#define SMALLCI_CASE(n)                                                                                                             \
  case n:                                                                                                                           \
    if (sums) hipLaunchKernelGGL((conv_smallci_kernel<n, true>), grid, dim3(256), 0, st, W, X, Y, (int)Co, (long long)L, part);     \
    else hipLaunchKernelGGL((conv_smallci_kernel<n, false>), grid, dim3(256), 0, st, W, X, Y, (int)Co, (long long)L, part);         \
    break;
    SMALLCI_CASE(1) SMALLCI_CASE(2) SMALLCI_CASE(3) SMALLCI_CASE(4) SMALLCI_CASE(5) SMALLCI_CASE(6) SMALLCI_CASE(7) SMALLCI_CASE(8)
#undef SMALLCI_CASE
  }
  if (sums)
    hipLaunchKernelGGL(smallci_stats_finalize_kernel, dim3(1), dim3(256), 0, st, part, (long long)grid.x * grid.y, (int)Ci, W, (int)Co,
                       static_cast<double*>(sums));
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_conv1x1_fwd_smallci_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci,
                                              int64_t L, void* stream) {
  return tg_fwd_smallci(W, X, Y, B, Co, Ci, L, nullptr, nullptr, stream);
}

// ... leaving sums (2 Co doubles: per output channel the sum and the sum of squares of Y over all B L points, from the INPUT's first
// and second moments) for regnet_bn_train_stats_from_sums_f32 / regnet_bn_relu_train_fwd_from_sums_f32; workspace:
// regnet_conv1x1_smallci_stats_workspace_bytes(B, Ci, L) bytes, 8-byte aligned
extern "C" int64_t regnet_conv1x1_smallci_stats_workspace_bytes(int64_t B, int64_t Ci, int64_t L) {
  if (B <= 0 || Ci < 1 || Ci > 8 || L < 4) return 0;
  return ((L / 4 + 255) / 256) * B * SMALLCI_NG(Ci) * (int64_t)sizeof(double);
}

extern "C" int regnet_conv1x1_fwd_smallci_stats_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci,
                                                    int64_t L, void* workspace, void* sums, void* stream) {
  if (!sums) return REGNET_ERR_NULL;
  return tg_fwd_smallci(W, X, Y, B, Co, Ci, L, workspace, sums, stream);
}

// Weight gradient of the same layers: dW[o][i] = sum_{b,l} dY[b][o][l] X[b][i][l] with CI <= 8 -- a reduction over the 1.3 GB
// gradient, not a contraction.  A wave owns four output channels and one slice of one scene's points: per step a lane loads a
// float4 of each of its four dY rows and of every X row and updates 4 x CI sums; the sums are reduced over the wave and written as
// one partial matrix per (scene, slice) -- added by the caller in a fixed order (deterministic, like the split weight gradient above).
template <int CI>
__global__ __launch_bounds__(256) void wgrad_smallci_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                           float* __restrict__ part, int Co, long long L, long long slice_len) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int o0 = (blockIdx.y * 4 + wave) * 4;
  if (o0 >= Co) return;
  const long long b = blockIdx.z, l0 = (long long)blockIdx.x * slice_len;
  const long long l1 = l0 + slice_len < L ? l0 + slice_len : L;
  const float* x = X + b * CI * L;
  const float* dy = dY + (b * Co + o0) * L;
  const int no = Co - o0 < 4 ? Co - o0 : 4;
  float acc[4][CI];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < CI; ++i) acc[k][i] = 0.f;
  for (long long l = l0 + 4 * lane; l < l1; l += 256) {
    float4 xv[CI], dv[4];
#pragma unroll
    for (int i = 0; i < CI; ++i) xv[i] = *reinterpret_cast<const float4*>(x + (long long)i * L + l);
#pragma unroll
    for (int k = 0; k < 4; ++k) dv[k] = k < no ? *reinterpret_cast<const float4*>(dy + (long long)k * L + l) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < CI; ++i)
        acc[k][i] += (dv[k].x * xv[i].x + dv[k].y * xv[i].y) + (dv[k].z * xv[i].z + dv[k].w * xv[i].w);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < CI; ++i) {
      float v = acc[k][i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      acc[k][i] = v;
    }
  if (lane == 0) {
    float* out = part + ((b * gridDim.x + blockIdx.x) * Co + o0) * CI;
    for (int k = 0; k < no; ++k)
#pragma unroll
      for (int i = 0; i < CI; ++i) out[k * CI + i] = acc[k][i];
  }
}

static long long smallci_slices(long long B, long long Co, long long L) {
  const long long groups = B * ((Co + 15) / 16);
  long long s = (1024 + groups - 1) / groups;                 // ~1024 workgroups
  const long long most = (L + 4095) / 4096;                   // at least 4096 points per slice
  if (s > most) s = most;
  return s < 1 ? 1 : s;
}

extern "C" int64_t regnet_conv1x1_wgrad_smallci_partials(int64_t B, int64_t Co, int64_t L) {
  return B <= 0 || Co <= 0 || L <= 0 ? 0 : B * smallci_slices(B, Co, L);
}

// part: (regnet_conv1x1_wgrad_smallci_partials(B, Co, L), Co, Ci) floats; dW = their sum over the first axis.
extern "C" int regnet_conv1x1_wgrad_smallci_f32(const float* dY, const float* X, float* part, int64_t B, int64_t Co, int64_t Ci,
                                                int64_t L, void* stream) {
  if (B < 0 || Co < 1 || Ci < 1 || Ci > 8 || L < 4 || (L % 4) || B >= 65536 || Co >= (1ll << 20)) return REGNET_ERR_SHAPE;
  if (B == 0) return REGNET_OK;
  if (!dY || !X || !part) return REGNET_ERR_NULL;
  if (!tg_aligned16(dY) || !tg_aligned16(X)) return REGNET_ERR_SHAPE;
  const long long S = smallci_slices(B, Co, L);
  long long slice_len = (L + S - 1) / S;
  slice_len = (slice_len + 255) / 256 * 256;                 // whole wave steps
  const dim3 grid((unsigned)S, (unsigned)((Co + 15) / 16), (unsigned)B);
  hipStream_t st = as_stream(stream);
  switch (Ci) {
#define SMALLCI_CASE(n) case n: hipLaunchKernelGGL(wgrad_smallci_kernel<n>, grid, dim3(256), 0, st, dY, X, part, (int)Co, (long long)L, slice_len); break;
    SMALLCI_CASE(1) SMALLCI_CASE(2) SMALLCI_CASE(3) SMALLCI_CASE(4) SMALLCI_CASE(5) SMALLCI_CASE(6) SMALLCI_CASE(7) SMALLCI_CASE(8)
#undef SMALLCI_CASE
  }
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// The mirror case: a HANDFUL of output channels (the score convolution of the segmentation head, 128 -> 1 with bias over 8 x 25 600
// points, pointnet2.py:51, :118): forward a reduction over the channels per point, input gradient an outer product -- both streams.
// (The weight gradient is wgrad_smallci_kernel with the operands' roles swapped.)  MIOpen ran them as an implicit GEMM with
// transposes: 0.18 ms forward.
template <int CO>
__global__ __launch_bounds__(256) void conv_smallco_fwd_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ X, float* __restrict__ Y, int Ci, long long L) {
  const long long l = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (l >= L) return;
  const float* x = X + (long long)blockIdx.y * Ci * L + l;
  float4 acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) { const float b = bias ? bias[o] : 0.f; acc[o] = make_float4(b, b, b, b); }
#pragma unroll 4
  for (int i = 0; i < Ci; ++i) {
    const float4 v = *reinterpret_cast<const float4*>(x + (long long)i * L);
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      const float w = W[o * Ci + i];
      acc[o].x = fmaf(w, v.x, acc[o].x); acc[o].y = fmaf(w, v.y, acc[o].y);
      acc[o].z = fmaf(w, v.z, acc[o].z); acc[o].w = fmaf(w, v.w, acc[o].w);
    }
  }
  float* y = Y + (long long)blockIdx.y * CO * L + l;
#pragma unroll
  for (int o = 0; o < CO; ++o) *reinterpret_cast<float4*>(y + (long long)o * L) = acc[o];
}

template <int CO>
__global__ __launch_bounds__(256) void conv_smallco_dgrad_kernel(const float* __restrict__ W, const float* __restrict__ dY,
                                                                float* __restrict__ dX, int Ci, long long L) {
  const long long l = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (l >= L) return;
  const float* dy = dY + (long long)blockIdx.y * CO * L + l;
  float4 d[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) d[o] = *reinterpret_cast<const float4*>(dy + (long long)o * L);
  float* dx = dX + (long long)blockIdx.y * Ci * L + l;
  for (int i = 0; i < Ci; ++i) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      const float w = W[o * Ci + i];
      acc.x = fmaf(w, d[o].x, acc.x); acc.y = fmaf(w, d[o].y, acc.y);
      acc.z = fmaf(w, d[o].z, acc.z); acc.w = fmaf(w, d[o].w, acc.w);
    }
    *reinterpret_cast<float4*>(dx + (long long)i * L) = acc;
  }
}

// dir 0: Y (B, Co, L) = W (Co, Ci) . X (B, Ci, L) + bias (Co, may be NULL);  dir 1: dX (B, Ci, L) = W^T . dY (B, Co, L).  1 <= Co <= 4.
extern "C" int regnet_conv1x1_smallco_f32(int dir, const float* W, const float* bias, const float* in, float* out, int64_t B,
                                          int64_t Co, int64_t Ci, int64_t L, void* stream) {
  if (B < 0 || Co < 1 || Co > 4 || Ci < 1 || Ci >= (1ll << 20) || L < 4 || (L % 4) || B >= 65536 || (dir != 0 && dir != 1)) return REGNET_ERR_SHAPE;
  if (B == 0) return REGNET_OK;
  if (!W || !in || !out) return REGNET_ERR_NULL;
  if (!tg_aligned16(in) || !tg_aligned16(out)) return REGNET_ERR_SHAPE;
  const dim3 grid((unsigned)((L / 4 + 255) / 256), (unsigned)B);
  hipStream_t st = as_stream(stream);
  switch (Co) {
#define SMALLCO_CASE(n) case n: \
      if (dir == 0) hipLaunchKernelGGL(conv_smallco_fwd_kernel<n>, grid, dim3(256), 0, st, W, bias, in, out, (int)Ci, (long long)L); \
      else hipLaunchKernelGGL(conv_smallco_dgrad_kernel<n>, grid, dim3(256), 0, st, W, in, out, (int)Ci, (long long)L); \
      break;
    SMALLCO_CASE(1) SMALLCO_CASE(2) SMALLCO_CASE(3) SMALLCO_CASE(4)
#undef SMALLCO_CASE
  }
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Shapes this build handles (everything else: the caller keeps its library path): channel counts multiples of 16,
// points a multiple of 4, 16-byte aligned buffers.
extern "C" int regnet_conv1x1_train_supported(int64_t Co, int64_t Ci, int64_t L) {
  return Co >= 32 && Ci >= 32 && (Co % 16) == 0 && (Ci % 16) == 0 && L >= 64 && (L % 4) == 0;
}

static int tg_fwd(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci, int64_t L, int32_t* ticket,
                  void* stream) {
  if (B < 0 || !regnet_conv1x1_train_supported(Co, Ci, L)) return REGNET_ERR_SHAPE;
  if (B == 0) return REGNET_OK;
  if (!W || !X || !Y) return REGNET_ERR_NULL;
  if (!tg_aligned16(W) || !tg_aligned16(X) || !tg_aligned16(Y)) return REGNET_ERR_SHAPE;
  TgArgs a = {};
  a.A = W; a.lda = Ci;                                   // row: (o, i) at W + o * Ci + i
  a.B = X; a.ldb = L; a.b_batch = Ci * L;                // kmaj: (l, i) at X[b] + i * L + l
  a.C = Y; a.ldc = L; a.c_batch = Co * L;
  a.M = (int)Co; a.N = (int)L; a.K = (int)Ci; a.slices = 1;
  return ticket ? tg_launch_stream<false>(a, B, ticket, as_stream(stream)) : tg_launch<false, true, 256>(a, B, as_stream(stream));
}

extern "C" int regnet_conv1x1_fwd_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci,
                                      int64_t L, void* stream) {
  return tg_fwd(W, X, Y, B, Co, Ci, L, nullptr, stream);
}

extern "C" int regnet_conv1x1_fwd_stream_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci,
                                             int64_t L, int32_t* ticket, void* stream) {
  if (!ticket) return REGNET_ERR_NULL;
  return tg_fwd(W, X, Y, B, Co, Ci, L, ticket, stream);
}

extern "C" int regnet_conv1x1_bnrelu_supported(int64_t Co, int64_t Ci, int64_t L) {
  return regnet_conv1x1_train_supported(Co, Ci, L) && Ci <= TG_AFF_MAXK && (L % 16) == 0;
}

extern "C" int regnet_conv1x1_fwd_bnrelu_stream_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci,
                                                    int64_t L, const float* scale, const float* shift, int relu,
                                                    int32_t* ticket, void* stream) {
  if (B < 0 || !regnet_conv1x1_bnrelu_supported(Co, Ci, L)) return REGNET_ERR_SHAPE;
  if (B == 0) return REGNET_OK;
  if (!W || !X || !Y || !scale || !shift || !ticket) return REGNET_ERR_NULL;
  if (!tg_aligned16(W) || !tg_aligned16(X) || !tg_aligned16(Y)) return REGNET_ERR_SHAPE;
  TgArgs a = {};
  a.A = W; a.lda = Ci;
  a.B = X; a.ldb = L; a.b_batch = Ci * L;
  a.C = Y; a.ldc = L; a.c_batch = Co * L;
  a.M = (int)Co; a.N = (int)L; a.K = (int)Ci; a.slices = 1;
  a.bscale = scale; a.bshift = shift; a.brelu = relu;
  return tg_launch_stream<false, true>(a, B, ticket, as_stream(stream));
}

// The forward with the statistics pass of the BatchNorm that FOLLOWS the convolution taken from the accumulators (STATS): `sums` (2 Co
// doubles: sum, sum of squares per output channel over all B L points) is zeroed and filled by this call;
// regnet_bn_train_stats_from_sums_f32 / regnet_bn_relu_train_fwd_from_sums_f32 (bn_train.hip) continue from it.  scale == NULL: the plain
// forward; otherwise the operand's own BatchNorm (+ ReLU) is applied on the fragment as regnet_conv1x1_fwd_bnrelu_stream_f32 does.
extern "C" int regnet_conv1x1_fwd_stats_supported(int64_t Co, int64_t Ci, int64_t L, int affine) {
  if (!regnet_conv1x1_train_supported(Co, Ci, L) || Co > TG_STAT_MAXM) return 0;
  return affine ? (Ci <= TG_STAT_AFF_MAXK && (L % 16) == 0) : 1;
}

extern "C" int regnet_conv1x1_fwd_stats_stream_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci,
                                                   int64_t L, const float* scale, const float* shift, int relu, int32_t* ticket,
                                                   void* sums, void* stream) {
  if (B < 0 || !regnet_conv1x1_fwd_stats_supported(Co, Ci, L, scale != nullptr)) return REGNET_ERR_SHAPE;
  if (B == 0) return REGNET_OK;
  if (!W || !X || !Y || !ticket || !sums || (scale && !shift)) return REGNET_ERR_NULL;
  if (!tg_aligned16(W) || !tg_aligned16(X) || !tg_aligned16(Y) || (reinterpret_cast<uintptr_t>(sums) & 7)) return REGNET_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(sums, 0, (size_t)Co * 2 * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  TgArgs a = {};
  a.A = W; a.lda = Ci;
  a.B = X; a.ldb = L; a.b_batch = Ci * L;
  a.C = Y; a.ldc = L; a.c_batch = Co * L;
  a.M = (int)Co; a.N = (int)L; a.K = (int)Ci; a.slices = 1;
  a.stat = static_cast<double*>(sums);
  if (!scale) return tg_launch_stream<false, false, true>(a, B, ticket, st);
  a.bscale = scale; a.bshift = shift; a.brelu = relu;
  return tg_launch_stream<false, true, true>(a, B, ticket, st);
}

static int tg_dgrad(const float* W, const float* dY, float* dX, int64_t B, int64_t Co, int64_t Ci, int64_t L, int32_t* ticket,
                    void* stream) {
  if (B < 0 || !regnet_conv1x1_train_supported(Co, Ci, L)) return REGNET_ERR_SHAPE;
  if (B == 0) return REGNET_OK;
  if (!W || !dY || !dX) return REGNET_ERR_NULL;
  if (!tg_aligned16(W) || !tg_aligned16(dY) || !tg_aligned16(dX)) return REGNET_ERR_SHAPE;
  TgArgs a = {};
  a.A = W; a.lda = Ci;                                   // kmaj: W^T element (i, o) at W + o * Ci + i
  a.B = dY; a.ldb = L; a.b_batch = Co * L;               // kmaj: (l, o) at dY[b] + o * L + l
  a.C = dX; a.ldc = L; a.c_batch = Ci * L;
  a.M = (int)Ci; a.N = (int)L; a.K = (int)Co; a.slices = 1;
  return ticket ? tg_launch_stream<true>(a, B, ticket, as_stream(stream)) : tg_launch<true, true, 256>(a, B, as_stream(stream));
}

extern "C" int regnet_conv1x1_dgrad_f32(const float* W, const float* dY, float* dX, int64_t B, int64_t Co, int64_t Ci,
                                        int64_t L, void* stream) {
  return tg_dgrad(W, dY, dX, B, Co, Ci, L, nullptr, stream);
}

extern "C" int regnet_conv1x1_dgrad_stream_f32(const float* W, const float* dY, float* dX, int64_t B, int64_t Co, int64_t Ci,
                                               int64_t L, int32_t* ticket, void* stream) {
  if (!ticket) return REGNET_ERR_NULL;
  return tg_dgrad(W, dY, dX, B, Co, Ci, L, ticket, stream);
}

// Slices of the point axis for the weight gradient: enough (slice x tile) workgroups to fill the chip, slices of at
// least 256 points and a multiple of 16; 1 when L does not divide.
extern "C" int64_t regnet_conv1x1_wgrad_slices(int64_t B, int64_t Co, int64_t Ci, int64_t L) {
  const int64_t tiles = ((Co + TG_BM - 1) / TG_BM) * ((Ci + 255) / 256);
  int64_t s = 1;
  while (L % (2 * s) == 0 && (L / (2 * s)) % 16 == 0 && L / (2 * s) >= 256 && s * tiles * B < TG_WGRAD_BLOCKS) s *= 2;
  return s;
}

extern "C" int64_t regnet_conv1x1_wgrad_workspace_bytes(int64_t B, int64_t Co, int64_t Ci, int64_t L) {
  const int64_t n = B * regnet_conv1x1_wgrad_slices(B, Co, Ci, L);
  return n > 1 ? n * Co * Ci * (int64_t)sizeof(float) : 0;
}

static int tg_wgrad(const float* dY, const float* X, float* dW, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                    const float* scale, const float* shift, int relu, bool affine, void* workspace, void* stream) {
  if (B <= 0 || !regnet_conv1x1_train_supported(Co, Ci, L) || (L % 16)) return REGNET_ERR_SHAPE;
  if (!dY || !X || !dW) return REGNET_ERR_NULL;
  const int64_t S = regnet_conv1x1_wgrad_slices(B, Co, Ci, L), n = B * S;
  if (n > 1 && !workspace) return REGNET_ERR_NULL;
  if (!tg_aligned16(dY) || !tg_aligned16(X) || !tg_aligned16(dW) || (n > 1 && !tg_aligned16(workspace))) return REGNET_ERR_SHAPE;
  TgArgs a = {};
  a.A = dY; a.lda = L; a.a_batch = Co * L; a.a_slice = L / S;      // row: (o, l)
  a.B = X;  a.ldb = L; a.b_batch = Ci * L; a.b_slice = L / S;      // row: (i, l)
  a.C = n > 1 ? (float*)workspace : dW; a.ldc = Ci; a.c_batch = S * Co * Ci; a.c_slice = Co * Ci;
  a.M = (int)Co; a.N = (int)Ci; a.K = (int)(L / S); a.slices = (int)S;
  a.bscale = scale; a.bshift = shift; a.brelu = relu;
  int rc = affine ? (Ci <= 128 ? tg_launch<false, false, 128, true>(a, B, as_stream(stream))
                               : tg_launch<false, false, 256, true>(a, B, as_stream(stream)))
                  : (Ci <= 128 ? tg_launch<false, false, 128>(a, B, as_stream(stream))
                               : tg_launch<false, false, 256>(a, B, as_stream(stream)));
  if (rc || n == 1) return rc;
  const long long total = Co * Ci;                                 // multiple of 4 (both multiples of 16)
  hipLaunchKernelGGL(tg_reduce_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                     (const float*)workspace, (int)n, total, dW);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_conv1x1_wgrad_f32(const float* dY, const float* X, float* dW, int64_t B, int64_t Co, int64_t Ci,
                                        int64_t L, void* workspace, void* stream) {
  return tg_wgrad(dY, X, dW, B, Co, Ci, L, nullptr, nullptr, 0, false, workspace, stream);
}

extern "C" int regnet_conv1x1_wgrad_bnrelu_f32(const float* dY, const float* X, float* dW, int64_t B, int64_t Co, int64_t Ci,
                                               int64_t L, const float* scale, const float* shift, int relu, void* workspace,
                                               void* stream) {
  if (!scale || !shift) return REGNET_ERR_NULL;
  return tg_wgrad(dY, X, dW, B, Co, Ci, L, scale, shift, relu, true, workspace, stream);
}
