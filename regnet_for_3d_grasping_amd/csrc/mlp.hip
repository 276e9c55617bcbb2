// mlp.hip -- the per-point shared-MLP contraction of the set-abstraction / feature-propagation
// layers on gfx950 matrix cores (fp32-input MFMA, exact fp32 products, fp32 accumulate).
//
// Reference behaviour restated (paths relative to /root/reference/multi_model/utils):
//   SharedMLP = [1x1 conv (bias-free) -> BatchNorm -> ReLU]*   pn2_utils/nn/modules/mlp.py:55-114, conv.py:6-76
//   SA block: group -> (xyz - centre | feature) -> SharedMLP -> max over K   pn2_utils/modules.py:39-56,:210-246
//   FP block: 3-NN weights -> interpolate -> cat(interp, skip) -> SharedMLP  pn2_utils/modules.py:104-131,:500-509
//   score head: conv_score(+bias) -> bn_score -> sigmoid                    pointnet2.py:116-119
//
// Layout: activations are CHANNELS-LAST, X[point][channel]; a 1x1 conv over P points is the GEMM
//   C[P x N] = X[P x K] * W[N x K]^T,  then  y = relu(scale[n] * c + shift[n])   (eval-mode BN folded
// into a per-channel affine).  Both operands sit in LDS as [row][k] tiles, so the A fragment
// (A[i][k], i = lane&31, k = lane>>5) and the B fragment (B[k][j] = W[j][k]) are read with the
// same ds_read_b128 pattern: a lane fetches 4 consecutive k of its row and feeds 4 MFMAs; the
// k order inside an 8-wide slab is therefore permuted identically for A and W (a sum is order
// independent up to fp32 rounding).
//
// Tile: 128 points x 128 channels x BK k per workgroup of 4 waves (2x2), each wave 64x64 =
// 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 VGPRs).  LDS rows are padded by 4 floats so the
// 16-lane service groups of ds_read_b128 hit 64 distinct banks (SQ_LDS_BANK_CONFLICT = 0).  Global->LDS goes through
// registers (the gather prologue needs per-row pointers), prefetching tile t+1 while tile t is
// multiplied; two LDS buffers, one barrier per k-tile.
#include <stdlib.h>
#include "common.h"
#include "gemm2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define MLP_THREADS 256
#ifndef MLP_BK
#define MLP_BK 16  // k-depth of one LDS tile: 16 -> 41 KB LDS and <=128 VGPRs = 4 workgroups/CU (measured
                   // 91 vs 77 TFLOP/s over the ScoreNet shapes against 32 -> 74 KB, 2 workgroups/CU: the
                   // extra resident waves cover the global-load latency of the short-K layers)
#endif
#define BK MLP_BK
#ifndef MLP_ABLATE
#define MLP_ABLATE 0  // timing experiments only (scripts/ablate): 1 no global loads after tile 0, 2 also no LDS
                      // refill / barrier, 3 no epilogue, 4 = 2 + 3; results are wrong for anything but 0
#endif
#ifndef MLP_TRACE
#define MLP_TRACE 0   // timing experiments only (scripts/ablate/g2_bench.cpp): per-workgroup s_memtime stamps + HW_ID, optional
                      // start-up stagger of the co-resident workgroups
#endif
#define LDS_LD (BK + 4)  // +4 floats: the 16-lane service groups of ds_read_b128 then hit 64 distinct banks
#define ROWS_PER_PASS (MLP_THREADS / (BK / 4))  // rows one staging pass of the 256 threads covers
#define STAGE_PASSES (BM / ROWS_PER_PASS)
#ifndef MLP_MIN_WAVES
#define MLP_MIN_WAVES (BK == 32 ? 2 : 4)
#endif

struct MlpArgs {
  // ---- A operand, plain mode: rows of a channels-last activation buffer
  const float* A;
  long long lda;
  int Ka;  // valid columns of A (multiple of 4); columns >= Ka read as zero
  // ---- A operand, gather mode (first layer of a set-abstraction block):
  //      row p = (scene b, centre m, neighbour k);  A[p] = [feat[b, nbr[p], 0:Cf] | xyz[nbr[p]] - xyz[ctr[p/group]] | 0]
  const float* feat;  // (B*Nsrc, ldf) channels-last features of the source level (may be NULL if Cf == 0)
  long long ldf;
  long long fb, fn, fc;  // generic strides of feat: element (b, n, c) at feat[b*fb + n*fn + c*fc]
  int Cf;
  int feat_vec;  // 1: channel stride 1, Cf % 4 == 0 and 16-byte aligned rows -> float4 loads
  const float* xyz;  // (B,3,Nsrc) strided
  long long xb, xc, xn;
  const long long* nbr;  // [P] neighbour ids (ball-query output, int64)
  const long long* ctr;  // [P/group] centre ids (FPS output, int64)
  int group;             // neighbours per centre (64)
  long long rows_per_scene;
  // ---- W operand: packed [Npad][Kpad] (zero padded), folded BN affine
  const float* W;
  int Kpad;
  const float* scale;
  const float* shift;
  // ---- gather mode with the FIRST MLP layer fused into the operand load (AMODE 2): the gathered
  //      row x = [feat | rel xyz] (Cf + 3 <= 8 inputs) is pushed through layer 1 on the VALU while
  //      it is staged,  A[p][k] = relu(scale1[k] * dot(W1[k][0:8], x) + shift1[k]),  so the
  //      (P x C1) activation of layer 1 never goes to HBM.  W1 is packed [C1][8] (zero padded).
  const float* W1;
  const float* scale1;
  const float* shift1;
  // ---- gather mode over PRE-MULTIPLIED source rows (AMODE 3).  The first layer of a set-abstraction
  //      block is linear in its gathered input, W1 [f_j | x_j - x_c] = (W1 [f_j | x_j]) - (W1x x_c), so it is
  //      evaluated once per SOURCE point (U = scale1 * W1 [f | x], rows b*Nsrc + j, via `feat`/`fb`/`fn`) and
  //      once per CENTRE (V = scale1 * W1x x_c - shift1) instead of once per (centre, neighbour) pair:
  //        A[p][k] = relu(U[b, nbr[p]][k] - V[p / group][k]).
  const float* V;
  long long ldv;
  // ---- output
  float* C;
  long long ldc;
  long long P;
  int N;
  int relu;
  int ksplit;      // > 1: deterministic split-K for skinny problems -- workgroup (tile, s) multiplies k-slice s and writes its RAW
                   // partial sums to C + s * P * ldc (C is then a workspace); splitk_finish_kernel adds the slices in order
  int wide_store;  // C rows 16-byte aligned (ldc % 4 == 0): interior tiles use the LDS-transposed dwordx4 epilogue
#if MLP_TRACE
  unsigned long long* trace;   // [grid][6]: hw_id, xcc_id, t_start, t_loop, t_loop_end, t_end
  int stagger_cycles;          // first-round workgroups wait (wave slot & 3) * stagger_cycles before starting
  int first_round;
#endif
};

__device__ __forceinline__ void xcd_tile(unsigned vblock, int& tm, int& tn, int tiles_m, int tiles_n) {
  // Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, a speed-only
  // observation).  Remap so that consecutive blocks ON ONE XCD walk the N-tiles of the same
  // M-tile: the A tile is then fetched into one L2 instead of eight.
  const long long total = (long long)tiles_m * tiles_n;
  const long long L = vblock;
  const long long q = total / 8, r = total % 8;
  const long long xcd = L % 8, s = L / 8;
  const long long t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
  tm = (int)(t / tiles_n);
  tn = (int)(t % tiles_n);
}

// The occupancy REQUEST of AMODE 0 cannot be met (the tile's LDS allows 3 workgroups' worth of waves per SIMD); what it is for is
// the register cap that comes with it.  The backend says so once per instantiation: silenced for THIS kernel only, so that a
// failed unroll / vectorise remark anywhere else in the file is still seen; build.py checks that the kernel does not spill.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
template <int AMODE, bool POOL>
// (AMODE 0 -- the region heads' small layers on the side stream -- is compiled for 5 waves per SIMD: 88 registers instead of
// "whatever 4 waves allow" (126), so that its workgroups find room on CUs whose SIMDs hold two 208-register chain waves)
__global__ __launch_bounds__(MLP_THREADS, AMODE == 2 ? 3 : (AMODE == 0 ? 5 : MLP_MIN_WAVES)) void mlp_gemm_kernel(const MlpArgs p) {
  // one LDS block: [2][BM][LDS_LD] A tiles, [2][BN][LDS_LD] W tiles; re-used by the epilogue as
  // per-wave transposition buffers (4 x 64 x 36 floats)
  __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDS_LD + 2 * BN * LDS_LD];
  float (*sA)[BM][LDS_LD] = reinterpret_cast<float (*)[BM][LDS_LD]>(smem);
  float (*sW)[BN][LDS_LD] = reinterpret_cast<float (*)[BN][LDS_LD]>(smem + 2 * BM * LDS_LD);
  static_assert(2 * BM * LDS_LD + 2 * BN * LDS_LD >= 4 * 64 * 36, "epilogue staging does not fit");
  __shared__ __attribute__((aligned(16))) float sW1[AMODE == 2 ? 256 * 10 : 4];  // [C1 <= 256][8 weights | scale | shift]

#if MLP_TRACE
  const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  if (p.stagger_cycles > 0 && (int)blockIdx.x < p.first_round) {
    const unsigned long long until = __builtin_readcyclecounter() + (unsigned long long)(hw_id & 3) * p.stagger_cycles;
    while (__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(16);
  }
  const unsigned long long tr_start = __builtin_readcyclecounter();
#endif
  constexpr bool GATHER = AMODE == 1 || AMODE == 2;
  constexpr bool FUSE1 = AMODE == 2;
  constexpr bool PREMUL = AMODE == 3;
  const int tid = threadIdx.x;
  if (FUSE1) {
    for (int k = tid; k < p.Kpad; k += MLP_THREADS) {
#pragma unroll
      for (int c = 0; c < 8; ++c) sW1[k * 10 + c] = p.W1[(long long)k * 8 + c];
      sW1[k * 10 + 8] = p.scale1[k];
      sW1[k * 10 + 9] = p.shift1[k];
    }
    __syncthreads();
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (int)((p.P + BM - 1) / BM);
  int tm, tn;
  const int ks = p.ksplit > 1 ? (int)(blockIdx.x % (unsigned)p.ksplit) : 0;
  xcd_tile(p.ksplit > 1 ? blockIdx.x / (unsigned)p.ksplit : blockIdx.x, tm, tn, tiles_m, tiles_n);
  const long long row0 = (long long)tm * BM;
  const int col0 = tn * BN;

  // staging map: thread -> (row sr + ROWS_PER_PASS*i, 4 columns starting at 4*c4)
  const int sr = tid / (BK / 4), c4 = tid % (BK / 4);

  // per-row source pointers
  const float* arow[STAGE_PASSES];
  bool arow_ok[STAGE_PASSES];
  float relx[STAGE_PASSES], rely[STAGE_PASSES], relz[STAGE_PASSES];
  float xin[STAGE_PASSES][8];  // FUSE1: the gathered layer-1 input row [feat | rel xyz | 0]
  const float* vrow[STAGE_PASSES];  // PREMUL: the centre's row of V
#pragma unroll
  for (int i = 0; i < STAGE_PASSES; ++i) {
    const long long row = row0 + sr + ROWS_PER_PASS * i;
    arow_ok[i] = row < p.P;
    const long long rs = arow_ok[i] ? row : 0;
    vrow[i] = nullptr;
    if (PREMUL) {
      const long long b = rs / p.rows_per_scene;
      arow[i] = p.feat + b * p.fb + p.nbr[rs] * p.fn;
      vrow[i] = p.V + (rs / p.group) * p.ldv;
      relx[i] = rely[i] = relz[i] = 0.f;
    } else if (GATHER) {
      const long long b = rs / p.rows_per_scene;
      const long long j = p.nbr[rs];
      const long long cj = p.ctr[rs / p.group];
      arow[i] = p.feat ? p.feat + b * p.fb + j * p.fn : p.xyz;  // never dereferenced when Cf == 0
      const float* xb = p.xyz + b * p.xb;
      relx[i] = xb[j * p.xn] - xb[cj * p.xn];
      rely[i] = xb[p.xc + j * p.xn] - xb[p.xc + cj * p.xn];
      relz[i] = xb[2 * p.xc + j * p.xn] - xb[2 * p.xc + cj * p.xn];
      if (FUSE1) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float x = 0.f;
          if (c < p.Cf) x = arow[i][(long long)c * p.fc];
          else if (c == p.Cf) x = relx[i];
          else if (c == p.Cf + 1) x = rely[i];
          else if (c == p.Cf + 2) x = relz[i];
          xin[i][c] = arow_ok[i] ? x : 0.f;
        }
      }
    } else {
      arow[i] = p.A + rs * p.lda;
      relx[i] = rely[i] = relz[i] = 0.f;
    }
  }
  const float* wrow[STAGE_PASSES];
#pragma unroll
  for (int i = 0; i < STAGE_PASSES; ++i) wrow[i] = p.W + (long long)(col0 + sr + ROWS_PER_PASS * i) * p.Kpad;

  // Staging registers.  (Kept as plain locals + an inlined helper macro: a by-reference lambda
  // made the compiler keep ra/rw in scratch and wait for every global load right after issuing it.)
  float4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;  // named scalars (arrays here ended up in scratch)
  float4 rv0, rv1, rv2, rv3;                        // PREMUL: the V chunk that goes with ra*
  ra0 = ra1 = ra2 = ra3 = rw0 = rw1 = rw2 = rw3 = make_float4(0.f, 0.f, 0.f, 0.f);
  rv0 = rv1 = rv2 = rv3 = make_float4(0.f, 0.f, 0.f, 0.f);
// All global loads are UNCONDITIONAL and their results are not touched until STORE_TILE: addresses
// are clamped into valid memory and the fix-ups (zero columns, relative-xyz columns of a gathered
// row) are applied right before the LDS write, one k-tile of MFMAs later.  (A select directly after
// the load made every wave sit out the full memory latency at the top of each k-tile: -11 %.)
//   * rows >= P read row 0: they produce accumulator rows the epilogue never stores;
//   * plain mode, columns >= Ka read columns 0..3 of the row: W is zero there (packed, zero padded).
#define LOAD_PASS(I, RA, RW, RV, KC)                                                                    \
  if ((I) < STAGE_PASSES) {                                                                             \
    constexpr int i = (I) < STAGE_PASSES ? (I) : 0;                                                     \
    if (PREMUL) {                                                                                       \
      const int kc_u = (KC) < p.Ka ? (KC) : 0;                                                          \
      RA = *reinterpret_cast<const float4*>(arow[i] + kc_u);                                            \
      RV = *reinterpret_cast<const float4*>(vrow[i] + kc_u);                                            \
    } else if (FUSE1) {                                                                                        \
      float e[4];                                                                                       \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                   \
        const float* w = &sW1[((KC) + t) * 10];                                                         \
        float a1 = w[0] * xin[i][0];                                                                    \
        _Pragma("unroll") for (int c = 1; c < 8; ++c) a1 += w[c] * xin[i][c];                           \
        e[t] = fmaxf(a1 * w[8] + w[9], 0.f);                                                            \
      }                                                                                                 \
      RA = make_float4(e[0], e[1], e[2], e[3]);                                                         \
    } else if (GATHER) {                                                                                \
      if (p.feat_vec) { /* wave-uniform branch */                                                       \
        RA = *reinterpret_cast<const float4*>(arow[i] + ((KC) + 4 <= p.Cf ? (KC) : 0));                 \
      } else if (p.Cf > 0) {                                                                            \
        RA.x = arow[i][(long long)min((KC) + 0, p.Cf - 1) * p.fc];                                      \
        RA.y = arow[i][(long long)min((KC) + 1, p.Cf - 1) * p.fc];                                      \
        RA.z = arow[i][(long long)min((KC) + 2, p.Cf - 1) * p.fc];                                      \
        RA.w = arow[i][(long long)min((KC) + 3, p.Cf - 1) * p.fc];                                      \
      }                                                                                                 \
    } else {                                                                                            \
      RA = *reinterpret_cast<const float4*>(arow[i] + ((KC) < p.Ka ? (KC) : 0));                        \
    }                                                                                                   \
    RW = *reinterpret_cast<const float4*>(wrow[i] + (KC));                                              \
  }
// gather mode: columns [0, Cf) are features, Cf..Cf+2 the relative xyz, the rest zero
#define FIX_PASS(I, RA, RV, KC)                                                                         \
  if (PREMUL && (I) < STAGE_PASSES) {                                                                   \
    RA = make_float4(fmaxf(RA.x - RV.x, 0.f), fmaxf(RA.y - RV.y, 0.f), fmaxf(RA.z - RV.z, 0.f),         \
                     fmaxf(RA.w - RV.w, 0.f));                                                          \
  }                                                                                                     \
  if (GATHER && !FUSE1 && (I) < STAGE_PASSES) {                                                         \
    constexpr int i = (I) < STAGE_PASSES ? (I) : 0;                                                     \
    float e[4] = {RA.x, RA.y, RA.z, RA.w};                                                              \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                     \
      const int col = (KC) + t;                                                                         \
      float x = col < p.Cf ? e[t] : 0.f;                                                                \
      x = (col == p.Cf) ? relx[i] : x;                                                                  \
      x = (col == p.Cf + 1) ? rely[i] : x;                                                              \
      x = (col == p.Cf + 2) ? relz[i] : x;                                                              \
      e[t] = x;                                                                                         \
    }                                                                                                   \
    RA = make_float4(e[0], e[1], e[2], e[3]);                                                           \
  }
#define LOAD_TILE(K0)                  \
  do {                                 \
    const int kc_ = (K0) + 4 * c4;     \
    LOAD_PASS(0, ra0, rw0, rv0, kc_)   \
    LOAD_PASS(1, ra1, rw1, rv1, kc_)   \
    LOAD_PASS(2, ra2, rw2, rv2, kc_)   \
    LOAD_PASS(3, ra3, rw3, rv3, kc_)   \
  } while (0)
#define STORE_PASS(I, RA, RW, BUF)                                                          \
  if ((I) < STAGE_PASSES) {                                                                 \
    *reinterpret_cast<float4*>(&sA[BUF][sr + ROWS_PER_PASS * (I)][4 * c4]) = RA;            \
    *reinterpret_cast<float4*>(&sW[BUF][sr + ROWS_PER_PASS * (I)][4 * c4]) = RW;            \
  }
#define STORE_TILE(BUF, K0)            \
  do {                                 \
    const int kf_ = (K0) + 4 * c4;     \
    FIX_PASS(0, ra0, rv0, kf_)         \
    FIX_PASS(1, ra1, rv1, kf_)         \
    FIX_PASS(2, ra2, rv2, kf_)         \
    FIX_PASS(3, ra3, rv3, kf_)         \
    STORE_PASS(0, ra0, rw0, BUF)       \
    STORE_PASS(1, ra1, rw1, BUF)       \
    STORE_PASS(2, ra2, rw2, BUF)       \
    STORE_PASS(3, ra3, rw3, BUF)       \
  } while (0)
  static_assert(STAGE_PASSES <= 4, "staging macros cover at most 4 passes");

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int KT_all = p.Kpad / BK;
  const int kt_per = p.ksplit > 1 ? (KT_all + p.ksplit - 1) / p.ksplit : KT_all;
  const int kt0 = ks * kt_per;
  const int KT = min(KT_all, kt0 + kt_per);   // this workgroup's k-tiles are [kt0, KT) (empty slices write zeros)
  if (kt0 < KT) {
    LOAD_TILE(kt0 * BK);
    STORE_TILE(0, kt0 * BK);
  }
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
#if MLP_TRACE
  const unsigned long long tr_loop = __builtin_readcyclecounter();
#endif
  for (int kt = kt0; kt < KT; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < KT && !(MLP_ABLATE == 1 || MLP_ABLATE == 2 || MLP_ABLATE == 4)) LOAD_TILE((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        a[mi] = *reinterpret_cast<const float4*>(&sA[buf][wr * 64 + mi * 32 + fr][kk * 8 + 4 * fh]);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        b[ni] = *reinterpret_cast<const float4*>(&sW[buf][wc * 64 + ni * 32 + fr][kk * 8 + 4 * fh]);
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
    if (MLP_ABLATE == 2 || MLP_ABLATE == 4) continue;
    if (kt + 1 < KT) STORE_TILE(buf ^ 1, (kt + 1) * BK);
    __syncthreads();
  }
  if (MLP_ABLATE >= 3) {   // keep the accumulators live without the store traffic
    float keep = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += acc[mi][ni][r];
    if (keep == 1.2345e-30f) p.C[0] = keep;
    return;
  }

#if MLP_TRACE
  const unsigned long long tr_loop_end = __builtin_readcyclecounter();
#endif
  // ---- epilogue: folded BN affine + ReLU (+ max over the 64 rows of a group) -------------
  // C/D layout of v_mfma_f32_32x32x2_f32: element r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
  const bool splitk = !POOL && p.ksplit > 1;
  float* const Cout = splitk ? p.C + (long long)ks * p.P * p.ldc : p.C;
  const int relu = splitk ? 0 : p.relu;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = col0 + wc * 64 + ni * 32 + fr;
    const bool col_ok = col < p.N;
    const float s = splitk ? 1.f : (col_ok ? p.scale[col] : 0.f);
    const float t = splitk ? 0.f : (col_ok ? p.shift[col] : 0.f);
    if (POOL) {
      float m = -__builtin_inff();
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float y = acc[mi][ni][r] * s + t;
          if (relu) y = fmaxf(y, 0.f);
          m = fmaxf(m, y);
        }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const long long g = (row0 + wr * 64) / 64;  // a wave's 64 rows are exactly one group
      if (col_ok && fh == 0 && row0 + wr * 64 < p.P) p.C[g * p.ldc + col] = m;
    } else {
      // Rows of one accumulator in register order: 0,1,2,3, 8,9,10,11, 16.., 24.. (+4 for the upper
      // lane half).  Walk them with pointer increments (no 64-bit multiply per store); interior
      // tiles take the unguarded path so the 32 stores of an accumulator issue back to back.
      const long long first_row = row0 + wr * 64 + 4 * fh;
      float* cp = Cout + first_row * p.ldc + col;
      const long long ld = p.ldc;
      const bool interior = (row0 + BM <= p.P) && (col0 + BN <= p.N);
      if (interior && p.wide_store) {
        // Transpose the 64x32 accumulator slab through this wave's LDS region and write 16 bytes
        // per lane: 8 dwordx4 stores (8 rows x 128 B each) instead of 32 dword stores.
        float* stage = smem + wave * (64 * 36);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float y = acc[mi][ni][r] * s + t;
            if (relu) y = fmaxf(y, 0.f);
            stage[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 36 + fr] = y;
          }
        float* cbase = Cout + (row0 + wr * 64) * p.ldc + (col0 + wc * 64 + ni * 32);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + (lane >> 3), q4 = lane & 7;
          const float4 v = *reinterpret_cast<const float4*>(&stage[row * 36 + q4 * 4]);
          *reinterpret_cast<float4*>(cbase + (long long)row * p.ldc + q4 * 4) = v;
        }
      } else if (interior) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float y = acc[mi][ni][r] * s + t;
            if (relu) y = fmaxf(y, 0.f);
            *cp = y;
            cp += ((r & 3) == 3) ? 5 * ld : ld;   // after rows 3, 11, 19, 27 jump to the next block of four
          }
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long long row = first_row + mi * 32 + (r & 3) + 8 * (r >> 2);
            float y = acc[mi][ni][r] * s + t;
            if (relu) y = fmaxf(y, 0.f);
            if (col_ok && row < p.P) *cp = y;
            cp += ((r & 3) == 3) ? 5 * ld : ld;
          }
        }
      }
    }
  }
#if MLP_TRACE
  if (p.trace && tid == 0) {
    unsigned long long* t = p.trace + (long long)blockIdx.x * 6;
    t[0] = hw_id; t[1] = xcc_id; t[2] = tr_start; t[3] = tr_loop; t[4] = tr_loop_end; t[5] = __builtin_readcyclecounter();
  }
#endif
}
#pragma clang diagnostic pop

#if MLP_TRACE
static unsigned long long* g_mlp_trace = nullptr;   // set by the harness before a launch
static int g_mlp_stagger = 0, g_mlp_first_round = 0;
#endif

static int launch_gemm(const MlpArgs& a_in, int amode, bool pool, hipStream_t st) {
  MlpArgs a = a_in;
#if MLP_TRACE
  a.trace = g_mlp_trace; a.stagger_cycles = g_mlp_stagger; a.first_round = g_mlp_first_round;
#endif
  a.wide_store = ((reinterpret_cast<uintptr_t>(a.C) & 15) == 0) && (a.ldc % 4 == 0);
  const long long tiles = ((a.P + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  if (tiles <= 0) return REGNET_OK;
  if (tiles >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  if (a.ksplit > 1 && (pool || tiles * a.ksplit >= (1ll << 31))) return REGNET_ERR_UNSUPPORTED;
  dim3 grid((unsigned)(a.ksplit > 1 ? tiles * a.ksplit : tiles)), block(MLP_THREADS);
  if (amode == 3 && pool) hipLaunchKernelGGL((mlp_gemm_kernel<3, true>), grid, block, 0, st, a);
  else if (amode == 3) hipLaunchKernelGGL((mlp_gemm_kernel<3, false>), grid, block, 0, st, a);
  else if (amode == 2 && pool) hipLaunchKernelGGL((mlp_gemm_kernel<2, true>), grid, block, 0, st, a);
  else if (amode == 2) hipLaunchKernelGGL((mlp_gemm_kernel<2, false>), grid, block, 0, st, a);
  else if (amode == 1 && pool) hipLaunchKernelGGL((mlp_gemm_kernel<1, true>), grid, block, 0, st, a);
  else if (amode == 1) hipLaunchKernelGGL((mlp_gemm_kernel<1, false>), grid, block, 0, st, a);
  else if (pool) hipLaunchKernelGGL((mlp_gemm_kernel<0, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((mlp_gemm_kernel<0, false>), grid, block, 0, st, a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#ifndef MLP_USE_GEMM2
#define MLP_USE_GEMM2 1   // 0: plain layers on mlp_gemm_kernel (the round-1 kernel; kept for the gather / pre-multiplied variants)
#endif
#define G2_CUS 256

// Tile choice (measured on the ScoreNet shapes, scripts/ablate/g2_bench.cpp): 256 x 128 x 8 waves (2 workgroups per CU)
// when that grid still fills the chip, 128 x 128 x 4 waves (3 per CU) for few rows or N <= 128.  The last partial
// round of tiles is cut into half-height slices (tail split) when it would occupy at most half the workgroup slots.
template <int TBM, int TBN, int WM, int WN, int STAGES, int OCC>
static int launch_gemm2_tile(G2Args g, bool pool, hipStream_t st) {
  g.tiles_m = (int)((g.P + TBM - 1) / TBM);
  g.tiles_n = (g.N + TBN - 1) / TBN;
  const long long tiles = (long long)g.tiles_m * g.tiles_n;
  if (tiles >= (1ll << 30)) return REGNET_ERR_UNSUPPORTED;
  const long long slots = (long long)G2_CUS * OCC;
  long long main_blocks = tiles, tail_tiles = 0;
  constexpr bool can_split = (TBM / WM / 32 == 2) && ((WM * 32 + TBN) / 16) % (WM * WN) == 0;
  if (!pool && can_split && tiles > slots) {
    const long long rem = tiles % slots;
    if (rem > 0 && rem * 2 <= slots) { main_blocks = tiles - rem; tail_tiles = rem; }
  }
  g.main_blocks = (int)main_blocks; g.tail_tiles = (int)tail_tiles; g.tail_split = 2;
  const dim3 grid((unsigned)(main_blocks + 2 * tail_tiles)), block(WM * WN * 64);
  if (pool) {
    if constexpr (TBM / WM == 64) hipLaunchKernelGGL((gemm2_kernel<TBM, TBN, WM, WN, STAGES, OCC, true>), grid, block, 0, st, g);
    else return REGNET_ERR_UNSUPPORTED;
  } else {
    hipLaunchKernelGGL((gemm2_kernel<TBM, TBN, WM, WN, STAGES, OCC, false>), grid, block, 0, st, g);
  }
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

static int launch_gemm2(const G2Args& g, bool pool, hipStream_t st) {
  // Tile choice, measured on the step's layer shapes BESIDE a level-1 sampling launch (one CU of every XCD held, the
  // condition these launches run in: scripts/bench_gemm2_layers.py with SIDE_BLOCKS=8, REGNET_G2_TILE=0/1/2): the hardware's
  // in-order workgroup rotation then loses a round to every tile that does not fit the first one, so a launch wants either
  // several rounds of its tile (>= 4 x the CUs) or the smallest tile.  (Alone on the chip the big tile wins from one round
  // up -- 149 vs 177 us at P = 8192, K = N = 1024 -- but beside the sampling it takes 277 vs 205 us.)
  static const int force_tile = getenv("REGNET_G2_TILE") ? atoi(getenv("REGNET_G2_TILE")) : -1;   // A/B measurements only
  if (force_tile == 0) return launch_gemm2_tile<256, 128, 4, 2, 3, 2>(g, pool, st);
  if (force_tile == 1) return launch_gemm2_tile<128, 128, 2, 2, 3, 3>(g, pool, st);
  if (force_tile == 2) return launch_gemm2_tile<64, 128, 1, 4, 3, 4>(g, pool, st);
  if (force_tile == 3 && !pool) return launch_gemm2_tile<128, 128, 4, 2, 3, 2>(g, pool, st);   // 8 waves of 32 x 64: slabs fit
  if (force_tile == 8 && !pool) return launch_gemm2_tile<128, 128, 2, 2, 3, 2>(g, pool, st);   // 4 waves of 64 x 64 at 2 workgroups per CU: slabs fit
  const long long nt = (g.N + 127) / 128;
  const long long t256 = ((g.P + 255) / 256) * nt, t128 = ((g.P + 127) / 128) * nt;
  if (g.N > 128 && t256 >= 4 * G2_CUS) return launch_gemm2_tile<256, 128, 4, 2, 3, 2>(g, pool, st);
  // two or more slabs of K and at least one 128 x 128 tile per CU: four waves of 64 x 64 at two workgroups per CU -- at two waves
  // per SIMD a wave has 256 registers, which hold the slab sums next to the 64 accumulator registers (180 VGPRs, no spill),
  // and a 64 x 64 wave tile reads half the LDS bytes per MFMA of the 64 x 32 one.  Same order of additions as the 64 x 128
  // tile: bit-identical.  Stand-alone on the step's shapes 4-7 % faster (profiles/r05_plain_layer_tiles.txt).
  static const bool t8 = !(getenv("REGNET_G2_T8") && getenv("REGNET_G2_T8")[0] == '0');                 // A/B measurements only
  if (t8 && !pool && G2_SLAB_KT > 0 && g.Kpad >= 2 * G2_SLAB_KT * G2_BK && t128 >= G2_CUS)
    return launch_gemm2_tile<128, 128, 2, 2, 3, 2>(g, pool, st);
  // (only the 64 x 128 tile has the registers for slab accumulation -- gemm2.h: G2Slab -- so a layer with two or more
  // slabs of K takes it even where the 128 x 128 tile would be ~6 % faster: P = 40 960, N = 512, K = 256 / 512 of the
  // ScoreNet forward, +18 us per step for 0.55e-5 of parity margin, profiles/r04_error_budget.txt)
  if (t128 >= 4 * G2_CUS && (G2_SLAB_KT == 0 || g.Kpad < 2 * G2_SLAB_KT * G2_BK || pool))
    return launch_gemm2_tile<128, 128, 2, 2, 3, 3>(g, pool, st);
  return launch_gemm2_tile<64, 128, 1, 4, 3, 4>(g, pool, st);
}

extern "C" int regnet_mlp_layer_f32(const float* A, int64_t lda, int64_t Ka, const float* W, int64_t Kpad,
                                    const float* scale, const float* shift, float* C, int64_t ldc, int64_t P,
                                    int64_t N, int relu, int pool_group, void* stream) {
  if (P < 0 || N <= 0 || Ka < 0 || Kpad <= 0 || Kpad % BK || Ka > Kpad || (Ka & 3) || (lda & 3)) return REGNET_ERR_SHAPE;
  if (pool_group != 0 && (pool_group != 64 || P % 64)) return REGNET_ERR_UNSUPPORTED;
  if (P == 0) return REGNET_OK;
  if (!A || !W || !scale || !shift || !C) return REGNET_ERR_NULL;
  if (!aligned16(A) || !aligned16(W)) return REGNET_ERR_SHAPE;
  static const bool use_gemm2 = MLP_USE_GEMM2 && !(getenv("REGNET_GEMM2") && getenv("REGNET_GEMM2")[0] == '0');   // debugging switch
  if (use_gemm2 && Kpad >= 2 * G2_BK) {   // LDS-DMA ring kernel (gemm2.h); needs two k-tiles for its prologue
    G2Args g = {};
    g.A = A; g.lda = lda; g.Ka = (int)Ka; g.W = W; g.Kpad = (int)Kpad; g.scale = scale; g.shift = shift;
    g.C = C; g.ldc = ldc; g.P = P; g.N = (int)N; g.relu = relu;
    return launch_gemm2(g, pool_group != 0, as_stream(stream));
  }
  MlpArgs a = {};
  a.A = A; a.lda = lda; a.Ka = (int)Ka;
  a.W = W; a.Kpad = (int)Kpad; a.scale = scale; a.shift = shift;
  a.C = C; a.ldc = ldc; a.P = P; a.N = (int)N; a.relu = relu; a.group = 64; a.rows_per_scene = 1;
  return launch_gemm(a, 0, pool_group != 0, as_stream(stream));
}

// out[p][n] = relu(scale[n] * (sum_s part[s][p][n]) + shift[n]); slices added in index order (deterministic).
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int S, long long P, int N,
                                                            long long ldp, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu,
                                                            float* __restrict__ C, long long ldc) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one output element per thread
  if (i >= P * N) return;
  const long long row = i / N;
  const int col = (int)(i - row * N);
  float acc = part[row * ldp + col];
  for (int s = 1; s < S; ++s) acc += part[(long long)s * P * ldp + row * ldp + col];
  float y = acc * scale[col] + shift[col];
  if (relu) y = fmaxf(y, 0.f);
  C[row * ldc + col] = y;
}

extern "C" int64_t regnet_mlp_splitk_workspace_bytes(int64_t P, int64_t N, int64_t ksplit) {
  return ksplit > 1 ? ksplit * P * ((N + 3) / 4 * 4) * (int64_t)sizeof(float) : 0;
}

extern "C" int regnet_mlp_layer_splitk_f32(const float* A, int64_t lda, int64_t Ka, const float* W, int64_t Kpad,
                                           const float* scale, const float* shift, float* C, int64_t ldc, int64_t P,
                                           int64_t N, int relu, int64_t ksplit, void* workspace, void* stream) {
  if (P < 0 || N <= 0 || Ka < 0 || Kpad <= 0 || Kpad % BK || Ka > Kpad || (Ka & 3) || (lda & 3) || ksplit < 1)
    return REGNET_ERR_SHAPE;
  if (ksplit > Kpad / BK || P * N >= (1ll << 40)) return REGNET_ERR_UNSUPPORTED;
  if (P == 0) return REGNET_OK;
  if (!A || !W || !scale || !shift || !C || (ksplit > 1 && !workspace)) return REGNET_ERR_NULL;
  if (!aligned16(A) || !aligned16(W) || (ksplit > 1 && !aligned16(workspace))) return REGNET_ERR_SHAPE;
  MlpArgs a = {};
  a.A = A; a.lda = lda; a.Ka = (int)Ka;
  a.W = W; a.Kpad = (int)Kpad; a.scale = scale; a.shift = shift;
  a.P = P; a.N = (int)N; a.relu = relu; a.group = 64; a.rows_per_scene = 1;
  if (ksplit == 1) {
    a.C = C; a.ldc = ldc;
    return launch_gemm(a, 0, false, as_stream(stream));
  }
  const long long ldp = (N + 3) / 4 * 4;
  a.C = (float*)workspace; a.ldc = ldp; a.ksplit = (int)ksplit;
  int rc = launch_gemm(a, 0, false, as_stream(stream));
  if (rc) return rc;
  const long long blocks = (P * N + 255) / 256;
  if (blocks >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     (const float*)workspace, (int)ksplit, (long long)P, (int)N, ldp, scale, shift, relu, C,
                     (long long)ldc);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_sa_layer1_f32(const float* feat, int64_t fb, int64_t fn, int64_t fc, int64_t Cf,
                                    const float* xyz, int64_t xb, int64_t xc, int64_t xn, const int64_t* nbr,
                                    const int64_t* ctr, int64_t B, int64_t M, int64_t group, const float* W,
                                    int64_t Kpad, const float* scale, const float* shift, float* C, int64_t ldc,
                                    int64_t N, int relu, void* stream) {
  if (B < 0 || M < 0 || group <= 0 || N <= 0 || Cf < 0 || Kpad <= 0 || Kpad % BK || Cf + 3 > Kpad) return REGNET_ERR_SHAPE;
  const long long P = B * M * group;
  if (P == 0) return REGNET_OK;
  if (!xyz || !nbr || !ctr || !W || !scale || !shift || !C || (Cf > 0 && !feat)) return REGNET_ERR_NULL;
  if (!aligned16(W)) return REGNET_ERR_SHAPE;
  MlpArgs a = {};
  a.feat = Cf > 0 ? feat : nullptr; a.fb = fb; a.fn = fn; a.fc = fc; a.Cf = (int)Cf;
  a.feat_vec = Cf > 0 && fc == 1 && (Cf & 3) == 0 && aligned16(feat) && !(fb & 3) && !(fn & 3);
  a.xyz = xyz; a.xb = xb; a.xc = xc; a.xn = xn; a.nbr = (const long long*)nbr; a.ctr = (const long long*)ctr;
  a.group = (int)group; a.rows_per_scene = M * group;
  a.W = W; a.Kpad = (int)Kpad; a.scale = scale; a.shift = shift;
  a.C = C; a.ldc = ldc; a.P = P; a.N = (int)N; a.relu = relu;
  return launch_gemm(a, 1, false, as_stream(stream));
}

extern "C" int regnet_sa_layer12_f32(const float* feat, int64_t fb, int64_t fn, int64_t fc, int64_t Cf,
                                     const float* xyz, int64_t xb, int64_t xc, int64_t xn, const int64_t* nbr,
                                     const int64_t* ctr, int64_t B, int64_t M, int64_t group, const float* W1,
                                     const float* scale1, const float* shift1, int64_t C1, const float* W,
                                     int64_t Kpad, const float* scale, const float* shift, float* C, int64_t ldc,
                                     int64_t N, int relu, int pool_group, void* stream) {
  if (B < 0 || M < 0 || group <= 0 || N <= 0 || Cf < 0 || Cf + 3 > 8 || C1 <= 0 || C1 > 256 || Kpad != C1 || Kpad % BK)
    return REGNET_ERR_SHAPE;
  if (pool_group != 0 && (pool_group != 64 || group != 64)) return REGNET_ERR_UNSUPPORTED;
  const long long P = B * M * group;
  if (P == 0) return REGNET_OK;
  if (!xyz || !nbr || !ctr || !W1 || !scale1 || !shift1 || !W || !scale || !shift || !C || (Cf > 0 && !feat))
    return REGNET_ERR_NULL;
  if (!aligned16(W) || !aligned16(W1)) return REGNET_ERR_SHAPE;
  MlpArgs a = {};
  a.feat = Cf > 0 ? feat : nullptr; a.fb = fb; a.fn = fn; a.fc = fc; a.Cf = (int)Cf;
  a.xyz = xyz; a.xb = xb; a.xc = xc; a.xn = xn; a.nbr = (const long long*)nbr; a.ctr = (const long long*)ctr;
  a.group = (int)group; a.rows_per_scene = M * group;
  a.W1 = W1; a.scale1 = scale1; a.shift1 = shift1;
  a.W = W; a.Kpad = (int)Kpad; a.scale = scale; a.shift = shift;
  a.C = C; a.ldc = ldc; a.P = P; a.N = (int)N; a.relu = relu;
  return launch_gemm(a, 2, pool_group != 0, as_stream(stream));
}

// Layer 2 of a set-abstraction block over pre-multiplied layer-1 rows (AMODE 3, see MlpArgs::V):
//   U (B*Nsrc, ldu) = scale1 * W1 [f | x] per source point, V (B*M, ldv) = scale1 * W1x x_c - shift1 per centre,
//   A[p][k] = relu(U[b*Nsrc + nbr[p]][k] - V[p / group][k]) for k < C1;  C = epilogue(A . W^T).
extern "C" int regnet_sa_premul_layer_f32(const float* U, int64_t ldu, const float* V, int64_t ldv, int64_t C1,
                                          const int64_t* nbr, int64_t B, int64_t Nsrc, int64_t M, int64_t group,
                                          const float* W, int64_t Kpad, const float* scale, const float* shift,
                                          float* C, int64_t ldc, int64_t N, int relu, int pool_group, void* stream) {
  if (B < 0 || M < 0 || Nsrc <= 0 || group <= 0 || N <= 0 || C1 <= 0 || (C1 & 3) || C1 > Kpad || Kpad % BK || (ldu & 3) ||
      (ldv & 3) || ldu < C1 || ldv < C1)
    return REGNET_ERR_SHAPE;
  if (pool_group != 0 && (pool_group != 64 || group != 64)) return REGNET_ERR_UNSUPPORTED;
  const long long P = B * M * group;
  if (P == 0) return REGNET_OK;
  if (!U || !V || !nbr || !W || !scale || !shift || !C) return REGNET_ERR_NULL;
  if (!aligned16(U) || !aligned16(V) || !aligned16(W)) return REGNET_ERR_SHAPE;
  MlpArgs a = {};
  a.feat = U; a.fb = Nsrc * ldu; a.fn = ldu; a.fc = 1; a.Ka = (int)C1;
  a.V = V; a.ldv = ldv; a.nbr = (const long long*)nbr;
  a.group = (int)group; a.rows_per_scene = M * group;
  a.W = W; a.Kpad = (int)Kpad; a.scale = scale; a.shift = shift;
  a.C = C; a.ldc = ldc; a.P = P; a.N = (int)N; a.relu = relu;
  return launch_gemm(a, 3, pool_group != 0, as_stream(stream));
}

// ---------------------------------------------------------------------------------------
// 3-NN interpolation + skip concat, channels-last:  out[p] = [sum_k w_k * sparse[idx_k] | dense[p] | 0]
// Weights from SQUARED distances: inv = 1/max(d2, eps), w = inv / sum(inv)  (modules.py:117-122).
// One workgroup handles ROWS rows; a thread walks channels so every gathered row is read coalesced.
#define IC_ROWS 4
__global__ __launch_bounds__(256) void interp_concat_kernel(
    const float* __restrict__ sparse, long long sb, long long sn, int Cs, const long long* __restrict__ idx,
    const float* __restrict__ dist2, float eps, const float* __restrict__ dense, long long db, long long dn,
    long long dc, int Cd, long long Nd, float* __restrict__ out, long long ldo, int Cout, long long P) {
  const long long row_base = (long long)blockIdx.x * IC_ROWS;
  for (int rr = 0; rr < IC_ROWS; ++rr) {
    const long long p = row_base + rr;
    if (p >= P) return;
    const long long b = p / Nd, n = p - b * Nd;
    const long long j0 = idx[p * 3], j1 = idx[p * 3 + 1], j2 = idx[p * 3 + 2];
    const float i0 = 1.0f / fmaxf(dist2[p * 3], eps), i1 = 1.0f / fmaxf(dist2[p * 3 + 1], eps),
                i2 = 1.0f / fmaxf(dist2[p * 3 + 2], eps);
    const float norm = (i0 + i1) + i2;
    const float w0 = i0 / norm, w1 = i1 / norm, w2 = i2 / norm;
    const float* s0 = sparse + b * sb + j0 * sn;
    const float* s1 = sparse + b * sb + j1 * sn;
    const float* s2 = sparse + b * sb + j2 * sn;
    float* o = out + p * ldo;
    for (int c = threadIdx.x; c < Cout; c += 256) {
      float v = 0.f;
      if (c < Cs) {
        float acc = 0.f;  // k order, as interpolate_kernel.cu:165-170
        acc += s0[c] * w0;
        acc += s1[c] * w1;
        acc += s2[c] * w2;
        v = acc;
      } else if (c < Cs + Cd) {
        v = dense[b * db + n * dn + (long long)(c - Cs) * dc];
      }
      o[c] = v;
    }
  }
}

extern "C" int regnet_interp_concat_f32(const float* sparse, int64_t sb, int64_t sn, int64_t Cs, const int64_t* idx,
                                        const float* dist2, float eps, const float* dense, int64_t db, int64_t dn,
                                        int64_t dc, int64_t Cd, int64_t B, int64_t Nd, float* out, int64_t ldo,
                                        int64_t Cout, void* stream) {
  if (B < 0 || Nd < 0 || Cs < 0 || Cd < 0 || Cout < Cs + Cd || ldo < Cout) return REGNET_ERR_SHAPE;
  const long long P = B * Nd;
  if (P == 0 || Cout == 0) return REGNET_OK;
  if (!sparse || !idx || !dist2 || !out || (Cd > 0 && !dense)) return REGNET_ERR_NULL;
  const long long blocks = (P + IC_ROWS - 1) / IC_ROWS;
  if (blocks >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(interp_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), sparse,
                     (long long)sb, (long long)sn, (int)Cs, (const long long*)idx, dist2, eps, dense, (long long)db,
                     (long long)dn, (long long)dc, (int)Cd, (long long)Nd, out, (long long)ldo, (int)Cout, P);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// ---------------------------------------------------------------------------------------
// First layer of a feature-propagation block evaluated BEFORE the interpolation.  The layer is linear in
// its input [sum_k w_k sparse[idx_k] | dense]  (pn2_utils/modules.py:117-131), so with Ys = Ws . sparse
// (one row per SPARSE point) and Yd = Wd . dense (one row per dense point, optional):
//   out[p][n] = relu(scale[n] * (sum_k w_k Ys[b, idx_k][n] + Yd[p][n]) + shift[n])
// (a skip input of <= 4 channels, e.g. rgb, is multiplied right here instead: + Wd4[n][0:4] . dense[p])
// -- the wide GEMM runs over the few sparse rows instead of the many dense ones.  Weights as above
// (from squared distances, k order of the reference's interpolation kernel).  Threads walk channels in
// float4 steps so every gathered row is one coalesced read.
__global__ __launch_bounds__(256) void interp_affine_kernel(
    const float* __restrict__ ys, long long sb, long long sn, const long long* __restrict__ idx,
    const float* __restrict__ dist2, float eps, const float* __restrict__ yd, long long ldd,
    const float* __restrict__ dsm, long long db, long long dn, long long dc, int Cdsm, const float* __restrict__ wd4,
    const float* __restrict__ scale, const float* __restrict__ shift, int relu, long long Nd, int C,
    float* __restrict__ out, long long ldo, long long P) {
  const int lanes_per_row = C / 4;                       // C % 4 == 0, <= 256
  const int rows_per_pass = 256 / lanes_per_row;         // launcher guarantees lanes_per_row divides 256
  const int r_in = threadIdx.x / lanes_per_row, c4 = threadIdx.x % lanes_per_row;
  const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * c4);
  const float4 sh = *reinterpret_cast<const float4*>(shift + 4 * c4);
  float4 wq[4];   // narrow skip input (<= 4 channels, e.g. rgb): its weight columns for this thread's 4 outputs
#pragma unroll
  for (int q = 0; q < 4; ++q) wq[q] = dsm ? *reinterpret_cast<const float4*>(wd4 + (4 * c4 + q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const long long row_base = (long long)blockIdx.x * (IC_ROWS * rows_per_pass);
#pragma unroll
  for (int rr = 0; rr < IC_ROWS; ++rr) {
    const long long p = row_base + (long long)rr * rows_per_pass + r_in;
    if (p >= P) continue;
    const long long b = p / Nd;
    const long long j0 = idx[p * 3], j1 = idx[p * 3 + 1], j2 = idx[p * 3 + 2];
    const float i0 = 1.0f / fmaxf(dist2[p * 3], eps), i1 = 1.0f / fmaxf(dist2[p * 3 + 1], eps),
                i2 = 1.0f / fmaxf(dist2[p * 3 + 2], eps);
    const float norm = (i0 + i1) + i2;
    const float w0 = i0 / norm, w1 = i1 / norm, w2 = i2 / norm;
    const float4 a0 = *reinterpret_cast<const float4*>(ys + b * sb + j0 * sn + 4 * c4);
    const float4 a1 = *reinterpret_cast<const float4*>(ys + b * sb + j1 * sn + 4 * c4);
    const float4 a2 = *reinterpret_cast<const float4*>(ys + b * sb + j2 * sn + 4 * c4);
    float4 v;
    v.x = (a0.x * w0 + a1.x * w1) + a2.x * w2;
    v.y = (a0.y * w0 + a1.y * w1) + a2.y * w2;
    v.z = (a0.z * w0 + a1.z * w1) + a2.z * w2;
    v.w = (a0.w * w0 + a1.w * w1) + a2.w * w2;
    if (yd) {
      const float4 d = *reinterpret_cast<const float4*>(yd + p * ldd + 4 * c4);
      v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
    }
    if (dsm) {
      const float* dp = dsm + b * db + (p - b * Nd) * dn;
      float d[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) d[c] = c < Cdsm ? dp[(long long)min(c, Cdsm - 1) * dc] : 0.f;
      v.x += ((wq[0].x * d[0] + wq[0].y * d[1]) + wq[0].z * d[2]) + wq[0].w * d[3];
      v.y += ((wq[1].x * d[0] + wq[1].y * d[1]) + wq[1].z * d[2]) + wq[1].w * d[3];
      v.z += ((wq[2].x * d[0] + wq[2].y * d[1]) + wq[2].z * d[2]) + wq[2].w * d[3];
      v.w += ((wq[3].x * d[0] + wq[3].y * d[1]) + wq[3].z * d[2]) + wq[3].w * d[3];
    }
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(out + p * ldo + 4 * c4) = v;
  }
}

extern "C" int regnet_interp_affine_f32(const float* ys, int64_t sb, int64_t sn, const int64_t* idx,
                                        const float* dist2, float eps, const float* yd, int64_t ldd,
                                        const float* dense_small, int64_t db, int64_t dn, int64_t dc,
                                        int64_t Cd_small, const float* Wd4, const float* scale, const float* shift,
                                        int relu, int64_t B, int64_t Nd, int64_t C, float* out, int64_t ldo,
                                        void* stream) {
  if (B < 0 || Nd < 0 || C <= 0 || (C & 3) || ldo < C || (ldo & 3) || (sn & 3) || (sb & 3) || (yd && (ldd < C || (ldd & 3))) ||
      (dense_small && (Cd_small < 1 || Cd_small > 4)))
    return REGNET_ERR_SHAPE;
  // a row is walked by C/4 lanes; the block holds a whole number of rows
  if (C / 4 > 256 || 256 % (C / 4) != 0) return REGNET_ERR_UNSUPPORTED;
  const long long P = B * Nd;
  if (P == 0) return REGNET_OK;
  if (!ys || !idx || !dist2 || !scale || !shift || !out || (dense_small && !Wd4)) return REGNET_ERR_NULL;
  if (!aligned16(ys) || !aligned16(out) || !aligned16(scale) || !aligned16(shift) || (yd && !aligned16(yd)) ||
      (dense_small && !aligned16(Wd4)))
    return REGNET_ERR_SHAPE;
  const long long rows_per_block = (long long)IC_ROWS * (256 / (C / 4));
  const long long blocks = (P + rows_per_block - 1) / rows_per_block;
  if (blocks >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(interp_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), ys, (long long)sb,
                     (long long)sn, (const long long*)idx, dist2, eps, yd, (long long)ldd, dense_small, (long long)db,
                     (long long)dn, (long long)dc, (int)Cd_small, Wd4, scale, shift, relu, (long long)Nd, (int)C, out,
                     (long long)ldo, P);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// ---------------------------------------------------------------------------------------
// Score head: score[p] = sigmoid(bn(dot(x[p], w) + bias))  (pointnet2.py:116-119), x channels-last.
// One wave per point: 64 lanes stride the channels, DPP-free shuffle reduction.
__global__ __launch_bounds__(256) void score_head_kernel(const float* __restrict__ x, long long ldx, int C,
                                                        const float* __restrict__ w, float bias, float bn_scale,
                                                        float bn_shift, float* __restrict__ score, long long P) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const int lane = threadIdx.x & 63;
  const float* row = x + p * ldx;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) acc += row[c] * w[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    const float v = (acc + bias) * bn_scale + bn_shift;
    score[p] = 1.0f / (1.0f + expf(-v));
  }
}

extern "C" int regnet_score_head_f32(const float* x, int64_t ldx, int64_t C, const float* w, float bias,
                                     float bn_scale, float bn_shift, float* score, int64_t P, void* stream) {
  if (P < 0 || C <= 0 || ldx < C) return REGNET_ERR_SHAPE;
  if (P == 0) return REGNET_OK;
  if (!x || !w || !score) return REGNET_ERR_NULL;
  const long long blocks = (P + 3) / 4;
  if (blocks >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(score_head_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, (long long)ldx,
                     (int)C, w, bias, bn_scale, bn_shift, score, (long long)P);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
