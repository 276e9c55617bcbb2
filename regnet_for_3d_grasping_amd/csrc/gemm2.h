// gemm2.h -- second-generation shared-MLP contraction for gfx950: LDS-DMA operand ring + counted waits.
//
//   C[P x N] = epilogue(X[P x K] . W[N x K]^T),   epilogue = relu(scale[n] * c + shift[n])  (+ max over 64 rows)
//
// Same maths, operand packing and MFMA fragment scheme as mlp.hip (v_mfma_f32_32x32x2_f32, both operands K-contiguous,
// one ds_read_b128 feeds 4 MFMAs); what changes is how the operands reach the LDS and how workgroups are scheduled:
//
//  * global -> LDS by `global_load_lds_dwordx4` (LDS-DMA): no staging VGPRs, no ds_write pass, and -- the point -- the
//    loads of k-tile t+2 stay in flight ACROSS the barrier of k-tile t (ring of STAGES buffers, `s_waitcnt vmcnt(N)`
//    with N > 0 in the steady state).  hipcc does not count LDS-DMA issued from inline asm, so every wait on it is
//    written by hand here (one statement = wait + s_barrier with a "memory" clobber; the ds_reads are ordinary C++
//    loads which the compiler may not move across it).
//  * an LDS-DMA writes 64 lanes x 16 B to CONSECUTIVE LDS addresses, so the image of a tile is dense ([row][16 floats],
//    64-byte rows, no padding) and the bank spreading is done by permuting which 16-byte chunk of its row a lane
//    FETCHES: physical chunk c' of row r holds logical chunk c' ^ ((r >> 2) & 3).  A fragment read of 16-lane service
//    group {fr..} then touches 16 distinct 16-byte slots of the 256-byte bank row (derivation in DESIGN.md §5.3).
//  * workgroup tile TBM x TBN with WM x WN waves; a wave owns (TBM/WM) x (TBN/WN) as 32x32 accumulators.
//  * slab accumulation (SLAB_KT > 0): the MFMAs accumulate SLAB_KT k-tiles (128 columns of K) into `acc`, which is then
//    added to a second register set and cleared -- K = 1024 becomes eight 128-long fp32 chains added once each instead
//    of one 1024-long chain (what a cache-blocked CPU sgemm does too: the reference's own arithmetic).  Measured on the
//    S8 scores: profiles/r04_error_budget.txt.  One v_add per accumulator register per 128 MFMAs.
//  * tail splitting: the last partial round of tiles (fewer tiles than the chip has workgroup slots) is cut into
//    narrower-M sub-tiles inside the SAME launch, so a launch does not end with most CUs idle for a whole tile time.
#pragma once
#include "common.h"

typedef float g2_f32x16 __attribute__((ext_vector_type(16)));

#define G2_BK 16

struct G2Args {
  const float* A; long long lda; int Ka;   // plain rows; columns >= Ka read as zero (W is zero-padded there)
  const float* W; int Kpad;                // packed [Npad][Kpad], Npad % TBN == 0, zero padded
  const float* scale; const float* shift;
  float* C; long long ldc;
  long long P; int N; int relu;
  int tiles_m, tiles_n;                    // full-size tile grid
  // tail split: blocks >= main_blocks work on the last `tail_tiles` tiles, each cut into `tail_split` row slices
  int main_blocks, tail_tiles, tail_split;
};

// One LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_dst + 16 * lane].
__device__ __forceinline__ void g2_glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int N> __device__ __forceinline__ void g2_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(N) : "memory");
}

__device__ __forceinline__ long long g2_xcd_linear(unsigned vblock, long long total) {
  // block b runs on XCD b % 8 (speed-only observation): give every XCD a CONTIGUOUS run of the tile list, so that
  // consecutive blocks on one XCD walk the N-tiles of one M-tile (the A tile is fetched into one L2, not eight)
  const long long q = total / 8, r = total % 8;
  const long long xcd = vblock % 8, s = vblock / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
}

// Rows [row0, row0 + ROWS) x columns [col0, col0 + TBN) with WM x WN waves: the k-loop + epilogue of one (sub-)tile.
//   ROWS = WM * TM * 32.  smem: ring of STAGES x (ROWS_MAX + TBN) x 64 bytes.
template <int ROWS_MAX, int TBN, int WM, int WN, int TM, int STAGES, bool POOL, int SLAB_KT>
__device__ __forceinline__ void g2_tile(const G2Args& p, float* smem, long long row0, int col0) {
  constexpr int NW = WM * WN;
  constexpr int ROWS = WM * TM * 32;
  constexpr int TN = TBN / WN / 32;
  constexpr int STAGE_FLOATS = (ROWS_MAX + TBN) * G2_BK;
  constexpr int GROUPS = (ROWS + TBN) / 16;           // 1 KiB pieces (16 rows x 64 B) of one k-tile
  constexpr int L = (GROUPS + NW - 1) / NW;          // LDS-DMA instructions per wave per k-tile
  static_assert(GROUPS % NW == 0, "every wave issues the same number of LDS-DMA pieces");
  constexpr int D = STAGES - 1;                      // k-tiles in flight ahead of the one being multiplied
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 31, fh = lane >> 5;

  // ---- per-lane source pointers of this wave's pieces (k = 0), advanced by 16 floats per k-tile
  const float* gp[L];
  unsigned dst[L];                                   // byte offset of the piece inside a stage
  bool isA[L];
  const int lrow = lane >> 2, lchunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
  for (int j = 0; j < L; ++j) {
    const int grp = wave + j * NW;                   // wave-uniform
    const int cr = grp * 16 + lrow;                  // combined row: [0, ROWS) = A rows, [ROWS, ROWS + TBN) = W rows
    isA[j] = grp * 16 < ROWS;
    if (isA[j]) {
      long long r = row0 + cr;
      if (r >= p.P) r = 0;                           // rows past the end: valid memory, never stored
      gp[j] = p.A + r * p.lda + 4 * lchunk;
    } else {
      gp[j] = p.W + (long long)(col0 + cr - ROWS) * p.Kpad + 4 * lchunk;
    }
    dst[j] = (unsigned)(grp * 16 * G2_BK * 4);
  }
  const unsigned smem_base = (unsigned)(uintptr_t)smem;   // LDS byte address of the ring (shared aperture: low 32 bits)
  const bool ragged_k = p.Ka < p.Kpad;

#define G2_ISSUE(KT_, STAGE_)                                                                                   \
  _Pragma("unroll") for (int j = 0; j < L; ++j) {                                                                \
    const float* src = gp[j] + (long long)(KT_) * G2_BK;                                                         \
    if (ragged_k && isA[j] && (KT_) * G2_BK + 4 * lchunk >= p.Ka) src = gp[j] - 4 * lchunk; /* zero columns */   \
    g2_glds16(src, smem_base + (unsigned)((STAGE_) * STAGE_FLOATS * 4) + dst[j]);                                \
  }

  g2_f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // slab accumulation: the sum of the slabs finished so far (see the header); `since` counts the open slab's k-tiles
  g2_f32x16 tot[SLAB_KT > 0 ? TM : 1][SLAB_KT > 0 ? TN : 1];
  if constexpr (SLAB_KT > 0) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[mi][ni][r] = 0.f;
  }
  int since = 0;
#define G2_SLAB()                                                                                                 \
  if constexpr (SLAB_KT > 0) {                                                                                    \
    if (++since == SLAB_KT) {                                                                                     \
      since = 0;                                                                                                  \
      _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                                           \
        _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)                                                         \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                        \
            tot[mi][ni][r] += acc[mi][ni][r];                                                                     \
            acc[mi][ni][r] = 0.f;                                                                                 \
          }                                                                                                       \
    }                                                                                                             \
  }

  // fragment offsets (floats) inside a stage: row * 16 + 4 * ((2 kk + fh) ^ ((fr >> 2) & 3))
  const int sw = (fr >> 2) & 3;
  int offA[2], offW[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    offA[kk] = (wm * TM * 32 + fr) * G2_BK + 4 * ((2 * kk + fh) ^ sw);
    offW[kk] = (ROWS + wn * TN * 32 + fr) * G2_BK + 4 * ((2 * kk + fh) ^ sw);
  }

  const int KT = p.Kpad / G2_BK;
#ifndef G2_PINNED
#define G2_PINNED 1   // 1: both k-halves' fragment reads are issued in front of the k-tile's MFMAs (scheduling barrier);
                      // 0: hipcc's order (it sinks the second half's reads to mid-way)
#endif
#if G2_PINNED
#define G2_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define G2_PIN()
#endif
#define G2_COMPUTE(STAGE_)                                                                                        \
  {                                                                                                               \
    const float* st = smem + (STAGE_) * STAGE_FLOATS;                                                             \
    float4 a[2][TM], b[2][TN];   /* all fragments of the k-tile first: the second half's LDS latency hides */     \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                            \
      _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                                           \
        a[kk][mi] = *reinterpret_cast<const float4*>(st + offA[kk] + mi * 32 * G2_BK);                            \
      _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)                                                           \
        b[kk][ni] = *reinterpret_cast<const float4*>(st + offW[kk] + ni * 32 * G2_BK);                            \
    }                                                                                                             \
    G2_PIN();                                                                                                     \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
      _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                                           \
        _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) {                                                       \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][mi].x, b[kk][ni].x, acc[mi][ni], 0, 0, 0);     \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][mi].y, b[kk][ni].y, acc[mi][ni], 0, 0, 0);     \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][mi].z, b[kk][ni].z, acc[mi][ni], 0, 0, 0);     \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][mi].w, b[kk][ni].w, acc[mi][ni], 0, 0, 0);     \
        }                                                                                                         \
  }
  // prologue: D k-tiles in flight (KT >= D is guaranteed by the launcher: Kpad >= 16 * D)
#pragma unroll
  for (int d = 0; d < D; ++d) { G2_ISSUE(d, d) }
  int stage = 0;
  // steady state: tile kt has landed (mine: counted wait; everybody's: barrier), nobody reads stage (kt - 1) % STAGES
  // any more, so tile kt + D may be fetched into it
  for (int kt = 0; kt < KT - D; ++kt) {
    g2_wait_barrier<(D - 1) * L>();
    int ns = stage + D; if (ns >= STAGES) ns -= STAGES;
    G2_ISSUE(kt + D, ns)
    G2_COMPUTE(stage)
    G2_SLAB()
    if (++stage == STAGES) stage = 0;
  }
  // drain: the last D tiles, nothing left to fetch
#pragma unroll
  for (int d = 0; d < D; ++d) {
    g2_wait_barrier<0>();
    G2_COMPUTE(stage)
    G2_SLAB()
    if (++stage == STAGES) stage = 0;
  }
  if constexpr (SLAB_KT > 0) {   // the open slab (all zeros when K is a multiple of the slab) joins the finished ones
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] += tot[mi][ni][r];
  }
#undef G2_SLAB
#undef G2_COMPUTE
#undef G2_ISSUE
  // ---- epilogue.  C/D layout: element r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = col0 + (wn * TN + ni) * 32 + fr;
    const bool col_ok = col < p.N;
    const float s = col_ok ? p.scale[col] : 0.f, t = col_ok ? p.shift[col] : 0.f;
    if (POOL) {
      static_assert(!POOL || TM == 2, "pooling: a wave's 64 rows are one neighbourhood");
      float m = -__builtin_inff();
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float y = acc[mi][ni][r] * s + t;
          if (p.relu) y = fmaxf(y, 0.f);
          m = fmaxf(m, y);
        }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const long long wrow = row0 + wm * 64;
      if (col_ok && fh == 0 && wrow < p.P) p.C[(wrow / 64) * p.ldc + col] = m;
    } else {
      const long long first_row = row0 + wm * TM * 32 + 4 * fh;
      float* cp = p.C + first_row * p.ldc + col;
      const long long ld = p.ldc;
      const bool interior = (row0 + ROWS <= p.P) && (col0 + TBN <= p.N);
      if (interior) {   // unguarded: the stores of an accumulator issue back to back
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float y = acc[mi][ni][r] * s + t;
            if (p.relu) y = fmaxf(y, 0.f);
            *cp = y;
            cp += ((r & 3) == 3) ? 5 * ld : ld;
          }
      } else {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float y = acc[mi][ni][r] * s + t;
            if (p.relu) y = fmaxf(y, 0.f);
            if (col_ok && first_row + mi * 32 + (r & 3) + 8 * (r >> 2) < p.P) *cp = y;
            cp += ((r & 3) == 3) ? 5 * ld : ld;
          }
      }
    }
  }
}

#ifndef G2_SLAB_KT
#define G2_SLAB_KT 8      // k-tiles (of 16 columns) per accumulation slab; 0 = one chain over all of K (rounds 2-3)
#endif
// slabs where the second accumulator set fits the register budget of the tile's occupancy: wave tiles of 32 accumulator
// registers (64 x 32: the 64 x 128 x 4-wave tile, 74 -> 104 of 128 VGPRs at 4 waves per SIMD; 32 x 64: the 128 x 128 x 8-wave
// tile), and wave tiles of 64 (64 x 64) when the launch asks for at most TWO waves per SIMD -- 256 registers per wave: the
// 128 x 128 x 4-wave tile at two workgroups per CU, 180 VGPRs, no spill (round 5).  At 3-4 waves per SIMD the 64 x 64 wave
// tiles (128 x 128 x 4 waves at 3 workgroups per CU: 118 -> 168 + 51 spilled; 256 x 128 x 8 waves: 116 of 128) keep one chain.
template <int TBM, int TBN, int WM, int WN, int WG_PER_CU> struct G2Slab {
  static constexpr int acc_tiles = (TBM / WM / 32) * (TBN / WN / 32);
  static constexpr int kt = (acc_tiles <= 2 || (acc_tiles <= 4 && WG_PER_CU * WM * WN <= 8)) ? G2_SLAB_KT : 0;
};

template <int TBM, int TBN, int WM, int WN, int STAGES, int WG_PER_CU, bool POOL>
__global__ __launch_bounds__(WM * WN * 64, WG_PER_CU * WM * WN / 4)
void gemm2_kernel(const G2Args p) {
  constexpr int SLAB = G2Slab<TBM, TBN, WM, WN, WG_PER_CU>::kt;
  constexpr int TM = TBM / WM / 32;
  __shared__ __attribute__((aligned(1024))) float smem[STAGES * (TBM + TBN) * G2_BK];
  const int bid = blockIdx.x;
  if (bid < p.main_blocks) {
    const long long t = g2_xcd_linear((unsigned)bid, p.main_blocks);
    const int tm = (int)(t / p.tiles_n), tn = (int)(t % p.tiles_n);
    g2_tile<TBM, TBN, WM, WN, TM, STAGES, POOL, SLAB>(p, smem, (long long)tm * TBM, tn * TBN);
  } else if constexpr (!POOL && TM == 2 && ((WM * 32 + TBN) / 16) % (WM * WN) == 0) {
    // tail: tile index = main_blocks + t (plain order), slice s of tail_split: half-height sub-tiles (TM = 1)
    const int t = (bid - p.main_blocks) / p.tail_split, s = (bid - p.main_blocks) % p.tail_split;
    const long long tile = (long long)p.main_blocks + t;
    const int tm = (int)(tile / p.tiles_n), tn = (int)(tile % p.tiles_n);
    g2_tile<TBM, TBN, WM, WN, 1, STAGES, false, SLAB>(p, smem, (long long)tm * TBM + (long long)s * (TBM / 2), tn * TBN);
  }
}
