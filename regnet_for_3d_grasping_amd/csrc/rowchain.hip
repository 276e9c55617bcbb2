// rowchain.hip -- chains of shared-MLP layers over the SAME rows in one kernel (gfx950): a wave keeps 16 points'
// activations in registers from the first layer to the last, only the weights move (L2 -> LDS ring -> MFMA operand).
//
// Reference behaviour restated (paths relative to /root/reference/multi_model/utils):
//   FP3 tail + segmentation head of PointNet2Seg (pointnet2.py:64-84, :116-119; pn2_utils/modules.py:500-509):
//     h1 (P x 256: first FP3 layer, produced by interp_affine) -> conv 256->256 -> conv 256->256 = the network's
//     256-channel point feature F (returned, and read by the region stage) -> SharedMLP 256->512->256->256->128
//     -> conv_score 128->1 + bn_score + sigmoid.        Every conv is bias-free 1x1 + eval BatchNorm (folded to a
//     per-channel affine) + ReLU (pn2_utils/nn/modules/mlp.py:55-114); dropout is identity in eval mode.
//   The layer-by-layer path (mlp.hip / gemm2.h) writes and re-reads every (204 800 x {256, 512}) activation of a
//   batch: 3.1 GB of HBM traffic per step for 0.2 TFLOP.  Here: h1 is read once, F and the scores are written once.
//
// Scheme (v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] . B[4x16]; A: lane l -> A[i = l & 15][k = l >> 4],
// B: lane l -> B[k = l >> 4][j = l & 15], D: register r of lane l = D[i = 4 (l >> 4) + r][j = l & 15]):
//   products are formed channel-major, D = W . X^T (rows = output channels, columns = the wave's 16 points).  Register r
//   of output tile ot then holds, in lane (g = l >> 4, j = l & 15), channel 16 ot + 4 g + r of point j -- which is
//   exactly a B operand of the next layer for the k-step "channels {16 kt + 4 g + r : g = 0..3}" (kt = ot).  The
//   matching A operand is component r of the 16-byte chunk W[out][16 kt + 4 g ..]: one ds_read_b128 feeds 4 MFMAs.
//   So a layer is   acc[ot] += mfma(Wfrag(ot, kt).r, xin[kt].r)   over kt, r, and BN + ReLU in place turn acc[] into
//   the next layer's xin[] -- no activation ever leaves the register file.
//   A wave owns 16 points: widths up to 512 -> (256 + 512) / 4 = 192 VGPRs for (input + output), 2 waves per SIMD.
//   A workgroup = 8 waves = 128 rows; its waves share the weight stream: 32 KiB stages ([32 output channels][256 k],
//   dense rows, 16-byte chunks XOR-swizzled by the row so the fragment reads are bank-conflict free) fetched by
//   LDS-DMA into a ring of 3, two stages in flight across every barrier (counted vmcnt, written by hand).  The stream
//   is laid out on the host in consumption order, already swizzled: the fetch is a linear copy.
//   One resident workgroup per CU draws 128-row blocks from a ticket counter (a block = one pass over the stream), so a
//   CU that another stream's kernel keeps busy simply takes fewer blocks.
#include "common.h"

typedef float rc_f32x4 __attribute__((ext_vector_type(4)));

#ifndef RC_WAVES
#define RC_WAVES 8                             // waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU)
#endif
#define RC_WG_PER_CU (8 / RC_WAVES)
#define RC_THREADS (RC_WAVES * 64)
#define RC_STAGE_FLOATS (32 * 256)             // 32 KiB
#ifndef RC_STAGES
#define RC_STAGES (RC_WAVES == 8 ? 3 : 2)      // ring depth (two workgroups per CU: 2 x 2 x 32 KiB)
#endif
#define RC_PIECES (32 / RC_WAVES)              // 1 KiB LDS-DMA pieces per wave per stage
#ifndef RC_TAIL_HALF
#define RC_TAIL_HALF 0                         // 1: the last 128 blocks of a launch are handed out as half blocks (waves 0-3;
                                               // measured SLOWER inside the pipeline: 1.74-1.83 vs 1.65-1.67 ms, A/B on one box)
#endif
#ifndef RC_SPREAD_FETCH
#define RC_SPREAD_FETCH 0                      // 1: a stage's LDS-DMA pieces are issued between its MFMA steps (measured 2 %
                                               // SLOWER per kernel, same step time); 0: all right behind the barrier
#endif
#ifndef RC_MIDSYNC
#define RC_MIDSYNC (RC_WAVES == 8 && !RC_SPREAD_FETCH)   // 1: fp_head_chain / sa_premul_chain meet in the MIDDLE of a stage (ring of 4)
#endif
#define RC_MID_STAGES (RC_MIDSYNC ? 4 : RC_STAGES)
#ifndef RC_SYNC_STEP
#define RC_SYNC_STEP 7                         // the step (0 .. 15) of a stage behind which its barrier + fetch sit
#endif
#define RC_AFFINE_MAX 4096                     // floats of folded BN affine kept in LDS
#define RC_INTERP_FLOATS 1536                  // + the interpolating prologue's tables: Wd4 (256 x 4) | scale1 (256) | shift1 (256)

struct RcArgs {
  const float* X;  long long ldx;              // (P, 256) first-layer activation (channels-last), 16-byte aligned rows
  float* F;        long long ldf;              // (P, 256) point feature out
  float* score;                                // (P)
  long long P;
  const float* stream;                         // weight stream: n_stages x 32 KiB, consumption order, swizzled
  int n_stages;                                // stages per pass over a row block (the chain's total)
  const float* affine;                         // per layer [scale(N) | shift(N)], concatenated in layer order
  int affine_floats;
  const float* wscore;                         // conv_score weight (128)
  float score_bias, score_bn_scale, score_bn_shift;
  int* ticket;                                 // work queue head (zeroed by the caller before the launch)
  long long n_blocks;                          // tickets to hand out: n_full whole blocks (8 waves) + half blocks (waves 0-3)
  long long n_full;                            // tickets < n_full are whole blocks; ticket t >= n_full: half block t - n_full
  long long blk_first, blk_count;              // this LAUNCH hands out blocks [blk_first, blk_first + blk_count) (a launch may be
                                               // split: the last partial round of blocks on a side stream, fused.py)
  // ---- interpolating prologue (fp_head_chain_kernel<true>): the first FP layer is evaluated right here, X is not read
  const float* ys; long long ys_sb, ys_sn;     // (B, Ns, 256) sparse rows already multiplied by the layer's W[:, :Cs]
  const long long* idx; const float* dist2;    // (P, 3) three nearest sparse points and their SQUARED distances
  float eps; long long Nd;                     // weights 1 / max(d2, eps), normalised (modules.py:117-122); points per scene
  const float* dsm; long long db, dn, dc; int Cdsm;   // narrow skip input (B, Cd <= 4, Nd), element strides (rgb)
  const float* tables;                         // Wd4 (256 x 4) | scale1 (256) | shift1 (256)
};

__device__ __forceinline__ void rc_glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void rc_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(N) : "memory");
}
// A workgroup barrier that orders LDS traffic only (__syncthreads() also waits for vmcnt(0): inside a pass that DRAINS the
// ring's two or three stages in flight -- a load latency of stall per use; it was used twice per 24-stage block).
__device__ __forceinline__ void rc_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The weight-stream ring.  Stage s of the (periodic) stream lives in ring slot s % STAGES.
//
// Two hand-over schemes (round 5):
//  MID = false (3 slots; rounds 2-4, still sa_premul_chain_kernel): the barrier sits at the
//    stage BOUNDARY -- acquire() waits for the next stage, everybody meets, the stage behind the ones in flight is fetched into
//    the slot consumed last.  Every stage then starts with the matrix pipe empty: barrier, four LDS-DMA issues with their
//    address arithmetic, the first fragment's LDS round trip -- on both waves of a SIMD at once, because the barrier released
//    them together (measured: 14 % of fp_head_chain_kernel's shader cycles per block are not MFMA issue, ~1 300 per stage).
//  MID = true (4 slots; fp_head_chain_kernel, sa3_premul_chain_kernel): the barrier sits in the MIDDLE of a stage (step
//    RC_SYNC_STEP of 16).  At that point every wave is inside stage s, so the slot of stage s - 1 is free: the fetch of stage
//    s + 3 goes there; and the counted wait in front of the barrier makes stage s + 1 readable for everybody.  The stage
//    boundary itself needs nothing: a wave runs from the last MFMA of stage s into the first fragment reads of stage s + 1
//    without meeting anyone, and the DMA issue + barrier happen between two MFMA steps whose operands are already in registers.
template <int STAGES, bool MID>
struct RcRing {
  static constexpr bool mid = MID;
  const float* src;        // this lane's source of piece 0 of the next stage to fetch
  const float* src_begin;  // ... of stream stage 0
  int fetch_idx;           // stream index (0 .. n_stages) of the next stage to fetch
  int n_stages;
  unsigned lds_fetch;      // LDS byte address of this wave's piece 0 in the slot the next fetch goes to
  unsigned lds_lo, lds_hi; // ... in slot 0 / one past the last slot
  int slot;                // ring slot of the stage to be consumed next

  __device__ __forceinline__ void init(const float* stream, int n, float* smem, int wave, int lane) {
    n_stages = n;
    src_begin = stream + (wave * RC_PIECES) * 256 + lane * 4;
    src = src_begin;
    fetch_idx = 0;
    lds_lo = (unsigned)(uintptr_t)smem + (unsigned)(wave * RC_PIECES * 1024);
    lds_hi = lds_lo + STAGES * RC_STAGE_FLOATS * 4;
    lds_fetch = lds_lo;
    slot = 0;
  }
  __device__ __forceinline__ void fetch_piece(int j) {   // j = 0 .. RC_PIECES - 1, in order; the last one advances the ring
    rc_glds16(src + j * 256, lds_fetch + j * 1024);
    if (j == RC_PIECES - 1) {
      src += RC_STAGE_FLOATS;
      if (++fetch_idx == n_stages) { fetch_idx = 0; src = src_begin; }
      lds_fetch += RC_STAGE_FLOATS * 4;
      if (lds_fetch == lds_hi) lds_fetch = lds_lo;
    }
  }
  __device__ __forceinline__ void fetch() {
#pragma unroll
    for (int j = 0; j < RC_PIECES; ++j) fetch_piece(j);
  }
  // Start of the ring: STAGES - 1 stages in flight; MID: stage 0 must be readable before its first fragment read (later
  // stages become readable at the barrier in the middle of the stage before them).
  __device__ __forceinline__ void prime() {
#pragma unroll
    for (int d = 0; d < STAGES - 1; ++d) fetch();
    if (MID) rc_wait_barrier<(STAGES - 2) * RC_PIECES>();
  }
  // Make the next stage readable (mine: counted wait; everybody's: barrier -- which also says that nobody reads the
  // slot consumed last any more, so the stage behind the ones in flight is fetched into it: right here, or -- with
  // RC_SPREAD_FETCH -- one LDS-DMA piece at a time between the stage's MFMA steps (RC_FETCH_AT)).  MID: nothing to wait for.
  // The counted wait stays correct with other vector memory operations outstanding (activation loads, feature
  // stores, issued after the newest fetch): loads return in order among loads, so "at most RC_PIECES operations
  // outstanding" implies that at most the RC_PIECES newest LOADS are -- every piece of the stage wanted here is older
  // than those; extra operations only make the wait longer.  DRAIN (vmcnt(0)) is kept as a debugging switch.
  template <bool DRAIN> __device__ __forceinline__ int acquire() {
    if (!MID) {
      if (DRAIN) rc_wait_barrier<0>();
      else rc_wait_barrier<(STAGES - 2) * RC_PIECES>();
#if !RC_SPREAD_FETCH
      fetch();
#endif
    }
    const int s = slot;
    slot = (slot + 1 == STAGES) ? 0 : slot + 1;
    return s;
  }
  // MID, called behind step `step` of the 16 steps of EVERY stage: in flight are the stages s + 1 .. s + STAGES - 2; all but
  // the newest STAGES - 3 of them must have landed (s + 1 is consumed next, without another meeting).
  __device__ __forceinline__ void at_step(int step) {
    if (MID && step == RC_SYNC_STEP) {
      rc_wait_barrier<(STAGES - 3) * RC_PIECES>();
      fetch();
    }
  }
};

typedef RcRing<RC_MID_STAGES, (RC_MIDSYNC != 0)> RcRingMid;    // fp_head_chain_kernel, sa3_premul_chain_kernel
typedef RcRing<RC_STAGES, false> RcRingEdge;                    // sa_premul_chain_kernel (and RC_MIDSYNC=0 builds)
#define RC_SA3_POOL_FLOATS (2 * RC_WAVES * 256)

// One LDS-DMA piece of the next fetch behind step STEP_ of a 16-step stage (pieces spread evenly over the steps).
#if RC_SPREAD_FETCH
#define RC_FETCH_AT(RING_, STEP_)                                                              \
  if (((STEP_) + 1) % (16 / RC_PIECES) == 0) (RING_).fetch_piece(((STEP_) + 1) / (16 / RC_PIECES) - 1);
#else
#define RC_FETCH_AT(RING_, STEP_) (RING_).at_step(STEP_);
#endif

// Per-lane fragment offsets (floats) inside a stage.  A-stages are [32 rows][256 k], B-stages [64 rows][128 k]; in both
// the logical 16-byte chunk 4 kt + g of row (16 t + j) is stored at physical chunk (4 kt + g) ^ j =
// 16 (kt >> 2) + ((4 (kt & 3)) ^ (g ^ j)): the 16 lanes of a ds_read_b128 service group then hit 16 distinct 16-byte
// slots of the 256-byte bank row (the row strides, 1024 and 512 bytes, are multiples of it).
struct RcFrag { int a[4], b[4]; };
__device__ __forceinline__ RcFrag rc_frag_offsets() {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  RcFrag f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f.a[q] = j * 256 + 4 * ((4 * q) ^ (g ^ j));
    f.b[q] = j * 128 + 4 * ((4 * q) ^ (g ^ j));
  }
  return f;
}

// One step = 8 MFMAs on two accumulators, alternating (v_mfma_f32_16x16x4_f32 issues every 32 cycles but its result is
// ready after 40: back-to-back MFMAs on ONE accumulator would each wait).  The order is pinned with scheduling
// barriers: left alone, hipcc clusters the four MFMAs of each accumulator and sinks the fragment reads to just in
// front of their first use, so the two waves of a SIMD -- released by the same barrier, running the same code -- sit
// out every LDS round trip together (67 TFLOP/s).  The stage loops below are software-pipelined at source level: the
// reads of step k + 1 are issued in front of the MFMAs of step k.
#define RC_PIN() __builtin_amdgcn_sched_barrier(0)
#define RC_MFMA8(W0, W1, X, A0, A1)                                        \
  A0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W0.x, X.x, A0, 0, 0, 0);       \
  A1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W1.x, X.x, A1, 0, 0, 0);       \
  RC_PIN();                                                                \
  A0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W0.y, X.y, A0, 0, 0, 0);       \
  A1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W1.y, X.y, A1, 0, 0, 0);       \
  RC_PIN();                                                                \
  A0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W0.z, X.z, A0, 0, 0, 0);       \
  A1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W1.z, X.z, A1, 0, 0, 0);       \
  RC_PIN();                                                                \
  A0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W0.w, X.w, A0, 0, 0, 0);       \
  A1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W1.w, X.w, A1, 0, 0, 0);       \
  RC_PIN();

// The same step with the roles swapped: the activation registers as the A operand (rows = points), the weight fragment as
// the B operand (columns = output channels) -> D[point 4 g + r][channel l & 15].
#define RC_MFMA8_T(W0, W1, X, A0, A1)                                      \
  A0 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.x, W0.x, A0, 0, 0, 0);       \
  A1 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.x, W1.x, A1, 0, 0, 0);       \
  RC_PIN();                                                                \
  A0 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.y, W0.y, A0, 0, 0, 0);       \
  A1 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.y, W1.y, A1, 0, 0, 0);       \
  RC_PIN();                                                                \
  A0 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.z, W0.z, A0, 0, 0, 0);       \
  A1 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.z, W1.z, A1, 0, 0, 0);       \
  RC_PIN();                                                                \
  A0 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.w, W0.w, A0, 0, 0, 0);       \
  A1 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.w, W1.w, A1, 0, 0, 0);       \
  RC_PIN();

// Two consecutive layers  A: 256 -> M  and  B: M -> N  as one loop over the 128-channel groups of M:
//     mid = actA(affA(W_A[group rows] . xin))          4 A-stages ([32 rows][256 k]),  32 registers
//     acc += W_B[:, group columns] . mid               N/64 B-stages ([64 rows][128 k])
// and xout = actB(affB(acc)) at the end.  Nothing in here is indexed by a loop counter except memory: xin, mid and
// the N/16 accumulators are compile-time register arrays (a register array written at a run-time index would be
// spilled), and the loop body -- 8 or 6 stages, 128 MFMAs each -- is small enough for the whole chain to stay in the
// instruction cache.  Stream order: for each group, its A-stages then its B-stages.
template <int M, int N, bool RELU_B, class Ring>
__device__ __forceinline__ void rc_pair(const rc_f32x4 (&xin)[16], rc_f32x4 (&xout)[N / 16], Ring& ring,
                                        const float* __restrict__ smem, const float* __restrict__ affA,
                                        const float* __restrict__ affB, bool active, const RcFrag& fo) {
  static_assert(M % 128 == 0 && N % 64 == 0, "groups of 128 mid channels; B-stages of 64 output channels");
  const int g4 = ((threadIdx.x & 63) >> 4) * 4;
#pragma unroll
  for (int ot = 0; ot < N / 16; ++ot) xout[ot] = rc_f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ob = 0; ob < M / 128; ++ob) {
    rc_f32x4 mid[8];
    // ---- layer A, channels [128 ob, 128 ob + 128)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rc_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const int slot = ring.template acquire<false>();
      {   // (waves without rows in the last pass compute on row 0's data: no branch around the MFMA stream)
        const float* st = smem + slot * RC_STAGE_FLOATS;
        rc_f32x4 w0n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0]);
        rc_f32x4 w1n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0] + 16 * 256);
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
          const rc_f32x4 w0 = w0n, w1 = w1n;
          if (kt + 1 < 16) {
            const float* wp = st + fo.a[(kt + 1) & 3] + 64 * ((kt + 1) >> 2);
            w0n = *reinterpret_cast<const rc_f32x4*>(wp);
            w1n = *reinterpret_cast<const rc_f32x4*>(wp + 16 * 256);
          }
          RC_PIN();
          const rc_f32x4 x = xin[kt];
          RC_MFMA8(w0, w1, x, acc0, acc1)
          RC_FETCH_AT(ring, kt)
        }
      }
      const float* a = affA + 128 * ob + 32 * u + g4;   // register r of tile t = channel 16 t + 4 g + r
      const rc_f32x4 s0 = *reinterpret_cast<const rc_f32x4*>(a), s1 = *reinterpret_cast<const rc_f32x4*>(a + 16);
      const rc_f32x4 t0 = *reinterpret_cast<const rc_f32x4*>(a + M), t1 = *reinterpret_cast<const rc_f32x4*>(a + M + 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0[r] = fmaxf(acc0[r] * s0[r] + t0[r], 0.f);
        acc1[r] = fmaxf(acc1[r] * s1[r] + t1[r], 0.f);
      }
      mid[2 * u] = acc0; mid[2 * u + 1] = acc1;
    }
    // ---- layer B, k-slice [128 ob, 128 ob + 128) for all N outputs
#pragma unroll
    for (int v = 0; v < N / 64; ++v) {
      const int slot = ring.template acquire<false>();
      {   // (waves without rows in the last pass compute on row 0's data: no branch around the MFMA stream)
        const float* st = smem + slot * RC_STAGE_FLOATS;
        rc_f32x4 w0n = *reinterpret_cast<const rc_f32x4*>(st + fo.b[0]);
        rc_f32x4 w1n = *reinterpret_cast<const rc_f32x4*>(st + fo.b[0] + 16 * 128);
#pragma unroll
        for (int step = 0; step < 16; ++step) {   // tile pair tp = step / 8 (tiles 4 v + 2 tp, + 1), k-step kt = step % 8
          constexpr int dummy = 0; (void)dummy;
          const int tp = step >> 3, kt = step & 7;
          const rc_f32x4 w0 = w0n, w1 = w1n;
          if (step + 1 < 16) {
            const int tpn = (step + 1) >> 3, ktn = (step + 1) & 7;
            const float* wp = st + (2 * tpn) * 16 * 128 + fo.b[ktn & 3] + 64 * (ktn >> 2);
            w0n = *reinterpret_cast<const rc_f32x4*>(wp);
            w1n = *reinterpret_cast<const rc_f32x4*>(wp + 16 * 128);
          }
          RC_PIN();
          const rc_f32x4 x = mid[kt];
          RC_MFMA8(w0, w1, x, xout[4 * v + 2 * tp], xout[4 * v + 2 * tp + 1])
          RC_FETCH_AT(ring, step)
        }
      }
    }
  }
#pragma unroll
  for (int ot = 0; ot < N / 16; ++ot) {
    const float* a = affB + 16 * ot + g4;
    const rc_f32x4 s = *reinterpret_cast<const rc_f32x4*>(a), t = *reinterpret_cast<const rc_f32x4*>(a + N);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y = xout[ot][r] * s[r] + t[r];
      xout[ot][r] = RELU_B ? fmaxf(y, 0.f) : y;
    }
  }
}

// FP3 tail + segmentation head: 256 -> 256 -> [F] 256 -> 512 -> 256 -> 256 -> 128 -> score, as three layer pairs.
// Stages per pass: 2 x (4 + 4) + 4 x (4 + 4) + 2 x (4 + 2) = 60.  Affine table (floats): layer i at the sum of 2 N of
// the layers before it: 0, 512, 1024, 2048, 2560, 3072 (total 3328).
// INTERP: the block's first layer (pn2_utils/modules.py:104-131 + the first SharedMLP layer, evaluated on the sparse rows:
// fused._fp_first_layer) happens in the prologue -- x0 = relu(scale1 (sum_k w_k Ys[idx_k] + Wd4 . rgb) + shift1) -- instead
// of in interp_affine_kernel, whose (P x 256) output (210 MB per batch of 8 written, then read here) never exists.
template <bool INTERP>
__global__ __launch_bounds__(RC_THREADS, 2) void fp_head_chain_kernel(const RcArgs p) {
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // ring | affine | wscore | ticket | (interp tables)
  float* const aff = smem + RC_MID_STAGES * RC_STAGE_FLOATS;
  float* const wsc = aff + RC_AFFINE_MAX;
  float* const itab = wsc + 128 + 4;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < p.affine_floats; i += RC_THREADS) aff[i] = p.affine[i];
  for (int i = tid; i < 128; i += RC_THREADS) wsc[i] = p.wscore[i];
  if (INTERP) for (int i = tid; i < RC_INTERP_FLOATS; i += RC_THREADS) itab[i] = p.tables[i];
#ifndef RC_NO_TABLE_SYNC
  __syncthreads();   // the tables are read after the ring's barriers, which do not wait for LDS writes (lgkmcnt)
#endif

  RcRingMid ring;
  ring.init(p.stream, p.n_stages, smem, wave, lane);
  const RcFrag fo = rc_frag_offsets();

  // Row blocks (128 rows = one pass of the 8 waves over the whole weight stream) are handed out through a ticket
  // counter: workgroups that start late -- their CU was busy with another stream's kernel, e.g. a 10 ms furthest-
  // point-sampling workgroup -- simply take fewer.  (A static split over 256 workgroups ran 2.86 ms inside the
  // pipeline against 1.55 ms stand-alone: the workgroups whose CU was taken ran as a second round.)
  int* const s_blk = reinterpret_cast<int*>(wsc + 128);
  const long long units_total = (p.P + 15) / 16;
  bool primed = false;
  for (;;) {
    if (tid == 0) *s_blk = atomicAdd(p.ticket, 1);
    rc_barrier_lds();                          // (thread 0's wave waited for its atomic; nobody else drains anything)
    const long long tick = __builtin_amdgcn_readfirstlane(*s_blk);
    if (tick >= p.blk_count) break;
    const long long blk = p.blk_first + tick;
    if (!primed) {   // prologue of the ring: RC_STAGES - 1 stages in flight
      ring.prime();
      primed = true;
    }
    // whole block: 8 waves x 16 rows; half block (the last round's worth of rows, so that the launch's tail is made of
    // half-length passes): waves 0-3 only -- one wave per SIMD, which then has the matrix pipe to itself
    const bool half = blk >= p.n_full;
    const long long unit = half ? p.n_full * RC_WAVES + (blk - p.n_full) * (RC_WAVES / 2) + wave : blk * RC_WAVES + wave;
    const bool active = unit < units_total && !(half && wave >= RC_WAVES / 2);   // wave-uniform
    long long row = unit * 16 + j;
    const bool row_ok = active && row < p.P;
    if (!row_ok) row = 0;
    // ---- h1: x0[kt] = X[row][16 kt + 4 g ..]
    rc_f32x4 x0[16];
    if (!INTERP) {
      const float* xr = p.X + row * p.ldx + 4 * g;
#pragma unroll
      for (int kt = 0; kt < 16; ++kt) x0[kt] = *reinterpret_cast<const rc_f32x4*>(xr + 16 * kt);
    } else {
      const long long b = row / p.Nd;
      const long long j0 = p.idx[row * 3], j1 = p.idx[row * 3 + 1], j2 = p.idx[row * 3 + 2];
      const float i0 = 1.0f / fmaxf(p.dist2[row * 3], p.eps), i1 = 1.0f / fmaxf(p.dist2[row * 3 + 1], p.eps),
                  i2 = 1.0f / fmaxf(p.dist2[row * 3 + 2], p.eps);
      const float norm = (i0 + i1) + i2;
      const float w0 = i0 / norm, w1 = i1 / norm, w2 = i2 / norm;
      float d[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.dsm) {
        const float* dp = p.dsm + b * p.db + (row - b * p.Nd) * p.dn;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = c < p.Cdsm ? dp[(long long)(c < p.Cdsm ? c : 0) * p.dc] : 0.f;
      }
      const float* r0 = p.ys + b * p.ys_sb + j0 * p.ys_sn + 4 * g;
      const float* r1 = p.ys + b * p.ys_sb + j1 * p.ys_sn + 4 * g;
      const float* r2 = p.ys + b * p.ys_sb + j2 * p.ys_sn + 4 * g;
#pragma unroll
      for (int kq = 0; kq < 8; ++kq) {     // two k-tiles' gathers (six 16-byte loads) in flight at a time
        rc_f32x4 a0[2], a1[2], a2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a0[t] = *reinterpret_cast<const rc_f32x4*>(r0 + 16 * (2 * kq + t));
          a1[t] = *reinterpret_cast<const rc_f32x4*>(r1 + 16 * (2 * kq + t));
          a2[t] = *reinterpret_cast<const rc_f32x4*>(r2 + 16 * (2 * kq + t));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int c0 = 16 * (2 * kq + t) + 4 * g;          // this lane's four channels of the k-tile
          const rc_f32x4 sc = *reinterpret_cast<const rc_f32x4*>(itab + 1024 + c0);
          const rc_f32x4 sh = *reinterpret_cast<const rc_f32x4*>(itab + 1280 + c0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const rc_f32x4 wq = *reinterpret_cast<const rc_f32x4*>(itab + 4 * (c0 + r));
            float v = (a0[t][r] * w0 + a1[t][r] * w1) + a2[t][r] * w2;
            v += ((wq.x * d[0] + wq.y * d[1]) + wq.z * d[2]) + wq.w * d[3];
            x0[2 * kq + t][r] = fmaxf(v * sc[r] + sh[r], 0.f);
          }
        }
        RC_PIN();
      }
    }
    rc_f32x4 x2[16];
    rc_pair<256, 256, true>(x0, x2, ring, smem, aff + 0, aff + 512, active, fo);
    // ---- F out
    if (row_ok) {
      float* fr = p.F + row * p.ldf + 4 * g;
#pragma unroll
      for (int ot = 0; ot < 16; ++ot) *reinterpret_cast<rc_f32x4*>(fr + 16 * ot) = x2[ot];
    }
    rc_f32x4 x4[16];
    rc_pair<512, 256, true>(x2, x4, ring, smem, aff + 1024, aff + 2048, active, fo);
    rc_f32x4 x6[8];
    rc_pair<256, 128, true>(x4, x6, ring, smem, aff + 2560, aff + 3072, active, fo);
    // ---- conv_score + bn_score + sigmoid
    float acc = 0.f;
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
      const rc_f32x4 w = *reinterpret_cast<const rc_f32x4*>(wsc + 16 * ot + 4 * g);
      acc += x6[ot].x * w.x; acc += x6[ot].y * w.y; acc += x6[ot].z * w.z; acc += x6[ot].w * w.w;
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (row_ok && g == 0) {
      const float v = (acc + p.score_bias) * p.score_bn_scale + p.score_bn_shift;
      p.score[row] = 1.0f / (1.0f + expf(-v));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's look-ahead fetches must land before the LDS is released
}

// ---------------------------------------------------------------------------------------------------------------------
// Level-2 set-abstraction block of PointNet2Seg on pre-multiplied layer-1 rows (pointnet2.py:40-42: 259 -> 256 -> 256
// -> 512 over 1024 x 64 rows per scene; pn2_utils/modules.py:39-56, :244-245), layers 2 + 3 + the max over the 64
// neighbours in one kernel:
//     x0[p] = relu(U[b, nbr[p]] - V[p / 64])        (layer 1, evaluated per source point / per centre: fused.sa_features)
//     x1 = relu(aff2(W2 . x0))   256 -> 256         channel-major, as in rc_pair's layer A
//     y  = relu(aff3(x1 . W3^T)) 256 -> 512         POINT-major: x1's registers are just as well the A operand
//                                                   (lane -> point l & 15, k -> channel 4 g + r), the weight fragment the B
//                                                   operand; D[point 4 g + r][channel j'] -- the max over the points is then
//                                                   a max over the 4 registers + two lane-group exchanges, no 16-lane reduction
//     out[p / 64] = max over the neighbourhood's 4 waves (LDS)
// The 537 MB (batch of 8) layer-2 activation the two-launch path writes and re-reads never exists.  24 stages per pass,
// all [32 output channels][256 k]; a block = 128 rows = 2 neighbourhoods.
struct ScArgs {
  const float* U; long long ldu, scene_stride;   // U row of source point j of scene b: U + b * scene_stride + j * ldu
  const float* V; long long ldv;                 // V row of neighbourhood g
  const long long* nbr;                          // (groups, 64)
  long long groups, groups_per_scene;
  const float* stream; int n_stages;
  const float* affine; int affine_floats;        // [scale2(256) | shift2(256) | scale3(512) | shift3(512)]
  int relu3;
  float* out; long long ldo;                     // (groups, 512)
  int* ticket; long long n_blocks, n_full;     // as RcArgs: whole blocks first, then half blocks
};

__global__ __launch_bounds__(RC_THREADS, 2) void sa_premul_chain_kernel(const ScArgs p) {
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // ring | affine (1536) | pool (8 x 512) | ticket
  float* const aff = smem + RC_STAGES * RC_STAGE_FLOATS;
  float* const pool = aff + 1536;
  int* const s_blk = reinterpret_cast<int*>(pool + RC_WAVES * 512);
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < p.affine_floats; i += RC_THREADS) aff[i] = p.affine[i];
  __syncthreads();

  // (the round-2 hand-over, three slots: this kernel's 174 VGPRs leave room for other streams' waves on its CUs -- the region
  // stage's and the geometry's small kernels -- as long as its LDS does too: with a fourth slot (150 KB) they lost that place,
  // their launches took 3 x longer and the step gained nothing from this kernel's 1.5 %)
  RcRingEdge ring;
  ring.init(p.stream, p.n_stages, smem, wave, lane);
  const RcFrag fo = rc_frag_offsets();
  const int g4 = 4 * g;

  bool primed = false;
  for (;;) {
    if (tid == 0) *s_blk = atomicAdd(p.ticket, 1);
    rc_barrier_lds();                          // (thread 0's wave waited for its atomic; nobody else drains anything)
    const long long blk = __builtin_amdgcn_readfirstlane(*s_blk);
    if (blk >= p.n_blocks) break;
    if (!primed) {
      ring.prime();
      primed = true;
    }
    const bool half = blk >= p.n_full;                         // half block: one neighbourhood on waves 0-3
    const long long grp0 = half ? p.n_full * (RC_WAVES / 4) + (blk - p.n_full) * (RC_WAVES / 8) : blk * (RC_WAVES / 4);
    const long long grp = grp0 + (wave >> 2);                  // this wave's neighbourhood
    const bool active = grp < p.groups && !(half && wave >= RC_WAVES / 2);   // wave-uniform
    const long long gs = active ? grp : 0;
    // ---- layer-1 rows of this wave's 16 points
    rc_f32x4 x0[16];
    {
      const long long b = gs / p.groups_per_scene;
      const long long src = p.nbr[gs * 64 + (wave & 3) * 16 + j];
      const float* ur = p.U + b * p.scene_stride + src * p.ldu + g4;
      const float* vr = p.V + gs * p.ldv + g4;
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {     // four loads of each kind in flight at a time (all 32 at once spill)
        rc_f32x4 u[4], v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          u[t] = *reinterpret_cast<const rc_f32x4*>(ur + 16 * (4 * kq + t));
          v[t] = *reinterpret_cast<const rc_f32x4*>(vr + 16 * (4 * kq + t));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) x0[4 * kq + t][r] = fmaxf(u[t][r] - v[t][r], 0.f);
        RC_PIN();
      }
    }
    // ---- layer 2: 8 stages, fully unrolled (x1 is a compile-time register array)
    rc_f32x4 x1[16];
#pragma unroll
    for (int st8 = 0; st8 < 8; ++st8) {
      rc_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const int slot = ring.template acquire<false>();
      {   // (waves without rows in the last pass compute on row 0's data: no branch around the MFMA stream)
        const float* st = smem + slot * RC_STAGE_FLOATS;
        rc_f32x4 w0n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0]);
        rc_f32x4 w1n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0] + 16 * 256);
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
          const rc_f32x4 w0 = w0n, w1 = w1n;
          if (kt + 1 < 16) {
            const float* wp = st + fo.a[(kt + 1) & 3] + 64 * ((kt + 1) >> 2);
            w0n = *reinterpret_cast<const rc_f32x4*>(wp);
            w1n = *reinterpret_cast<const rc_f32x4*>(wp + 16 * 256);
          }
          RC_PIN();
          const rc_f32x4 x = x0[kt];
          RC_MFMA8(w0, w1, x, acc0, acc1)
          RC_FETCH_AT(ring, kt)
        }
      }
      const float* a = aff + 32 * st8 + g4;
      const rc_f32x4 s0 = *reinterpret_cast<const rc_f32x4*>(a), s1 = *reinterpret_cast<const rc_f32x4*>(a + 16);
      const rc_f32x4 t0 = *reinterpret_cast<const rc_f32x4*>(a + 256), t1 = *reinterpret_cast<const rc_f32x4*>(a + 256 + 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0[r] = fmaxf(acc0[r] * s0[r] + t0[r], 0.f);
        acc1[r] = fmaxf(acc1[r] * s1[r] + t1[r], 0.f);
      }
      x1[2 * st8] = acc0; x1[2 * st8 + 1] = acc1;
    }
    // ---- layer 3 + max over the points, one stage (32 channels) at a time
    for (int s3 = 0; s3 < 16; ++s3) {
      rc_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const int slot = ring.template acquire<false>();
      {   // (waves without rows in the last pass compute on row 0's data: no branch around the MFMA stream)
        const float* st = smem + slot * RC_STAGE_FLOATS;
        rc_f32x4 w0n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0]);
        rc_f32x4 w1n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0] + 16 * 256);
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
          const rc_f32x4 w0 = w0n, w1 = w1n;
          if (kt + 1 < 16) {
            const float* wp = st + fo.a[(kt + 1) & 3] + 64 * ((kt + 1) >> 2);
            w0n = *reinterpret_cast<const rc_f32x4*>(wp);
            w1n = *reinterpret_cast<const rc_f32x4*>(wp + 16 * 256);
          }
          RC_PIN();
          const rc_f32x4 x = x1[kt];
          RC_MFMA8_T(w0, w1, x, acc0, acc1)
          RC_FETCH_AT(ring, kt)
        }
        // folded BN affine (+ ReLU) per channel (32 s3 + 16 t + j), max over this wave's 16 points: registers, then lane groups
        const float sc0 = aff[512 + 32 * s3 + j], sc1 = aff[512 + 32 * s3 + 16 + j];
        const float sh0 = aff[1024 + 32 * s3 + j], sh1 = aff[1024 + 32 * s3 + 16 + j];
        float m0 = -__builtin_inff(), m1 = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          m0 = fmaxf(m0, acc0[r] * sc0 + sh0);
          m1 = fmaxf(m1, acc1[r] * sc1 + sh1);
        }
        if (p.relu3) { m0 = fmaxf(m0, 0.f); m1 = fmaxf(m1, 0.f); }
        m0 = fmaxf(m0, __shfl_xor(m0, 16, 64)); m1 = fmaxf(m1, __shfl_xor(m1, 16, 64));
        m0 = fmaxf(m0, __shfl_xor(m0, 32, 64)); m1 = fmaxf(m1, __shfl_xor(m1, 32, 64));
        if (g == 0) {
          pool[wave * 512 + 32 * s3 + j] = m0;
          pool[wave * 512 + 32 * s3 + 16 + j] = m1;
        }
      }
    }
    // ---- max over the 4 waves of each neighbourhood (the ring's next barriers separate this from the next pass's writes)
    rc_barrier_lds();
#pragma unroll
    for (int n = 0; n < RC_WAVES / 4; ++n) {
      const long long gn = grp0 + n;
      if (gn < p.groups && !(half && n >= RC_WAVES / 8))
        for (int c = tid; c < 512; c += RC_THREADS) {
          const float* q = pool + (4 * n) * 512 + c;
          p.out[gn * p.ldo + c] = fmaxf(fmaxf(q[0], q[512]), fmaxf(q[1024], q[1536]));
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// Level-3 set-abstraction block of PointNet2Seg on pre-multiplied layer-1 rows (pointnet2.py:40-42: 515 -> 512 -> 512 ->
// 1024 over 256 x 64 rows per scene; pn2_utils/modules.py:39-56, :244-245): layers 2 + 3 + the max over the 64 neighbours
// in one kernel, like sa_premul_chain_kernel but with 512-wide activations -- 128 registers per 16 points, so layer 1 and
// layer 2 cannot both be resident.  Layer 2 therefore runs as two K-halves:
//     for kh in {0, 1}:  x0h = relu(U[b, nbr[p]][256 kh ..] - V[p / 64][256 kh ..])       64 registers, re-gathered per half
//                        x1 += W2[:, 256 kh ..] . x0h                                      128 accumulator registers (all 512)
//     x1 = relu(aff2(x1));   y = relu(aff3(x1 . W3^T)) point-major, 32 channels per step, two K-half stages each;  max.
// The 268 MB (batch of 8) layer-2 activation that regnet_sa_premul_layer_f32 wrote and the pooling GEMM re-read never
// exists.  96 stages per pass ([32 output channels][256 k] each): W2 as (kh, 16 row blocks), W3 as (32 row blocks, kh).
struct Sc3Args {
  const float* U; long long ldu, scene_stride;
  const float* V; long long ldv;
  const long long* nbr;
  long long groups, groups_per_scene;
  const float* stream; int n_stages;
  const float* affine; int affine_floats;        // [scale2(512) | shift2(512) | scale3(1024) | shift3(1024)]
  int relu3;
  float* out; long long ldo;                     // (groups, 1024)
  int* ticket; long long n_blocks;
};

__global__ __launch_bounds__(RC_THREADS, 2) void sa3_premul_chain_kernel(const Sc3Args p) {
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // ring | affine (3072) | pool (2 x 8 x 256) | ticket
  float* const aff = smem + RC_MID_STAGES * RC_STAGE_FLOATS;
  float* const pool = aff + 3072;
  int* const s_blk = reinterpret_cast<int*>(pool + RC_SA3_POOL_FLOATS);
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < p.affine_floats; i += RC_THREADS) aff[i] = p.affine[i];
  __syncthreads();

  RcRingMid ring;
  ring.init(p.stream, p.n_stages, smem, wave, lane);
  const RcFrag fo = rc_frag_offsets();
  const int g4 = 4 * g;

  bool primed = false;
  for (;;) {
    if (tid == 0) *s_blk = atomicAdd(p.ticket, 1);
    rc_barrier_lds();                          // (thread 0's wave waited for its atomic; nobody else drains anything)
    const long long blk = __builtin_amdgcn_readfirstlane(*s_blk);
    if (blk >= p.n_blocks) break;
    if (!primed) {
      ring.prime();
      primed = true;
    }
    const long long grp0 = blk * (RC_WAVES / 4);
    const long long grp = grp0 + (wave >> 2);                  // this wave's neighbourhood
    const bool active = grp < p.groups;                        // wave-uniform
    const long long gs = active ? grp : 0;
    const long long b = gs / p.groups_per_scene;
    const long long src = p.nbr[gs * 64 + (wave & 3) * 16 + j];
    const float* const ur = p.U + b * p.scene_stride + src * p.ldu + g4;
    const float* const vr = p.V + gs * p.ldv + g4;

    // ---- layer 2 as two K-halves; x1[t] register r, lane (g, j) = channel 16 t + 4 g + r of point j
    rc_f32x4 x1[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) x1[t] = rc_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < 2; ++kh) {
      rc_f32x4 x0[16];
      {
        const float* urh = ur + 256 * kh;
        const float* vrh = vr + 256 * kh;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {     // four loads of each kind in flight at a time
          rc_f32x4 u[4], v[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            u[t] = *reinterpret_cast<const rc_f32x4*>(urh + 16 * (4 * kq + t));
            v[t] = *reinterpret_cast<const rc_f32x4*>(vrh + 16 * (4 * kq + t));
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) x0[4 * kq + t][r] = fmaxf(u[t][r] - v[t][r], 0.f);
          RC_PIN();
        }
      }
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int slot = ring.template acquire<false>();
        const float* st = smem + slot * RC_STAGE_FLOATS;
        rc_f32x4 w0n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0]);
        rc_f32x4 w1n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0] + 16 * 256);
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
          const rc_f32x4 w0 = w0n, w1 = w1n;
          if (kt + 1 < 16) {
            const float* wp = st + fo.a[(kt + 1) & 3] + 64 * ((kt + 1) >> 2);
            w0n = *reinterpret_cast<const rc_f32x4*>(wp);
            w1n = *reinterpret_cast<const rc_f32x4*>(wp + 16 * 256);
          }
          RC_PIN();
          const rc_f32x4 x = x0[kt];
          RC_MFMA8(w0, w1, x, x1[2 * rg], x1[2 * rg + 1])
          RC_FETCH_AT(ring, kt)
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const float* a = aff + 16 * t + g4;
      const rc_f32x4 sc = *reinterpret_cast<const rc_f32x4*>(a), sh = *reinterpret_cast<const rc_f32x4*>(a + 512);
#pragma unroll
      for (int r = 0; r < 4; ++r) x1[t][r] = fmaxf(x1[t][r] * sc[r] + sh[r], 0.f);
    }
    // ---- layer 3 + max over the points: 32 channels per step, the two K-halves as two stages
    for (int s3 = 0; s3 < 32; ++s3) {
      rc_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int slot = ring.template acquire<false>();
        const float* st = smem + slot * RC_STAGE_FLOATS;
        rc_f32x4 w0n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0]);
        rc_f32x4 w1n = *reinterpret_cast<const rc_f32x4*>(st + fo.a[0] + 16 * 256);
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
          const rc_f32x4 w0 = w0n, w1 = w1n;
          if (kt + 1 < 16) {
            const float* wp = st + fo.a[(kt + 1) & 3] + 64 * ((kt + 1) >> 2);
            w0n = *reinterpret_cast<const rc_f32x4*>(wp);
            w1n = *reinterpret_cast<const rc_f32x4*>(wp + 16 * 256);
          }
          RC_PIN();
          const rc_f32x4 x = x1[16 * kh + kt];
          RC_MFMA8_T(w0, w1, x, acc0, acc1)
          RC_FETCH_AT(ring, kt)
        }
      }
      // folded BN affine (+ ReLU) per channel (32 s3 + 16 t + j), max over this wave's 16 points: registers, then lane groups
      const float sc0 = aff[1024 + 32 * s3 + j], sc1 = aff[1024 + 32 * s3 + 16 + j];
      const float sh0 = aff[2048 + 32 * s3 + j], sh1 = aff[2048 + 32 * s3 + 16 + j];
      float m0 = -__builtin_inff(), m1 = -__builtin_inff();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        m0 = fmaxf(m0, acc0[r] * sc0 + sh0);
        m1 = fmaxf(m1, acc1[r] * sc1 + sh1);
      }
      if (p.relu3) { m0 = fmaxf(m0, 0.f); m1 = fmaxf(m1, 0.f); }
      m0 = fmaxf(m0, __shfl_xor(m0, 16, 64)); m1 = fmaxf(m1, __shfl_xor(m1, 16, 64));
      m0 = fmaxf(m0, __shfl_xor(m0, 32, 64)); m1 = fmaxf(m1, __shfl_xor(m1, 32, 64));
      // The per-wave maxima of 256 channels (8 steps) at a time, double-buffered: after every eighth step the four waves of a
      // neighbourhood meet (LDS-only barrier) and 512 threads write 2 x 256 pooled values; the buffer is written again two
      // chunks later, behind the next chunk's barrier.  (One 8 x 1024 buffer reduced at the end of the pass -- 32 KB -- left no
      // room for the ring's fourth slot.)
      const int chunk = s3 >> 3, cl = 32 * (s3 & 7);
      float* const pw = pool + ((chunk & 1) * RC_WAVES + wave) * 256 + cl;
      if (g == 0) {
        pw[j] = m0;
        pw[16 + j] = m1;
      }
      if ((s3 & 7) == 7) {
        rc_barrier_lds();
        const int n = tid >> 8, c = tid & 255;                // RC_THREADS = 512: neighbourhood n of the block, channel c of the chunk
        const long long gn = grp0 + n;
        if (gn < p.groups) {
          const float* q = pool + ((chunk & 1) * RC_WAVES + 4 * n) * 256 + c;
          p.out[gn * p.ldo + 256 * chunk + c] = fmaxf(fmaxf(q[0], q[256]), fmaxf(q[512], q[768]));
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static bool rc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// > 64 KiB of dynamic LDS needs the function attribute, once per (kernel, device) -- one process may drive several GPUs
static int rc_allow_lds(const void* kernel, size_t bytes) {
  static unsigned long long done[4] = {0ull, 0ull, 0ull, 0ull};   // bit per device id < 64, per kernel
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
  const int which = kernel == reinterpret_cast<const void*>(fp_head_chain_kernel<false>) ? 0
                  : kernel == reinterpret_cast<const void*>(sa_premul_chain_kernel) ? 1
                  : kernel == reinterpret_cast<const void*>(sa3_premul_chain_kernel) ? 2 : 3;
  if (dev >= 0 && dev < 64 && (done[which] >> dev) & 1ull) return REGNET_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0 && dev < 64) done[which] |= 1ull << dev;
  return REGNET_OK;
}

extern "C" int64_t regnet_sa_premul_chain_stream_floats(void) { return 24ll * RC_STAGE_FLOATS; }

extern "C" int regnet_sa_premul_chain_f32(const float* U, int64_t ldu, const float* V, int64_t ldv, const int64_t* nbr,
                                          int64_t B, int64_t Nsrc, int64_t M, const float* stream, int64_t n_stages,
                                          const float* affine, int64_t affine_floats, int relu3, float* out, int64_t ldo,
                                          int32_t* ticket, void* stream_handle) {
  if (B < 0 || M < 0 || Nsrc <= 0 || ldu < 256 || ldv < 256 || (ldu & 3) || (ldv & 3) || ldo < 512 || n_stages != 24 ||
      affine_floats != 1536)
    return REGNET_ERR_SHAPE;
  const long long groups = B * M;
  if (groups == 0) return REGNET_OK;
  if (!U || !V || !nbr || !stream || !affine || !out || !ticket) return REGNET_ERR_NULL;
  if (!rc_aligned16(U) || !rc_aligned16(V) || !rc_aligned16(stream) || !rc_aligned16(affine)) return REGNET_ERR_SHAPE;
  ScArgs a = {};
  a.U = U; a.ldu = ldu; a.scene_stride = Nsrc * ldu; a.V = V; a.ldv = ldv; a.nbr = (const long long*)nbr;
  a.groups = groups; a.groups_per_scene = M; a.stream = stream; a.n_stages = (int)n_stages;
  a.affine = affine; a.affine_floats = (int)affine_floats; a.relu3 = relu3; a.out = out; a.ldo = ldo;
  {   // the last ~half round of blocks is handed out as half blocks (see the kernel)
    const long long gpb = RC_WAVES / 4, blocks = (groups + gpb - 1) / gpb;
    const long long split = (RC_WAVES == 8 && RC_TAIL_HALF) ? (blocks < 128 ? blocks : 128) : 0;
    a.n_full = blocks - split;
    const long long rest = groups - a.n_full * gpb;            // neighbourhoods left for half blocks (one each)
    a.n_blocks = a.n_full + (rest > 0 ? rest : 0);
  }
  a.ticket = ticket;
  const int cus = 256 * RC_WG_PER_CU;
  const long long wgs = a.n_blocks < cus ? a.n_blocks : cus;
  const size_t lds = (size_t)(RC_STAGES * RC_STAGE_FLOATS + 1536 + RC_WAVES * 512 + 4) * sizeof(float);
  int rc_attr = rc_allow_lds(reinterpret_cast<const void*>(sa_premul_chain_kernel), lds);
  if (rc_attr) return rc_attr;
  hipLaunchKernelGGL(sa_premul_chain_kernel, dim3((unsigned)wgs), dim3(RC_THREADS), lds, as_stream(stream_handle), a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int64_t regnet_sa3_premul_chain_stream_floats(void) { return 96ll * RC_STAGE_FLOATS; }

extern "C" int regnet_sa3_premul_chain_f32(const float* U, int64_t ldu, const float* V, int64_t ldv, const int64_t* nbr,
                                           int64_t B, int64_t Nsrc, int64_t M, const float* stream, int64_t n_stages,
                                           const float* affine, int64_t affine_floats, int relu3, float* out, int64_t ldo,
                                           int32_t* ticket, void* stream_handle) {
  if (B < 0 || M < 0 || Nsrc <= 0 || ldu < 512 || ldv < 512 || (ldu & 3) || (ldv & 3) || ldo < 1024 || n_stages != 96 ||
      affine_floats != 3072 || RC_WAVES != 8)
    return REGNET_ERR_SHAPE;
  const long long groups = B * M;
  if (groups == 0) return REGNET_OK;
  if (!U || !V || !nbr || !stream || !affine || !out || !ticket) return REGNET_ERR_NULL;
  if (!rc_aligned16(U) || !rc_aligned16(V) || !rc_aligned16(stream) || !rc_aligned16(affine)) return REGNET_ERR_SHAPE;
  Sc3Args a = {};
  a.U = U; a.ldu = ldu; a.scene_stride = Nsrc * ldu; a.V = V; a.ldv = ldv; a.nbr = (const long long*)nbr;
  a.groups = groups; a.groups_per_scene = M; a.stream = stream; a.n_stages = (int)n_stages;
  a.affine = affine; a.affine_floats = (int)affine_floats; a.relu3 = relu3; a.out = out; a.ldo = ldo;
  a.n_blocks = (groups + 1) / 2;
  a.ticket = ticket;
  const long long wgs = a.n_blocks < 256 ? a.n_blocks : 256;
  const size_t lds = (size_t)(RC_MID_STAGES * RC_STAGE_FLOATS + 3072 + RC_SA3_POOL_FLOATS + 4) * sizeof(float);
  int rc_attr = rc_allow_lds(reinterpret_cast<const void*>(sa3_premul_chain_kernel), lds);
  if (rc_attr) return rc_attr;
  hipLaunchKernelGGL(sa3_premul_chain_kernel, dim3((unsigned)wgs), dim3(RC_THREADS), lds, as_stream(stream_handle), a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int64_t regnet_fp_head_chain_stream_floats(void) { return 60ll * RC_STAGE_FLOATS; }

static long long rc_fp_head_blocks(long long P, long long* n_full) {
  const long long units = (P + 15) / 16, blocks = (units + RC_WAVES - 1) / RC_WAVES;
  const long long split = (RC_WAVES == 8 && RC_TAIL_HALF) ? (blocks < 128 ? blocks : 128) : 0;
  const long long full = blocks - split;
  const long long rest = units - full * RC_WAVES;              // 16-row units left for half blocks (4 each)
  if (n_full) *n_full = full;
  return full + (rest > 0 ? (rest + RC_WAVES / 2 - 1) / (RC_WAVES / 2) : 0);
}

extern "C" int64_t regnet_fp_head_chain_blocks(int64_t P) { return P <= 0 ? 0 : rc_fp_head_blocks(P, nullptr); }

static int rc_launch_fp_head(RcArgs& a, bool interp, int64_t block_first, int64_t block_count, void* stream_handle) {
  a.n_blocks = rc_fp_head_blocks(a.P, &a.n_full);
  if (block_count < 0) { block_first = 0; block_count = a.n_blocks; }
  if (block_first < 0 || block_first + block_count > a.n_blocks) return REGNET_ERR_SHAPE;
  if (block_count == 0) return REGNET_OK;
  a.blk_first = block_first; a.blk_count = block_count;
  const int cus = 256 * RC_WG_PER_CU;
  const long long wgs = block_count < cus ? block_count : cus;
  const size_t lds = (size_t)(RC_MID_STAGES * RC_STAGE_FLOATS + RC_AFFINE_MAX + 128 + 4 + (interp ? RC_INTERP_FLOATS : 0)) * sizeof(float);
  const void* kernel = interp ? reinterpret_cast<const void*>(fp_head_chain_kernel<true>)
                              : reinterpret_cast<const void*>(fp_head_chain_kernel<false>);
  int rc_attr = rc_allow_lds(kernel, lds);
  if (rc_attr) return rc_attr;
  if (interp) hipLaunchKernelGGL(fp_head_chain_kernel<true>, dim3((unsigned)wgs), dim3(RC_THREADS), lds, as_stream(stream_handle), a);
  else hipLaunchKernelGGL(fp_head_chain_kernel<false>, dim3((unsigned)wgs), dim3(RC_THREADS), lds, as_stream(stream_handle), a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_fp_head_chain_f32(const float* X, int64_t ldx, const float* stream, int64_t n_stages,
                                        const float* affine, int64_t affine_floats, const float* wscore,
                                        float score_bias, float score_bn_scale, float score_bn_shift, float* F,
                                        int64_t ldf, float* score, int64_t P, int32_t* ticket, int64_t block_first,
                                        int64_t block_count, void* stream_handle) {
  if (P < 0 || ldx < 256 || ldf < 256 || (ldx & 3) || (ldf & 3) || n_stages != 60 || affine_floats != 3328)
    return REGNET_ERR_SHAPE;
  if (P == 0) return REGNET_OK;
  if (!X || !stream || !affine || !wscore || !F || !score || !ticket) return REGNET_ERR_NULL;
  if (!rc_aligned16(X) || !rc_aligned16(F) || !rc_aligned16(stream) || !rc_aligned16(affine) || !rc_aligned16(wscore))
    return REGNET_ERR_SHAPE;
  RcArgs a = {};
  a.X = X; a.ldx = ldx; a.F = F; a.ldf = ldf; a.score = score; a.P = P;
  a.stream = stream; a.n_stages = (int)n_stages; a.affine = affine; a.affine_floats = (int)affine_floats;
  a.wscore = wscore; a.score_bias = score_bias; a.score_bn_scale = score_bn_scale; a.score_bn_shift = score_bn_shift;
  a.ticket = ticket;
  return rc_launch_fp_head(a, false, block_first, block_count, stream_handle);
}

extern "C" int regnet_fp_head_chain_interp_f32(const float* Ys, int64_t ys_sb, int64_t ys_sn, const int64_t* idx,
                                               const float* dist2, float eps, const float* dense_small, int64_t db,
                                               int64_t dn, int64_t dc, int64_t Cd_small, const float* tables,
                                               int64_t B, int64_t Nd, const float* stream, int64_t n_stages,
                                               const float* affine, int64_t affine_floats, const float* wscore,
                                               float score_bias, float score_bn_scale, float score_bn_shift, float* F,
                                               int64_t ldf, float* score, int32_t* ticket, int64_t block_first,
                                               int64_t block_count, void* stream_handle) {
  if (B < 0 || Nd < 0 || ldf < 256 || (ldf & 3) || n_stages != 60 || affine_floats != 3328 || (ys_sb & 3) || (ys_sn & 3) ||
      ys_sn < 256 || (dense_small && (Cd_small < 1 || Cd_small > 4)))
    return REGNET_ERR_SHAPE;
  const long long P = B * Nd;
  if (P == 0) return REGNET_OK;
  if (!Ys || !idx || !dist2 || !tables || !stream || !affine || !wscore || !F || !score || !ticket) return REGNET_ERR_NULL;
  if (!rc_aligned16(Ys) || !rc_aligned16(F) || !rc_aligned16(stream) || !rc_aligned16(affine) || !rc_aligned16(wscore) ||
      !rc_aligned16(tables))
    return REGNET_ERR_SHAPE;
  RcArgs a = {};
  a.F = F; a.ldf = ldf; a.score = score; a.P = P;
  a.stream = stream; a.n_stages = (int)n_stages; a.affine = affine; a.affine_floats = (int)affine_floats;
  a.wscore = wscore; a.score_bias = score_bias; a.score_bn_scale = score_bn_scale; a.score_bn_shift = score_bn_shift;
  a.ticket = ticket;
  a.ys = Ys; a.ys_sb = ys_sb; a.ys_sn = ys_sn; a.idx = (const long long*)idx; a.dist2 = dist2; a.eps = eps; a.Nd = Nd;
  a.dsm = dense_small; a.db = db; a.dn = dn; a.dc = dc; a.Cdsm = (int)Cd_small; a.tables = tables;
  return rc_launch_fp_head(a, true, block_first, block_count, stream_handle);
}
