// sa_split.hip -- EXPERIMENT, off by default (fused.SPLIT_PRODUCTS): sa_chain.hip's level-1 set-abstraction block (gather -> layer 1
// -> layer 2 -> layer 3 -> max over the 64 neighbours; pn2_utils/modules.py:39-56, :210-246, pointnet2.py:40-42: 6 -> 128 -> 128 ->
// 256 over 5120 x 64 rows per scene) with fp32-FAITHFUL products on the bf16 matrix pipe (see tsplit.hip: x = x1 + x2 + x3 in bf16
// pieces, six of the nine piece products, fp32 accumulation; v_mfma_f32_32x32x16_bf16 issues 13.9x the fp32 instruction's flops).
//
// Same register chaining as sa_chain.hip, in the 32x32x16 operand layout (A: lane l = row l & 31, k = 8 (l >> 5) .. + 7; B: column
// l & 31, same k; D: register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31):
//   layer 1   D1[channel][point] = W1 . x on the matrix pipe too (K = 8 of 16: lane half 1 supplies zeros) -- the same layout as layer
//             2's output, so its BatchNorm + ReLU + pieces, made per k-block right in front of their MFMAs, are the B operand of
//   layer 2   D2[channel][point] = W2 . h1: eight k-blocks x four channel tiles, W2's three planes in LDS for the workgroup's life
//   BN + ReLU in place; registers 8 j .. 8 j + 7 of tile dt are channels 32 dt + 16 j + (i & 3) + 8 (i >> 2) + 4 fh: the eight k of lane
//             half fh in k-block 2 dt + j of the A operand of
//   layer 3   D3[point][channel] = h2 . W3^T, whose planes are stored with that k order (ss_planes_kernel) and stream through two LDS
//             buffers by LDS-DMA, one 32-channel tile at a time; the max over the points is a max over the accumulator's
//             registers + one exchange.
// A wave owns 32 points (one point tile): 64 accumulator + 96 operand registers; waves w and w + 4 hold the two halves of a
// neighbourhood and meet in LDS per output tile; a second half that is all padding (<= 32 members: `count`) skips its MFMAs.
// Workgroups are persistent (a ticket per block of four neighbourhoods); the next block's gather runs underneath layer 3.
// Built for the level-1 block's shapes only: 3 colour channels + 3 relative coordinates in, C3 = 256 out.
#include "common.h"

typedef float ss_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ss_bf16x8 __attribute__((ext_vector_type(8)));

#ifndef SS_ABLATE
#define SS_ABLATE 0          // measurement builds only (scripts/ablate/sa_split_ablate.sh; their results are wrong on purpose): 1 no layer-3
#endif                       // MFMAs, 2 no layer-2 MFMAs, 4 / 5 no piece conversion of layer 2's / layer 1's output, 7 no W3 stream, 8 no W3
                             // stream and no barriers, 9 barriers without the vmcnt wait, 10 no gather of the next block's points
#ifndef SS_CHAINS
#define SS_CHAINS 2         // layer 3's accumulators per wave: even / odd k-blocks as two dependency chains, or one
#endif
#define SS_WAVES 8
#define SS_THREADS (SS_WAVES * 64)
#define SS_C 128
#define SS_TILES 8           // C3 = 256

struct SsArgs {
  const float* feat; long long fb, fn, fc; int Cf;
  const float* xyz; long long xb, xc, xn;
  const long long* nbr; const long long* ctr;
  const long long* count;      // (groups) members per neighbourhood (slots >= count repeat slot 0), or NULL
  const long long* order;      // (groups) processing order (neighbourhoods with <= 32 members together), or NULL
  long long groups, groups_per_scene;
  const float* W1; const float* scale1; const float* shift1;      // [128][8], [128], [128]
  const __bf16* W2p; const float* scale2; const float* shift2;    // [3][128][128]
  const __bf16* W3p; const float* scale3; const float* shift3;    // [3][C3][128], k permuted per 16-block
  int C3, relu3;
  float* out; long long ldo;
  int* ticket;                 // work-queue head, zeroed by the caller
};

__device__ __forceinline__ void ss_split(const float (&x)[8], ss_bf16x8& p1, ss_bf16x8& p2, ss_bf16x8& p3) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r = x[i] - (float)a;
    const __bf16 b = (__bf16)r;
    p1[i] = a; p2[i] = b; p3[i] = (__bf16)(r - (float)b);
  }
}

#define SS_SIX(ACC, A1, A2, A3, B1, B2, B3)                                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, B1, ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B3, ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B2, ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B1, ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B2, ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, ACC, 0, 0, 0);

// LDS images, chunk-major: a plane row is 128 bf16 = 16 chunks of 16 bytes, and an image of R rows stores chunk c of row r at
// (c R + r) 16 bytes.  The 32 lanes of an operand fragment (consecutive rows, the same chunk) read 512 consecutive bytes: no bank
// conflicts without a swizzle, and every fragment address of the kernel is ONE per-lane base (16 fr + fh R 16) plus a compile-time
// constant.  (A first version kept rows contiguous and XOR-swizzled the chunk: 8 k-blocks x 3 regions of distinct address
// registers, which the compiler computed once per launch and spilled.)  W2's image: [plane][16][128 rows]; a W3 tile's:
// [plane][16][32 rows], and the planes are stored in global memory AS these images (ss_planes_kernel), tile after tile, so that
// staging is a flat copy.
// global -> LDS without registers: lane l's 16 bytes land at lds_dst + 16 l (lds_dst wave-uniform); counted by vmcnt
__device__ __forceinline__ void ss_glds16(const void* gbase, unsigned lane_off, unsigned lds_dst) {      // gbase wave-uniform
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(SS_THREADS, 2) void sa_chain_split_kernel(const SsArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sW2 = smem;                                   // [3][128 rows][256 B]            98 304
  unsigned char* const sW3 = sW2 + 3 * SS_C * 256;                   // [2][3][32 rows][256 B]          49 152
  float* const sW1 = reinterpret_cast<float*>(sW3 + 2 * 3 * 32 * 256);   // three bf16 planes [128][8]    6 144
  float* const sS1 = sW1 + SS_C * 12;
  float* const sT1 = sS1 + SS_C;
  float* const sS2 = sT1 + SS_C;
  float* const sT2 = sS2 + SS_C;
  float* const sPool = sT2 + SS_C;                                   // [2][4][32]
  float* const sS3 = sPool + 2 * 4 * 32 + 4;                         // layer 3's folded BatchNorm (a global load per tile and wave, used
  float* const sT3 = sS3 + 512;                                      // at once, cost a load latency per output tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const unsigned char* const frag2 = sW2 + (fh * SS_C + fr) * 16;      // this lane's 16 bytes of chunk fh, row fr of an image
  const unsigned char* const frag3 = sW3 + (fh * 32 + fr) * 16;

  // ---- stage W2's image (a flat copy of 6144 chunks), the first two W3 tiles, the small tables
  for (int i = tid; i < 3 * SS_C * 16; i += SS_THREADS)
    *reinterpret_cast<float4*>(sW2 + 16 * i) = *reinterpret_cast<const float4*>(p.W2p + 8 * i);
  // A W3 tile's image is 24 pieces of 1 KB; wave w fetches pieces w, 8 + w, 16 + w by LDS-DMA (lane l: 16 bytes at 16 l of the piece).
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned w3lane = (unsigned)lane * 16u;
  const unsigned w3lds = (unsigned)(uintptr_t)sW3 + (unsigned)(wave_s * 1024);
#define SS_FETCH_W3(TILE, BUF)                                                                                   \
  _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                  \
    ss_glds16(p.W3p + ((long long)(TILE) * 3 * 32 * SS_C + (q * 8 + wave_s) * 512), w3lane,                     \
              w3lds + (unsigned)((BUF) * 3 * 32 * 256 + q * 8 * 1024));
  SS_FETCH_W3(0, 0)
  SS_FETCH_W3(1, 1)
  // W1 (128 x 8) as three bf16 planes of 16-byte rows [plane][128][8] (sW1: 6 KB), its folded BatchNorm in sS1 / sT1
  for (int c = tid; c < SS_C; c += SS_THREADS) {
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = p.W1[c * 8 + k];
    ss_bf16x8 q1, q2, q3;
    ss_split(w, q1, q2, q3);
    ss_bf16x8* pl = reinterpret_cast<ss_bf16x8*>(sW1);
    pl[c] = q1; pl[SS_C + c] = q2; pl[2 * SS_C + c] = q3;
    sS1[c] = p.scale1[c];
    sT1[c] = p.shift1[c];
    sS2[c] = p.scale2[c];
    sT2[c] = p.shift2[c];
  }
  for (int c = tid; c < p.C3; c += SS_THREADS) { sS3[c] = p.scale3[c]; sT3[c] = p.shift3[c]; }

  // ---- persistent: blocks of four neighbourhoods from a ticket counter; wave w holds half (w >> 2) of neighbourhood 4 blk + (w & 3),
  // lane -> point fr (both lane halves the same).  The NEXT block's gather is issued underneath this block's layer 3, and the W3
  // tiles form a cyclic stream: the fetches behind the last two tiles' barriers are tiles 0 and 1 of the next block.  Nothing but the
  // ticket sits between two blocks.
  // Everything that depends on the neighbourhood only (slot, processing order, scene, centre, member count) is the same for the 64
  // lanes of a wave: kept in scalar registers (v_readfirstlane of a broadcast load), so that the vector registers carried under
  // layer 3 are one point index and six inputs.
  int* const s_tick = reinterpret_cast<int*>(sPool + 2 * 4 * 32);
  const int blocks = (int)((p.groups + 3) / 4);
  const int half = wave_s >> 2, quarter = wave_s & 3;
  constexpr int tiles = SS_TILES;      // C3 / 32, unrolled: what the next block's gather carries from tile to tile stays in place
  const unsigned per_scene = (unsigned)p.groups_per_scene;
  if (tid == 0) s_tick[0] = atomicAdd(p.ticket, 1);
  __syncthreads();
  int blk = __builtin_amdgcn_readfirstlane(s_tick[0]);
  float x[8];
  int gs = 0;
  bool valid = false, work = false;      // work: this wave's half holds real members (a second half of <= 32 members is all padding)
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] = 0.f;
  if (blk < blocks) {
    const long long slot = (long long)blk * 4 + quarter;
    valid = slot < p.groups;
    gs = valid ? (p.order ? __builtin_amdgcn_readfirstlane((int)p.order[slot]) : (int)slot) : 0;
    work = valid && !(half == 1 && p.count && __builtin_amdgcn_readfirstlane((int)p.count[gs]) <= 32);
    const unsigned b = (unsigned)gs / per_scene;
    const float* xb = p.xyz + (long long)b * p.xb;
    const float* fb = p.feat + (long long)b * p.fb;
    const long long cj = __builtin_amdgcn_readfirstlane((int)p.ctr[gs]);
    const long long j = (int)p.nbr[(long long)gs * 64 + half * 32 + fr];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x[c] = fb[j * p.fn + c * p.fc];
      x[3 + c] = xb[c * p.xc + j * p.xn] - xb[c * p.xc + cj * p.xn];
    }
  }
  __syncthreads();      // (s_tick[0] has been read by everybody)
  float m_hold = 0.f;
  bool pvalid = false;      // (false until a tile is pending)
  int pet = 0;
  float* porow = p.out;

  while (blk < blocks) {
    if (tid == 0) s_tick[0] = atomicAdd(p.ticket, 1);      // read behind the first output tile's barrier
    // ---- layer 2 (layer 1 on the fly)
    ss_f32x16 acc2[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[dt][r] = 0.f;
    // ---- layer 1 on the matrix pipe as well (K = 8 of 16: lane half 1 supplies zeros): D1[channel][point], the same layout as layer
    // 2's output -- so its BatchNorm + ReLU + pieces are the B operand of layer 2 with the k order of layer 3's A operand (W2's
    // planes are stored with it).  (A first version evaluated layer 1 on the VALU inside layer 2's loop, as sa_chain.hip does: its 24
    // broadcast LDS reads per k-block and wave made the workgroup's ONE LDS pipe the bound of layer 2 -- 0.51 ms where the MFMAs
    // need 0.28.)
#ifndef SS_SKEW
#define SS_SKEW 8
#endif
    // The two waves of a SIMD (w and w + 4) leave the block's last barrier together and run the same code: their piece conversions
    // and fragment reads (matrix pipe idle) and their MFMA runs (both queueing) would coincide for all of layer 2.  The second one
    // starts half a k-block late, so that one converts while the other multiplies.
    if (half == 1 && SS_SKEW > 0) __builtin_amdgcn_s_sleep(SS_SKEW);
    if (work) {
      ss_bf16x8 x1, x2, x3;
      {
        float xv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) xv[c] = fh == 0 ? x[c] : 0.f;
        ss_split(xv, x1, x2, x3);
      }
      ss_f32x16 acc1[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[dt][r] = 0.f;
      }
      const ss_bf16x8* pl = reinterpret_cast<const ss_bf16x8*>(sW1);
      ss_bf16x8 c1[4], c2[4], c3[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        ss_bf16x8 z;
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f;
        c1[dt] = fh == 0 ? pl[32 * dt + fr] : z;
        c2[dt] = fh == 0 ? pl[SS_C + 32 * dt + fr] : z;
        c3[dt] = fh == 0 ? pl[2 * SS_C + 32 * dt + fr] : z;
      }
#define SS_L1(AP, BP) _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) acc1[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AP[dt], BP, acc1[dt], 0, 0, 0);
      SS_L1(c3, x1) SS_L1(c1, x3) SS_L1(c2, x2) SS_L1(c2, x1) SS_L1(c1, x2) SS_L1(c1, x1)
#undef SS_L1
      // ---- layer 2; the pieces of k-block kb = 2 dt + j (registers 8 j .. 8 j + 7 of layer 1's tile dt, BatchNorm + ReLU applied) are
      // made right in front of its MFMAs: 12 registers live instead of 96
#pragma unroll
      for (int kb = 0; kb < SS_C / 16; ++kb) {
        const int dt1 = kb >> 1, j1 = kb & 1;
        ss_bf16x8 h1, h2, h3;
        {
          float v[8];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int ch = 32 * dt1 + 16 * j1 + 8 * q + 4 * fh;
            const float4 sc = *reinterpret_cast<const float4*>(&sS1[ch]);
            const float4 sh = *reinterpret_cast<const float4*>(&sT1[ch]);
            v[4 * q + 0] = fmaxf(acc1[dt1][8 * j1 + 4 * q + 0] * sc.x + sh.x, 0.f);
            v[4 * q + 1] = fmaxf(acc1[dt1][8 * j1 + 4 * q + 1] * sc.y + sh.y, 0.f);
            v[4 * q + 2] = fmaxf(acc1[dt1][8 * j1 + 4 * q + 2] * sc.z + sh.z, 0.f);
            v[4 * q + 3] = fmaxf(acc1[dt1][8 * j1 + 4 * q + 3] * sc.w + sh.w, 0.f);
          }
          if (SS_ABLATE != 5) ss_split(v, h1, h2, h3);
          else { h1 = *reinterpret_cast<const ss_bf16x8*>(&v[0]); h2 = h1; h3 = h1; }
        }
        ss_bf16x8 a1[4], a2[4], a3[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const unsigned char* f = frag2 + (2 * kb * SS_C + 32 * dt) * 16;      // chunk 2 kb + fh, row 32 dt + fr
          a1[dt] = *reinterpret_cast<const ss_bf16x8*>(f);
          a2[dt] = *reinterpret_cast<const ss_bf16x8*>(f + SS_C * 256);
          a3[dt] = *reinterpret_cast<const ss_bf16x8*>(f + 2 * SS_C * 256);
        }
#define SS_L2(AP, BP) _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) acc2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AP[dt], BP, acc2[dt], 0, 0, 0);
        if (SS_ABLATE != 2) { SS_L2(a3, h1) SS_L2(a1, h3) SS_L2(a2, h2) SS_L2(a2, h1) SS_L2(a1, h2) SS_L2(a1, h1) }
#undef SS_L2
      }
    }
    // ---- BN + ReLU of layer 2, then its three pieces as the A operand of layer 3: block 2 dt + j = registers 8 j .. 8 j + 7
    ss_bf16x8 g1[8], g2[8], g3[8];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ch = 32 * dt + 16 * j + 8 * q + 4 * fh;           // registers 8 j + 4 q .. + 3: channels ch .. ch + 3
          const float4 sc = *reinterpret_cast<const float4*>(&sS2[ch]);
          const float4 sh = *reinterpret_cast<const float4*>(&sT2[ch]);
          v[4 * q + 0] = fmaxf(acc2[dt][8 * j + 4 * q + 0] * sc.x + sh.x, 0.f);
          v[4 * q + 1] = fmaxf(acc2[dt][8 * j + 4 * q + 1] * sc.y + sh.y, 0.f);
          v[4 * q + 2] = fmaxf(acc2[dt][8 * j + 4 * q + 2] * sc.z + sh.z, 0.f);
          v[4 * q + 3] = fmaxf(acc2[dt][8 * j + 4 * q + 3] * sc.w + sh.w, 0.f);
        }
        if (SS_ABLATE != 4) ss_split(v, g1[2 * dt + j], g2[2 * dt + j], g3[2 * dt + j]);
        else { g1[2 * dt + j] = *reinterpret_cast<const ss_bf16x8*>(&v[0]); g2[2 * dt + j] = g1[2 * dt + j]; g3[2 * dt + j] = g1[2 * dt + j]; }
      }

    // ---- layer 3, one 32-channel output tile at a time (two accumulators: even / odd k-blocks, so that a wave's MFMAs are two
    // dependency chains instead of one)
    //
    // The W3 tiles are a cyclic stream through two LDS buffers, filled by LDS-DMA (no registers, no ds_write).  A tile's barrier sits
    // right behind the tile's LAST FRAGMENT READ, not behind its last MFMA: what it orders is LDS traffic.  In front of it a wave
    // waits for its own pieces of tile et + 1 (fetched behind the previous barrier: a whole tile of time); behind it the buffer of
    // tile et is free, and tile et + 2 goes there.  The tail MFMAs, the epilogue and the next tile's first reads run without
    // anybody waiting for anybody.
    //
    // The next block's gather is three DEPENDENT load levels (processing order -> centre, member count, point index ->
    // coordinates and colours): one level behind each of the first three barriers, consumed behind the next one -- where the wave
    // has just waited for vmcnt(0) anyway (the compiler's own counted waits do not see the DMA issues between its loads).
    float* orow = p.out + (long long)gs * p.ldo;
    int nblk = blocks, ngs = 0, ncj = 0, nj = 0, ncount = 64;
    bool nvalid = false, nwork = false;
    float xr[6], nc[3];
#pragma unroll
    for (int c = 0; c < 6; ++c) xr[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) nc[c] = 0.f;
#define SS_NEXT_GATHER                                                                                                           \
  if (et == 0) {          /* the ticket was written before this barrier */                                                       \
    nblk = __builtin_amdgcn_readfirstlane(s_tick[0]);                                                                            \
    if (nblk < blocks) {                                                                                                         \
      const long long slot = (long long)nblk * 4 + quarter;                                                                      \
      nvalid = slot < p.groups;                                                                                                  \
      ngs = nvalid ? (p.order ? (int)p.order[slot] : (int)slot) : 0;                                                             \
    }                                                                                                                            \
  }                                                                                                                              \
  if (et == 1 && nblk < blocks) {                                                                                                \
    ngs = __builtin_amdgcn_readfirstlane(ngs);                                                                                   \
    ncj = (int)p.ctr[ngs];                                                                                                       \
    nj = (int)p.nbr[(long long)ngs * 64 + half * 32 + fr];                                                                       \
    if (half == 1 && p.count) ncount = (int)p.count[ngs];                                                                        \
  }                                                                                                                              \
  if (et == 2 && nblk < blocks && SS_ABLATE != 10) {                                                                             \
    const unsigned nb = (unsigned)ngs / per_scene;                                                                               \
    const float* xb = p.xyz + (long long)nb * p.xb;                                                                              \
    const float* fb = p.feat + (long long)nb * p.fb;                                                                             \
    const long long cj = __builtin_amdgcn_readfirstlane(ncj);                                                                    \
    nwork = nvalid && __builtin_amdgcn_readfirstlane(ncount) > 32;                                                               \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                                              \
      xr[c] = fb[(long long)nj * p.fn + c * p.fc];                                                                               \
      xr[3 + c] = xb[c * p.xc + (long long)nj * p.xn];                                                                           \
      nc[c] = xb[c * p.xc + cj * p.xn];                                                                                          \
    }                                                                                                                            \
  }
#define SS_TILE_BARRIER                                                                                                          \
  {                                                                                                                              \
    if (SS_ABLATE == 9) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
    else if (SS_ABLATE != 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                            \
    SS_NEXT_GATHER                                                                                                               \
    if (SS_ABLATE != 7 && SS_ABLATE != 8 && (et + 2 < tiles || nblk < blocks)) {                                                 \
      const int nt = et + 2 < tiles ? et + 2 : et + 2 - tiles;                                                                   \
      SS_FETCH_W3(nt, buf)                                                                                                       \
    }                                                                                                                            \
    /* the previous tile's pooled maximum: its second half was written before this barrier */                                    \
    if (half == 0 && fh == 0 && pvalid) porow[pet * 32 + fr] = fmaxf(m_hold, sPool[((pet & 1) * 4 + quarter) * 32 + fr]);        \
  }
#pragma unroll
    for (int et = 0; et < tiles; ++et) {
      const int buf = et & 1;
      __builtin_amdgcn_sched_barrier(0);      // (a tile's reads are not to be hoisted into the tile before: registers)
      ss_f32x16 acc3[SS_CHAINS];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc3[0][r] = 0.f; acc3[SS_CHAINS - 1][r] = 0.f; }
      const unsigned char* wt = frag3 + buf * 3 * 32 * 256;
      if (work && SS_ABLATE != 1) {
#pragma unroll
        for (int kb = 0; kb < 8; kb += 2) {
          const int off0 = 2 * kb * 32 * 16, off1 = (2 * kb + 2) * 32 * 16;      // chunks 2 kb + fh, 2 kb + 2 + fh of row fr
          const ss_bf16x8 u1 = *reinterpret_cast<const ss_bf16x8*>(wt + off0);
          const ss_bf16x8 u2 = *reinterpret_cast<const ss_bf16x8*>(wt + 32 * 256 + off0);
          const ss_bf16x8 u3 = *reinterpret_cast<const ss_bf16x8*>(wt + 2 * 32 * 256 + off0);
          const ss_bf16x8 v1 = *reinterpret_cast<const ss_bf16x8*>(wt + off1);
          const ss_bf16x8 v2 = *reinterpret_cast<const ss_bf16x8*>(wt + 32 * 256 + off1);
          const ss_bf16x8 v3 = *reinterpret_cast<const ss_bf16x8*>(wt + 2 * 32 * 256 + off1);
          if (kb == 6) SS_TILE_BARRIER
          acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g3[kb], u1, acc3[0], 0, 0, 0);
          acc3[SS_CHAINS - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g3[kb + 1], v1, acc3[SS_CHAINS - 1], 0, 0, 0);
          acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb], u3, acc3[0], 0, 0, 0);
          acc3[SS_CHAINS - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb + 1], v3, acc3[SS_CHAINS - 1], 0, 0, 0);
          acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[kb], u2, acc3[0], 0, 0, 0);
          acc3[SS_CHAINS - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[kb + 1], v2, acc3[SS_CHAINS - 1], 0, 0, 0);
          acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[kb], u1, acc3[0], 0, 0, 0);
          acc3[SS_CHAINS - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[kb + 1], v1, acc3[SS_CHAINS - 1], 0, 0, 0);
          acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb], u2, acc3[0], 0, 0, 0);
          acc3[SS_CHAINS - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb + 1], v2, acc3[SS_CHAINS - 1], 0, 0, 0);
          acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb], u1, acc3[0], 0, 0, 0);
          acc3[SS_CHAINS - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb + 1], v1, acc3[SS_CHAINS - 1], 0, 0, 0);
        }
      } else SS_TILE_BARRIER
      // lane l holds channel 32 et + fr of 16 points (+ the other 16 in lane l ^ 32)
      const float sc = sS3[et * 32 + fr], sh = sT3[et * 32 + fr];
      float m = -__builtin_inff();
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, (SS_CHAINS == 2 ? acc3[0][r] + acc3[SS_CHAINS - 1][r] : acc3[0][r]) * sc + sh);
      if (p.relu3) m = fmaxf(m, 0.f);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      if (!work) m = -__builtin_inff();          // (an all-padding half: copies of slot 0, which the first half holds)
      if (half == 1 && fh == 0) sPool[(buf * 4 + quarter) * 32 + fr] = m;
      m_hold = m; pet = et; porow = orow; pvalid = valid;      // combined and stored behind the NEXT tile's barrier
    }
#undef SS_TILE_BARRIER
#undef SS_NEXT_GATHER
    // ---- hand over to the next block
    blk = nblk; gs = ngs; valid = nvalid; work = nwork;
#pragma unroll
    for (int c = 0; c < 3; ++c) { x[c] = xr[c]; x[3 + c] = xr[3 + c] - nc[c]; }
  }
  {      // the last tile of the last block
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (half == 0 && fh == 0 && pvalid) porow[pet * 32 + fr] = fmaxf(m_hold, sPool[((pet & 1) * 4 + quarter) * 32 + fr]);
  }
}

// W (rows x 128) fp32 -> its three bf16 planes as LDS images of R rows (W2: one image of 128 rows; W3: one of 32 rows per output
// tile), [image][plane][chunk c][R rows][8], in the k order of the operand they meet: position 16 b + 8 fh + i of a row holds channel
// 16 b + (i & 3) + 8 (i >> 2) + 4 fh (the order in which an accumulator's registers supply k).
__global__ __launch_bounds__(256) void ss_planes_kernel(const float* __restrict__ W, long long ldw, int rows, int R, __bf16* __restrict__ P) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)rows * SS_C) return;
  const int r = (int)(i / SS_C), pos = (int)(i % SS_C);
  const int bb = pos >> 4, fh = (pos >> 3) & 1, ii = pos & 7;
  const int ch = 16 * bb + (ii & 3) + 8 * (ii >> 2) + 4 * fh;
  const float x = W[(long long)r * ldw + ch];
  const __bf16 a = (__bf16)x;
  const float rr = x - (float)a;
  const __bf16 b = (__bf16)rr;
  const long long image = r / R, plane = (long long)R * SS_C;
  const long long at = image * 3 * plane + ((long long)(pos >> 3) * R + r % R) * 8 + ii;
  P[at] = a;
  P[at + plane] = b;
  P[at + 2 * plane] = (__bf16)(rr - (float)b);
}

// planes: 3 * (128 + C3) * 128 bf16 (W2's, then W3's); built by this call when build_planes != 0 (once per weight version: the caller
// caches them), otherwise taken as they are.
extern "C" int64_t regnet_sa_chain3_split_plane_bytes(int64_t C3) { return 3 * (SS_C + C3) * SS_C * 2; }

extern "C" int regnet_sa_chain3_split_f32(const float* feat, int64_t fb, int64_t fn, int64_t fc, int64_t Cf, const float* xyz,
                                          int64_t xb, int64_t xc, int64_t xn, const int64_t* nbr, const int64_t* ctr,
                                          const int64_t* count, const int64_t* order, int64_t B, int64_t M, int64_t group, const float* W1, const float* scale1, const float* shift1,
                                          const float* W2, int64_t ldw2, const float* scale2, const float* shift2, const float* W3,
                                          int64_t ldw3, const float* scale3, const float* shift3, int64_t C3, int relu3, void* planes,
                                          int build_planes, float* out, int64_t ldo, int32_t* ticket, void* stream) {
  if (B < 0 || M < 0 || Cf < 0 || C3 <= 0 || ldo < C3) return REGNET_ERR_SHAPE;
  if (group != 64 || Cf != 3 || (C3 & 31) || ldw2 < SS_C || ldw3 < SS_C || B * M >= (1ll << 29)) return REGNET_ERR_UNSUPPORTED;
  const long long groups = B * M;
  if (groups == 0) return REGNET_OK;
  if (!xyz || !nbr || !ctr || !W1 || !scale1 || !shift1 || !W2 || !scale2 || !shift2 || !W3 || !scale3 || !shift3 || !out || !planes || !ticket ||
      (Cf > 0 && !feat))
    return REGNET_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(planes) & 15)) return REGNET_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  __bf16* p2 = reinterpret_cast<__bf16*>(planes);
  __bf16* p3 = p2 + 3ll * SS_C * SS_C;
  if (build_planes) {
    hipLaunchKernelGGL(ss_planes_kernel, dim3((SS_C * SS_C + 255) / 256), dim3(256), 0, st, W2, (long long)ldw2, SS_C, SS_C, p2);
    hipLaunchKernelGGL(ss_planes_kernel, dim3((unsigned)((C3 * SS_C + 255) / 256)), dim3(256), 0, st, W3, (long long)ldw3, (int)C3, 32, p3);
  }
  SsArgs a = {};
  a.feat = Cf > 0 ? feat : nullptr; a.fb = fb; a.fn = fn; a.fc = fc; a.Cf = (int)Cf;
  a.xyz = xyz; a.xb = xb; a.xc = xc; a.xn = xn;
  a.nbr = (const long long*)nbr; a.ctr = (const long long*)ctr; a.groups = groups; a.groups_per_scene = M;
  a.count = (const long long*)count; a.order = (const long long*)order;
  a.W1 = W1; a.scale1 = scale1; a.shift1 = shift1; a.W2p = p2; a.scale2 = scale2; a.shift2 = shift2;
  a.W3p = p3; a.scale3 = scale3; a.shift3 = shift3; a.C3 = (int)C3; a.relu3 = relu3; a.out = out; a.ldo = ldo; a.ticket = ticket;
  if (C3 != 32 * SS_TILES) return REGNET_ERR_UNSUPPORTED;     // (tiles even: the cyclic W3 stream lands tile 0 of the next block in buffer 0)
  const size_t lds = 3 * SS_C * 256 + 2 * 3 * 32 * 256 + (SS_C * 12 + 4 * SS_C + 2 * 4 * 32 + 4 + 2 * 512) * sizeof(float);
  static unsigned long long opted = 0ull;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  if (dev < 0 || dev >= 64 || !((opted >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sa_chain_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64) opted |= 1ull << dev;
  }
  const long long blocks = (groups + 3) / 4;
  hipLaunchKernelGGL(sa_chain_split_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(SS_THREADS), lds, st, a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
