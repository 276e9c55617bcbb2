// sa_split.hip -- EXPERIMENT, off by default (fused.SPLIT_PRODUCTS): sa_chain.hip's level-1 set-abstraction block (gather -> layer 1
// -> layer 2 -> layer 3 -> max over the 64 neighbours; pn2_utils/modules.py:39-56, :210-246, pointnet2.py:40-42: 6 -> 128 -> 128 ->
// 256 over 5120 x 64 rows per scene) with fp32-FAITHFUL products on the bf16 matrix pipe (see tsplit.hip: x = x1 + x2 + x3 in bf16
// pieces, six of the nine piece products, fp32 accumulation; v_mfma_f32_32x32x16_bf16 issues 13.9x the fp32 instruction's flops).
//
// Same register chaining as sa_chain.hip, in the 32x32x16 operand layout (A: lane l = row l & 31, k = 8 (l >> 5) .. + 7; B: column
// l & 31, same k; D: register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31):
//   layer 1   VALU, per (point, channel): lane (point fr, half fh) evaluates channels 16 b + 8 fh + i of k-block b -- the B operand of
//   layer 2   D2[channel][point] = W2 . h1: eight k-blocks x four channel tiles, W2's three planes in LDS for the workgroup's life
//   BN + ReLU in place; registers 8 j .. 8 j + 7 of tile dt are channels 32 dt + 16 j + (i & 3) + 8 (i >> 2) + 4 fh: the eight k of lane
//             half fh in k-block 2 dt + j of the A operand of
//   layer 3   D3[point][channel] = h2 . W3^T, whose planes are stored with that k order (sa_split_permute_w3) and stream through LDS
//             one 32-channel tile at a time; the max over the points is a max over the accumulator's registers + one exchange.
// A wave owns 32 points (one point tile): 64 accumulator + 96 operand registers; waves w and w + 4 hold the two halves of a
// neighbourhood and meet in LDS per output tile.  First version: every neighbourhood runs both halves (sa_chain.hip skips padded
// tiles: 26 % of its work).
#include "common.h"

typedef float ss_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ss_bf16x8 __attribute__((ext_vector_type(8)));

#ifndef SS_ABLATE
#define SS_ABLATE 0          // measurement builds only (scripts/ablate/sa_split_ablate.sh): 1 no layer-3 MFMAs, 2 no layer-2 MFMAs, 3 no
#endif                       // layer-1 / piece VALU work in layer 2, 4 no conversion of layer 2's output
#define SS_WAVES 8
#define SS_THREADS (SS_WAVES * 64)
#define SS_C 128

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

// LDS images: a plane row is 128 bf16 = 16 chunks of 16 bytes; chunk c of row r is stored at position c ^ (r & 15), so that the 16
// lanes of a ds_read_b128 service group (consecutive rows, same chunk) hit 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int ss_chunk(int row, int c) { return row * 256 + ((c ^ (row & 15)) << 4); }

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

  // ---- stage W2's planes (6144 chunks), the first W3 tile (1536 chunks), the small tables
  for (int i = tid; i < 3 * SS_C * 16; i += SS_THREADS) {
    const int plane = i / (SS_C * 16), rem = i % (SS_C * 16), row = rem >> 4, c = rem & 15;
    const float4 v = *reinterpret_cast<const float4*>(p.W2p + ((long long)plane * SS_C + row) * SS_C + 8 * c);
    *reinterpret_cast<float4*>(sW2 + plane * SS_C * 256 + ss_chunk(row, c)) = v;
  }
  float4 w3n[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {      // piece tid + 512 q of the tile's 1536: plane q (512 pieces = 32 rows x 16 chunks)
    const int row = tid >> 4, c = tid & 15;
    w3n[q] = *reinterpret_cast<const float4*>(p.W3p + ((long long)q * p.C3 + row) * SS_C + 8 * c);
    *reinterpret_cast<float4*>(sW3 + q * 32 * 256 + ss_chunk(row, c)) = w3n[q];
  }
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
  // lane -> point fr (both lane halves the same).  The NEXT block's gather is issued underneath this block's layer 3 (its two
  // dependent load levels -- indices, then coordinates -- behind two different output tiles), and the W3 tiles form a cyclic stream:
  // the last tile's prefetch is tile 0 of the next block.  Nothing but the ticket and one barrier sits between two blocks.
  int* const s_tick = reinterpret_cast<int*>(sPool + 2 * 4 * 32);
  const long long blocks = (p.groups + 3) / 4;
  const int half = wave >> 2;
  const int tiles = p.C3 / 32;
  if (tid == 0) s_tick[0] = atomicAdd(p.ticket, 1);
  __syncthreads();
  long long blk = s_tick[0];
  float x[8];
  long long gs = 0;
  bool valid = false, work = false;      // work: this wave's half holds real members (a second half of <= 32 members is all padding)
  if (blk < blocks) {
    const long long slot = blk * 4 + (wave & 3);
    valid = slot < p.groups;
    gs = valid ? (p.order ? p.order[slot] : slot) : 0;
    work = valid && !(half == 1 && p.count && p.count[gs] <= 32);
    const long long b = gs / p.groups_per_scene;
    const float* xb = p.xyz + b * p.xb;
    const long long cj = p.ctr[gs];
    const float cx = xb[cj * p.xn], cy = xb[p.xc + cj * p.xn], cz = xb[2 * p.xc + cj * p.xn];
    const long long j = p.nbr[gs * 64 + half * 32 + fr];
    const float rx = xb[j * p.xn] - cx, ry = xb[p.xc + j * p.xn] - cy, rz = xb[2 * p.xc + j * p.xn] - cz;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float v = 0.f;
      if (c < p.Cf) v = p.feat[b * p.fb + j * p.fn + (long long)c * p.fc];
      else if (c == p.Cf) v = rx;
      else if (c == p.Cf + 1) v = ry;
      else if (c == p.Cf + 2) v = rz;
      x[c] = v;
    }
  }
  __syncthreads();      // (s_tick[0] has been read by everybody)

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
          const int off = SS_ABLATE == 6 ? ss_chunk(32 * dt + fr, fh) : ss_chunk(32 * dt + fr, 2 * kb + fh);
          a1[dt] = *reinterpret_cast<const ss_bf16x8*>(sW2 + off);
          a2[dt] = *reinterpret_cast<const ss_bf16x8*>(sW2 + SS_C * 256 + off);
          a3[dt] = *reinterpret_cast<const ss_bf16x8*>(sW2 + 2 * SS_C * 256 + off);
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
    float* orow = p.out + gs * p.ldo;
    long long nblk = blocks, ngs = 0, nj = 0, ncj = 0;
    bool nvalid = false, nwork = false;
    float xr[8], ncx = 0.f, ncy = 0.f, ncz = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) xr[c] = 0.f;
    for (int et = 0; et < tiles; ++et) {
      const int buf = et & 1;
      {   // the next tile of the cyclic stream (after the last tile: tile 0, for the next block)
        const int nt = et + 1 < tiles ? et + 1 : 0;
#pragma unroll
        for (int q = 0; q < 3; ++q)
          w3n[q] = *reinterpret_cast<const float4*>(p.W3p + ((long long)q * p.C3 + 32 * nt + (tid >> 4)) * SS_C + 8 * (tid & 15));
      }
      if (et == 1) {          // next block: its ticket (written before the barrier of tile 0), then the first level of its gather
        nblk = s_tick[0];
        if (nblk < blocks) {
          const long long slot = nblk * 4 + (wave & 3);
          nvalid = slot < p.groups;
          ngs = nvalid ? (p.order ? p.order[slot] : slot) : 0;
          nwork = nvalid && !(half == 1 && p.count && p.count[ngs] <= 32);
          ncj = p.ctr[ngs];
          nj = p.nbr[ngs * 64 + half * 32 + fr];
        }
      }
      if (et == 4 && nblk < blocks) {   // second level: coordinates and features of the next block's points
        const long long nb = ngs / p.groups_per_scene;
        const float* xb = p.xyz + nb * p.xb;
        ncx = xb[ncj * p.xn]; ncy = xb[p.xc + ncj * p.xn]; ncz = xb[2 * p.xc + ncj * p.xn];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float v = 0.f;
          if (c < p.Cf) v = p.feat[nb * p.fb + nj * p.fn + (long long)c * p.fc];
          else if (c < p.Cf + 3) v = xb[(long long)(c - p.Cf) * p.xc + nj * p.xn];
          xr[c] = v;
        }
      }
      ss_f32x16 acc3a, acc3b;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc3a[r] = 0.f; acc3b[r] = 0.f; }
      const unsigned char* wt = sW3 + buf * 3 * 32 * 256;
      if (work && SS_ABLATE != 1)
#pragma unroll
      for (int kb = 0; kb < 8; kb += 2) {
        const int off0 = ss_chunk(fr, 2 * kb + fh), off1 = ss_chunk(fr, 2 * kb + 2 + fh);
        const ss_bf16x8 u1 = *reinterpret_cast<const ss_bf16x8*>(wt + off0);
        const ss_bf16x8 u2 = *reinterpret_cast<const ss_bf16x8*>(wt + 32 * 256 + off0);
        const ss_bf16x8 u3 = *reinterpret_cast<const ss_bf16x8*>(wt + 2 * 32 * 256 + off0);
        const ss_bf16x8 v1 = *reinterpret_cast<const ss_bf16x8*>(wt + off1);
        const ss_bf16x8 v2 = *reinterpret_cast<const ss_bf16x8*>(wt + 32 * 256 + off1);
        const ss_bf16x8 v3 = *reinterpret_cast<const ss_bf16x8*>(wt + 2 * 32 * 256 + off1);
        acc3a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g3[kb], u1, acc3a, 0, 0, 0);
        acc3b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g3[kb + 1], v1, acc3b, 0, 0, 0);
        acc3a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb], u3, acc3a, 0, 0, 0);
        acc3b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb + 1], v3, acc3b, 0, 0, 0);
        acc3a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[kb], u2, acc3a, 0, 0, 0);
        acc3b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[kb + 1], v2, acc3b, 0, 0, 0);
        acc3a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[kb], u1, acc3a, 0, 0, 0);
        acc3b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[kb + 1], v1, acc3b, 0, 0, 0);
        acc3a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb], u2, acc3a, 0, 0, 0);
        acc3b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb + 1], v2, acc3b, 0, 0, 0);
        acc3a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb], u1, acc3a, 0, 0, 0);
        acc3b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[kb + 1], v1, acc3b, 0, 0, 0);
      }
      // lane l holds channel 32 et + fr of 16 points (+ the other 16 in lane l ^ 32)
      const float sc = sS3[et * 32 + fr], sh = sT3[et * 32 + fr];
      float m = -__builtin_inff();
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, (acc3a[r] + acc3b[r]) * sc + sh);
      if (p.relu3) m = fmaxf(m, 0.f);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      if (!work) m = -__builtin_inff();          // (an all-padding half: copies of slot 0, which the first half holds)
      if (half == 1 && fh == 0) sPool[(buf * 4 + (wave & 3)) * 32 + fr] = m;
#pragma unroll
      for (int q = 0; q < 3; ++q)
        *reinterpret_cast<float4*>(sW3 + (buf ^ 1) * 3 * 32 * 256 + q * 32 * 256 + ss_chunk(tid >> 4, tid & 15)) = w3n[q];
      __syncthreads();
      if (half == 0 && fh == 0 && valid) orow[et * 32 + fr] = fmaxf(m, sPool[(buf * 4 + (wave & 3)) * 32 + fr]);
    }
    // ---- hand over to the next block
    blk = nblk; gs = ngs; valid = nvalid; work = nwork;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float v = xr[c];
      if (c == p.Cf) v -= ncx; else if (c == p.Cf + 1) v -= ncy; else if (c == p.Cf + 2) v -= ncz;
      x[c] = v;
    }
  }
}

// W3 (C3 x 128) fp32 -> three bf16 planes [3][C3][128] with layer 3's k order: position 16 b + 8 fh + i of a row holds channel
// 16 b + (i & 3) + 8 (i >> 2) + 4 fh (the order in which an accumulator's registers supply k).  W2 (128 x 128): the same (its B
// operand is layer 1's accumulator).
__global__ __launch_bounds__(256) void ss_planes_kernel(const float* __restrict__ W, long long ldw, int rows, int permute, __bf16* __restrict__ P) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)rows * SS_C) return;
  const int r = (int)(i / SS_C), pos = (int)(i % SS_C);
  int ch = pos;
  if (permute) {
    const int bb = pos >> 4, fh = (pos >> 3) & 1, ii = pos & 7;
    ch = 16 * bb + (ii & 3) + 8 * (ii >> 2) + 4 * fh;
  }
  const float x = W[(long long)r * ldw + ch];
  const __bf16 a = (__bf16)x;
  const float rr = x - (float)a;
  const __bf16 b = (__bf16)rr;
  P[i] = a;
  P[(long long)rows * SS_C + i] = b;
  P[2ll * rows * SS_C + i] = (__bf16)(rr - (float)b);
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
  if (group != 64 || Cf + 3 > 8 || (C3 & 31) || ldw2 < SS_C || ldw3 < SS_C) return REGNET_ERR_UNSUPPORTED;
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
    hipLaunchKernelGGL(ss_planes_kernel, dim3((SS_C * SS_C + 255) / 256), dim3(256), 0, st, W2, (long long)ldw2, SS_C, 1, p2);
    hipLaunchKernelGGL(ss_planes_kernel, dim3((unsigned)((C3 * SS_C + 255) / 256)), dim3(256), 0, st, W3, (long long)ldw3, (int)C3, 1, p3);
  }
  SsArgs a = {};
  a.feat = Cf > 0 ? feat : nullptr; a.fb = fb; a.fn = fn; a.fc = fc; a.Cf = (int)Cf;
  a.xyz = xyz; a.xb = xb; a.xc = xc; a.xn = xn;
  a.nbr = (const long long*)nbr; a.ctr = (const long long*)ctr; a.groups = groups; a.groups_per_scene = M;
  a.count = (const long long*)count; a.order = (const long long*)order;
  a.W1 = W1; a.scale1 = scale1; a.shift1 = shift1; a.W2p = p2; a.scale2 = scale2; a.shift2 = shift2;
  a.W3p = p3; a.scale3 = scale3; a.shift3 = shift3; a.C3 = (int)C3; a.relu3 = relu3; a.out = out; a.ldo = ldo; a.ticket = ticket;
  if (((C3 / 32) & 1) || C3 > 512) return REGNET_ERR_UNSUPPORTED;     // (the cyclic W3 stream lands tile 0 of the next block in buffer 0)
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
