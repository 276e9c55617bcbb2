// sa_chain.hip -- a whole narrow-input set-abstraction block (gather -> layer 1 -> layer 2 -> layer 3 -> max
// over the 64 neighbours) in ONE kernel whose activations never leave the register file (gfx950).
//
// Reference behaviour restated (paths relative to /root/reference/multi_model/utils): QueryGrouper.forward
// pn2_utils/modules.py:39-56 (group, xyz - centre, cat), SharedMLP = [1x1 conv -> BatchNorm -> ReLU]*
// pn2_utils/nn/modules/mlp.py:55-114, max over K pn2_utils/modules.py:244-245 -- the level-1 block of
// PointNet2Seg (pointnet2.py:40-42: 6 gathered inputs -> 128 -> 128 -> 256 over 5120 x 64 rows per scene).
//
// Why the products are computed TRANSPOSED.  v_mfma_f32_32x32x2_f32 computes D[32x32] += A[32x2] * B[2x32] with
//   A: lane l supplies A[i = l & 31][k = l >> 5]       B: lane l supplies B[k = l >> 5][j = l & 31]
//   D: element r of lane l is D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31].
// With rows = output CHANNELS and columns = POINTS (D = W . X^T), the accumulator of one layer IS a valid B
// operand of the next: register r of the layer-2 accumulator holds, in lane l, channel d0(r) + 4 (l >> 5) of
// point l & 31 -- exactly "k = l >> 5 selects one of two channels, j = l & 31 the point".  The two channels of
// such a step are d0 and d0 + 4 instead of consecutive ones; a dot product does not care, the A operand (the
// next layer's weights) just reads the same two columns: element (r & 3) of the 16-byte chunk
// W[e][32 dt + 8 (r >> 2) + 4 (l >> 5) ..], i.e. the same ds_read_b128 fragment pattern as mlp.hip.  So:
//   layer 1 (<= 8 inputs)  VALU, per (point, channel), straight into the B operand of layer 2
//   layer 2                64 k-steps x (4 channel tiles x 2 point tiles) MFMAs -> 128 accumulator VGPRs
//   BN + ReLU              in place on those registers
//   layer 3                per 32-channel output tile: 64 k-steps x 2 point tiles, the registers above as the A operand
//                          (A and B fragments have the same lane layout), so D3 is [point][channel] again and the
//   max over 64 points     is a max over accumulator registers + one cross-half exchange
// A wave owns one neighbourhood (64 points); the (P x 128) activations of layers 1 and 2 (1.3 GB each per batch
// of 8 scenes) are never written.  W2 stays in LDS for the life of the workgroup, W3 streams through a double
// buffer one 32-channel tile at a time (the 8 waves of a workgroup share it, one barrier per tile).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CH_WAVES 8
#define CH_THREADS (CH_WAVES * 64)
#define CH_C 128            // width of layers 1 and 2
#define CH_LD (CH_C + 4)    // LDS row stride: the 16-lane service groups of ds_read_b128 hit 64 distinct banks

struct ChainArgs {
  const float* feat; long long fb, fn, fc; int Cf;
  const float* xyz; long long xb, xc, xn;
  const long long* nbr;   // (groups, 64)
  const long long* ctr;   // (groups)
  const long long* count; // (groups) members per neighbourhood (slots >= count repeat slot 0), or NULL
  const long long* order; // (groups) the order in which the workgroups take the neighbourhoods, or NULL
  long long groups, groups_per_scene;
  const float* W1;        // [128][8]  columns [feat | rel xyz | 0]
  const float* scale1; const float* shift1;
  const float* W2;        // [128][128] row-major
  const float* scale2; const float* shift2;
  const float* W3;        // [C3][128] row-major, C3 % 32 == 0
  const float* scale3; const float* shift3;
  int C3, relu3;
  float* out; long long ldo;   // (groups, C3)
};

// Layers 2 and 3 + pooling of one neighbourhood per wave, for NPT = 1 or 2 tiles of 32 points.  A neighbourhood with
// at most 32 members fills only the first tile: its slots 32..63 are copies of slot 0 (the ball query pads with the
// first hit), which cannot change a maximum, so the second tile's MFMAs are skipped altogether.  Both instantiations
// execute the same barriers (one per W3 tile), so waves of one workgroup may take different ones.
//
// Paired neighbourhoods (role 1 = host, role 2 = guest).  Two neighbourhoods with 33..48 members each need THREE point
// tiles between them, not four: waves w and w + 4 of a workgroup -- the two waves of one SIMD -- form a pair when both of
// theirs are of that kind.  The host (w < 4) runs NPT = 2 with its second tile holding its own slots 32..47 in rows 0..15
// and the GUEST's slots 32..47 in rows 16..31; the guest runs NPT = 1 on its slots 0..31.  Rows are accumulator registers
// in the layer-3 product (rows 0..15 = registers 0..7), so the host pools the two halves of that tile separately at no
// cost, leaves the guest's partial maximum in LDS (sPart, double-buffered by W3 tile) and the guest -- the lighter wave --
// folds it into its own after the tile's barrier.  Every point's activations are the same MFMA chain as before and a
// maximum does not care about grouping: bit-identical outputs, 3/4 of the matrix work for such a pair, and the SIMD's two
// waves still add up to the same load on every SIMD of the workgroup (3 tiles each).
template <int NPT>
__device__ __forceinline__ void chain_group(const ChainArgs& p, const float* __restrict__ sW2, float (*sW3)[32 * CH_LD],
                                            const float* __restrict__ sW1, const float* __restrict__ sS2,
                                            const float* __restrict__ sT2, float (*sPart)[4][32], const int role,
                                            const float (&x)[2][8], long long gs, bool valid,
                                            float4 w3a, float4 w3b, int tid, int lane, int fr, int fh, int t_row0,
                                            int t_c4) {
  // ---- layer 2 (layer 1 on the fly): acc2[dt][pt] = W2[32 dt .., :] . h1[:, 32 pt ..] ------------------------------
  f32x16 acc2[4][NPT];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[dt][pt][r] = 0.f;
#pragma unroll 2
  for (int s = 0; s < CH_C / 8; ++s) {
    float4 a[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) a[dt] = *reinterpret_cast<const float4*>(&sW2[(dt * 32 + fr) * CH_LD + 8 * s + 4 * fh]);
    float h[NPT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* w = &sW1[(8 * s + 4 * fh + j) * 12];   // one address per half-wave: broadcast reads
      const float4 w0 = *reinterpret_cast<const float4*>(w), w1 = *reinterpret_cast<const float4*>(w + 4);
      const float2 st = *reinterpret_cast<const float2*>(w + 8);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        float v = w0.x * x[pt][0];
        v += w0.y * x[pt][1]; v += w0.z * x[pt][2]; v += w0.w * x[pt][3];
        v += w1.x * x[pt][4]; v += w1.y * x[pt][5]; v += w1.z * x[pt][6]; v += w1.w * x[pt][7];
        h[pt][j] = fmaxf(v * st.x + st.y, 0.f);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        acc2[dt][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt].x, h[pt][0], acc2[dt][pt], 0, 0, 0);
        acc2[dt][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt].y, h[pt][1], acc2[dt][pt], 0, 0, 0);
        acc2[dt][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt].z, h[pt][2], acc2[dt][pt], 0, 0, 0);
        acc2[dt][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt].w, h[pt][3], acc2[dt][pt], 0, 0, 0);
      }
  }
  // ---- BN + ReLU of layer 2, in place: register r of lane l is channel 32 dt + (r & 3) + 8 (r >> 2) + 4 fh ---------
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 sc = *reinterpret_cast<const float4*>(&sS2[dt * 32 + 8 * q + 4 * fh]);
      const float4 sh = *reinterpret_cast<const float4*>(&sT2[dt * 32 + 8 * q + 4 * fh]);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        acc2[dt][pt][4 * q + 0] = fmaxf(acc2[dt][pt][4 * q + 0] * sc.x + sh.x, 0.f);
        acc2[dt][pt][4 * q + 1] = fmaxf(acc2[dt][pt][4 * q + 1] * sc.y + sh.y, 0.f);
        acc2[dt][pt][4 * q + 2] = fmaxf(acc2[dt][pt][4 * q + 2] * sc.z + sh.z, 0.f);
        acc2[dt][pt][4 * q + 3] = fmaxf(acc2[dt][pt][4 * q + 3] * sc.w + sh.w, 0.f);
      }
    }

  // ---- layer 3, one 32-channel output tile at a time ------------------------------------------------------------------
  const int tiles = p.C3 / 32;
  float* orow = p.out + gs * p.ldo;
  for (int et = 0; et < tiles; ++et) {
    const int buf = et & 1;
    if (et + 1 < tiles) {   // next W3 tile: registers now, LDS after this tile's MFMAs
      w3a = reinterpret_cast<const float4*>(p.W3)[(et + 1) * 1024 + tid];
      w3b = reinterpret_cast<const float4*>(p.W3)[(et + 1) * 1024 + tid + CH_THREADS];
    }
    f32x16 acc3[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[pt][r] = 0.f;
    const float* wt = &sW3[buf][fr * CH_LD + 4 * fh];
    // Layer 3 is formed the other way round, D3[point][channel] = h2 . W3^T: the layer-2 registers serve as the A
    // operand just as well (lane -> point l & 31, k -> channel pair), and the max over the points then is a max over
    // the accumulator's registers + one cross-half exchange instead of a 32-lane reduction per register (which cost
    // 6 % of the kernel: VALU work is not free next to MFMAs; measured with scripts/ablate/chain_ablate.cpp).
    // Software-pipelined at source level and pinned: the weight fragment of step k + 1 is read in front of the MFMAs of
    // step k.  (hipcc's own order reads a fragment right before its first use; the two waves of a SIMD are released by
    // the same barrier and run the same code, so they would sit out every LDS round trip together.)
    float4 wn = *reinterpret_cast<const float4*>(wt);
#pragma unroll
    for (int step = 0; step < 16; ++step) {
      const int dt = step >> 2, q = step & 3;
      const float4 w = wn;
      if (step + 1 < 16) wn = *reinterpret_cast<const float4*>(wt + ((step + 1) >> 2) * 32 + 8 * ((step + 1) & 3));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) acc3[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc2[dt][pt][4 * q + 0], w.x, acc3[pt], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) acc3[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc2[dt][pt][4 * q + 1], w.y, acc3[pt], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) acc3[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc2[dt][pt][4 * q + 2], w.z, acc3[pt], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) acc3[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc2[dt][pt][4 * q + 3], w.w, acc3[pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // lane l holds channel et*32 + (l & 31) of 16 points per point tile (+ the other 16 in lane l ^ 32); registers 0..7
    // are rows 0..15 of a tile, registers 8..15 rows 16..31
    float hold;
    {
      const float sc = p.scale3[et * 32 + fr], sh = p.shift3[et * 32 + fr];
      float m = -__builtin_inff(), m2 = -__builtin_inff();
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc3[0][r] * sc + sh);
      if (NPT == 2) {
#pragma unroll
        for (int r = 0; r < 8; ++r) m = fmaxf(m, acc3[NPT - 1][r] * sc + sh);
#pragma unroll
        for (int r = 8; r < 16; ++r) m2 = fmaxf(m2, acc3[NPT - 1][r] * sc + sh);
        if (role == 1) {   // rows 16..31 of the second tile are the guest's
          m2 = fmaxf(m2, __shfl_xor(m2, 32, 64));
          if (fh == 0) sPart[buf][tid >> 6][fr] = m2;
        } else {
          m = fmaxf(m, m2);
        }
      }
      if (p.relu3) m = fmaxf(m, 0.f);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      hold = m;
      if (role != 2 && fh == 0 && valid) orow[et * 32 + fr] = m;
    }
    if (et + 1 < tiles) {
      *reinterpret_cast<float4*>(&sW3[buf ^ 1][t_row0 * CH_LD + t_c4 * 4]) = w3a;
      *reinterpret_cast<float4*>(&sW3[buf ^ 1][(t_row0 + 16) * CH_LD + t_c4 * 4]) = w3b;
    }
    __syncthreads();
    if (NPT == 1 && role == 2) {   // the host's next write to sPart[buf] is two barriers away
      const float m = fmaxf(hold, sPart[buf][(tid >> 6) - 4][fr]);
      if (fh == 0 && valid) orow[et * 32 + fr] = m;
    }
  }
}

// Measurement hook: scripts/wg_timeline.py builds a variant with -DCH_TRACE_H='"<repo>/scripts/ablate/chain_trace.h"', which
// defines the three macros (per-workgroup start / end stamps + placement) and its own debug export; the product compiles none.
#ifdef CH_TRACE_H
#include CH_TRACE_H
#else
#define CH_TRACE_BEGIN()
#define CH_TRACE_MID()
#define CH_TRACE_END()
#endif

__global__ __launch_bounds__(CH_THREADS, 2) void sa_chain_kernel(const ChainArgs p) {
  CH_TRACE_BEGIN();
  __shared__ __attribute__((aligned(16))) float sW2[CH_C * CH_LD];
  __shared__ __attribute__((aligned(16))) float sW3[2][32 * CH_LD];
  __shared__ __attribute__((aligned(16))) float sW1[CH_C * 12];   // [c][w0..w7 | scale | shift | 0 0]
  __shared__ __attribute__((aligned(16))) float sS2[CH_C], sT2[CH_C];
  __shared__ float sPart[2][4][32];   // [W3 tile parity][host wave][channel]: a paired guest's partial maxima
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;

  // ---- stage the weights -----------------------------------------------------------------------------------------
  for (int i = tid; i < CH_C * (CH_C / 4); i += CH_THREADS) {
    const float4 v = reinterpret_cast<const float4*>(p.W2)[i];
    *reinterpret_cast<float4*>(&sW2[(i / (CH_C / 4)) * CH_LD + (i % (CH_C / 4)) * 4]) = v;
  }
  for (int c = tid; c < CH_C; c += CH_THREADS) {
#pragma unroll
    for (int k = 0; k < 8; ++k) sW1[c * 12 + k] = p.W1[c * 8 + k];
    sW1[c * 12 + 8] = p.scale1[c];
    sW1[c * 12 + 9] = p.shift1[c];
    sW1[c * 12 + 10] = 0.f;
    sW1[c * 12 + 11] = 0.f;
    sS2[c] = p.scale2[c];
    sT2[c] = p.shift2[c];
  }
  // W3 tile: 32 rows x 128 floats = 1024 float4, two per thread
  const int t_row0 = tid / (CH_C / 4), t_c4 = tid % (CH_C / 4);   // rows t_row0 and t_row0 + 16
  float4 w3a = reinterpret_cast<const float4*>(p.W3)[tid];
  float4 w3b = reinterpret_cast<const float4*>(p.W3)[tid + CH_THREADS];
  *reinterpret_cast<float4*>(&sW3[0][t_row0 * CH_LD + t_c4 * 4]) = w3a;
  *reinterpret_cast<float4*>(&sW3[0][(t_row0 + 16) * CH_LD + t_c4 * 4]) = w3b;

  // ---- gather this wave's neighbourhood: lane -> point fr of point tile pt (both half-waves hold the same points)
  const long long slot = (long long)blockIdx.x * CH_WAVES + wave;
  const bool valid = slot < p.groups;
  const long long gs = valid ? (p.order ? p.order[slot] : slot) : 0;
  // pairing (see chain_group): waves w and w + 4 when both neighbourhoods have 33..48 members; all of it wave-uniform
  int role = 0;
  long long gm = gs;
  const long long c_me = p.count ? p.count[gs] : 64;
  if (p.count && valid) {
    const long long pslot = wave < 4 ? slot + 4 : slot - 4;
    if (pslot < p.groups) {
      const long long gp = p.order ? p.order[pslot] : pslot;
      const long long c_p = p.count[gp];
      if (c_me > 32 && c_me <= 48 && c_p > 32 && c_p <= 48) { role = wave < 4 ? 1 : 2; gm = gp; }
    }
  }
  const int npt = (c_me <= 32 || role == 2) ? 1 : 2;
  const long long b = gs / p.groups_per_scene;
  const float* xb = p.xyz + b * p.xb;
  const long long cj = p.ctr[gs];
  const float cx = xb[cj * p.xn], cy = xb[p.xc + cj * p.xn], cz = xb[2 * p.xc + cj * p.xn];
  float x[2][8];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    // a host's second tile: rows 0..15 its own slots 32..47, rows 16..31 the guest's slots 32..47 (the guest's scene and centre)
    const bool theirs = pt == 1 && role == 1 && fr >= 16;
    const long long g = theirs ? gm : gs;
    const long long bg = theirs ? gm / p.groups_per_scene : b;
    const float* xg = p.xyz + bg * p.xb;
    float ox = cx, oy = cy, oz = cz;
    if (theirs) {
      const long long cm = p.ctr[gm];
      ox = xg[cm * p.xn]; oy = xg[p.xc + cm * p.xn]; oz = xg[2 * p.xc + cm * p.xn];
    }
    const long long j = p.nbr[g * 64 + pt * 32 + (theirs ? fr - 16 : fr)];
    const float rx = xg[j * p.xn] - ox, ry = xg[p.xc + j * p.xn] - oy, rz = xg[2 * p.xc + j * p.xn] - oz;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float v = 0.f;
      if (c < p.Cf) v = p.feat[bg * p.fb + j * p.fn + (long long)c * p.fc];
      else if (c == p.Cf) v = rx;
      else if (c == p.Cf + 1) v = ry;
      else if (c == p.Cf + 2) v = rz;
      x[pt][c] = v;
    }
  }
  __syncthreads();
  CH_TRACE_MID();

  if (npt == 2) chain_group<2>(p, sW2, sW3, sW1, sS2, sT2, sPart, role, x, gs, valid, w3a, w3b, tid, lane, fr, fh, t_row0, t_c4);
  else chain_group<1>(p, sW2, sW3, sW1, sS2, sT2, sPart, role, x, gs, valid, w3a, w3b, tid, lane, fr, fh, t_row0, t_c4);
  CH_TRACE_END();
}

static bool aligned16c(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int regnet_sa_chain3_f32(const float* feat, int64_t fb, int64_t fn, int64_t fc, int64_t Cf, const float* xyz,
                                    int64_t xb, int64_t xc, int64_t xn, const int64_t* nbr, const int64_t* ctr,
                                    const int64_t* count, const int64_t* order, int64_t B, int64_t M, int64_t group,
                                    const float* W1, const float* scale1,
                                    const float* shift1, int64_t C1, const float* W2, int64_t K2pad,
                                    const float* scale2, const float* shift2, int64_t C2, const float* W3,
                                    int64_t K3pad, const float* scale3, const float* shift3, int64_t C3, int relu3,
                                    float* out, int64_t ldo, void* stream) {
  if (B < 0 || M < 0 || Cf < 0 || C3 <= 0 || ldo < C3) return REGNET_ERR_SHAPE;
  if (group != 64 || Cf + 3 > 8 || C1 != CH_C || C2 != CH_C || K2pad != CH_C || K3pad != CH_C || (C3 & 31))
    return REGNET_ERR_UNSUPPORTED;
  const long long groups = B * M;
  if (groups == 0) return REGNET_OK;
  if (!xyz || !nbr || !ctr || !W1 || !scale1 || !shift1 || !W2 || !scale2 || !shift2 || !W3 || !scale3 || !shift3 ||
      !out || (Cf > 0 && !feat))
    return REGNET_ERR_NULL;
  if (!aligned16c(W2) || !aligned16c(W3) || !aligned16c(scale3) || !aligned16c(shift3)) return REGNET_ERR_SHAPE;
  const long long blocks = (groups + CH_WAVES - 1) / CH_WAVES;
  if (blocks >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  ChainArgs a = {};
  a.feat = Cf > 0 ? feat : nullptr; a.fb = fb; a.fn = fn; a.fc = fc; a.Cf = (int)Cf;
  a.xyz = xyz; a.xb = xb; a.xc = xc; a.xn = xn;
  a.nbr = (const long long*)nbr; a.ctr = (const long long*)ctr; a.groups = groups; a.groups_per_scene = M;
  a.count = (const long long*)count; a.order = (const long long*)order;
  a.W1 = W1; a.scale1 = scale1; a.shift1 = shift1; a.W2 = W2; a.scale2 = scale2; a.shift2 = shift2;
  a.W3 = W3; a.scale3 = scale3; a.shift3 = shift3; a.C3 = (int)C3; a.relu3 = relu3;
  a.out = out; a.ldo = ldo;
  hipLaunchKernelGGL(sa_chain_kernel, dim3((unsigned)blocks), dim3(CH_THREADS), 0, as_stream(stream), a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
