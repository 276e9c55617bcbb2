// sa_chain2.hip -- layers 2 and 3 + the max over the neighbourhood of a WIDE set-abstraction block (levels 2 and 3
// of PointNet2Seg: 256 -> 256 -> 512 and 512 -> 512 -> 1024 over 64-point neighbourhoods) in one kernel whose
// layer-2 activation stays in the register file (gfx950).
//
// Reference behaviour restated (paths relative to /root/reference/multi_model/utils): SharedMLP layers 2..3 =
// [1x1 conv -> BatchNorm -> ReLU] (pn2_utils/nn/modules/mlp.py:55-114) and the max over K of
// PointNetSAModule.forward (pn2_utils/modules.py:244-245).  Layer 1 arrives pre-multiplied per source point
// (mlp.hip, AMODE 3 / fused.sa_features): a1[p][c] = relu(U[nbr[p]][c] - V[centre][c]).
//
// Same register chaining as sa_chain.hip, with the 16x16x4 fp32 MFMA so that a wave's slice of the layer-2
// activation fits its registers:  v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] . B[4x16],
//   A: lane l supplies A[i = l & 15][k = l >> 4]      B: lane l supplies B[k = l >> 4][j = l & 15]
//   D: register r of lane l is D[i = 4 (l >> 4) + r][j = l & 15].
// Layer 2 is formed channel-major (D2[d][p] = W2 . a1^T): register r of tile dt holds channel 16 dt + 4 g + r
// (g = l >> 4) of point l & 15 -- which is a valid A or B operand of a k-step whose four channels are
// {r, 4 + r, 8 + r, 12 + r} of the tile; the matching weight fragment is element r of the 16-byte chunk
// W[e][16 dt + 4 g ..].  Layer 3 takes those registers as the A operand, D3[p][e] = a2 . W3^T, so the max over the
// points is a max over the 4 accumulator registers, two lane exchanges (g) and the waves sharing a neighbourhood.
// A wave owns NT tiles of 16 points with (C2 / 16) x NT x 4 = 128 accumulator VGPRs: C2 = 256 -> 32 points (two
// waves per neighbourhood), C2 = 512 -> 16 points (four).  Both weight matrices stream through LDS, shared by the
// 8 waves of a workgroup: W2 as column panels [C2][16 k] (one per 16-channel slab of the input, whose gathered rows
// are read once), W3 as row panels [16 e][C2 k]; 128 MFMAs per wave between barriers, as in mlp.hip, but with a
// quarter of its load/store instructions per MFMA and no activation traffic at all.
//
// STATUS: correct (tests/test_gpu_mlp.py) but measured SLOWER than the two generic launches it replaces -- 2.20 vs
// 1.97 ms at level 2 (C2 = 256), 2.30 vs 1.97 ms at level 3 (C2 = 512, where hipcc spills 165 VGPRs of the
// accumulator) -- so fused.py does not use it by default (REGNET_CHAIN_WIDTHS=256,512 switches it on).  With one
// 8-wave workgroup per CU every barrier stalls the whole CU; 4-wave workgroups (two per CU) measured the same.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define C2_WAVES 8
#define C2_THREADS (C2_WAVES * 64)
#define C2_PLD 24   // row stride of a W2 column panel (16 k + 8): the ds_read_b128 service groups hit 64 distinct banks

struct Chain2Args {
  const float* U; long long ldu, scene_stride;   // U row of source point j of scene b: U + b*scene_stride + j*ldu
  const float* V; long long ldv;                 // V row of neighbourhood g: V + g*ldv
  const long long* nbr;                          // (groups, 64)
  long long groups, groups_per_scene;
  int C1;                                        // width of layer 1 = K of layer 2 (multiple of 16)
  const float* W2; int K2pad;                    // [>= C2][K2pad]
  const float* scale2; const float* shift2;
  const float* W3; int K3pad;                    // [>= C3][K3pad >= C2]
  const float* scale3; const float* shift3;
  int C3, relu3;
  float* out; long long ldo;                     // (groups, C3)
};

template <int C2>
__global__ __launch_bounds__(C2_THREADS, 2) void sa_chain2_kernel(const Chain2Args p) {
  constexpr int NT = 512 / C2;             // 16-point tiles per wave
  constexpr int WPG = 4 / NT;              // waves per neighbourhood
  constexpr int GPB = C2_WAVES / WPG;      // neighbourhoods per workgroup
  constexpr int DT = C2 / 16;              // channel tiles of layer 2
  constexpr int QLD = C2 + 24;             // row stride of a W3 row panel (same bank argument: = 24 mod 64)
  constexpr int PANEL2 = C2 * C2_PLD, PANEL3 = 16 * QLD;
  constexpr int BUF = PANEL2 > PANEL3 ? PANEL2 : PANEL3;
  constexpr int W_PER_THREAD = C2 * 4 / C2_THREADS;   // float4 per thread per panel (both kinds): 2 or 4
  __shared__ __attribute__((aligned(16))) float sW[2 * BUF];
  __shared__ __attribute__((aligned(16))) float sPool[2][C2_WAVES][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lp = lane & 15, g = lane >> 4;

  const long long grp = (long long)blockIdx.x * GPB + wave / WPG;
  const bool valid = grp < p.groups;
  const long long gs = valid ? grp : 0;
  const int part = wave % WPG;
  const long long scene = gs / p.groups_per_scene;
  const float* urow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const long long j = p.nbr[gs * 64 + (part * NT + t) * 16 + lp];
    urow[t] = p.U + scene * p.scene_stride + j * p.ldu + 4 * g;
  }
  const float* vrow = p.V + gs * p.ldv + 4 * g;

  // ---- layer 2: one 16-channel slab of the gathered input per barrier ------------------------------------------------
  f32x4 acc2[DT][NT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc2[dt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 wr[W_PER_THREAD];
  // W2 column panel of slab s: element (row d, chunk c4) <- W2[d][16 s + 4 c4 ..]; thread -> (d = idx / 4, c4 = idx % 4)
#define LOAD_PANEL2(S)                                                                                   \
  _Pragma("unroll") for (int k = 0; k < W_PER_THREAD; ++k) {                                             \
    const int idx = tid + k * C2_THREADS;                                                                \
    wr[k] = *reinterpret_cast<const float4*>(p.W2 + (long long)(idx >> 2) * p.K2pad + 16 * (S) + 4 * (idx & 3)); \
  }
#define STORE_PANEL2(BUFI)                                                                               \
  _Pragma("unroll") for (int k = 0; k < W_PER_THREAD; ++k) {                                             \
    const int idx = tid + k * C2_THREADS;                                                                \
    *reinterpret_cast<float4*>(&sW[(BUFI) * BUF + (idx >> 2) * C2_PLD + 4 * (idx & 3)]) = wr[k];        \
  }
  const int slabs = p.C1 / 16;
  float4 un[NT], vn;
  LOAD_PANEL2(0)
#pragma unroll
  for (int t = 0; t < NT; ++t) un[t] = *reinterpret_cast<const float4*>(urow[t]);
  vn = *reinterpret_cast<const float4*>(vrow);
  STORE_PANEL2(0)
  __syncthreads();
  for (int s = 0; s < slabs; ++s) {
    const int buf = s & 1;
    float4 a1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
      a1[t] = make_float4(fmaxf(un[t].x - vn.x, 0.f), fmaxf(un[t].y - vn.y, 0.f), fmaxf(un[t].z - vn.z, 0.f),
                          fmaxf(un[t].w - vn.w, 0.f));
    if (s + 1 < slabs) {
      LOAD_PANEL2(s + 1)
#pragma unroll
      for (int t = 0; t < NT; ++t) un[t] = *reinterpret_cast<const float4*>(urow[t] + 16 * (s + 1));
      vn = *reinterpret_cast<const float4*>(vrow + 16 * (s + 1));
    }
    const float* pw = &sW[buf * BUF + lp * C2_PLD + 4 * g];
    // four channel tiles at a time: the 16 NT MFMAs walk the 4 NT accumulators round-robin (a dependent 16x16x4 MFMA
    // issues every 40 cycles, an independent one every 32) while the NEXT four fragments are already being read; the
    // scheduling fences keep the compiler from hoisting all DT fragment reads (128 VGPRs) to the top
    float4 af[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const float4*>(pw + i * 16 * C2_PLD);
#pragma unroll
    for (int d0 = 0; d0 < DT; d0 += 4) {
      constexpr int dummy = 0;
      const int cur = (d0 >> 2) & 1;
      if (d0 + 4 < DT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[cur ^ 1][i] = *reinterpret_cast<const float4*>(pw + (d0 + 4 + i) * 16 * C2_PLD);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[d0 + i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i].x, a1[t].x, acc2[d0 + i][t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[d0 + i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i].y, a1[t].y, acc2[d0 + i][t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[d0 + i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i].z, a1[t].z, acc2[d0 + i][t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[d0 + i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i].w, a1[t].w, acc2[d0 + i][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      (void)dummy;
    }
    if (s + 1 < slabs) STORE_PANEL2(buf ^ 1)
    __syncthreads();
  }
  // ---- BN + ReLU of layer 2 in place: register r of tile dt is channel 16 dt + 4 g + r ----------------------------------
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const float4 sc = *reinterpret_cast<const float4*>(p.scale2 + 16 * dt + 4 * g);
    const float4 sh = *reinterpret_cast<const float4*>(p.shift2 + 16 * dt + 4 * g);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc2[dt][t][0] = fmaxf(acc2[dt][t][0] * sc.x + sh.x, 0.f);
      acc2[dt][t][1] = fmaxf(acc2[dt][t][1] * sc.y + sh.y, 0.f);
      acc2[dt][t][2] = fmaxf(acc2[dt][t][2] * sc.z + sh.z, 0.f);
      acc2[dt][t][3] = fmaxf(acc2[dt][t][3] * sc.w + sh.w, 0.f);
    }
    if ((dt & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // do not hoist all DT affine loads (register budget)
  }

  // ---- layer 3: one 16-channel output tile per barrier; W3 row panel [16][C2] ----------------------------------------------
  // element (row e, chunk c) <- W3[16 et + e][4 c ..]; thread -> (e = idx / (C2 / 4), c = idx % (C2 / 4))
#define LOAD_PANEL3(ET)                                                                                  \
  _Pragma("unroll") for (int k = 0; k < W_PER_THREAD; ++k) {                                             \
    const int idx = tid + k * C2_THREADS;                                                                \
    wr[k] = *reinterpret_cast<const float4*>(p.W3 + (long long)(16 * (ET) + idx / (C2 / 4)) * p.K3pad + 4 * (idx % (C2 / 4))); \
  }
#define STORE_PANEL3(BUFI)                                                                               \
  _Pragma("unroll") for (int k = 0; k < W_PER_THREAD; ++k) {                                             \
    const int idx = tid + k * C2_THREADS;                                                                \
    *reinterpret_cast<float4*>(&sW[(BUFI) * BUF + (idx / (C2 / 4)) * QLD + 4 * (idx % (C2 / 4))]) = wr[k]; \
  }
  const int tiles = p.C3 / 16;
  LOAD_PANEL3(0)
  STORE_PANEL3(0)   // buffer 0 was last read two barriers ago (slabs - 2) or never: free
  __syncthreads();
  float* orow = p.out + gs * p.ldo;
  for (int et = 0; et < tiles; ++et) {
    const int buf = et & 1;
    if (et + 1 < tiles) LOAD_PANEL3(et + 1)
    // two accumulators per tile (even / odd channel tiles): a dependent 16x16x4 MFMA can issue only every 40 cycles,
    // an independent one every 32
    f32x4 acc3[NT], acc3b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc3[t] = acc3b[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* qw = &sW[buf * BUF + lp * QLD + 4 * g];
    float4 wf[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[0][i] = *reinterpret_cast<const float4*>(qw + 16 * i);
#pragma unroll
    for (int d0 = 0; d0 < DT; d0 += 4) {
      const int cur = (d0 >> 2) & 1;
      if (d0 + 4 < DT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[cur ^ 1][i] = *reinterpret_cast<const float4*>(qw + 16 * (d0 + 4 + i));
      }
#pragma unroll
      for (int i = 0; i < 4; i += 2)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc2[d0 + i][t][0], wf[cur][i].x, acc3[t], 0, 0, 0);
          acc3b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc2[d0 + i + 1][t][0], wf[cur][i + 1].x, acc3b[t], 0, 0, 0);
          acc3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc2[d0 + i][t][1], wf[cur][i].y, acc3[t], 0, 0, 0);
          acc3b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc2[d0 + i + 1][t][1], wf[cur][i + 1].y, acc3b[t], 0, 0, 0);
          acc3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc2[d0 + i][t][2], wf[cur][i].z, acc3[t], 0, 0, 0);
          acc3b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc2[d0 + i + 1][t][2], wf[cur][i + 1].z, acc3b[t], 0, 0, 0);
          acc3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc2[d0 + i][t][3], wf[cur][i].w, acc3[t], 0, 0, 0);
          acc3b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc2[d0 + i + 1][t][3], wf[cur][i + 1].w, acc3b[t], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc3[t] += acc3b[t];
    // lane l holds channel 16 et + (l & 15) of points 4 g + r of each tile
    {
      const float sc = p.scale3[16 * et + lp], sh = p.shift3[16 * et + lp];
      float m = -__builtin_inff();
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc3[t][r] * sc + sh);
      if (p.relu3) m = fmaxf(m, 0.f);
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      if (g == 0) sPool[buf][wave][lp] = m;
    }
    if (et + 1 < tiles) STORE_PANEL3(buf ^ 1)
    __syncthreads();
    // the waves sharing a neighbourhood combine their maxima (first wave of the neighbourhood stores)
    if (part == 0 && g == 0 && valid) {
      float m = sPool[buf][wave][lp];
#pragma unroll
      for (int w = 1; w < WPG; ++w) m = fmaxf(m, sPool[buf][wave + w][lp]);
      orow[16 * et + lp] = m;
    }
  }
}

static bool aligned16d(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int regnet_sa_chain_premul_f32(const float* U, int64_t ldu, const float* V, int64_t ldv, int64_t C1,
                                          const int64_t* nbr, int64_t B, int64_t Nsrc, int64_t M, int64_t group,
                                          const float* W2, int64_t K2pad, const float* scale2, const float* shift2,
                                          int64_t C2, const float* W3, int64_t K3pad, const float* scale3,
                                          const float* shift3, int64_t C3, int relu3, float* out, int64_t ldo,
                                          void* stream) {
  if (B < 0 || M < 0 || Nsrc <= 0 || C1 <= 0 || C3 <= 0 || ldo < C3 || ldu < C1 || ldv < C1 || (ldu & 3) || (ldv & 3) ||
      K2pad < C1 || K3pad < C2 || (K2pad & 3) || (K3pad & 3))
    return REGNET_ERR_SHAPE;
  if (group != 64 || (C2 != 256 && C2 != 512) || (C1 & 15) || (C3 & 15)) return REGNET_ERR_UNSUPPORTED;
  const long long groups = B * M;
  if (groups == 0) return REGNET_OK;
  if (!U || !V || !nbr || !W2 || !scale2 || !shift2 || !W3 || !scale3 || !shift3 || !out) return REGNET_ERR_NULL;
  if (!aligned16d(U) || !aligned16d(V) || !aligned16d(W2) || !aligned16d(W3) || !aligned16d(scale2) || !aligned16d(shift2))
    return REGNET_ERR_SHAPE;
  Chain2Args a = {};
  a.U = U; a.ldu = ldu; a.scene_stride = Nsrc * ldu; a.V = V; a.ldv = ldv; a.nbr = (const long long*)nbr;
  a.groups = groups; a.groups_per_scene = M; a.C1 = (int)C1;
  a.W2 = W2; a.K2pad = (int)K2pad; a.scale2 = scale2; a.shift2 = shift2;
  a.W3 = W3; a.K3pad = (int)K3pad; a.scale3 = scale3; a.shift3 = shift3; a.C3 = (int)C3; a.relu3 = relu3;
  a.out = out; a.ldo = ldo;
  const long long gpb = C2 == 256 ? C2_WAVES / 2 : C2_WAVES / 4;
  const long long blocks = (groups + gpb - 1) / gpb;
  if (blocks >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  if (C2 == 256) hipLaunchKernelGGL((sa_chain2_kernel<256>), dim3((unsigned)blocks), dim3(C2_THREADS), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((sa_chain2_kernel<512>), dim3((unsigned)blocks), dim3(C2_THREADS), 0, as_stream(stream), a);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
