// region.hip -- region grouping between ScoreNet and the grasp-region network (gfx950).
//
// Built with -ffp-contract=off (membership tests are discontinuous in the fp32 arithmetic).
// Reference behaviour restated (paths relative to /root/reference):
//   radius grouping   dataset_utils/get_regiondataset.py:279-295, :311-352
//   gripper box crop  multi_model/gripper_region_network.py:508-544
//   feature gather+max multi_model/gripper_region_network.py:388-395 + utils/pointnet2.py:167
//
// The reference builds dense (centres x N) masks with torch and calls torch.nonzero per centre
// from a Python loop (one device sync each).  Here one wavefront owns one centre / grasp, walks
// the points 64 at a time and appends the members in ascending index order with a ballot +
// prefix-popcount, so the host receives every candidate list and count with a single sync and
// only has to draw the numpy random positions.
#include "common.h"

#define RG_WAVES 4

// cand[(b*NC + c)*cap + pos] = pos-th point (ascending) of scene b with d2(point, centre c) <= thr.
// One workgroup per centre; its four waves own consecutive quarters of the cloud.  Pass 1 counts the
// members of each quarter, pass 2 re-tests (the cloud is L2-resident by then) and writes them behind the
// earlier quarters' totals.  Four 64-point chunks are in flight per wave and iteration.
#define RG_UNROLL 4
__device__ __forceinline__ unsigned long long rg_hits(const float* __restrict__ p, int64_t pn, int j, int end,
                                                      float cx, float cy, float cz, float thr) {
  bool hit = false;
  if (j < end) {
    const float* r = p + (int64_t)j * pn;
    hit = sqdist3(r[0], r[1], r[2], cx, cy, cz) <= thr;  // point minus centre, inclusive
  }
  return __ballot(hit);
}

__global__ __launch_bounds__(RG_WAVES * 64) void radius_group_kernel(
    const float* __restrict__ pc, int64_t pb, int64_t pn, const float* __restrict__ ctr, int64_t cb, int64_t cn,
    int N, int NC, float thr, int64_t cap, int32_t* __restrict__ cand, int32_t* __restrict__ count) {
  __shared__ int wave_cnt[RG_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x;
  const int b = blockIdx.y;
  const float* p = pc + (int64_t)b * pb;
  const float* q = ctr + (int64_t)b * cb + (int64_t)c * cn;
  const float cx = q[0], cy = q[1], cz = q[2];
  int32_t* out = cand + ((int64_t)b * NC + c) * cap;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int per = ((N + RG_WAVES - 1) / RG_WAVES + 63) / 64 * 64;
  const int beg = min(N, wave * per), end = min(N, beg + per);
  int cnt = 0;
  for (int j0 = beg; j0 < end; j0 += 64 * RG_UNROLL) {
    unsigned long long m[RG_UNROLL];
#pragma unroll
    for (int u = 0; u < RG_UNROLL; ++u) m[u] = rg_hits(p, pn, j0 + 64 * u + lane, end, cx, cy, cz, thr);
#pragma unroll
    for (int u = 0; u < RG_UNROLL; ++u) cnt += (int)__popcll(m[u]);
  }
  if (lane == 0) wave_cnt[wave] = cnt;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < RG_WAVES; ++w) {
    const int v = wave_cnt[w];
    base += w < wave ? v : 0;
    total += v;
  }
  if (threadIdx.x == 0) count[(int64_t)b * NC + c] = total;
  if (cnt == 0 || base >= cap) return;   // wave-uniform
  for (int j0 = beg; j0 < end; j0 += 64 * RG_UNROLL) {
    unsigned long long m[RG_UNROLL];
#pragma unroll
    for (int u = 0; u < RG_UNROLL; ++u) m[u] = rg_hits(p, pn, j0 + 64 * u + lane, end, cx, cy, cz, thr);
#pragma unroll
    for (int u = 0; u < RG_UNROLL; ++u) {
      if ((m[u] >> lane) & 1ull) {
        const int pos = base + (int)__popcll(m[u] & lt_mask);
        if (pos < cap) out[pos] = j0 + 64 * u + lane;
      }
      base += (int)__popcll(m[u]);
    }
  }
}

extern "C" int regnet_radius_group_f32(const float* pc, int64_t pb, int64_t pn, const float* centres, int64_t cb,
                                       int64_t cn, int64_t B, int64_t N, int64_t NC, float d2_threshold,
                                       int64_t cap, int32_t* cand, int32_t* count, void* stream) {
  if (B < 0 || N < 0 || NC < 0 || cap < 0) return REGNET_ERR_SHAPE;
  if (N >= (int64_t)1 << 31 || B > 65535) return REGNET_ERR_UNSUPPORTED;
  if (B == 0 || NC == 0) return REGNET_OK;
  if (!centres || !count || (cap > 0 && !cand) || (N > 0 && !pc)) return REGNET_ERR_NULL;
  if (NC >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  dim3 grid((unsigned)NC, (unsigned)B);
  hipLaunchKernelGGL(radius_group_kernel, grid, dim3(RG_WAVES * 64), 0, as_stream(stream), pc, pb, pn, centres, cb,
                     cn, (int)N, (int)NC, d2_threshold, cap, cand, count);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Positive-point selection of the centre picker (dataset_utils/get_regiondataset.py:354-434: the
// reference masks `score > thr` and calls torch.nonzero per scene).  One workgroup per scene walks the
// points in order: index[b, 0:count[b]] = ascending ids of the positives; xyz_out (B,3,N) holds their
// coordinates in the same order, and every slot behind them repeats the FIRST positive (so a
// furthest-point-sampling launch over a common, longer prefix of all scenes gives each scene the result of
// sampling its own positives: a copy of the start point is at distance 0 from the selected set for ever).
// Two passes, two barriers (round 5; before: N / 1024 rounds of three barriers each, 370 us per call beside the matrix
// kernels): every wave owns a contiguous segment of the scene, counts its hits with ballots, the segment totals are scanned,
// and the wave walks its segment again writing each hit at (segment base + hits so far + hits in lower lanes).
#define SP_WAVES 8    // (eight waves of 40 registers fit beside two ~200-register chain waves per SIMD; sixteen needed 160 free registers)
__global__ __launch_bounds__(SP_WAVES * 64) void select_positive_kernel(const float* __restrict__ pc, int64_t pb, int64_t pn,
                                                                       const float* __restrict__ score, int64_t sb, int N,
                                                                       float thr, int64_t* __restrict__ index,
                                                                       float* __restrict__ xyz_out, int32_t* __restrict__ count) {
  __shared__ int wave_cnt[SP_WAVES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const float* p = pc + (int64_t)b * pb;
  const float* sc = score + (int64_t)b * sb;
  int64_t* idx = index + (int64_t)b * N;
  float* xo = xyz_out + (int64_t)b * 3 * N;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int seg = ((N + SP_WAVES - 1) / SP_WAVES + 63) / 64 * 64;      // segment length: whole 64-point steps
  const int lo = wave * seg, hi = lo + seg < N ? lo + seg : N;
  int mine = 0;
  for (int j0 = lo; j0 < hi; j0 += 64) {
    const int j = j0 + lane;
    mine += (int)__popcll(__ballot(j < hi && sc[j] > thr));
  }
  if (lane == 0) wave_cnt[wave] = mine;
  __syncthreads();
  int before = 0, cnt = 0;
#pragma unroll
  for (int w = 0; w < SP_WAVES; ++w) {
    const int v = wave_cnt[w];
    before += w < wave ? v : 0;
    cnt += v;
  }
  for (int j0 = lo; j0 < hi; j0 += 64) {
    const int j = j0 + lane;
    const bool hit = j < hi && sc[j] > thr;
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int pos = before + (int)__popcll(m & lt_mask);
      const float* r = p + (int64_t)j * pn;
      idx[pos] = j;
      xo[pos] = r[0];
      xo[N + pos] = r[1];
      xo[2 * N + pos] = r[2];
    }
    before += (int)__popcll(m);
  }
  if (tid == 0) count[b] = cnt;
  if (cnt == 0) return;
  __threadfence_block();
  __syncthreads();
  const float fx = xo[0], fy = xo[N], fz = xo[2 * N];   // written by this workgroup above
  for (int j = cnt + tid; j < N; j += SP_WAVES * 64) {
    xo[j] = fx;
    xo[N + j] = fy;
    xo[2 * N + j] = fz;
  }
}

extern "C" int regnet_select_positive_f32(const float* pc, int64_t pb, int64_t pn, const float* score, int64_t sb,
                                          int64_t B, int64_t N, float threshold, int64_t* index, float* xyz_out,
                                          int32_t* count, void* stream) {
  if (B < 0 || N < 0) return REGNET_ERR_SHAPE;
  if (N >= (int64_t)1 << 31 || B >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (B == 0) return REGNET_OK;
  if (!count || (N > 0 && (!pc || !score || !index || !xyz_out))) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(select_positive_kernel, dim3((unsigned)B), dim3(SP_WAVES * 64), 0, as_stream(stream), pc, pb, pn, score, sb,
                     (int)N, threshold, index, xyz_out, count);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Gripper closing-box membership.  For grasp i: t = R_i * (p - c_i) with rows of R_i =
// [approach; axis_y; minor_normal] (gripper_region_network.py:506-509); member iff
// 0 < t.x < xlim_i, |t.y| < ylim_i, |t.z| < zlim (all strict, :523-528).
__global__ __launch_bounds__(RG_WAVES * 64) void box_crop_kernel(
    const float* __restrict__ pts, int64_t gb, int64_t gn, const float* __restrict__ centre,
    const float* __restrict__ rot, const float* __restrict__ xlim, const float* __restrict__ ylim, float zlim,
    int n, int G, int32_t* __restrict__ cand, int32_t* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * RG_WAVES + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* p = pts + (int64_t)i * gb;
  const float cx = centre[i * 3 + 0], cy = centre[i * 3 + 1], cz = centre[i * 3 + 2];
  float m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = rot[(int64_t)i * 9 + k];
  const float xl = xlim[i], yl = ylim[i];
  int32_t* out = cand + (int64_t)i * G;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int cnt = 0;
  for (int j0 = 0; j0 < G; j0 += 64) {
    const int j = j0 + lane;
    bool hit = false;
    if (j < G) {
      const float* r = p + (int64_t)j * gn;
      const float dx = r[0] - cx, dy = r[1] - cy, dz = r[2] - cz;
      const float tx = (m[0] * dx + m[1] * dy) + m[2] * dz;
      const float ty = (m[3] * dx + m[4] * dy) + m[5] * dz;
      const float tz = (m[6] * dx + m[7] * dy) + m[8] * dz;
      hit = tx > 0.f && tx < xl && ty > -yl && ty < yl && tz > -zlim && tz < zlim;
    }
    const unsigned long long mask = __ballot(hit);
    if (hit) out[cnt + (int)__popcll(mask & lt_mask)] = j;
    cnt += (int)__popcll(mask);
  }
  if (lane == 0) count[i] = cnt;
}

extern "C" int regnet_box_crop_f32(const float* group_points, int64_t gb, int64_t gn, const float* centre,
                                   const float* rot, const float* xlim, const float* ylim, float zlim, int64_t n,
                                   int64_t G, int32_t* cand, int32_t* count, void* stream) {
  if (n < 0 || G < 0) return REGNET_ERR_SHAPE;
  if (G >= (int64_t)1 << 31 || n >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (n == 0) return REGNET_OK;
  if (!centre || !rot || !xlim || !ylim || !count || (G > 0 && (!group_points || !cand))) return REGNET_ERR_NULL;
  dim3 grid((unsigned)((n + RG_WAVES - 1) / RG_WAVES));
  hipLaunchKernelGGL(box_crop_kernel, grid, dim3(RG_WAVES * 64), 0, as_stream(stream), group_points, gb, gn, centre,
                     rot, xlim, ylim, zlim, (int)n, (int)G, cand, count);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Gripper frame of every predicted grasp (gripper_region_network.py:447-506): centre (n,3) and the rotation whose rows are
// [approach; axis_y; minor_normal].  One thread per grasp instead of ~35 torch launches on (n,3) tensors (0.45 ms of host
// time per batch, n = 64 per scene).  Individually rounded fp32 in the reference's order of operations (this file is built
// with -ffp-contract=off): v / (|v| + eps) with the zero-norm fallback axis, R1 = rotation by theta about y,
// approach = unit(column 0 of [axis_x axis_y axis_z] R1).
struct Vec3 { float x, y, z; };
__device__ __forceinline__ Vec3 unit3(Vec3 v, float eps, int fallback_axis) {
  const float norm = sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z) + eps;
  if (norm == 0.f) return Vec3{fallback_axis == 0 ? 1.f : 0.f, fallback_axis == 1 ? 1.f : 0.f, fallback_axis == 2 ? 1.f : 0.f};
  return Vec3{v.x / norm, v.y / norm, v.z / norm};
}
__device__ __forceinline__ Vec3 cross3(Vec3 a, Vec3 b) {
  return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

__global__ __launch_bounds__(64) void gripper_frame_kernel(const float* __restrict__ grasp, int64_t ld, int n,
                                                          float* __restrict__ centre, float* __restrict__ rot) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float* g = grasp + (int64_t)i * ld;
  centre[i * 3 + 0] = g[0]; centre[i * 3 + 1] = g[1]; centre[i * 3 + 2] = g[2];
  const float c = cosf(g[6]), s = sinf(g[6]);
  const Vec3 ay = unit3(Vec3{g[3], g[4], g[5]}, 1e-12f, 1);
  const Vec3 ax = unit3(Vec3{ay.y, -ay.x, 0.f}, 1e-12f, 0);
  const Vec3 az = unit3(cross3(ax, ay), 0.f, 2);
  // column 0 of [ax ay az] * R1, R1 = [[c,0,-s],[0,1,0],[s,0,c]]: ax * c + ay * 0 + az * s, summed in k order
  const Vec3 col0{(ax.x * c + ay.x * 0.f) + az.x * s, (ax.y * c + ay.y * 0.f) + az.y * s, (ax.z * c + ay.z * 0.f) + az.z * s};
  const Vec3 ap = unit3(col0, 1e-12f, 0);
  const Vec3 mn = cross3(ap, ay);
  float* r = rot + (int64_t)i * 9;
  r[0] = ap.x; r[1] = ap.y; r[2] = ap.z;
  r[3] = ay.x; r[4] = ay.y; r[5] = ay.z;
  r[6] = mn.x; r[7] = mn.y; r[8] = mn.z;
}

extern "C" int regnet_gripper_frame_f32(const float* grasp, int64_t ld, int64_t n, float* centre, float* rot, void* stream) {
  if (n < 0 || ld < 7) return REGNET_ERR_SHAPE;
  if (n >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (n == 0) return REGNET_OK;
  if (!grasp || !centre || !rot) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(gripper_frame_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, as_stream(stream), grasp, ld, (int)n,
                     centre, rot);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Stage-2 decode without labels (gripper_region_network.py:69-90 with `ground is None`): per centre, the arg-max anchor's
// regression becomes the grasp -- centre = delta * radius + centre point, closing axis = (delta + template) / sqrt(|.|^2 + 1e-12),
// theta = pi * (delta + template), the remaining channels passed through (their sigmoid, pointnet2.py:187, applied here when
// the head hands over raw values).  One thread per centre; every operation individually rounded (this file is compiled with
// -ffp-contract=off) in the order of the ~25 small torch kernels it replaces.
__global__ __launch_bounds__(64) void stage2_decode_kernel(const float* __restrict__ cls, const float* __restrict__ reg, int A, int C,
                                                          const float* __restrict__ centre, int64_t centre_ld,
                                                          const float* __restrict__ tmpl, float radius, int sigmoid_tail, int n,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float* cl = cls + (int64_t)i * A;
  int pick = 0;
  float best = cl[0];
  for (int a = 1; a < A; ++a)
    if (cl[a] > best) { best = cl[a]; pick = a; }          // first maximum, as torch.max(dim) returns it
  const float* r = reg + ((int64_t)i * A + pick) * C;
  const float* c = centre + (int64_t)i * centre_ld;
  const float* t = tmpl + pick * 4;
  float* o = out + (int64_t)i * C;
  o[0] = r[0] * radius + c[0];
  o[1] = r[1] * radius + c[1];
  o[2] = r[2] * radius + c[2];
  const float ax = r[3] + t[0], ay = r[4] + t[1], az = r[5] + t[2];
  const float norm = sqrtf(((ax * ax + ay * ay) + az * az) + 1e-12f);
  o[3] = ax / norm; o[4] = ay / norm; o[5] = az / norm;
  o[6] = 3.14159265358979323846f * (r[6] + t[3]);
  for (int k = 7; k < C; ++k) o[k] = sigmoid_tail ? 1.f / (1.f + expf(-r[k])) : r[k];
}

extern "C" int regnet_stage2_decode_f32(const float* cls, const float* reg, int64_t A, int64_t C, const float* centre,
                                        int64_t centre_ld, const float* tmpl, float radius, int sigmoid_tail, int64_t n,
                                        float* out, void* stream) {
  if (n < 0 || A <= 0 || C < 7 || centre_ld < 3) return REGNET_ERR_SHAPE;
  if (n >= (int64_t)1 << 31 || A >= 1024 || C >= 1024) return REGNET_ERR_UNSUPPORTED;
  if (n == 0) return REGNET_OK;
  if (!cls || !reg || !centre || !tmpl || !out) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(stage2_decode_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, as_stream(stream), cls, reg, (int)A,
                     (int)C, centre, centre_ld, tmpl, radius, sigmoid_tail, (int)n, out);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Refine decode without labels (gripper_region_network.py:201-215): final = stage-2 grasp + refine deltas (centre deltas times
// radius), predicted class = arg-max of the two class scores, flags[0] = class 1, flags[1] = class 1 and final score channel 7
// above the threshold.  One thread per grasp.
__global__ __launch_bounds__(64) void refine_decode_kernel(const float* __restrict__ grasp, int64_t grasp_ld,
                                                          const float* __restrict__ cls, const float* __restrict__ reg, int C,
                                                          float radius, float score_thre, int n, float* __restrict__ final_grasp,
                                                          uint8_t* __restrict__ flags) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float* g = grasp + (int64_t)i * grasp_ld;
  const float* r = reg + (int64_t)i * C;
  float* o = final_grasp + (int64_t)i * C;
  for (int k = 0; k < 3; ++k) o[k] = g[k] + r[k] * radius;
  for (int k = 3; k < C; ++k) o[k] = g[k] + r[k];
  const bool one = cls[2 * i + 1] > cls[2 * i];            // torch.max(dim=-1)[1]: the first maximum, so class 0 on a tie
  flags[i] = one ? 1 : 0;
  flags[n + i] = (one && o[7] > score_thre) ? 1 : 0;
}

extern "C" int regnet_refine_decode_f32(const float* grasp, int64_t grasp_ld, const float* cls, const float* reg, int64_t C,
                                        float radius, float score_thre, int64_t n, float* final_grasp, uint8_t* flags,
                                        void* stream) {
  if (n < 0 || C < 8 || grasp_ld < C) return REGNET_ERR_SHAPE;
  if (n >= (int64_t)1 << 30 || C >= 1024) return REGNET_ERR_UNSUPPORTED;
  if (n == 0) return REGNET_OK;
  if (!grasp || !cls || !reg || !final_grasp || !flags) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(refine_decode_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, as_stream(stream), grasp, grasp_ld, cls,
                     reg, (int)C, radius, score_thre, (int)n, final_grasp, flags);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// The drawn positions of a box crop resolved to indices (gripper_region_network.py:540-548): index[i][r] = the position
// inside the group of the r-th drawn candidate, index_inall[i][r] = that member's index in the scene, both -1 for a grasp
// without a valid crop (<= 5 points in the box).  One launch instead of 2 gathers, 3 `where`s and their temporaries.
__global__ __launch_bounds__(256) void crop_pick_kernel(const int32_t* __restrict__ cand, int G, const int64_t* __restrict__ pos,
                                                       int R, const uint8_t* __restrict__ valid,
                                                       const int64_t* __restrict__ group_index, int64_t gi_stride, long long total,
                                                       int64_t* __restrict__ index, int64_t* __restrict__ index_inall) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long i = t / R;
  int64_t a = -1, b = -1;
  if (valid[i]) {
    const int64_t p = pos[t];
    if (p >= 0 && p < G) {
      a = cand[i * G + p];
      if (a >= 0 && a < G) b = group_index[i * gi_stride + a];
    }
  }
  index[t] = a;
  index_inall[t] = b;
}

extern "C" int regnet_crop_pick(const int32_t* cand, int64_t G, const int64_t* pos, int64_t R, const uint8_t* valid,
                                const int64_t* group_index, int64_t gi_stride, int64_t n, int64_t* index,
                                int64_t* index_inall, void* stream) {
  if (n < 0 || G <= 0 || R <= 0) return REGNET_ERR_SHAPE;
  if (G >= (int64_t)1 << 31 || R >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (n == 0) return REGNET_OK;
  if (!cand || !pos || !valid || !group_index || !index || !index_inall) return REGNET_ERR_NULL;
  const long long total = (long long)n * R;
  hipLaunchKernelGGL(crop_pick_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), cand, (int)G, pos,
                     (int)R, valid, group_index, gi_stride, total, index, index_inall);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Collision scan of predicted grasps against the view cloud (test.py:147 -> utils.py:391-401 -> eval_score/eval.py:4-12
// -> evaluation_data_generator.py:188-236, EvalDataTest.finger_hand_view).  The reference loops over the grasps in
// Python and, for each, multiplies the whole cloud by the grasp's 4x4 global->local matrix and builds five boolean
// masks with torch ops (~25 launches and a host sync per grasp; 4000 grasps per scene in test.py).  Here: one workgroup
// per grasp walks the cloud once and counts the points (0) in the closing slab -bottom < x < depth, (1) behind the hand,
// (2) inside either finger -- the three numbers the reference's thresholds are applied to.  Arithmetic: individually
// rounded fp32 in source order, x = ((t00*px + t01*py) + t02*pz) + t03 (this file is built with -ffp-contract=off), the
// same as oracle/collision_oracle.py.
#define GC_T 256

struct GraspBox {   // the reference's box constants as float32 (what torch compares a float32 tensor against)
  float x_lo, x_hi, half_thickness, half_width, half_space, back_x;
};

// local coordinates of point j for the matrix rows held in registers; returns false when outside the closing slab
#define GC_LOCAL(j)                                                          \
  const float* p = points + (int64_t)(j) * pn;                               \
  const float px = p[0], py = p[pc], pz = p[2 * pc];                         \
  const float x = ((t00 * px + t01 * py) + t02 * pz) + t03;                  \
  const bool close = x > box.x_lo && x < x_hi;                               \
  const float y = ((t10 * px + t11 * py) + t12 * pz) + t13;                  \
  const float z = ((t20 * px + t21 * py) + t22 * pz) + t23;                  \
  const bool zc = close && z < box.half_thickness && z > -box.half_thickness;

__device__ __forceinline__ int block_sum_int(int v, int* red) {   // red: GC_T / 64 ints of LDS; all threads get the sum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int s = 0;
#pragma unroll
  for (int w = 0; w < GC_T / 64; ++w) s += red[w];
  return s;
}

// counts[b] = { closing slab, behind the hand, inside a finger, between the fingers (closing region) }
__global__ __launch_bounds__(GC_T) void grasp_collision_kernel(const float* __restrict__ points, int64_t pn, int64_t pc,
                                                               int N, const float* __restrict__ T, GraspBox box,
                                                               const float* __restrict__ x_hi_per_grasp,
                                                               int32_t* __restrict__ counts) {
  __shared__ int red[GC_T / 64];
  const float* t = T + (int64_t)blockIdx.x * 16;
  const float t00 = t[0], t01 = t[1], t02 = t[2], t03 = t[3];
  const float t10 = t[4], t11 = t[5], t12 = t[6], t13 = t[7];
  const float t20 = t[8], t21 = t[9], t22 = t[10], t23 = t[11];
  const float x_hi = x_hi_per_grasp ? x_hi_per_grasp[blockIdx.x] : box.x_hi;
  int c_close = 0, c_back = 0, c_finger = 0, c_region = 0;
  for (int j = threadIdx.x; j < N; j += GC_T) {
    GC_LOCAL(j)
    const bool back = zc && y < box.half_width && y > -box.half_width && x < box.back_x;
    const bool finger = zc && ((y < box.half_width && y > box.half_space) || (y > -box.half_width && y < -box.half_space));
    const bool region = zc && y < box.half_space && y > -box.half_space;
    c_close += close;
    c_back += back;
    c_finger += finger;
    c_region += region;
  }
  c_close = block_sum_int(c_close, red);
  c_back = block_sum_int(c_back, red);
  c_finger = block_sum_int(c_finger, red);
  c_region = block_sum_int(c_region, red);
  if (threadIdx.x == 0) {
    int32_t* o = counts + (int64_t)blockIdx.x * 4;
    o[0] = c_close; o[1] = c_back; o[2] = c_finger; o[3] = c_region;
  }
}

// Antipodal statistics of the closing region against a cloud with normals (evaluation_data_generator.py:392-418 on the
// region selected at :521-534): y extent of the region, then |n_y| (normal in the grasp frame) summed over the points
// within `depth = min((y_max - y_min) / 3, neighbour_depth)` of either extreme.
// stats[b] = { y_max, y_min, sum_left, sum_right }, side_counts[b] = { n_left, n_right }.
__global__ __launch_bounds__(GC_T) void grasp_antipodal_kernel(const float* __restrict__ points, int64_t pn, int64_t pc,
                                                               const float* __restrict__ normals, int64_t nn, int64_t nc,
                                                               int N, const float* __restrict__ T, GraspBox box,
                                                               const float* __restrict__ x_hi_per_grasp,
                                                               float neighbour_depth, float* __restrict__ stats,
                                                               int32_t* __restrict__ side_counts) {
  __shared__ float redf[2][GC_T / 64];
  __shared__ double redd[2][GC_T / 64];
  __shared__ int red[GC_T / 64];
  const float* t = T + (int64_t)blockIdx.x * 16;
  const float t00 = t[0], t01 = t[1], t02 = t[2], t03 = t[3];
  const float t10 = t[4], t11 = t[5], t12 = t[6], t13 = t[7];
  const float t20 = t[8], t21 = t[9], t22 = t[10], t23 = t[11];
  const float x_hi = x_hi_per_grasp ? x_hi_per_grasp[blockIdx.x] : box.x_hi;
  float ymax = -__builtin_inff(), ymin = __builtin_inff();
  for (int j = threadIdx.x; j < N; j += GC_T) {
    GC_LOCAL(j)
    if (zc && y < box.half_space && y > -box.half_space) { ymax = fmaxf(ymax, y); ymin = fminf(ymin, y); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { ymax = fmaxf(ymax, __shfl_xor(ymax, o)); ymin = fminf(ymin, __shfl_xor(ymin, o)); }
  if ((threadIdx.x & 63) == 0) { redf[0][threadIdx.x >> 6] = ymax; redf[1][threadIdx.x >> 6] = ymin; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < GC_T / 64; ++w) { ymax = fmaxf(ymax, redf[0][w]); ymin = fminf(ymin, redf[1][w]); }
  const float third = (ymax - ymin) / 3.0f;
  const float depth = third < neighbour_depth ? third : neighbour_depth;   // torch.min(a, b)
  const float left_edge = ymax - depth, right_edge = ymin + depth;
  double s_left = 0.0, s_right = 0.0;
  int n_left = 0, n_right = 0;
  for (int j = threadIdx.x; j < N; j += GC_T) {
    GC_LOCAL(j)
    if (!(zc && y < box.half_space && y > -box.half_space)) continue;
    const float* q = normals + (int64_t)j * nn;
    const float ny = fabsf((t10 * q[0] + t11 * q[nc]) + t12 * q[2 * nc]);
    if (y > left_edge) { s_left += (double)ny; n_left += 1; }
    if (y < right_edge) { s_right += (double)ny; n_right += 1; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s_left += __shfl_xor(s_left, o); s_right += __shfl_xor(s_right, o); }
  if ((threadIdx.x & 63) == 0) { redd[0][threadIdx.x >> 6] = s_left; redd[1][threadIdx.x >> 6] = s_right; }
  n_left = block_sum_int(n_left, red);     // (its barriers also publish redd)
  n_right = block_sum_int(n_right, red);
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int w = 0; w < GC_T / 64; ++w) { a += redd[0][w]; b += redd[1][w]; }
    float* o = stats + (int64_t)blockIdx.x * 4;
    o[0] = ymax; o[1] = ymin; o[2] = (float)a; o[3] = (float)b;
    side_counts[(int64_t)blockIdx.x * 2] = n_left;
    side_counts[(int64_t)blockIdx.x * 2 + 1] = n_right;
  }
}

static inline int grasp_args_ok(int64_t N, int64_t B) {
  if (N < 0 || B < 0) return REGNET_ERR_SHAPE;
  if (N >= (int64_t)1 << 31 || B >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  return REGNET_OK;
}

extern "C" int regnet_grasp_collision_counts_f32(const float* points, int64_t pn, int64_t pc, int64_t N, const float* T,
                                                 int64_t B, float x_lo, float x_hi, const float* x_hi_per_grasp,
                                                 float half_thickness, float half_width, float half_space, float back_x,
                                                 int32_t* counts, void* stream) {
  const int rc = grasp_args_ok(N, B);
  if (rc != REGNET_OK) return rc;
  if (B == 0) return REGNET_OK;
  if (!T || !counts || (N > 0 && !points)) return REGNET_ERR_NULL;
  const GraspBox box{x_lo, x_hi, half_thickness, half_width, half_space, back_x};
  hipLaunchKernelGGL(grasp_collision_kernel, dim3((unsigned)B), dim3(GC_T), 0, as_stream(stream), points, pn, pc, (int)N, T,
                     box, x_hi_per_grasp, counts);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_grasp_antipodal_stats_f32(const float* points, int64_t pn, int64_t pc, const float* normals, int64_t nn,
                                                int64_t nc, int64_t N, const float* T, int64_t B, float x_lo, float x_hi,
                                                const float* x_hi_per_grasp, float half_thickness, float half_width,
                                                float half_space, float back_x, float neighbour_depth, float* stats,
                                                int32_t* side_counts, void* stream) {
  const int rc = grasp_args_ok(N, B);
  if (rc != REGNET_OK) return rc;
  if (B == 0) return REGNET_OK;
  if (!T || !stats || !side_counts || (N > 0 && (!points || !normals))) return REGNET_ERR_NULL;
  const GraspBox box{x_lo, x_hi, half_thickness, half_width, half_space, back_x};
  hipLaunchKernelGGL(grasp_antipodal_kernel, dim3((unsigned)B), dim3(GC_T), 0, as_stream(stream), points, pn, pc, normals,
                     nn, nc, (int)N, T, box, x_hi_per_grasp, neighbour_depth, stats, side_counts);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// out[r, ch] = max_g feat[rows[r, g], ch]: the grouped-feature gather fused with MaxPool1d(G).
// feat is (num_rows, F) row-major (= all_feature.view(B*N, F)); one workgroup per output row,
// threads across channels so every gathered row is one coalesced F*4-byte read.  (F not a multiple of 4 or unaligned rows.)
__global__ __launch_bounds__(256) void gather_max_kernel(const float* __restrict__ feat, int64_t num_rows, int F,
                                                        const int64_t* __restrict__ rows, int G,
                                                        float* __restrict__ out) {
  const int64_t r = blockIdx.x;
  const int64_t* idx = rows + r * G;
  for (int ch = threadIdx.x; ch < F; ch += 256) {
    float m = -__builtin_inff();
    for (int g = 0; g < G; ++g) {
      const int64_t row = idx[g];
      if (row >= 0 && row < num_rows) m = fmaxf(m, feat[row * F + ch]);
    }
    out[r * F + ch] = m;
  }
}

// The same for 16-byte rows (F % 4 == 0, aligned feat), round 5: a lane reads a float4, a wave one 1 KB row piece, the four
// waves of a workgroup take every fourth group member with FOUR rows in flight each (16 independent 1 KB reads per workgroup,
// 8 workgroups per CU), and the waves' partial maxima meet in 4 KB of LDS.  A maximum does not depend on the order: same bits.
// Optional addressing by scene: row id rid = row_ids ? row_ids[r] : r selects the index list rows[rid]; its entries are LOCAL
// to scene rid / per_scene and scene_stride rows apart (per_scene == 0: global row ids as above) -- the reference's
// `index + b * N` (gripper_region_network.py:388, :334) formed in the address instead of as an (R, G) int64 tensor.
__global__ __launch_bounds__(256) void gather_max_v4_kernel(const float* __restrict__ feat, int64_t num_rows, int F,
                                                           const int64_t* __restrict__ rows, const int64_t* __restrict__ row_ids,
                                                           int G, int64_t per_scene, int64_t scene_stride,
                                                           float* __restrict__ out) {
  __shared__ float4 part[4][64];
  const int64_t r = blockIdx.x;
  const int64_t rid = row_ids ? row_ids[r] : r;
  const int64_t* idx = rows + rid * G;
  const int64_t base = per_scene > 0 ? (rid / per_scene) * scene_stride : 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float ninf = -__builtin_inff();
  for (int c0 = 0; c0 < F; c0 += 256) {
    const int ch = c0 + 4 * lane;
    float4 m = make_float4(ninf, ninf, ninf, ninf);
    if (ch < F) {
      for (int g = wave; g < G; g += 16) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int gg = g + 4 * u;
          int64_t row = gg < G ? idx[gg] : -1;
          if (row >= 0) row += base;
          v[u] = (row >= 0 && row < num_rows) ? *reinterpret_cast<const float4*>(feat + row * F + ch)
                                              : make_float4(ninf, ninf, ninf, ninf);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          m.x = fmaxf(m.x, v[u].x); m.y = fmaxf(m.y, v[u].y); m.z = fmaxf(m.z, v[u].z); m.w = fmaxf(m.w, v[u].w);
        }
      }
    }
    part[wave][lane] = m;
    __syncthreads();
    {
      const int t = threadIdx.x;                                  // channel c0 + t: component t & 3 of lane t >> 2
      const float* p0 = reinterpret_cast<const float*>(&part[0][0]);
      float y = fmaxf(fmaxf(p0[t], p0[256 + t]), fmaxf(p0[512 + t], p0[768 + t]));
      if (c0 + t < F) out[r * F + c0 + t] = y;
    }
    __syncthreads();
  }
}

static int launch_gather_max(const float* feat, int64_t num_rows, int64_t F, const int64_t* rows, const int64_t* row_ids,
                             int64_t R, int64_t G, int64_t per_scene, int64_t scene_stride, float* out, void* stream) {
  if (num_rows < 0 || F < 0 || R < 0 || G <= 0 || per_scene < 0 || scene_stride < 0) return REGNET_ERR_SHAPE;
  if (F >= (int64_t)1 << 31 || G >= (int64_t)1 << 31 || R >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (R == 0 || F == 0) return REGNET_OK;
  if (!feat || !rows || !out) return REGNET_ERR_NULL;
  const bool v4 = (F % 4 == 0) && ((reinterpret_cast<uintptr_t>(feat) & 15) == 0);
  if (!v4) {
    if (row_ids || per_scene > 0) return REGNET_ERR_UNSUPPORTED;    // (the scene-addressed form exists for 16-byte rows only)
    hipLaunchKernelGGL(gather_max_kernel, dim3((unsigned)R), dim3(256), 0, as_stream(stream), feat, num_rows, (int)F,
                       rows, (int)G, out);
  } else {
    hipLaunchKernelGGL(gather_max_v4_kernel, dim3((unsigned)R), dim3(256), 0, as_stream(stream), feat, num_rows, (int)F,
                       rows, row_ids, (int)G, per_scene, scene_stride, out);
  }
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_gather_max_f32(const float* feat, int64_t num_rows, int64_t F, const int64_t* rows, int64_t R,
                                     int64_t G, float* out, void* stream) {
  return launch_gather_max(feat, num_rows, F, rows, nullptr, R, G, 0, 0, out, stream);
}

extern "C" int regnet_gather_max_scene_f32(const float* feat, int64_t num_rows, int64_t F, const int64_t* rows,
                                           const int64_t* row_ids, int64_t R, int64_t G, int64_t per_scene,
                                           int64_t scene_stride, float* out, void* stream) {
  if (per_scene <= 0) return REGNET_ERR_SHAPE;
  return launch_gather_max(feat, num_rows, F, rows, row_ids, R, G, per_scene, scene_stride, out, stream);
}

// Training twin of gather_max_kernel: also records WHICH row gave the maximum (the backward scatters R x F values instead
// of going through the reference's materialised (R, G, F) gather and its zero-filled gradient), and a negative row id
// counts from the end like the reference's advanced indexing does (all_feature.view(-1, F)[index]).
__global__ __launch_bounds__(256) void gather_max_arg_kernel(const float* __restrict__ feat, int64_t num_rows, int F,
                                                            const int64_t* __restrict__ rows, int G,
                                                            float* __restrict__ out, int64_t* __restrict__ arg) {
  const int64_t r = blockIdx.x;
  const int64_t* idx = rows + r * G;
  for (int ch = threadIdx.x; ch < F; ch += 256) {
    float m = -__builtin_inff();
    int64_t am = -1;
    for (int g = 0; g < G; ++g) {
      int64_t row = idx[g];
      if (row < 0) row += num_rows;
      if (row >= 0 && row < num_rows) {
        const float v = feat[row * F + ch];
        if (v > m || am < 0) { m = v; am = row; }     // first maximum wins, as torch.max over the group axis
      }
    }
    out[r * F + ch] = m;
    arg[r * F + ch] = am;
  }
}

extern "C" int regnet_gather_max_arg_f32(const float* feat, int64_t num_rows, int64_t F, const int64_t* rows, int64_t R,
                                         int64_t G, float* out, int64_t* arg, void* stream) {
  if (num_rows < 0 || F < 0 || R < 0 || G <= 0) return REGNET_ERR_SHAPE;
  if (F >= (int64_t)1 << 31 || G >= (int64_t)1 << 31 || R >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (R == 0 || F == 0) return REGNET_OK;
  if (!feat || !rows || !out || !arg) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(gather_max_arg_kernel, dim3((unsigned)R), dim3(256), 0, as_stream(stream), feat, num_rows, (int)F,
                     rows, (int)G, out, arg);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// Backward of gather_max_arg_kernel: grad[arg[r][f]][f] += dy[r][f] (rows that gave a maximum; arg < 0: an empty group).  The
// destination is addressed as row (scene b = row / scene_rows, point n = row % scene_rows) -> b * batch_stride + n * row_stride
// + f * ch_stride, so the same kernel adds into a (rows, F) matrix (row_stride F, ch_stride 1) or straight into the
// channel-first (B, F, N) gradient of the feature map (batch_stride F N, row_stride 1, ch_stride N) -- no (rows, F) gradient,
// no transpose of it.  R x F float atomics (a few hundred thousand): groups of neighbouring centres share points.
__global__ __launch_bounds__(256) void scatter_max_grad_kernel(const float* __restrict__ dy, const int64_t* __restrict__ arg,
                                                              long long total, int F, long long scene_rows,
                                                              long long batch_stride, long long row_stride,
                                                              long long ch_stride, float* __restrict__ grad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long row = arg[i];
  if (row < 0) return;
  const int f = (int)(i % F);
  const long long b = row / scene_rows, n = row - b * scene_rows;
  unsafeAtomicAdd(grad + b * batch_stride + n * row_stride + (long long)f * ch_stride, dy[i]);
}

extern "C" int regnet_scatter_max_grad_f32(const float* dy, const int64_t* arg, int64_t R, int64_t F, int64_t scene_rows,
                                           int64_t batch_stride, int64_t row_stride, int64_t ch_stride, float* grad,
                                           void* stream) {
  if (R < 0 || F < 0 || scene_rows <= 0) return REGNET_ERR_SHAPE;
  if (R == 0 || F == 0) return REGNET_OK;
  if (!dy || !arg || !grad) return REGNET_ERR_NULL;
  const long long total = (long long)R * F, blocks = (total + 255) / 256;
  if (blocks >= (1ll << 31) || F >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(scatter_max_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dy, arg, total, (int)F,
                     (long long)scene_rows, (long long)batch_stride, (long long)row_stride, (long long)ch_stride, grad);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// out[r] = -(x[r][0] + ... + x[r][K-1]): the gradient of the per-centre term V of a pre-multiplied first layer
// (pn2_utils/modules.py: Y = U[nbr] - V, so dV = -sum over the K neighbours of dY).  One wave per 64 / K ... rows of K
// contiguous floats; torch's own `neg` + 4-D `sum` took 0.78 ms for the level-2 block's 537 MB gradient (0.7 TB/s).
__global__ __launch_bounds__(256) void rowsum_neg_kernel(const float* __restrict__ x, long long rows, int K,
                                                        float* __restrict__ out) {
  // K is a multiple of 4 and <= 256: K / 4 lanes per row, float4 loads
  const int lpr = K / 4;
  const int rows_per_block = 256 / lpr;
  const long long r = (long long)blockIdx.x * rows_per_block + threadIdx.x / lpr;
  const int c = threadIdx.x % lpr;
  float s = 0.f;
  if (r < rows && threadIdx.x < rows_per_block * lpr) {
    const float4 v = *reinterpret_cast<const float4*>(x + r * K + 4 * c);
    s = (v.x + v.y) + (v.z + v.w);
  }
  for (int off = 1; off < lpr; off <<= 1) s += __shfl_xor(s, off, 64);    // lpr is a power of two <= 64
  if (c == 0 && r < rows && threadIdx.x < rows_per_block * lpr) out[r] = -s;
}

extern "C" int regnet_rowsum_neg_f32(const float* x, int64_t rows, int64_t K, float* out, void* stream) {
  if (rows < 0 || K < 4 || K > 256 || (K & (K - 1))) return REGNET_ERR_SHAPE;
  if (rows == 0) return REGNET_OK;
  if (!x || !out) return REGNET_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(x) & 15)) return REGNET_ERR_SHAPE;
  const long long per_block = 256 / (K / 4);
  const long long blocks = (rows + per_block - 1) / per_block;
  if (blocks >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(rowsum_neg_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, (long long)rows, (int)K, out);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

// ---- resampled groups ---------------------------------------------------------------------------------------------------
// get_regiondataset.py:331-352: every (scene, centre) candidate list is resampled to exactly G entries at positions drawn
// on the host (numpy's RNG), then the member indices and their points are gathered; a centre without candidates gets -1
// everywhere.  One thread per (scene, centre, slot): replaces clamp + gather + cast + 2 x where + gather + where.
__global__ __launch_bounds__(256) void resample_groups_kernel(const float* __restrict__ pc, int64_t pb, int64_t pn, int C,
                                                              const int* __restrict__ cand, int64_t cap,
                                                              const int64_t* __restrict__ pos, long long total, int G,
                                                              int64_t groups_per_scene, int64_t N, int* __restrict__ bad,
                                                              int64_t* __restrict__ index,
                                                              float* __restrict__ points) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long grp = t / G;               // scene * groups_per_scene + centre
  const int64_t p = pos[t];
  float* o = points + t * C;
  if (p < 0) {
    index[t] = -1;
    for (int c = 0; c < C; ++c) o[c] = -1.0f;
    return;
  }
  // (the torch.gather this kernel replaced raised on an out-of-range position; a position beyond the candidate list or
  //  a candidate beyond the cloud -- a count / capacity mismatch upstream -- is flagged and reads nothing)
  const int64_t j = p < cap ? (int64_t)cand[grp * cap + p] : -1;
  if (j < 0 || j >= N) {
    if (bad) atomicOr(bad, 1);
    index[t] = -1;
    for (int c = 0; c < C; ++c) o[c] = -1.0f;
    return;
  }
  index[t] = j;
  const float* src = pc + (grp / groups_per_scene) * pb + j * pn;
  for (int c = 0; c < C; ++c) o[c] = src[c];
}

extern "C" int regnet_resample_groups_f32(const float* pc, int64_t pb, int64_t pn, int64_t C, const int32_t* cand,
                                          int64_t cap, const int64_t* pos, int64_t B, int64_t Nc, int64_t G, int64_t N,
                                          int32_t* out_of_range, int64_t* index, float* points, void* stream) {
  if (B < 0 || Nc < 0 || G < 0 || C <= 0 || cap < 0) return REGNET_ERR_SHAPE;
  const long long total = (long long)B * Nc * G;
  if (total >= (1ll << 40) || G >= (int64_t)1 << 31 || C > 64) return REGNET_ERR_UNSUPPORTED;
  if (total == 0) return REGNET_OK;
  if (!pc || !cand || !pos || !index || !points) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(resample_groups_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), pc, pb,
                     pn, (int)C, cand, cap, pos, total, (int)G, Nc, N, out_of_range, index, points);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
