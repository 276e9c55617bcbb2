// losses.hip -- the two grasp losses of the TRAINING iteration as a handful of launches (with their gradients).
//
// Reference behaviour restated (paths relative to /root/reference/multi_model): GripperRegionNetwork.compute_loss with labels
// (gripper_region_network.py:46-184) and compute_loss_refine with labels (:186-309) are ~160 and ~200 small tensor operations
// (+ as many again under autograd) on a few hundred rows -- decode, cosine similarities, four smooth-L1 terms, a class-balanced
// cross entropy, monitoring terms.  On a GPU they are launch-bound and sit on the host-paced critical path of the iteration
// (DESIGN.md par. 12.5).  Here each loss is: ONE row kernel that computes everything that depends on a row alone -- decodes,
// picks, per-row loss and monitoring terms AND the gradient of the row's regression terms -- one device->host read for the
// decisions the reference takes on the host anyway (class balancing draws from numpy's stream), ONE cross-entropy kernel over
// the drawn rows, one column sum.  Same formulas, fp32, every operation individually rounded (-ffp-contract=off).
//
// smooth-L1 is torch's default (beta = 1): 0.5 x^2 for |x| < 1, |x| - 0.5 otherwise; its derivative x resp. sign(x).
#include "common.h"

__device__ __forceinline__ float sl1(float x) { const float a = fabsf(x); return a < 1.f ? 0.5f * x * x : a - 0.5f; }
__device__ __forceinline__ float sl1_grad(float x) { return fabsf(x) < 1.f ? x : (x > 0.f ? 1.f : -1.f); }
// 1 - cos(a, b) as compute_cos_sim (gripper_region_network.py:589-610) and torch's cosine_embedding_loss (target 1) write it:
// dot / sqrt((|a|^2 + 1e-12) (|b|^2 + 1e-12))
__device__ __forceinline__ float one_minus_cos(const float* a, const float* b) {
  const float dot = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
  const float na = ((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]) + 1e-12f;
  const float nb = ((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]) + 1e-12f;
  return 1.f - dot / sqrtf(na * nb);
}

#define S2_TERMS 12   // per-row terms of the stage-2 loss: see stage2_loss_rows_kernel

// Stage-2 loss, per labelled centre i (gripper_region_network.py:69-90, :92-181):
//   pick = arg-max class (first maximum), next_grasp = its decoded regression (as regnet_stage2_decode_f32);
//   g8   = the anchor whose template axis is closest (1 - cos, first minimum) to the label's axis;
//   with g = reg[i, g8, :], a = anchor g8 (centre | template), n = sqrt(|g[3:6] + a[3:6]|^2 + 1e-12):
//     terms[0] = sum_3 SL1(g[0:3] - (gt[0:3] - a[0:3]) / radius)      terms[1] = sum_3 SL1(g[3:6] * n - (gt[3:6] - a[3:6]))
//     terms[2] = SL1(g[6] - (gt[6] - a[6]) / pi)                      terms[3] = sum_3 SL1(g[7:10] - gt[7:10])
//   monitoring of the arg-max decode against the label: terms[4] = sum_3 SL1(centre), [5] = 1 - cos(axis), [6] = SL1(theta),
//   [7] = sum_3 SL1(score);  terms[8] = (g8 == pick);  terms[9..11] = 0 (the cross entropy is added by ce_rows_kernel).
//   dreg[i, a, :] = d(w0 terms[0] + w1 terms[1] + w2 terms[2] + w3 terms[3]) / d reg[i, a, :]  (zero for a != g8).
__global__ __launch_bounds__(64) void stage2_loss_rows_kernel(const float* __restrict__ cls, const float* __restrict__ reg, int A, int C,
                                                             const float* __restrict__ centre, int64_t centre_ld,
                                                             const float* __restrict__ tmpl, const float* __restrict__ label,
                                                             int64_t label_ld, float radius, float w0, float w1, float w2,
                                                             float w3, const int64_t* __restrict__ rows, int m,
                                                             float* __restrict__ next_grasp,
                                                             int32_t* __restrict__ pick_out, int32_t* __restrict__ g8_out,
                                                             float* __restrict__ a_gt, float* __restrict__ terms,
                                                             float* __restrict__ dreg) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= m) return;
  const int64_t row = rows ? rows[i] : i;         // the labelled centre's row in cls / reg / centre / label / dreg; outputs are compact
  const float* cl = cls + row * A;
  const float* c = centre + row * centre_ld;
  const float* gt = label + row * label_ld;
  int pick = 0;
  float best = cl[0];
  for (int a = 1; a < A; ++a)
    if (cl[a] > best) { best = cl[a]; pick = a; }
  int g8 = 0;
  float smin = one_minus_cos(tmpl, gt + 3);
  for (int a = 1; a < A; ++a) {
    const float sa = one_minus_cos(tmpl + 4 * a, gt + 3);
    if (sa < smin) { smin = sa; g8 = a; }
  }
  pick_out[i] = pick;
  g8_out[i] = g8;
  float* t = terms + (int64_t)i * S2_TERMS;
  // ---- arg-max decode (next_grasp) and its monitoring terms
  {
    const float* r = reg + (row * A + pick) * C;
    const float* tp = tmpl + pick * 4;
    float* o = next_grasp + (int64_t)i * C;
    o[0] = r[0] * radius + c[0]; o[1] = r[1] * radius + c[1]; o[2] = r[2] * radius + c[2];
    const float ax = r[3] + tp[0], ay = r[4] + tp[1], az = r[5] + tp[2];
    const float norm = sqrtf(((ax * ax + ay * ay) + az * az) + 1e-12f);
    o[3] = ax / norm; o[4] = ay / norm; o[5] = az / norm;
    o[6] = 3.14159265358979323846f * (r[6] + tp[3]);
    for (int k = 7; k < C; ++k) o[k] = r[k];
    t[4] = (sl1(o[0] - gt[0]) + sl1(o[1] - gt[1])) + sl1(o[2] - gt[2]);
    t[5] = one_minus_cos(o + 3, gt + 3);
    t[6] = sl1(o[6] - gt[6]);
    t[7] = (sl1(o[7] - gt[7]) + sl1(o[8] - gt[8])) + sl1(o[9] - gt[9]);
  }
  t[8] = g8 == pick ? 1.f : 0.f;
  t[9] = t[10] = t[11] = 0.f;
  // ---- the label's anchor: regression terms and their gradient
  const float* g = reg + (row * A + g8) * C;
  const float* tp = tmpl + g8 * 4;
  float* ag = a_gt + (int64_t)i * 7;
  ag[0] = c[0]; ag[1] = c[1]; ag[2] = c[2]; ag[3] = tp[0]; ag[4] = tp[1]; ag[5] = tp[2]; ag[6] = tp[3];
  float* d = dreg + row * A * C;
  for (int k = 0; k < A * C; ++k) d[k] = 0.f;
  d += g8 * C;
  float s0 = 0.f, s1 = 0.f, s3 = 0.f;
  for (int k = 0; k < 3; ++k) {
    const float e = g[k] - (gt[k] - c[k]) / radius;
    s0 += sl1(e);
    d[k] = w0 * sl1_grad(e);
  }
  const float ax = g[3] + tp[0], ay = g[4] + tp[1], az = g[5] + tp[2];
  const float n = sqrtf(((ax * ax + ay * ay) + az * az) + 1e-12f);
  const float axis[3] = {ax, ay, az};
  float ge[3], gdot = 0.f;            // SL1'(e_k) and sum_k SL1'(e_k) g_k
  for (int k = 0; k < 3; ++k) {
    const float e = g[3 + k] * n - (gt[3 + k] - tp[k]);
    s1 += sl1(e);
    ge[k] = sl1_grad(e);
    gdot += ge[k] * g[3 + k];
  }
  for (int k = 0; k < 3; ++k) d[3 + k] = w1 * (ge[k] * n + gdot * (axis[k] / n));
  const float e6 = g[6] - (gt[6] - tp[3]) / 3.14159265358979323846f;
  d[6] = w2 * sl1_grad(e6);
  for (int k = 7; k < 10; ++k) {
    const float e = g[k] - gt[k];
    s3 += sl1(e);
    d[k] = w3 * sl1_grad(e);
  }
  t[0] = s0; t[1] = s1; t[2] = sl1(e6); t[3] = s3;
}

// Cross entropy over drawn rows (torch.nn.CrossEntropyLoss, reduction mean, gripper_region_network.py:131, :262): j = idx[k] in
// the compact numbering, r = rows[j] (or j) its row of cls (., A): loss[k] = logsumexp(cls[r]) - cls[r, target[j]];
// dcls[r, :] = (softmax - onehot) * scale (the drawn rows are distinct: without replacement per class).  dcls must be
// zero-filled by the caller.
__global__ __launch_bounds__(64) void ce_rows_kernel(const float* __restrict__ cls, int A, const int32_t* __restrict__ target,
                                                    const int64_t* __restrict__ idx, const int64_t* __restrict__ rows, int nb,
                                                    float scale, float* __restrict__ loss, float* __restrict__ dcls) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= nb) return;
  const int64_t j = idx[k];                       // position in the compact (labelled) numbering: target[j]
  const int64_t r = rows ? rows[j] : j;           // its row in cls / dcls
  const float* x = cls + r * A;
  float mx = x[0];
  for (int a = 1; a < A; ++a) mx = fmaxf(mx, x[a]);
  float se = 0.f;
  for (int a = 0; a < A; ++a) se += expf(x[a] - mx);
  const float lse = logf(se) + mx;
  const int tg = target[j];
  loss[k] = lse - x[tg];
  for (int a = 0; a < A; ++a) dcls[r * A + a] = (expf(x[a] - lse) - (a == tg ? 1.f : 0.f)) * scale;
}

#define RF_TERMS 20   // per-row terms of the refine loss: see refine_loss_rows_kernel

// Refine loss, per valid crop i (gripper_region_network.py:201-231, :259-296):
//   final = grasp + deltas (first three times radius); class = arg-max of the two scores (class 0 on a tie);
//   flags[0] = class 1, flags[1] = class 1 and final[7] > score_thre,
//   flags[2] = label-positive: |grasp centre - gt centre| < 0.025 and 1 - cos(axes) < 0.5 and |theta - gt theta| < 1.047.
//   regression terms of a label-positive row (zero otherwise), e = reg - target:
//     terms[0] = sum_3 SL1(reg[0:3] - (gt - grasp)[0:3] / radius)   terms[1] = sum_3 SL1(reg[3:6] - (gt - grasp)[3:6])
//     terms[2] = SL1(reg[6] - (gt - grasp)[6])                      terms[3] = sum_3 SL1(reg[7:10] - (gt - grasp)[7:10])
//   monitoring against the label -- stage-2 grasp of class-1 rows: terms[4..7] = (sum_3 SL1 centre, 1 - cos axis, SL1 theta,
//   sum_3 SL1 score); final grasp of class-1 rows: terms[8..11]; final grasp of score-kept rows: terms[12..15];
//   terms[16..19] = (gt 1 & class 1, gt 0 & class 0, gt 0 & class 1, gt 1 & class 0).
//   dreg[i, :] = SL1'(e) of a label-positive row (unscaled: the caller knows the number of positives only after reading flags).
__global__ __launch_bounds__(64) void refine_loss_rows_kernel(const float* __restrict__ grasp, int64_t grasp_ld,
                                                             const float* __restrict__ cls, const float* __restrict__ reg,
                                                             const float* __restrict__ label, int64_t label_ld, int C,
                                                             float radius, float score_thre, int m, float* __restrict__ final_grasp,
                                                             uint8_t* __restrict__ flags, float* __restrict__ terms,
                                                             float* __restrict__ dreg) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= m) return;
  const float* g = grasp + (int64_t)i * grasp_ld;
  const float* r = reg + (int64_t)i * C;
  const float* gt = label + (int64_t)i * label_ld;
  float* o = final_grasp + (int64_t)i * C;
  for (int k = 0; k < 3; ++k) o[k] = g[k] + r[k] * radius;
  for (int k = 3; k < C; ++k) o[k] = g[k] + r[k];
  const bool one = cls[2 * i + 1] > cls[2 * i];
  const bool kept = one && o[7] > score_thre;
  const float ox = g[0] - gt[0], oy = g[1] - gt[1], oz = g[2] - gt[2];
  const bool near = sqrtf((ox * ox + oy * oy) + oz * oz) < 0.025f;
  const bool aligned = one_minus_cos(g + 3, gt + 3) < 0.5f;
  const bool same_angle = fabsf(g[6] - gt[6]) < 1.047f;
  const bool pos = near && aligned && same_angle;
  flags[i] = one ? 1 : 0;
  flags[m + i] = kept ? 1 : 0;
  flags[2 * m + i] = pos ? 1 : 0;
  float* t = terms + (int64_t)i * RF_TERMS;
  float* d = dreg + (int64_t)i * C;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < C; ++k) {
    float e = 0.f;
    if (pos && k < 10) {
      const float tgt = k < 3 ? (gt[k] - g[k]) / radius : gt[k] - g[k];
      e = r[k] - tgt;
      s[k < 3 ? 0 : k < 6 ? 1 : k < 7 ? 2 : 3] += sl1(e);
    }
    d[k] = sl1_grad(e);
  }
  t[0] = s[0]; t[1] = s[1]; t[2] = s[2]; t[3] = s[3];
  const float* preds[3] = {g, o, o};
  const bool use[3] = {one, one, kept};
  for (int q = 0; q < 3; ++q) {
    const float* p = preds[q];
    float* tq = t + 4 + 4 * q;
    if (use[q]) {
      tq[0] = (sl1(p[0] - gt[0]) + sl1(p[1] - gt[1])) + sl1(p[2] - gt[2]);
      tq[1] = one_minus_cos(p + 3, gt + 3);
      tq[2] = sl1(p[6] - gt[6]);
      tq[3] = (sl1(p[7] - gt[7]) + sl1(p[8] - gt[8])) + sl1(p[9] - gt[9]);
    } else {
      tq[0] = tq[1] = tq[2] = tq[3] = 0.f;
    }
  }
  t[16] = (pos && one) ? 1.f : 0.f;
  t[17] = (!pos && !one) ? 1.f : 0.f;
  t[18] = (!pos && one) ? 1.f : 0.f;
  t[19] = (pos && !one) ? 1.f : 0.f;
}

extern "C" int regnet_stage2_loss_rows_f32(const float* cls, const float* reg, int64_t A, int64_t C, const float* centre,
                                           int64_t centre_ld, const float* tmpl, const float* label, int64_t label_ld,
                                           float radius, const float* weights4, const int64_t* rows, int64_t m,
                                           float* next_grasp, int32_t* pick, int32_t* g8, float* a_gt, float* terms, float* dreg,
                                           void* stream) {
  if (m < 0 || A <= 0 || C != 10 || centre_ld < 3 || label_ld < 10) return REGNET_ERR_SHAPE;
  if (m >= (int64_t)1 << 30 || A > 64) return REGNET_ERR_UNSUPPORTED;
  if (m == 0) return REGNET_OK;
  if (!cls || !reg || !centre || !tmpl || !label || !weights4 || !next_grasp || !pick || !g8 || !a_gt || !terms || !dreg)
    return REGNET_ERR_NULL;
  hipLaunchKernelGGL(stage2_loss_rows_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, as_stream(stream), cls, reg, (int)A,
                     (int)C, centre, centre_ld, tmpl, label, label_ld, radius, weights4[0], weights4[1], weights4[2], weights4[3],
                     rows, (int)m, next_grasp, pick, g8, a_gt, terms, dreg);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_ce_rows_f32(const float* cls, int64_t A, const int32_t* target, const int64_t* idx, const int64_t* rows,
                                  int64_t nb, float scale, float* loss, float* dcls, void* stream) {
  if (nb < 0 || A <= 0 || A > 64) return REGNET_ERR_SHAPE;
  if (nb >= (int64_t)1 << 30) return REGNET_ERR_UNSUPPORTED;
  if (nb == 0) return REGNET_OK;
  if (!cls || !target || !idx || !loss || !dcls) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, as_stream(stream), cls, (int)A, target, idx,
                     rows, (int)nb, scale, loss, dcls);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_refine_loss_rows_f32(const float* grasp, int64_t grasp_ld, const float* cls, const float* reg,
                                           const float* label, int64_t label_ld, int64_t C, float radius, float score_thre,
                                           int64_t m, float* final_grasp, uint8_t* flags, float* terms, float* dreg, void* stream) {
  if (m < 0 || C != 10 || grasp_ld < 10 || label_ld < 10) return REGNET_ERR_SHAPE;
  if (m >= (int64_t)1 << 30) return REGNET_ERR_UNSUPPORTED;
  if (m == 0) return REGNET_OK;
  if (!grasp || !cls || !reg || !label || !final_grasp || !flags || !terms || !dreg) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(refine_loss_rows_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, as_stream(stream), grasp, grasp_ld, cls,
                     reg, label, label_ld, (int)C, radius, score_thre, (int)m, final_grasp, flags, terms, dreg);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}


// Training labels of the centres (dataset_utils/get_regiondataset.py:45-134 + :136-199, `use_theta`): every centre is matched to
// the ground-truth grasp whose contact point is nearest -- squared distance by the reference's expansion -2 a.b + |b|^2 + |a|^2 in
// fp32, compared as float64, first minimum -- kept when that distance is <= max_sq, and the grasp's 4x4 frame [x | y | z | c] is
// re-expressed as (c, y with y_x >= 0, theta = atan2(x_z, z_z) mirrored to pi - theta when y was flipped and wrapped to (-pi, pi]
// by the reference's four steps, score, antipodal score, centre score).  Centres without a grasp get the reference's filler:
// -1 everywhere except the axis, which its sign flip turns into +1.  ~60 small tensor launches (two of them 0.2 / 0.4 ms long on
// 128 workgroups) of the host-paced stretch of the training iteration as one: a wave per centre, lanes stride over the grasps.
// packed: (B, Gmax, 19) = 16 row-major frame entries | score | antipodal | centre score; gcount (B): grasps of the scene.
__global__ __launch_bounds__(256) void label_match_kernel(const float* __restrict__ packed, const int32_t* __restrict__ gcount, int Gmax,
                                                          const float* __restrict__ centre, int64_t centre_sb, int64_t centre_sn,
                                                          int B, int Nc, float depth, double max_sq, float* __restrict__ out,
                                                          int32_t* __restrict__ wide_row) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= B * Nc) return;
  const int b = w / Nc, c = w - b * Nc;
  const float* a = centre + b * centre_sb + c * centre_sn;
  const float a0 = a[0], a1 = a[1], a2 = a[2];
  const float aa = (a0 * a0 + a1 * a1) + a2 * a2;
  const float* rec = packed + (int64_t)b * Gmax * 19;
  const int G = gcount[b];
  double best = __builtin_huge_val();
  int besti = 0x7fffffff;
  for (int g = lane; g < G; g += 64) {
    const float* f = rec + (int64_t)g * 19;
    float cp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float t = f[4 * k] * depth;              // approach_k * depth
      cp[k] = (f[4 * k + 3] + t) - t;                // the reference shifts the contact point along the approach and back
    }
    const float dot = (a0 * cp[0] + a1 * cp[1]) + a2 * cp[2];
    const float bb = (cp[0] * cp[0] + cp[1] * cp[1]) + cp[2] * cp[2];
    float d = -2.f * dot;
    d = d + bb;
    d = d + aa;
    const double dd = (double)d;
    if (dd < best) { best = dd; besti = g; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(besti, off);
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane != 0) return;
  float* o = out + (int64_t)w * 10;
  const bool has = G > 0 && best <= max_sq;
  float fx[3], fy[3], fz[3], fc[3], sc[3];
  if (has) {
    const float* f = rec + (int64_t)besti * 19;
#pragma unroll
    for (int k = 0; k < 3; ++k) { fx[k] = f[4 * k]; fy[k] = f[4 * k + 1]; fz[k] = f[4 * k + 2]; fc[k] = f[4 * k + 3]; sc[k] = f[16 + k]; }
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) fx[k] = fy[k] = fz[k] = fc[k] = sc[k] = -1.f;
  }
  const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
  const bool missing = fx[0] == -1.f && fx[1] == -1.f && fx[2] == -1.f;
  float th = atan2f(fx[2], fz[2]);
  const bool flip = fy[0] < 0.f;
  if (flip) th = pi - th;
  if (th >= two_pi) th = th - two_pi;
  if (th <= -two_pi) th = th + two_pi;
  if (th > pi) th = th - two_pi;
  if (th <= -pi) th = th + two_pi;
  if (missing) th = -1.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = fc[k]; o[3 + k] = flip ? -fy[k] : fy[k]; o[7 + k] = sc[k]; }
  o[6] = th;
  wide_row[w] = sc[1] != -1.f ? 1 : 0;
}

extern "C" int regnet_label_match_f32(const float* packed, const int32_t* gcount, int64_t Gmax, const float* centre,
                                      int64_t centre_sb, int64_t centre_sn, int64_t B, int64_t Nc, float depth, double max_sq,
                                      float* out, int32_t* wide_row, void* stream) {
  if (B < 0 || Nc < 0 || Gmax < 0) return REGNET_ERR_SHAPE;
  if (B * Nc == 0) return REGNET_OK;
  if (!packed || !gcount || !centre || !out || !wide_row) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(label_match_kernel, dim3((unsigned)((B * Nc + 3) / 4)), dim3(256), 0, as_stream(stream), packed, gcount,
                     (int)Gmax, centre, centre_sb, centre_sn, (int)B, (int)Nc, depth, max_sq, out, wide_row);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
