// dataset.hip -- the per-item work of the reference's ScoreDataset.__getitem__ on the device (gfx950).
//
// dataset_utils/scoredataset.py:60-81: a record's cloud is resampled to exactly all_points_num points
// (np.random.choice, without replacement when the record has enough points), the colours of table points (label 0) and
// object points are scaled by random per-channel gains (_noise_color, :52-58: table gain = rand(3), object gain =
// 1 - rand(3) / 5), the rows become [xyz | rgb] and the score is squashed with tanh.  The draws come from numpy's
// generator (np_random_dev.hip consumes the same stream on the device); this kernel is the gather + jitter + tanh:
// HBM-bound, 8 floats read per picked point at random rows, 8 written -- one thread per output point.
// The gains are float64 and numpy (>= 2, NEP 50) multiplies float32 colours by a float64 scalar IN DOUBLE before
// rounding back to float32; so does this kernel.
#include "common.h"

__global__ __launch_bounds__(256) void dataset_resample_kernel(const float* __restrict__ cloud, const float* __restrict__ color,
                                                               const float* __restrict__ score, const float* __restrict__ label,
                                                               long long M, const long long* __restrict__ pick, long long N,
                                                               const double* __restrict__ rand6, float* __restrict__ pc,
                                                               float* __restrict__ score_out, float* __restrict__ label_out,
                                                               int* __restrict__ bad) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= N) return;
  long long j = pick[t];
  if (j < 0 || j >= M) {          // cannot happen with positions drawn for this record; never read out of bounds
    if (bad) atomicOr(bad, 1);
    j = 0;
  }
  const float lab = label[j];
  float* o = pc + t * 6;
  o[0] = cloud[j * 3 + 0]; o[1] = cloud[j * 3 + 1]; o[2] = cloud[j * 3 + 2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double gain = lab == 0.0f ? rand6[c] : 1.0 - rand6[3 + c] / 5.0;
    o[3 + c] = (float)((double)color[j * 3 + c] * gain);
  }
  score_out[t] = tanhf(score[j]);
  label_out[t] = lab;
}

extern "C" int regnet_dataset_resample_f32(const float* cloud, const float* color, const float* score, const float* label,
                                           int64_t M, const int64_t* pick, int64_t N, const double* rand6, float* pc,
                                           float* score_out, float* label_out, int32_t* out_of_range, void* stream) {
  if (M <= 0 || N < 0) return REGNET_ERR_SHAPE;
  if (N == 0) return REGNET_OK;
  if (!cloud || !color || !score || !label || !pick || !rand6 || !pc || !score_out || !label_out) return REGNET_ERR_NULL;
  if ((N + 255) / 256 >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dataset_resample_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), cloud, color,
                     score, label, (long long)M, (const long long*)pick, (long long)N, rand6, pc, score_out, label_out,
                     out_of_range);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
