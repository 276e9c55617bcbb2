// np_random.hip -- HOST code: numpy-compatible random draws for the region stage.
//
// The reference resamples every centre's candidate list / every grasp's in-box points with
// np.random.choice from Python loops (dataset_utils/get_regiondataset.py:331-337,
// multi_model/gripper_region_network.py:532-544): ~1500 interpreter-level RNG calls per batch of 8
// scenes, which became the host-side bottleneck of the pipeline.  These entry points consume the
// SAME MT19937 stream with the SAME algorithms numpy's legacy RandomState uses (numpy 1.17+ .. 2.x):
//   choice(n, size, replace=True)  -> randint(0, n, size): 32-bit masked rejection
//                                     (buffered_bounded_masked_uint32: draw & mask until <= n-1)
//   choice(n, size, replace=False) -> permutation(n)[:size]: arange + Fisher-Yates from the top,
//                                     j = random_interval(i) (32-bit masked rejection)
// so that `np.random.seed(s)` followed by the native draws leaves numpy's generator in exactly the
// state the reference's loop would (tests/test_np_random.py checks outputs and final state against
// numpy itself).  The caller passes the state from np.random.get_state() and writes it back.
#include <stdint.h>
#include <stdlib.h>

#include "../../include/regnet_hip.h"

// host-only translation unit that hipcc also parses for the device: the AVX2 clones exist in the host pass only
#if defined(__HIP_DEVICE_COMPILE__)
#define HOST_SIMD_CLONES
#else
#define HOST_SIMD_CLONES __attribute__((target_clones("avx2", "default")))
#endif

namespace {

// MT19937 state (numpy's layout: 624 raw words + position) plus a block of tempered outputs: the generator is advanced
// 624 words at a time and the whole block is tempered in one vectorisable loop; consumers then filter the block with
// branch-free compaction (the masked-rejection loops of numpy are ~50 % unpredictable branches when written naively,
// which cost more than the generator itself: 5 ns -> ~1.5 ns per accepted draw on the bench host).
struct MT {
  uint32_t* key;  // 624 words
  int pos;
  uint32_t out[624];  // tempered key[], valid for indices >= the position at which it was last filled
};

HOST_SIMD_CLONES void mt_temper(const uint32_t* key, uint32_t* out) {
  for (int i = 0; i < 624; ++i) {
    uint32_t y = key[i];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    out[i] = y;
  }
}

HOST_SIMD_CLONES void mt_twist(uint32_t* key) {
  const int N = 624, M = 397;
  const uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;
  uint32_t y;
  int i;
  for (i = 0; i < N - M; i++) {
    y = (key[i] & UPPER) | (key[i + 1] & LOWER);
    key[i] = key[i + M] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & MATRIX_A);
  }
  for (; i < N - 1; i++) {
    y = (key[i] & UPPER) | (key[i + 1] & LOWER);
    key[i] = key[i + (M - N)] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & MATRIX_A);
  }
  y = (key[N - 1] & UPPER) | (key[0] & LOWER);
  key[N - 1] = key[M - 1] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & MATRIX_A);
}

inline void mt_refill(MT& s) {   // precondition: s.pos == 624
  mt_twist(s.key);
  mt_temper(s.key, s.out);
  s.pos = 0;
}

inline uint32_t mask_for(uint32_t max) {  // smallest 2^k - 1 >= max
  uint32_t m = max;
  m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
  return m;
}

// randint(0, n, size): values in [0, n-1]; every raw word is consumed in order, rejected ones (> n-1 after masking) are
// overwritten in place by the next candidate
inline void draw_with_replacement(MT& s, uint32_t n, int64_t size, int64_t* out) {
  const uint32_t rng = n - 1;
  if (rng == 0) {
    for (int64_t i = 0; i < size; ++i) out[i] = 0;  // no variates consumed
    return;
  }
  const uint32_t mask = mask_for(rng);
  int64_t i = 0;
  while (i < size) {
    if (s.pos == 624) mt_refill(s);
    int p = s.pos;
    for (; p < 624 && i < size; ++p) {
      const uint32_t v = s.out[p] & mask;
      out[i] = v;
      i += (v <= rng);
    }
    s.pos = p;
  }
}

// permutation(n)[:size]: Fisher-Yates from the top, j = random_interval(i) by masked rejection; a rejected candidate
// swaps element i with itself and leaves i unchanged
inline void draw_without_replacement(MT& s, uint32_t n, int64_t size, int64_t* out, int64_t* scratch) {
  for (uint32_t i = 0; i < n; ++i) scratch[i] = i;
  uint32_t i = n - 1;
  while (i >= 1) {
    if (s.pos == 624) mt_refill(s);
    int p = s.pos;
    for (; p < 624 && i >= 1; ++p) {
      const uint32_t v = s.out[p] & mask_for(i);
      const uint32_t ok = v <= i;
      const uint32_t j = ok ? v : i;
      const int64_t t = scratch[j]; scratch[j] = scratch[i]; scratch[i] = t;
      i -= ok;
    }
    s.pos = p;
  }
  for (int64_t k = 0; k < size; ++k) out[k] = scratch[k];
}

}  // namespace

// Rows are processed in order.  mode 0 (radius groups, get_regiondataset.py:333-337):
//   n >= size -> without replacement; 0 < n < size -> with replacement; n == 0 -> row of -1.
// mode 1 (gripper crops, gripper_region_network.py:533-544):
//   n > size -> without replacement; 5 < n <= size -> with replacement; n <= 5 -> row of 0, valid[r] = 0.
extern "C" int regnet_np_choice_rows(uint32_t* mt_key, int32_t* mt_pos, const int32_t* counts, int64_t rows,
                                     int64_t size, int mode, int64_t* out, uint8_t* valid) {
  if (!mt_key || !mt_pos || (rows > 0 && (!counts || !out)) || rows < 0 || size < 0) return REGNET_ERR_NULL;
  if (*mt_pos < 0 || *mt_pos > 624) return REGNET_ERR_SHAPE;
  MT s;
  s.key = mt_key;
  s.pos = *mt_pos;
  mt_temper(s.key, s.out);   // words at positions >= pos are still to be consumed
  int64_t maxn = 0;
  for (int64_t r = 0; r < rows; ++r) {
    if (counts[r] < 0) return REGNET_ERR_SHAPE;
    if (counts[r] > maxn) maxn = counts[r];
  }
  int64_t* scratch = (int64_t*)malloc(sizeof(int64_t) * (size_t)(maxn > 0 ? maxn : 1));
  if (!scratch) return REGNET_ERR_UNSUPPORTED;
  for (int64_t r = 0; r < rows; ++r) {
    const int64_t n = counts[r];
    int64_t* o = out + r * size;
    bool ok = true;
    if (mode == 0) {
      if (n >= size && n > 0) draw_without_replacement(s, (uint32_t)n, size, o, scratch);
      else if (n > 0) draw_with_replacement(s, (uint32_t)n, size, o);
      else { for (int64_t i = 0; i < size; ++i) o[i] = -1; ok = false; }
    } else {
      // the reference re-tests len(index) > 5 AFTER resampling (:538), i.e. on `size` in the first case
      if (n > size) { draw_without_replacement(s, (uint32_t)n, size, o, scratch); ok = size > 5; }
      else if (n > 5) draw_with_replacement(s, (uint32_t)n, size, o);
      else { for (int64_t i = 0; i < size; ++i) o[i] = 0; ok = false; }
    }
    if (valid) valid[r] = ok ? 1 : 0;
  }
  free(scratch);
  *mt_pos = s.pos;
  return REGNET_OK;
}
