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
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

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

// ---- AVX-512 forms (host pass only; chosen at run time): the same arithmetic 16 words at a time.  The twist's second loop
// reads words it has just written, 227 positions back: further than a vector.  Consumption: candidates are masked, compared and
// COMPRESSED in a register (vpcompressd), widened to int64 and stored as two full vectors -- the slots behind the accepted
// ones are overwritten by the next store, so a chunk is only taken while 16 more outputs fit in the row.
#if !defined(__HIP_DEVICE_COMPILE__)
#define TGT512 __attribute__((target("avx512f,avx512vl,avx512dq,popcnt")))
bool has_avx512() {
  static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") &&
                        __builtin_cpu_supports("avx512dq") && getenv("REGNET_NP_RANDOM_SCALAR") == nullptr;
  return v;
}
TGT512 inline void twist_range_512(uint32_t* key, int a, int b, int off) {
  const __m512i upper = _mm512_set1_epi32((int)0x80000000u), lower = _mm512_set1_epi32(0x7fffffff);
  const __m512i one = _mm512_set1_epi32(1), mat = _mm512_set1_epi32((int)0x9908b0dfu), zero = _mm512_setzero_si512();
  int i = a;
  for (; i + 16 <= b; i += 16) {
    const __m512i x0 = _mm512_loadu_si512(key + i), x1 = _mm512_loadu_si512(key + i + 1), xm = _mm512_loadu_si512(key + i + off);
    const __m512i y = _mm512_or_si512(_mm512_and_si512(x0, upper), _mm512_and_si512(x1, lower));
    const __m512i mag = _mm512_and_si512(_mm512_sub_epi32(zero, _mm512_and_si512(y, one)), mat);
    _mm512_storeu_si512(key + i, _mm512_xor_si512(_mm512_xor_si512(xm, _mm512_srli_epi32(y, 1)), mag));
  }
  for (; i < b; ++i) {
    const uint32_t y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
    key[i] = key[i + off] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & 0x9908b0dfu);
  }
}
TGT512 void mt_twist_512(uint32_t* key) {
  const int N = 624, M = 397;
  twist_range_512(key, 0, N - M, M);
  twist_range_512(key, N - M, N - 1, M - N);
  const uint32_t y = (key[N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
  key[N - 1] = key[M - 1] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & 0x9908b0dfu);
}
TGT512 void mt_temper_512(const uint32_t* key, uint32_t* out) {
  const __m512i b = _mm512_set1_epi32((int)0x9d2c5680u), c = _mm512_set1_epi32((int)0xefc60000u);
  for (int i = 0; i < 624; i += 16) {
    __m512i y = _mm512_loadu_si512(key + i);
    y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
    y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 7), b));
    y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 15), c));
    y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
    _mm512_storeu_si512(out + i, y);
  }
}
// whole 16-candidate chunks of cand[p ..] while 16 more outputs fit behind out[*pi]; -> the new p
TGT512 int wr_chunks_512(const uint32_t* cand, int p, uint32_t mask, uint32_t rng, int64_t* out, int64_t* pi, int64_t size) {
  const __m512i vmask = _mm512_set1_epi32((int)mask), vrng = _mm512_set1_epi32((int)rng);
  int64_t i = *pi;
  while (p + 16 <= 624 && i + 16 <= size) {
    const __m512i v = _mm512_and_si512(_mm512_loadu_si512(cand + p), vmask);
    const __mmask16 k = _mm512_cmple_epu32_mask(v, vrng);
    const __m512i c = _mm512_maskz_compress_epi32(k, v);
    _mm512_storeu_si512(out + i, _mm512_cvtepu32_epi64(_mm512_castsi512_si256(c)));
    _mm512_storeu_si512(out + i + 8, _mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(c, 1)));
    i += __builtin_popcount((unsigned)k);
    p += 16;
  }
  *pi = i;
  return p;
}
#else
inline bool has_avx512() { return false; }
inline void mt_twist_512(uint32_t*) {}
inline void mt_temper_512(const uint32_t*, uint32_t*) {}
inline int wr_chunks_512(const uint32_t*, int p, uint32_t, uint32_t, int64_t*, int64_t*, int64_t) { return p; }
#endif

inline void mt_temper_any(const uint32_t* key, uint32_t* out) {
  if (has_avx512()) mt_temper_512(key, out);
  else mt_temper(key, out);
}

inline void mt_refill(MT& s) {   // precondition: s.pos == 624
  if (has_avx512()) { mt_twist_512(s.key); mt_temper_512(s.key, s.out); }
  else { mt_twist(s.key); mt_temper(s.key, s.out); }
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
    if (has_avx512()) p = wr_chunks_512(s.out, p, mask, rng, out, &i, size);
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
inline void draw_without_replacement(MT& s, uint32_t n, int64_t size, int64_t* out, uint32_t* scratch) {
  for (uint32_t i = 0; i < n; ++i) scratch[i] = i;
  uint32_t i = n - 1;
  // mask = smallest 2^k - 1 >= i (numpy recomputes it from i in every call of random_interval): i falls by at most one per step,
  // so it halves at most once per step -- a compare and a conditional move on the loop's dependency chain (i -> mask -> candidate ->
  // accepted -> i) instead of five shift-or pairs; 32-bit scratch (n < 2^31)
  uint32_t mask = mask_for(i);
  while (i >= 1) {
    if (s.pos == 624) mt_refill(s);
    int p = s.pos;
    for (; p < 624 && i >= 1; ++p) {
      mask = (i <= (mask >> 1)) ? (mask >> 1) : mask;
      const uint32_t v = s.out[p] & mask;
      const uint32_t ok = v <= i;
      const uint32_t j = ok ? v : i;
      const uint32_t t = scratch[j]; scratch[j] = scratch[i]; scratch[i] = t;
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
  mt_temper_any(s.key, s.out);   // words at positions >= pos are still to be consumed
  int64_t maxn = 0;
  for (int64_t r = 0; r < rows; ++r) {
    if (counts[r] < 0) return REGNET_ERR_SHAPE;
    if (counts[r] > maxn) maxn = counts[r];
  }
  uint32_t* scratch = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(maxn > 0 ? maxn : 1));
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
