// np_random_dev.hip -- numpy's legacy MT19937 stream consumed ON THE DEVICE (gfx950).
//
// The reference resamples every centre's candidate list and every grasp's in-box points with np.random.choice from
// Python loops (dataset_utils/get_regiondataset.py:331-337, multi_model/gripper_region_network.py:532-544).  Round 1-2
// moved those draws into native HOST code (np_random.hip) -- which still cost a device->host copy of the counts, 2.6 ms
// of host work and a host->device copy of the positions per batch, with the GPU's region stream idle meanwhile.  Here
// the generator state lives in device memory and the draws are made by kernels, bit for bit the stream numpy would
// produce (same algorithms as np_random.hip, which tests/test_np_random.py pins against numpy itself):
//   choice(n, size, replace=True)  -> randint(0, n, size): per value, next_uint32 & mask until <= n - 1
//   choice(n, size, replace=False) -> permutation(n)[:size]: Fisher-Yates from the top, j = random_interval(i)
//                                     (next_uint32 & mask_for(i) until <= i), swap(a[i], a[j])
// The stream is inherently serial -- a row's first word is wherever the previous row stopped, and rejection sampling
// makes that data dependent -- so ONE workgroup walks it; what is parallel is everything inside a step:
//   np_choice_scan_kernel (2 waves): wave 1 twists + tempers MT block b+1 (624 words, three dependent sweeps) while
//     wave 0 scans block b, 64 words per step: acceptance flags by ballot, ranks by mbcnt, the accepted values stored
//     by their lanes.  With replacement the flags do not depend on the state; for the shuffle they do (i shrinks by
//     one per accepted word and the mask with it), which a fixpoint over the ballot resolves: lane p's decision only
//     depends on the lanes before it, so iterating "flags from the previous flags' prefix counts" reaches the serial
//     answer (lane p is final after p + 1 rounds; typically 2-3 rounds).  Shuffle rows only record their j's here;
//   np_shuffle_rows_kernel (one wave per shuffle row, rows in parallel): plays the swaps in LDS and writes a[:size].
// State in / out: 624 raw words + position, numpy's own layout (np.random.get_state()[1:3]).
#include "common.h"

namespace {
constexpr int MT_N = 624;
constexpr int MT_M = 397;

__device__ __forceinline__ unsigned mt_mix(unsigned a, unsigned b) {
  const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
__device__ __forceinline__ unsigned mask_for(unsigned m) { return m ? (0xffffffffu >> __clz(m)) : 0u; }   // 2^k - 1 >= m
__device__ __forceinline__ int rank_below(unsigned long long bits) {   // set bits of `bits` in the lanes below mine
  return __builtin_amdgcn_mbcnt_hi((unsigned)(bits >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bits, 0u));
}
__device__ __forceinline__ void wave_sync() {   // one wave: LDS executes its instructions in order; keep the compiler from reordering
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct ScanArgs {
  unsigned* key; int* pos;                  // generator state, device memory, updated
  const int* counts; long long rows; int size; int mode;
  long long* out; unsigned char* valid;     // (rows, size) positions; (rows) or null
  int* jbuf; long long jcap;                // the shuffle rows' accepted draws, row after row
  int* row_joff;                            // (rows): offset of the row's draws in jbuf, -1 = not a shuffle row
  int* status;                              // bit 0: jbuf too small (results of shuffle rows invalid)
};

enum { ROW_NONE = 0, ROW_REPLACE = 1, ROW_SHUFFLE = 2 };

constexpr int MT_PAD = 640;   // 10 x 64: every lane-indexed read of a block stays inside its array

// Block cur -> block cur ^ 1 by ONE wave, straight-line (mt19937_gen of numpy's randomkit, out of place): all reads of the old
// block are issued up front, the three dependent sweeps (new[i] needs new[i - 227]) exchange through LDS, and the
// tempered words are written from registers.  Word i = 64 c + lane lives in lane `lane`, register c.
__device__ __forceinline__ void mt_next_block(const unsigned* __restrict__ o, unsigned* __restrict__ nw,
                                              unsigned* __restrict__ t, int lane) {
  constexpr int D = MT_N - MT_M;   // 227
  unsigned mix[10], val[10];
#pragma unroll
  for (int c = 0; c < 10; ++c) {
    const int i = 64 * c + lane;
    mix[c] = mt_mix(o[i], o[i + 1 < MT_PAD ? i + 1 : i]);   // (i = 623 is redone below: it mixes with the NEW word 0)
    val[c] = 0u;
  }
  unsigned far[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { const int i = 64 * c + lane; far[c] = o[i + MT_M < MT_PAD ? i + MT_M : i]; }
#pragma unroll
  for (int c = 0; c < 4; ++c) {          // sweep 1: i < 227
    const int i = 64 * c + lane;
    if (i < D) { val[c] = far[c] ^ mix[c]; nw[i] = val[c]; }
  }
  wave_sync();
#pragma unroll
  for (int c = 3; c < 8; ++c) {          // sweep 2: 227 <= i < 454
    const int i = 64 * c + lane;
    if (i >= D && i < 2 * D) { val[c] = nw[i - D] ^ mix[c]; nw[i] = val[c]; }
  }
  wave_sync();
#pragma unroll
  for (int c = 7; c < 10; ++c) {         // sweep 3: 454 <= i < 623
    const int i = 64 * c + lane;
    if (i >= 2 * D && i < MT_N - 1) { val[c] = nw[i - D] ^ mix[c]; nw[i] = val[c]; }
  }
  wave_sync();
  if (lane == (MT_N - 1) % 64) {         // i = 623 (register 9): mixes old[623] with NEW[0]
    val[9] = nw[MT_M - 1] ^ mt_mix(o[MT_N - 1], nw[0]);
    nw[MT_N - 1] = val[9];
  }
#pragma unroll
  for (int c = 0; c < 10; ++c) {
    const int i = 64 * c + lane;
    if (i < MT_N) t[i] = mt_temper(val[c]);
  }
}

__global__ __launch_bounds__(128) void np_choice_scan_kernel(const ScanArgs p) {
  __shared__ unsigned raw[2][MT_PAD];
  __shared__ unsigned tmp[2][MT_PAD];
  __shared__ int s_done;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < MT_PAD; i += 128) {
    const unsigned k = i < MT_N ? p.key[i] : 0u;
    raw[0][i] = k; raw[1][i] = 0u;
    tmp[0][i] = mt_temper(k); tmp[1][i] = 0u;
  }
  if (tid == 0) s_done = 0;
  __syncthreads();

  int cur = 0;
  int pos = __builtin_amdgcn_readfirstlane(*p.pos);
  // ---- scanner state (wave 0; everything wave-uniform)
  long long r = 0;                 // current row
  long long rbase = -64;           // rows [rbase, rbase + 64) are prefetched in cnt_reg
  int cnt_reg = 0;
  int kind = ROW_NONE, need = 0;   // accepted words the current row still wants
  unsigned rng = 0, mask = 0;      // replace rows: n - 1 and its mask
  int icur = 0;                    // shuffle rows: current i (n - 1 down to 1)
  long long wr = 0;                // where the next accepted value goes (out index / jbuf index)
  long long joff = 0;              // jbuf fill
  const long long size = p.size;

  for (;;) {
    if (wave == 1) {
      mt_next_block(raw[cur], raw[cur ^ 1], tmp[cur ^ 1], lane);
    } else {
      // ---- scanner: rows consume the tempered words of block cur from pos on; the block sits in registers
      const unsigned* words = tmp[cur];
      unsigned w[10];
#pragma unroll
      for (int c = 0; c < 10; ++c) w[c] = words[64 * c + lane];
      for (;;) {
        while (need == 0) {          // open the next row (rows that draw nothing are finished right here)
          if (r == p.rows) break;
          if (r - rbase >= 64) {
            rbase = r;
            cnt_reg = (rbase + lane < p.rows) ? p.counts[rbase + lane] : 0;
          }
          const int n = __builtin_amdgcn_readlane(cnt_reg, (int)(r - rbase));
          bool ok = true;
          long long fill = 0;
          kind = ROW_NONE;
          if (p.mode == 0) {         // radius groups (get_regiondataset.py:333-337)
            if (n >= size && n > 0) kind = ROW_SHUFFLE;
            else if (n > 0) kind = ROW_REPLACE;
            else { fill = -1; ok = false; }
          } else {                   // gripper crops (gripper_region_network.py:533-544); :538 re-tests the RESAMPLED length
            if (n > size) { kind = ROW_SHUFFLE; ok = size > 5; }
            else if (n > 5) kind = ROW_REPLACE;
            else ok = false;
          }
          if (lane == 0 && p.valid) p.valid[r] = ok ? 1 : 0;
          int row_off = -1;
          if (kind == ROW_SHUFFLE) {
            if (n >= 2) {
              need = n - 1; icur = n - 1; wr = joff; row_off = (int)joff; joff += n - 1;
              if (joff > p.jcap) {   // cannot happen with the capacity the binding allocates; never write out of bounds
                if (lane == 0) atomicOr(p.status, 1);
                kind = ROW_NONE; need = 0; row_off = -1; joff -= n - 1;
              }
            } else {                 // permutation(1) = [0]: nothing drawn (size is 1 here)
              kind = ROW_NONE;
            }
          } else if (kind == ROW_REPLACE) {
            rng = (unsigned)(n - 1);
            if (rng == 0) kind = ROW_NONE;          // randint(0, 1): zeros, no variates consumed
            else { mask = mask_for(rng); need = (int)size; wr = r * size; }
          }
          if (lane == 0) p.row_joff[r] = row_off;
          if (need == 0) {           // constant row
            for (long long k = lane; k < size; k += 64) p.out[r * size + k] = fill;
            ++r;
          }
        }
        if (need == 0) break;        // all rows done
        if (pos >= MT_N) break;      // block exhausted
        if (kind == ROW_REPLACE) {
          // the flags do not depend on the state: one pass over the REST OF THE BLOCK (ten ballots, scalar prefix counts)
          unsigned long long acc[10];
          int pre[11];
          pre[0] = 0;
#pragma unroll
          for (int c = 0; c < 10; ++c) {
            const int idx = 64 * c + lane;
            acc[c] = __ballot(idx >= pos && idx < MT_N && (w[c] & mask) <= rng);
            pre[c + 1] = pre[c] + __popcll(acc[c]);
          }
          const int want = need;
#pragma unroll
          for (int c = 0; c < 10; ++c) {
            if (pre[c] < want) {     // (else: the row ended in an earlier chunk)
              const int rk = rank_below(acc[c]);
              const bool mine = (acc[c] >> lane) & 1ull;
              const int k = want - pre[c];                       // accepted words this chunk may still give the row (>= 1)
              if (mine && rk < k) p.out[wr + pre[c] + rk] = (long long)(w[c] & mask);
              if (pre[c + 1] >= want) {                          // the row ends here: at the chunk's k-th accepted word
                const int e = __builtin_ctzll(__ballot(mine && rk == k - 1));
                pos = 64 * c + e + 1;
              }
            }
          }
          if (pre[10] < want) { wr += pre[10]; need -= pre[10]; pos = MT_N; }
          else { need = 0; ++r; }
          continue;
        }
        // ---- shuffle row: i falls by one per accepted word (and the mask with it): 64 words per step, iterated to the
        //      serial answer (lane p's decision only depends on the lanes before it)
        const int idx = pos + lane;
        const bool in = idx < MT_N;
        const unsigned ww = words[in ? idx : MT_N - 1];
        unsigned v;
        bool ok;
        unsigned long long accs, prev = 0ull;
        for (;;) {
          const int ip = icur - rank_below(prev);
          v = ww & mask_for(ip > 0 ? (unsigned)ip : 0u);
          ok = in && ip >= 1 && v <= (unsigned)ip;
          accs = __ballot(ok);
          if (accs == prev) break;
          prev = accs;
        }
        int cnt = __popcll(accs);
        int used = MT_N - pos < 64 ? MT_N - pos : 64;       // words this step consumes
        const int rk = rank_below(accs);
        if (cnt >= need) {                                   // the row ends inside this step: at its need-th accepted word
          const int e = __builtin_ctzll(__ballot(ok && rk == need - 1));
          accs &= (e == 63) ? ~0ull : ((1ull << (e + 1)) - 1ull);
          cnt = need;
          used = e + 1;
        }
        if ((accs >> lane) & 1ull) p.jbuf[wr + rk] = (int)v;
        wr += cnt; need -= cnt; icur -= cnt; pos += used;
        if (need == 0) ++r;
      }
      if (need == 0 && r == p.rows && lane == 0) s_done = 1;
    }
    __syncthreads();
    if (s_done) break;
    cur ^= 1;
    pos = 0;
  }
  for (int i = tid; i < MT_N; i += 128) p.key[i] = raw[cur][i];
  if (tid == 0) *p.pos = pos;
}

// One wave per row; rows that are not shuffle rows leave at once.  a[] lives in LDS (lds_cap ints) or, for longer
// candidate lists, in the row's slice of `scratch` (global memory, device-scope accesses).  The n - 1 swaps
// (a[i], a[j_i]), i = n - 1 .. 1, are replayed 64 at a time: step q of a chunk only depends on an earlier step p of the
// chunk if p's j is q's i or q's j (the i's are distinct and above every later index), so every lane finds its LATEST
// such predecessor with 64 readlanes, and the chunk runs as a few rounds of mutually independent swaps -- one round
// almost always (collisions need two of 64 draws to hit the same element of n) -- instead of 64 dependent round trips.
template <bool IN_LDS>
__device__ __forceinline__ void shuffle_replay(int* a, const int* __restrict__ jbuf, int joff, int n, int lane) {
  const int steps = n - 1;
  for (int t0 = 0; t0 < steps; t0 += 64) {
    const int m = steps - t0 < 64 ? steps - t0 : 64;
    const bool live = lane < m;
    const int j = live ? jbuf[joff + t0 + lane] : -1 - lane;     // (dead lanes: distinct negatives, never equal to anything)
    const int i = live ? n - 1 - (t0 + lane) : -100 - lane;
    int cp = -1;                                                  // latest earlier step of the chunk this one must wait for
    for (int q = 0; q < m; ++q) {
      const int jq = __builtin_amdgcn_readlane(j, q);
      if (q < lane && (jq == j || jq == i)) cp = q;
    }
    int s = 0;
    while (s < m) {
      const unsigned long long blocked = __ballot(live && lane >= s && cp >= s);
      const int f = blocked ? __builtin_ctzll(blocked) : m;
      if (live && lane >= s && lane < f) {
        if (IN_LDS) { const int ai = a[i], aj = a[j]; a[j] = ai; a[i] = aj; }
        else {
          const int ai = __hip_atomic_load(a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int aj = __hip_atomic_load(a + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a + j, ai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a + i, aj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (IN_LDS) wave_sync();
      else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");    // the next round may read what this one wrote
      s = f;
    }
  }
}

__global__ __launch_bounds__(64) void np_shuffle_rows_kernel(const int* __restrict__ counts, const int* __restrict__ row_joff,
                                                             const int* __restrict__ jbuf, int size,
                                                             long long* __restrict__ out, int lds_cap,
                                                             int* __restrict__ scratch) {
  extern __shared__ int a_lds[];
  const long long r = blockIdx.x;
  const int joff = row_joff[r];
  if (joff < 0) return;
  const int n = counts[r], lane = threadIdx.x;
  int* a_glb = scratch + (long long)joff + r;   // n ints (the row's n - 1 draws start at joff; r rows before it)
  if (n <= lds_cap) {
    for (int k = lane; k < n; k += 64) a_lds[k] = k;
    wave_sync();
    shuffle_replay<true>(a_lds, jbuf, joff, n, lane);
    for (int k = lane; k < size; k += 64) out[r * (long long)size + k] = a_lds[k];
  } else {
    for (int k = lane; k < n; k += 64) __hip_atomic_store(a_glb + k, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    shuffle_replay<false>(a_glb, jbuf, joff, n, lane);
    for (int k = lane; k < size; k += 64)
      out[r * (long long)size + k] = __hip_atomic_load(a_glb + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// np.random.rand / random_sample: count doubles, two words each -- (a >> 5, b >> 6) -> (a 2^26 + b) / 2^53 (randomkit's rk_double).
__global__ __launch_bounds__(64) void np_rand_doubles_kernel(unsigned* __restrict__ key, int* __restrict__ pos_ptr, int count,
                                                             double* __restrict__ out) {
  __shared__ unsigned raw[2][MT_PAD];
  __shared__ unsigned tmp[2][MT_PAD];
  const int lane = threadIdx.x;
  for (int i = lane; i < MT_PAD; i += 64) {
    const unsigned k = i < MT_N ? key[i] : 0u;
    raw[0][i] = k; raw[1][i] = 0u; tmp[0][i] = mt_temper(k); tmp[1][i] = 0u;
  }
  wave_sync();
  int cur = 0, pos = __builtin_amdgcn_readfirstlane(*pos_ptr);
  unsigned have = 0;
  bool half = false;
  for (int k = 0; k < 2 * count; ++k) {
    if (pos >= MT_N) {
      mt_next_block(raw[cur], raw[cur ^ 1], tmp[cur ^ 1], lane);
      wave_sync();
      cur ^= 1;
      pos = 0;
    }
    const unsigned w = tmp[cur][pos++];
    if (!half) { have = w >> 5; half = true; }
    else {
      if (lane == 0) out[k >> 1] = ((double)have * 67108864.0 + (double)(w >> 6)) / 9007199254740992.0;
      half = false;
    }
  }
  wave_sync();
  for (int i = lane; i < MT_N; i += 64) key[i] = raw[cur][i];
  if (lane == 0) *pos_ptr = pos;
}

}  // namespace

extern "C" int64_t regnet_np_choice_rows_dev_workspace_ints(int64_t rows, int64_t max_count) {
  // row_joff (rows) | jbuf (rows * max_count) | scratch (rows * max_count + rows) | status (1)
  if (rows < 0 || max_count < 0) return -1;
  return rows + rows * max_count + (rows * max_count + rows) + 1;
}

extern "C" int regnet_np_choice_rows_dev(uint32_t* d_mt_key, int32_t* d_mt_pos, const int32_t* d_counts, int64_t rows,
                                         int64_t size, int64_t max_count, int mode, int64_t* d_out, uint8_t* d_valid,
                                         int32_t* d_workspace, void* stream) {
  if (rows < 0 || size < 0 || max_count < 0 || (mode != 0 && mode != 1)) return REGNET_ERR_SHAPE;
  if (size >= (1ll << 31) || max_count >= (1ll << 31) || rows * (max_count + 1) >= (1ll << 31)) return REGNET_ERR_UNSUPPORTED;
  if (rows == 0) return REGNET_OK;
  if (!d_mt_key || !d_mt_pos || !d_counts || !d_out || !d_workspace) return REGNET_ERR_NULL;
  ScanArgs a = {};
  a.key = d_mt_key; a.pos = d_mt_pos; a.counts = d_counts; a.rows = rows; a.size = (int)size; a.mode = mode;
  a.out = (long long*)d_out; a.valid = d_valid;
  a.row_joff = d_workspace;
  a.jbuf = d_workspace + rows; a.jcap = rows * max_count;
  int* scratch = a.jbuf + rows * max_count;
  a.status = scratch + rows * max_count + rows;
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(a.status, 0, sizeof(int), s) != hipSuccess) return (int)hipGetLastError();
  hipLaunchKernelGGL(np_choice_scan_kernel, dim3(1), dim3(128), 0, s, a);
  REGNET_LAUNCH_CHECK();
  const int lds_cap = 12288;   // 48 KiB: three shuffle rows per CU side by side (longer lists replay in global memory)
  hipLaunchKernelGGL(np_shuffle_rows_kernel, dim3((unsigned)rows), dim3(64), lds_cap * sizeof(int), s, d_counts,
                     a.row_joff, a.jbuf, (int)size, (long long*)d_out, lds_cap, scratch);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_np_rand_doubles_dev(uint32_t* d_mt_key, int32_t* d_mt_pos, int64_t count, double* d_out, void* stream) {
  if (count < 0 || count >= (1ll << 30)) return REGNET_ERR_SHAPE;
  if (count == 0) return REGNET_OK;
  if (!d_mt_key || !d_mt_pos || !d_out) return REGNET_ERR_NULL;
  hipLaunchKernelGGL(np_rand_doubles_kernel, dim3(1), dim3(64), 0, as_stream(stream), d_mt_key, d_mt_pos, (int)count, d_out);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
