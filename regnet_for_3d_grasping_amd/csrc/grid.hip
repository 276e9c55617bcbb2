// grid.hip -- uniform-grid acceleration of the two all-pairs scans of the set-abstraction path,
// 3-NN search and ball query, for gfx950.  Built with -ffp-contract=off: the distances are the same
// individually rounded fp32 expressions as in geometry.hip, so results stay bit-identical to the
// brute-force kernels / the oracle; the grid only decides WHICH pairs are evaluated.
//
// Reference behaviour (multi_model/utils/pn2_utils/csrc): interpolate_kernel.cu:28-77 (3-NN, strict <,
// earlier key wins ties), ball_query_kernel.cu:31-74 (first K in-radius points in index order).
//
// Structure: the build kernels bin the SOURCE points (keys / the dense cloud) into cubic cells of edge
// h over their bounding box with a counting sort (counters in a caller-provided workspace) and write
// them cell-contiguous as float4 (x, y, z, original index).
//   3-NN      one thread per query walks Chebyshev shells of cells around its own cell until the third
//             best squared distance is strictly inside the searched cube; candidates are ranked by
//             (distance, index), which is exactly the brute-force order.
//   ball query one wave per centroid visits the cells overlapping the ball, appends in-radius indices
//             and keeps the 64 smallest with a 128-wide bitonic merge in registers; ascending output.
#include "common.h"

#define GRID_MAX_CELLS 65536
#define GRID_HDR_FLOATS 16

struct GridHeader {   // per scene, at the start of its workspace slab (GRID_HDR_FLOATS * 4 bytes)
  float lo[3];
  float h;
  float inv_h;
  int dim[3];
  int cells;
  float eps;   // bound on the fp32 rounding of a cell-edge coordinate in this grid
  int pad[6];
};

static_assert(sizeof(GridHeader) == GRID_HDR_FLOATS * 4, "header layout");

__host__ __device__ inline long long grid_slab_bytes(long long N) {
  // header | cell_start[GRID_MAX_CELLS + 1] | sorted float4[N]
  long long b = GRID_HDR_FLOATS * 4 + (GRID_MAX_CELLS + 1) * 4ll;
  b = (b + 15) / 16 * 16;
  return b + N * 16ll;
}
__device__ __forceinline__ int* grid_cell_start(char* slab) { return reinterpret_cast<int*>(slab + GRID_HDR_FLOATS * 4); }
__device__ __forceinline__ float4* grid_sorted(char* slab) {
  long long off = GRID_HDR_FLOATS * 4 + (GRID_MAX_CELLS + 1) * 4ll;
  off = (off + 15) / 16 * 16;
  return reinterpret_cast<float4*>(slab + off);
}

__device__ __forceinline__ int cell_coord(float v, float lo, float inv_h, int dim) {
  int c = (int)floorf((v - lo) * inv_h);
  return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

// ---- build ------------------------------------------------------------------------------------
// Four short launches per build: bounds + header + zeroed counters (one workgroup per scene),
// histogram (all CUs), exclusive scan of the counters (one workgroup per scene), scatter (all CUs).
// cell_start[c + 1] is cell c's counter: the histogram counts into it, the scan turns it into the
// cell's first slot, and the scatter's atomicAdd hands out the slots -- leaving it at the cell's end =
// the start of cell c + 1, so after the scatter the array is the usual CSR offset table and no
// second cursor array is needed.  (The order of points inside a cell depends on the atomics; neither
// search depends on it.)
__device__ __forceinline__ int cell_of(const GridHeader& H, float x, float y, float z) {
  return (cell_coord(z, H.lo[2], H.inv_h, H.dim[2]) * H.dim[1] + cell_coord(y, H.lo[1], H.inv_h, H.dim[1])) * H.dim[0] +
         cell_coord(x, H.lo[0], H.inv_h, H.dim[0]);
}

// GRID_WG threads per scene for the two one-workgroup-per-scene kernels (bounds, scan): 512, not 1024 -- eight waves of <= 32
// registers find room on a CU whose SIMDs hold two ~200-register chain waves each; sixteen did not (the geometry of the batches
// ahead then waited for whole CUs).  Results do not depend on it (min / max, integer sums).
#define GRID_WG 512
__global__ __launch_bounds__(GRID_WG) void grid_bounds_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sc,
                                                           int64_t sn, int N, float h_req, char* __restrict__ ws,
                                                           long long slab) {
  __shared__ float red[6][GRID_WG / 64];
  __shared__ GridHeader hdr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* base = xyz + (int64_t)blockIdx.x * sb;
  char* my = ws + (long long)blockIdx.x * slab;
  int* cell_start = grid_cell_start(my);

  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int j = tid; j < N; j += GRID_WG) {
    const float x = base[(int64_t)j * sn], y = base[sc + (int64_t)j * sn], z = base[2 * sc + (int64_t)j * sn];
    lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
    lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
    lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
    if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
  }
  __syncthreads();
  if (tid == 0) {
    float e[3], scale = 0.f;
    for (int a = 0; a < 3; ++a) {
      float l = red[a][0], hgh = red[3 + a][0];
      for (int w = 1; w < GRID_WG / 64; ++w) { l = fminf(l, red[a][w]); hgh = fmaxf(hgh, red[3 + a][w]); }
      hdr.lo[a] = l;
      e[a] = fmaxf(hgh - l, 0.f);
      scale = fmaxf(scale, fmaxf(fabsf(l), fabsf(hgh)));
    }
    hdr.eps = 4e-6f * scale + 1e-30f;
    float h = h_req;
    if (!(h > 0.f)) {  // automatic: about one point per cell, never more than ~4N cells for flat / thin clouds
      const float n = (float)(N > 0 ? N : 1);
      const float vol = fmaxf(e[0], 1e-6f) * fmaxf(e[1], 1e-6f) * fmaxf(e[2], 1e-6f);
      const float area = fmaxf(fmaxf(e[0] * e[1], e[1] * e[2]), e[0] * e[2]);
      const float len = fmaxf(fmaxf(e[0], e[1]), e[2]);
      h = fmaxf(fmaxf(cbrtf(vol / n), sqrtf(area / (4.f * n))), len / (4.f * n));
      h = fmaxf(h, 1e-6f);
    }
    for (;;) {
      long long cells = 1;
      for (int a = 0; a < 3; ++a) {
        int d = (int)floorf(e[a] / h) + 1;
        d = d < 1 ? 1 : (d > 4096 ? 4096 : d);
        hdr.dim[a] = d;
        cells *= d;
      }
      if (cells <= GRID_MAX_CELLS) { hdr.cells = (int)cells; break; }
      h *= 1.26f;
    }
    hdr.h = h;
    hdr.inv_h = 1.0f / h;
  }
  __syncthreads();
  if (tid < (int)(sizeof(GridHeader) / 4)) reinterpret_cast<int*>(my)[tid] = reinterpret_cast<const int*>(&hdr)[tid];
  const int cells = hdr.cells;
  for (int c = tid; c <= cells; c += GRID_WG) cell_start[c] = 0;
}

__global__ __launch_bounds__(256) void grid_count_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sc,
                                                         int64_t sn, int N, char* __restrict__ ws, long long slab) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  char* my = ws + (long long)blockIdx.y * slab;
  const GridHeader H = *reinterpret_cast<const GridHeader*>(my);
  const float* base = xyz + (int64_t)blockIdx.y * sb;
  const int c = cell_of(H, base[(int64_t)j * sn], base[sc + (int64_t)j * sn], base[2 * sc + (int64_t)j * sn]);
  atomicAdd(&grid_cell_start(my)[c + 1], 1);
}

__global__ __launch_bounds__(GRID_WG) void grid_scan_kernel(char* __restrict__ ws, long long slab) {
  __shared__ unsigned wsum[GRID_WG / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* my = ws + (long long)blockIdx.x * slab;
  const int cells = reinterpret_cast<const GridHeader*>(my)->cells;
  int* cell_start = grid_cell_start(my);
  const int per = (cells + GRID_WG - 1) / GRID_WG;
  const int beg = tid * per, end = min(cells, beg + per);
  unsigned local = 0;
  for (int c = beg; c < end; ++c) local += (unsigned)cell_start[c + 1];
  unsigned incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned run = incl - local;
  for (int w = 0; w < GRID_WG / 64; ++w) run += (w < wave) ? wsum[w] : 0u;
  for (int c = beg; c < end; ++c) {   // each thread rewrites only its own chunk, after having read it
    const unsigned cnt = (unsigned)cell_start[c + 1];
    cell_start[c + 1] = (int)run;     // first slot of cell c
    run += cnt;
  }
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const float* __restrict__ xyz, int64_t sb, int64_t sc,
                                                           int64_t sn, int N, char* __restrict__ ws, long long slab) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  char* my = ws + (long long)blockIdx.y * slab;
  const GridHeader H = *reinterpret_cast<const GridHeader*>(my);
  const float* base = xyz + (int64_t)blockIdx.y * sb;
  const float x = base[(int64_t)j * sn], y = base[sc + (int64_t)j * sn], z = base[2 * sc + (int64_t)j * sn];
  const int pos = atomicAdd(&grid_cell_start(my)[cell_of(H, x, y, z) + 1], 1);
  grid_sorted(my)[pos] = make_float4(x, y, z, __int_as_float(j));
}

// ---- 3-NN over the grid -------------------------------------------------------------------------
__device__ __forceinline__ void nn3_insert(float d, int j, float& d0, float& d1, float& d2, int& i0, int& i1,
                                           int& i2) {
  // rank by (distance, index): identical to the brute-force scan's "strict <, ascending key order"
  const bool b2 = d < d2 || (d == d2 && j < i2);
  if (!b2) return;
  const bool b1 = d < d1 || (d == d1 && j < i1);
  const bool b0 = d < d0 || (d == d0 && j < i0);
  d2 = b1 ? d1 : d;
  i2 = b1 ? i1 : j;
  d1 = b0 ? d0 : (b1 ? d : d1);
  i1 = b0 ? i0 : (b1 ? j : i1);
  d0 = b0 ? d : d0;
  i0 = b0 ? j : i0;
}

__global__ __launch_bounds__(256) void three_nn_grid_kernel(const float* __restrict__ query, int64_t qb, int64_t qc,
                                                            int64_t qn, int N1, const char* __restrict__ ws,
                                                            long long slab, int64_t* __restrict__ index,
                                                            float* __restrict__ dist2) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N1) return;
  char* my = const_cast<char*>(ws) + (long long)b * slab;
  const GridHeader H = *reinterpret_cast<const GridHeader*>(my);
  const int* cell_start = grid_cell_start(my);
  const float4* sorted = grid_sorted(my);
  const float* qbase = query + (int64_t)b * qb;
  const float qx = qbase[(int64_t)i * qn], qy = qbase[qc + (int64_t)i * qn], qz = qbase[2 * qc + (int64_t)i * qn];
  const int cx = cell_coord(qx, H.lo[0], H.inv_h, H.dim[0]);
  const int cy = cell_coord(qy, H.lo[1], H.inv_h, H.dim[1]);
  const int cz = cell_coord(qz, H.lo[2], H.inv_h, H.dim[2]);
  float d0 = __builtin_inff(), d1 = d0, d2 = d0;
  int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff;
  const int rmax = max(max(max(cx, H.dim[0] - 1 - cx), max(cy, H.dim[1] - 1 - cy)), max(cz, H.dim[2] - 1 - cz));
  for (int R = 0; R <= rmax; ++R) {
    const int z0 = max(cz - R, 0), z1 = min(cz + R, H.dim[2] - 1);
    const int y0 = max(cy - R, 0), y1 = min(cy + R, H.dim[1] - 1);
    const int x0 = max(cx - R, 0), x1 = min(cx + R, H.dim[0] - 1);
    for (int z = z0; z <= z1; ++z) {
      const bool zface = (z == cz - R) || (z == cz + R);
      for (int y = y0; y <= y1; ++y) {
        const bool yface = (y == cy - R) || (y == cy + R);
        const int row = (z * H.dim[1] + y) * H.dim[0];
        if (zface || yface) {   // the whole x-run of this row belongs to the shell: contiguous cells
          const int beg = cell_start[row + x0], end = cell_start[row + x1 + 1];
          for (int k = beg; k < end; ++k) {
            const float4 p = sorted[k];
            nn3_insert(sqdist3(qx, qy, qz, p.x, p.y, p.z), __float_as_int(p.w), d0, d1, d2, i0, i1, i2);
          }
        } else {                // only the two end cells of the row
          if (cx - R >= 0) {
            const int beg = cell_start[row + cx - R], end = cell_start[row + cx - R + 1];
            for (int k = beg; k < end; ++k) {
              const float4 p = sorted[k];
              nn3_insert(sqdist3(qx, qy, qz, p.x, p.y, p.z), __float_as_int(p.w), d0, d1, d2, i0, i1, i2);
            }
          }
          if (R > 0 && cx + R < H.dim[0]) {
            const int beg = cell_start[row + cx + R], end = cell_start[row + cx + R + 1];
            for (int k = beg; k < end; ++k) {
              const float4 p = sorted[k];
              nn3_insert(sqdist3(qx, qy, qz, p.x, p.y, p.z), __float_as_int(p.w), d0, d1, d2, i0, i1, i2);
            }
          }
        }
      }
    }
    // every key not yet visited lies outside the cube of cells [c-R, c+R]; its distance to the query is
    // at least the distance to the nearest cube face that is still inside the grid
    float reach = __builtin_inff();
    if (cx - R > 0) reach = fminf(reach, qx - (H.lo[0] + (float)(cx - R) * H.h));
    if (cx + R < H.dim[0] - 1) reach = fminf(reach, (H.lo[0] + (float)(cx + R + 1) * H.h) - qx);
    if (cy - R > 0) reach = fminf(reach, qy - (H.lo[1] + (float)(cy - R) * H.h));
    if (cy + R < H.dim[1] - 1) reach = fminf(reach, (H.lo[1] + (float)(cy + R + 1) * H.h) - qy);
    if (cz - R > 0) reach = fminf(reach, qz - (H.lo[2] + (float)(cz - R) * H.h));
    if (cz + R < H.dim[2] - 1) reach = fminf(reach, (H.lo[2] + (float)(cz + R + 1) * H.h) - qz);
    reach = reach * 0.9999f - (H.eps + 4e-6f * fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz)));   // conservative against fp32 rounding of cell edges
    if (reach > 0.f && d2 < reach * reach) break;  // strict: an unvisited key at the same distance could still win a tie
  }
  const int64_t o = ((int64_t)b * N1 + i) * 3;
  index[o + 0] = i0; index[o + 1] = i1; index[o + 2] = i2;
  dist2[o + 0] = d0; dist2[o + 1] = d1; dist2[o + 2] = d2;
}

// ---- ball query over the grid ---------------------------------------------------------------------
// A wave owns one centroid.  `best` (one key per lane, ascending by lane) holds the 64 smallest
// in-radius indices seen so far; new hits are appended to a 64-entry LDS row in ballot order and, when
// the row is full (and once at the end), merged in with a 128-key bitonic network held in registers
// (element e = lane -> best, e = 64 + lane -> the pending key).  0xffffffff = empty slot.
__device__ __forceinline__ void bitonic128_keep64(unsigned& best, unsigned& pend, int lane) {
#pragma unroll
  for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 64) {                       // partner of element `lane` is element 64 + lane: same lane
        const unsigned lo = min(best, pend), hi = max(best, pend);
        best = lo;                         // k == 128: the whole sequence is ascending
        pend = hi;
      } else {
        const unsigned ob = __shfl_xor(best, j, 64), op = __shfl_xor(pend, j, 64);
        const bool lower = (lane & j) == 0;             // this element is the lower index of its pair
        const bool up_b = (lane & k) == 0;              // sort direction of element `lane`
        const bool up_p = ((64 + lane) & k) == 0;       // ... and of element 64 + lane
        best = (lower == up_b) ? min(best, ob) : max(best, ob);
        pend = (lower == up_p) ? min(pend, op) : max(pend, op);
      }
    }
  }
}

__global__ __launch_bounds__(256) void ball_query_grid_kernel(const float* __restrict__ ctr, int64_t cb, int64_t cc,
                                                              int64_t cn, int N2, float r2, float radius, int K,
                                                              const char* __restrict__ ws, long long slab,
                                                              int64_t* __restrict__ index,
                                                              int64_t* __restrict__ count) {
  __shared__ unsigned pendbuf[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + wave;
  const int b = blockIdx.y;
  if (c >= N2) return;   // wave-uniform
  char* my = const_cast<char*>(ws) + (long long)b * slab;
  const GridHeader H = *reinterpret_cast<const GridHeader*>(my);
  const int* cell_start = grid_cell_start(my);
  const float4* sorted = grid_sorted(my);
  const float* cbase = ctr + (int64_t)b * cb;
  const float qx = cbase[(int64_t)c * cn], qy = cbase[cc + (int64_t)c * cn], qz = cbase[2 * cc + (int64_t)c * cn];
  const float pad = radius * 1.0001f + H.eps + 4e-6f * fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz));   // conservative cell range of the ball
  const int x0 = cell_coord(qx - pad, H.lo[0], H.inv_h, H.dim[0]), x1 = cell_coord(qx + pad, H.lo[0], H.inv_h, H.dim[0]);
  const int y0 = cell_coord(qy - pad, H.lo[1], H.inv_h, H.dim[1]), y1 = cell_coord(qy + pad, H.lo[1], H.inv_h, H.dim[1]);
  const int z0 = cell_coord(qz - pad, H.lo[2], H.inv_h, H.dim[2]), z1 = cell_coord(qz + pad, H.lo[2], H.inv_h, H.dim[2]);
  unsigned best = 0xffffffffu;
  int npend = 0, total = 0;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  unsigned* row_buf = pendbuf[wave];
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      const int row = (z * H.dim[1] + y) * H.dim[0];
      const int beg = cell_start[row + x0], end = cell_start[row + x1 + 1];   // the x-run of cells is contiguous
      for (int k0 = beg; k0 < end; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        unsigned j = 0xffffffffu;
        if (k < end) {
          const float4 p = sorted[k];
          hit = sqdist3(p.x, p.y, p.z, qx, qy, qz) < r2;   // point minus centroid, strict, as the reference
          j = (unsigned)__float_as_int(p.w);
        }
        const unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        const int nh = (int)__popcll(mask);
        const int rank = (int)__popcll(mask & lt_mask);
        const int room = 64 - npend;
        if (hit && rank < room) row_buf[npend + rank] = j;
        if (nh >= room) {                 // row full: merge it, then stage the leftover hits
          unsigned pend = row_buf[lane];
          bitonic128_keep64(best, pend, lane);
          if (hit && rank >= room) row_buf[rank - room] = j;
          npend = nh - room;
        } else {
          npend += nh;
        }
        total += nh;
      }
    }
  if (npend > 0) {
    unsigned pend = lane < npend ? row_buf[lane] : 0xffffffffu;
    bitonic128_keep64(best, pend, lane);
  }
  // lane l now holds the l-th smallest in-radius index; slots [cnt, K) repeat the first hit (0 if none)
  const int cnt = min(total, K);
  const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)best);
  int64_t* out = index + ((int64_t)b * N2 + c) * K;
  if (lane < K) out[lane] = lane < cnt ? (int64_t)best : (cnt > 0 ? (int64_t)first : 0);
  if (lane == 0) count[(int64_t)b * N2 + c] = cnt;
}

// ---- surface normals over the grid ----------------------------------------------------------------------
// Reference: dataset_utils/eval_score/eval_utils/pointcloud.py:27-43 -- open3d's estimate_normals with
// KDTreeSearchParamHybrid(radius, max_nn), normalize_normals, orient_normals_towards_camera_location.  open3d is not
// part of the reference tree; the restated algorithm is its documented one: neighbours = the max_nn nearest points
// with squared distance (double, ((dx*dx)+(dy*dy))+(dz*dz)) < float(radius*radius), the query point included;
// fewer than 3 -> normal (0,0,1); else the eigenvector of the neighbours' covariance with the smallest eigenvalue;
// a zero vector becomes (0,0,1); the result is normalised and flipped to face the camera.
//
// normal_cov_kernel: a wave owns one point, visits the cells overlapping its ball, keeps the 64 nearest
// (distance, index) pairs with the same 128-wide bitonic merge as the ball query (pairs instead of bare indices) and
// reduces the second moments of the first max_nn of them, taken relative to the query point, in double.
// normal_eigen_kernel: a thread per point diagonalises the 3x3 covariance with cyclic Jacobi rotations in double.
struct NnKey {
  unsigned long long d;   // bit pattern of the non-negative double squared distance (orders like the value)
  unsigned j;             // original index: ties rank by index, so the selection does not depend on the cell order
};
__device__ __forceinline__ bool nn_less(const NnKey& a, const NnKey& b) { return a.d < b.d || (a.d == b.d && a.j < b.j); }
__device__ __forceinline__ NnKey nn_shfl_xor(const NnKey& a, int m) {
  NnKey r;
  r.d = __shfl_xor(a.d, m, 64);
  r.j = __shfl_xor(a.j, m, 64);
  return r;
}
__device__ __forceinline__ void nn_bitonic128_keep64(NnKey& best, NnKey& pend, int lane) {
#pragma unroll
  for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 64) {
        if (nn_less(pend, best)) { const NnKey t = best; best = pend; pend = t; }
      } else {
        const NnKey ob = nn_shfl_xor(best, j), op = nn_shfl_xor(pend, j);
        const bool lower = (lane & j) == 0;
        const bool up_b = (lane & k) == 0;
        const bool up_p = ((64 + lane) & k) == 0;
        const bool take_min_b = lower == up_b, take_min_p = lower == up_p;
        if (take_min_b ? nn_less(ob, best) : nn_less(best, ob)) best = ob;
        if (take_min_p ? nn_less(op, pend) : nn_less(pend, op)) pend = op;
      }
    }
  }
}

#define NN_EMPTY 0xffffffffffffffffull

__global__ __launch_bounds__(256) void normal_cov_kernel(const float* __restrict__ xyz, int N, double r2, float radius,
                                                         int max_nn, const char* __restrict__ ws,
                                                         double* __restrict__ cov, int* __restrict__ count) {
  __shared__ NnKey pendbuf[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;   // wave-uniform
  const GridHeader H = *reinterpret_cast<const GridHeader*>(ws);
  const int* cell_start = grid_cell_start(const_cast<char*>(ws));
  const float4* sorted = grid_sorted(const_cast<char*>(ws));
  const float qx = xyz[(int64_t)i * 3], qy = xyz[(int64_t)i * 3 + 1], qz = xyz[(int64_t)i * 3 + 2];
  const float pad = radius * 1.0001f + H.eps + 4e-6f * fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz));
  const int x0 = cell_coord(qx - pad, H.lo[0], H.inv_h, H.dim[0]), x1 = cell_coord(qx + pad, H.lo[0], H.inv_h, H.dim[0]);
  const int y0 = cell_coord(qy - pad, H.lo[1], H.inv_h, H.dim[1]), y1 = cell_coord(qy + pad, H.lo[1], H.inv_h, H.dim[1]);
  const int z0 = cell_coord(qz - pad, H.lo[2], H.inv_h, H.dim[2]), z1 = cell_coord(qz + pad, H.lo[2], H.inv_h, H.dim[2]);
  NnKey best;
  best.d = NN_EMPTY; best.j = 0xffffffffu;
  int npend = 0, total = 0;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  NnKey* row_buf = pendbuf[wave];
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      const int row = (z * H.dim[1] + y) * H.dim[0];
      const int beg = cell_start[row + x0], end = cell_start[row + x1 + 1];
      for (int k0 = beg; k0 < end; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        NnKey key;
        key.d = NN_EMPTY; key.j = 0xffffffffu;
        if (k < end) {
          const float4 p = sorted[k];
          const double dx = (double)p.x - (double)qx, dy = (double)p.y - (double)qy, dz = (double)p.z - (double)qz;
          const double d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
          hit = d2 < r2;
          key.d = (unsigned long long)__double_as_longlong(d2);
          key.j = (unsigned)__float_as_int(p.w);
        }
        const unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        const int nh = (int)__popcll(mask);
        const int rank = (int)__popcll(mask & lt_mask);
        const int room = 64 - npend;
        if (hit && rank < room) row_buf[npend + rank] = key;
        if (nh >= room) {
          NnKey pend = row_buf[lane];
          nn_bitonic128_keep64(best, pend, lane);
          if (hit && rank >= room) row_buf[rank - room] = key;
          npend = nh - room;
        } else {
          npend += nh;
        }
        total += nh;
      }
    }
  if (npend > 0) {
    NnKey pend;
    pend.d = NN_EMPTY; pend.j = 0xffffffffu;
    if (lane < npend) pend = row_buf[lane];
    nn_bitonic128_keep64(best, pend, lane);
  }
  // lane l holds the l-th nearest neighbour; the first `cnt` lanes contribute
  const int cnt = min(total, max_nn);
  double s[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
  if (lane < cnt) {
    const int64_t j = (int64_t)best.j;
    const double dx = (double)xyz[j * 3] - (double)qx, dy = (double)xyz[j * 3 + 1] - (double)qy,
                 dz = (double)xyz[j * 3 + 2] - (double)qz;
    s[0] = dx; s[1] = dy; s[2] = dz;
    s[3] = dx * dx; s[4] = dx * dy; s[5] = dx * dz; s[6] = dy * dy; s[7] = dy * dz; s[8] = dz * dz;
  }
#pragma unroll
  for (int a = 0; a < 9; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[a] += __shfl_xor(s[a], off, 64);
  }
  if (lane == 0) {
    count[i] = cnt;
    double* c = cov + (int64_t)i * 6;
    if (cnt > 0) {
      const double inv = 1.0 / (double)cnt;
      const double mx = s[0] * inv, my = s[1] * inv, mz = s[2] * inv;
      c[0] = s[3] * inv - mx * mx; c[1] = s[4] * inv - mx * my; c[2] = s[5] * inv - mx * mz;
      c[3] = s[6] * inv - my * my; c[4] = s[7] * inv - my * mz; c[5] = s[8] * inv - mz * mz;
    } else {
      c[0] = c[1] = c[2] = c[3] = c[4] = c[5] = 0.;
    }
  }
}

// Eigenvector of the smallest eigenvalue of the symmetric matrix [[a0,a1,a2],[a1,a3,a4],[a2,a4,a5]]: cyclic Jacobi.
// A zero matrix returns (1,0,0) (the first column of the identity, what a QR-based solver leaves untouched).
__device__ inline void sym3_min_eigenvector(const double* a, double* n) {
  double A[3][3] = {{a[0], a[1], a[2]}, {a[1], a[3], a[4]}, {a[2], a[4], a[5]}};
  double V[3][3] = {{1., 0., 0.}, {0., 1., 0.}, {0., 0., 1.}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    if (off == 0. || off <= 1e-300 + 1e-22 * diag) break;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const double apq = A[p][q];
      if (apq == 0.) continue;
      const double theta = (A[q][q] - A[p][p]) / (2. * apq);
      const double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
      const double c = 1. / sqrt(t * t + 1.), sn = t * c;
      const int r = 3 - p - q;
      const double app = A[p][p], aqq = A[q][q], arp = A[r][p], arq = A[r][q];
      A[p][p] = app - t * apq;
      A[q][q] = aqq + t * apq;
      A[p][q] = A[q][p] = 0.;
      A[r][p] = A[p][r] = c * arp - sn * arq;
      A[r][q] = A[q][r] = sn * arp + c * arq;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - sn * vkq;
        V[k][q] = sn * vkp + c * vkq;
      }
    }
  }
  int m = 0;
  if (A[1][1] < A[m][m]) m = 1;
  if (A[2][2] < A[m][m]) m = 2;
  n[0] = m == 0 ? V[0][0] : (m == 1 ? V[0][1] : V[0][2]);
  n[1] = m == 0 ? V[1][0] : (m == 1 ? V[1][1] : V[1][2]);
  n[2] = m == 0 ? V[2][0] : (m == 1 ? V[2][1] : V[2][2]);
}

__global__ __launch_bounds__(256) void normal_eigen_kernel(const float* __restrict__ xyz, int N,
                                                           const double* __restrict__ cov, const int* __restrict__ count,
                                                           double cx, double cy, double cz, float* __restrict__ normals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double n[3] = {0., 0., 1.};
  if (count[i] >= 3) {
    sym3_min_eigenvector(cov + (int64_t)i * 6, n);
    if (n[0] == 0. && n[1] == 0. && n[2] == 0.) { n[2] = 1.; }
  }
  double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  if (len > 0.) { n[0] /= len; n[1] /= len; n[2] /= len; }
  const double rx = cx - (double)xyz[(int64_t)i * 3], ry = cy - (double)xyz[(int64_t)i * 3 + 1],
               rz = cz - (double)xyz[(int64_t)i * 3 + 2];
  if (n[0] * rx + n[1] * ry + n[2] * rz < 0.) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
  normals[(int64_t)i * 3] = (float)n[0];
  normals[(int64_t)i * 3 + 1] = (float)n[1];
  normals[(int64_t)i * 3 + 2] = (float)n[2];
}

// ---- C ABI ------------------------------------------------------------------------------------------
extern "C" int64_t regnet_grid_workspace_bytes(int64_t B, int64_t N) { return B * grid_slab_bytes(N); }

static int build_grid(const float* xyz, int64_t sb, int64_t sc, int64_t sn, int64_t B, int64_t N, float h, void* ws,
                      hipStream_t st) {
  const long long slab = grid_slab_bytes(N);
  const dim3 per_point((unsigned)((N + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(grid_bounds_kernel, dim3((unsigned)B), dim3(GRID_WG), 0, st, xyz, sb, sc, sn, (int)N, h, (char*)ws, slab);
  if (N > 0) hipLaunchKernelGGL(grid_count_kernel, per_point, dim3(256), 0, st, xyz, sb, sc, sn, (int)N, (char*)ws, slab);
  hipLaunchKernelGGL(grid_scan_kernel, dim3((unsigned)B), dim3(GRID_WG), 0, st, (char*)ws, slab);
  if (N > 0) hipLaunchKernelGGL(grid_scatter_kernel, per_point, dim3(256), 0, st, xyz, sb, sc, sn, (int)N, (char*)ws, slab);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_three_nn_grid_f32(const float* query, int64_t qb, int64_t qc, int64_t qn, const float* key,
                                        int64_t kb, int64_t kc, int64_t kn, int64_t B, int64_t N1, int64_t N2,
                                        int64_t* index, float* dist2, void* workspace, void* stream) {
  if (N2 < 3 || B < 0 || N1 < 0) return REGNET_ERR_SHAPE;
  if (N1 >= (int64_t)1 << 31 || N2 >= (int64_t)1 << 31 || B > 65535) return REGNET_ERR_UNSUPPORTED;
  if (B == 0 || N1 == 0) return REGNET_OK;
  if (!query || !key || !index || !dist2 || !workspace) return REGNET_ERR_NULL;
  hipStream_t st = as_stream(stream);
  int rc = build_grid(key, kb, kc, kn, B, N2, 0.f, workspace, st);
  if (rc) return rc;
  dim3 grid((unsigned)((N1 + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(three_nn_grid_kernel, grid, dim3(256), 0, st, query, qb, qc, qn, (int)N1, (const char*)workspace,
                     grid_slab_bytes(N2), index, dist2);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int regnet_ball_query_grid_f32(const float* xyz, int64_t sb, int64_t sc, int64_t sn, const float* centroids,
                                          int64_t cb, int64_t cc, int64_t cn, int64_t B, int64_t N1, int64_t N2,
                                          float radius, int64_t K, int64_t* index, int64_t* count, void* workspace,
                                          void* stream) {
  if (K <= 0 || B < 0 || N1 < 0 || N2 < 0) return REGNET_ERR_SHAPE;
  if (K > 64 || !(radius > 0.f) || N1 >= (int64_t)1 << 31 || N2 >= (int64_t)1 << 31 || B > 65535)
    return REGNET_ERR_UNSUPPORTED;
  if (B == 0 || N2 == 0) return REGNET_OK;
  if (!centroids || !index || !count || !workspace || (N1 > 0 && !xyz)) return REGNET_ERR_NULL;
  hipStream_t st = as_stream(stream);
  int rc = build_grid(xyz, sb, sc, sn, B, N1, radius, workspace, st);
  if (rc) return rc;
  const float r2 = radius * radius;
  dim3 grid((unsigned)((N2 + 3) / 4), (unsigned)B);
  hipLaunchKernelGGL(ball_query_grid_kernel, grid, dim3(256), 0, st, centroids, cb, cc, cn, (int)N2, r2, radius, (int)K,
                     (const char*)workspace, grid_slab_bytes(N1), index, count);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}

extern "C" int64_t regnet_normals_workspace_bytes(int64_t N) {
  const long long grid = (grid_slab_bytes(N) + 15) / 16 * 16;
  return grid + N * 6 * 8ll + N * 4ll;
}

extern "C" int regnet_estimate_normals_f32(const float* xyz, int64_t N, double radius, int64_t max_nn, double cam_x,
                                           double cam_y, double cam_z, float* normals, int32_t* count, void* workspace,
                                           void* stream) {
  if (N < 0 || max_nn <= 0) return REGNET_ERR_SHAPE;
  if (max_nn > 64 || !(radius > 0.) || N >= (int64_t)1 << 31) return REGNET_ERR_UNSUPPORTED;
  if (N == 0) return REGNET_OK;
  if (!xyz || !normals || !workspace) return REGNET_ERR_NULL;
  hipStream_t st = as_stream(stream);
  int rc = build_grid(xyz, 0, 1, 3, 1, N, (float)radius, workspace, st);
  if (rc) return rc;
  const long long grid = (grid_slab_bytes(N) + 15) / 16 * 16;
  double* cov = reinterpret_cast<double*>((char*)workspace + grid);
  int* cnt = count ? count : reinterpret_cast<int*>((char*)workspace + grid + N * 6 * 8ll);
  const double r2 = (double)(float)(radius * radius);   // the search takes the squared radius as a float
  hipLaunchKernelGGL(normal_cov_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, xyz, (int)N, r2, (float)radius,
                     (int)max_nn, (const char*)workspace, cov, cnt);
  hipLaunchKernelGGL(normal_eigen_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, xyz, (int)N, cov, cnt, cam_x,
                     cam_y, cam_z, normals);
  REGNET_LAUNCH_CHECK();
  return REGNET_OK;
}
