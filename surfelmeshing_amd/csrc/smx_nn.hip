// smx_nn.hip -- batched radius-neighbor search for the mesher (gfx950).
//
// Replaces, for batched queries, CompressedOctree::FindNearestSurfelsWithinRadius
// (APP/octree.cc:313-470): up to K nearest points with dist^2 <= r^2 in ascending
// order, optional exclusion by a per-point state byte (octree.cc:330-335).  The
// pointer-chasing octree is replaced by a sparse uniform grid of SORTED CELL KEYS
// (DESIGN.md "Neighbor search"):
//   build:  bounding box -> 64-bit key per point (4x4x4-cell brick index, cell inside the brick)
//           -> LSD radix sort of (key, index) (LDS histograms, wave-ballot ranks: no global atomic
//           per point) -> (x, y, z, index) records in key order -> one hash-table entry per
//           OCCUPIED brick (start, end).  The grid is never materialised, so the cell size is
//           the caller's at any scene extent.
//   query:  queries are keyed by their brick and radix-sorted as well, so that the queries of one
//           brick form a tile (<= 64 queries, one wavefront).  The tile stages the points of the
//           bricks its search balls overlap in LDS ONCE (the points are binned by cell key, so that
//           is a handful of contiguous ranges); then the wavefront answers the tile's queries one
//           after the other from LDS, 64 candidates per step, the running top-K one entry per lane
//           in registers (ballot / shuffle insertion), and writes each finished row out coalesced.
// All workspace is owned by the handle and reused from call to call; a query with device-resident
// inputs and outputs enqueues kernels only (no allocation, no synchronisation).  A build reads back
// 32 bytes twice (bounding box; brick and point counts).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "smx_common.hpp"

using namespace smx;

namespace {

constexpr int kBlock = 256;
constexpr int kScanPerThread = 8;
constexpr int kScanPerBlock = kBlock * kScanPerThread;

__device__ __forceinline__ uint32_t order_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float order_unkey(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// Points with a non-finite coordinate are left out of the index (no finite ball contains them); the
// reconstruction parks merged surfel slots that way (smx_recon_build_neighbor_index).
__device__ __forceinline__ bool indexable(float x, float y, float z) {
  return fabsf(x) <= 3.0e38f && fabsf(y) <= 3.0e38f && fabsf(z) <= 3.0e38f;
}

__global__ void __launch_bounds__(kBlock)
k_bbox(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
       uint32_t* __restrict__ bb /* per workgroup: min xyz, max xyz as order keys */) {
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0, 0, 0};
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    if (!indexable(x[i], y[i], z[i])) continue;
    const uint32_t k[3] = {order_key(x[i]), order_key(y[i]), order_key(z[i])};
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], k[a]); mx[a] = max(mx[a], k[a]); }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off));
      mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off));
    }
  }
  // one partial per workgroup (no atomics on six addresses from every wavefront), reduced by k_bbox_finish
  __shared__ uint32_t part[kBlock / 64][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { part[threadIdx.x >> 6][a] = mn[a]; part[threadIdx.x >> 6][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    uint32_t v = part[0][threadIdx.x];
    for (int w = 1; w < kBlock / 64; ++w) v = threadIdx.x < 3 ? min(v, part[w][threadIdx.x]) : max(v, part[w][threadIdx.x]);
    bb[6 * blockIdx.x + threadIdx.x] = v;
  }
}
__global__ void __launch_bounds__(kBlock)
k_bbox_finish(const uint32_t* __restrict__ partial, int nparts, uint32_t* __restrict__ bb) {
  __shared__ uint32_t part[kBlock / 64][6];
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < nparts; i += kBlock)
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], partial[6 * i + a]); mx[a] = max(mx[a], partial[6 * i + 3 + a]); }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off));
      mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { part[threadIdx.x >> 6][a] = mn[a]; part[threadIdx.x >> 6][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    uint32_t v = part[0][threadIdx.x];
    for (int w = 1; w < kBlock / 64; ++w) v = threadIdx.x < 3 ? min(v, part[w][threadIdx.x]) : max(v, part[w][threadIdx.x]);
    bb[threadIdx.x] = v;
  }
}

// Cells are grouped into bricks of 4x4x4: the sort key of a point is (brick index, cell inside the brick), so the
// points of a brick -- the unit the query stages in LDS -- are contiguous, and inside it ordered by cell.
constexpr int kBrickShift = 2, kBrickCells = 1 << kBrickShift, kLocalBits = 3 * kBrickShift;
struct Grid {
  float min[3];
  float cell;
  int dim[3];    // cells per axis
  int bdim[3];   // bricks per axis
  int key_bits;  // bits of the largest point key + 1 (the "not indexable" sentinel is one past the largest key)
  int brick_bits;
  unsigned long long sentinel;  // key of points that are not indexed: sorts behind every valid key
};

__device__ __forceinline__ int cell_coord(float p, float mn, float cell, int dim) {
  int c = (int)floorf((p - mn) / cell);
  return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}
// The bricks' order in the sorted index -- and with it the order in which the self-query tiles are worked on -- follows a
// Z-order (Morton) curve (SMX_NN_MORTON 1; 0 = rows, columns, planes: rounds 2 - 6).  A tile stages the bricks AROUND its own, and
// with linear brick numbers the bricks of the next row lie a row of tiles away and those of the next plane hundreds of thousands:
// whatever one tile staged was gone from the L2 when its neighbours came (hit rate 51 %, PMC traffic 1.74 x the algorithmic bytes,
// unmoved by every tile-mapping variant of round 6).  Results do not depend on the order (rows are sorted by (d^2, index)).
#ifndef SMX_NN_MORTON
#define SMX_NN_MORTON 1
#endif
__host__ __device__ __forceinline__ unsigned long long spread3(unsigned long long v) {   // bit k of the 21-bit v -> bit 3 k
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__device__ __forceinline__ unsigned long long brick_index(const Grid& g, int bx, int by, int bz) {
#if SMX_NN_MORTON
  (void)g;
  return spread3((unsigned long long)bx) | (spread3((unsigned long long)by) << 1) | (spread3((unsigned long long)bz) << 2);
#else
  return ((unsigned long long)bz * (unsigned long long)g.bdim[1] + (unsigned long long)by) * (unsigned long long)g.bdim[0] +
         (unsigned long long)bx;
#endif
}

__global__ void __launch_bounds__(kBlock)
k_point_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n, Grid g,
             unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    unsigned long long k = g.sentinel;
    if (indexable(x[i], y[i], z[i])) {
      const int cx = cell_coord(x[i], g.min[0], g.cell, g.dim[0]);
      const int cy = cell_coord(y[i], g.min[1], g.cell, g.dim[1]);
      const int cz = cell_coord(z[i], g.min[2], g.cell, g.dim[2]);
      const uint32_t local = (uint32_t)(((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3));
      k = (brick_index(g, cx >> kBrickShift, cy >> kBrickShift, cz >> kBrickShift) << kLocalBits) | local;
    }
    keys[i] = k;
    vals[i] = i;
  }
}

// ---- exclusive scan (u32), multi-level, workspace supplied by the caller ------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_scan_block(uint32_t* __restrict__ data, size_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t wave_tot[kBlock / 64];
  const size_t base = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kScanPerThread;
  uint32_t v[kScanPerThread];
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) { v[j] = (base + j < n) ? data[base + j] : 0; mine += v[j]; }
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off);
    if (lane >= (uint32_t)off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) { if ((uint32_t)w < wave) wave_off += wave_tot[w]; total += wave_tot[w]; }
  uint32_t run = wave_off + incl - mine;
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) { if (base + j < n) data[base + j] = run; run += v[j]; }
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock)
k_scan_add(uint32_t* __restrict__ data, size_t n, const uint32_t* __restrict__ offsets) {
  const size_t base = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kScanPerThread;
  const uint32_t off = offsets[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) if (base + j < n) data[base + j] += off;
}

size_t scan_workspace_elems(size_t n) {
  size_t total = 0;
  for (;;) {
    n = (n + kScanPerBlock - 1) / kScanPerBlock;
    if (n == 0) n = 1;
    total += n;
    if (n == 1) break;
  }
  return total;
}
// In-place exclusive scan of data[0 .. n) (n >= 1); *total_out = where the grand total ends up (device memory).
void exclusive_scan_inplace(uint32_t* data, size_t n, uint32_t* ws, hipStream_t st, uint32_t** total_out) {
  const size_t nblocks = (n + kScanPerBlock - 1) / kScanPerBlock;
  hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nblocks), dim3(kBlock), 0, st, data, n, ws);
  if (nblocks > 1) {
    exclusive_scan_inplace(ws, nblocks, ws + nblocks, st, total_out);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nblocks), dim3(kBlock), 0, st, data, n, ws);
  } else if (total_out) {
    *total_out = ws;  // one block: its total is the grand total
  }
}

// ---- LSD radix sort of (u64 key, u32 value), 8 bits per pass ----------------------------------------------------------
// (Round 4 measured 11-bit digits -- 3 passes instead of 5 for the 33-bit keys of C5, 4096-key tiles: index build 4.1 ->
// 5.4 ms.  A pass scatters a tile's keys in runs of tile / radix: 8 keys = 64 bytes with 256 digits, 2 keys = 16 bytes with
// 2048 -- half of every 32-byte sector and the 4-byte values worse; fewer passes do not make up for it.)
// A tile of 2048 keys per workgroup.  Histogram: LDS atomics, one store per (digit, tile).  Ranks: wave w owns the
// 512 consecutive keys [w*512, (w+1)*512) of the tile and walks them 64 at a time; lanes with equal digits find each
// other with 8 ballots, so the rank of a key among the equal digits before it is a popcount -- stable, no atomics.
constexpr int kSortItems = 8, kSortTile = kBlock * kSortItems, kRadix = 256;

__global__ void __launch_bounds__(kBlock)
k_rs_hist(const unsigned long long* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ hist, uint32_t nblocks) {
  __shared__ uint32_t h[kRadix];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const uint32_t i = base + r * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(kBlock)
k_rs_scatter(const unsigned long long* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
             unsigned long long* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n, int shift,
             const uint32_t* __restrict__ hist_scanned, uint32_t nblocks) {
  __shared__ uint32_t wcount[kBlock / 64][kRadix];  // per wave: equal digits seen so far, then exclusive over the waves
  __shared__ uint32_t gbase[kRadix];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) wcount[w][threadIdx.x] = 0;
  gbase[threadIdx.x] = hist_scanned[(size_t)threadIdx.x * nblocks + blockIdx.x];
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile + wave * (kSortTile / (kBlock / 64));
  unsigned long long k[kSortItems];
  uint32_t v[kSortItems], lr[kSortItems];
  volatile uint32_t* wc = wcount[wave];
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const uint32_t i = base + r * 64 + lane;
    const bool valid = i < n;
    k[r] = valid ? keys_in[i] : 0ull;
    v[r] = valid ? vals_in[i] : 0u;
    const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bb = __ballot(bit);
      m &= bit ? bb : ~bb;
    }
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const uint32_t old = wc[d];                                    // every lane of the group reads the same counter ...
    __builtin_amdgcn_wave_barrier();                               // ... before
    if (valid && rank == 0) wc[d] = old + (uint32_t)__popcll(m);   // ... its first lane advances it (one wave, LDS in order)
    __builtin_amdgcn_wave_barrier();
    lr[r] = old + rank;
  }
  __syncthreads();
  {  // exclusive prefix over the waves, per digit
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { const uint32_t c = wcount[w][threadIdx.x]; wcount[w][threadIdx.x] = run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const uint32_t i = base + r * 64 + lane;
    if (i >= n) continue;
    const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
    const uint32_t pos = gbase[d] + wcount[wave][d] + lr[r];
    keys_out[pos] = k[r];
    vals_out[pos] = v[r];
  }
}

// (x, y, z, index) records in key order
__global__ void __launch_bounds__(kBlock)
k_gather_records(const uint32_t* __restrict__ order, const float* __restrict__ x, const float* __restrict__ y,
                 const float* __restrict__ z, uint32_t n, float4* __restrict__ sorted) {
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
    const uint32_t i = order[j];
    sorted[j] = make_float4(x[i], y[i], z[i], __uint_as_float(i));
  }
}

// ---- hash table of the occupied bricks -------------------------------------------------------------------------------
struct BrickSlot { unsigned long long key; uint32_t start, end; };   // key = brick index + 1 (0 = empty slot)
__device__ __forceinline__ uint32_t hash_brick(unsigned long long b, uint32_t mask) {
  b ^= b >> 33; b *= 0xff51afd7ed558ccdull; b ^= b >> 33; b *= 0xc4ceb9fe1a85ec53ull; b ^= b >> 33;
  return (uint32_t)b & mask;
}
__device__ __forceinline__ bool find_brick(const BrickSlot* __restrict__ table, uint32_t mask, unsigned long long brick,
                                           uint32_t& start, uint32_t& end) {
  uint32_t s = hash_brick(brick, mask);
  for (;;) {
    const uint4 e = *reinterpret_cast<const uint4*>(&table[s]);
    const unsigned long long key = ((unsigned long long)e.y << 32) | e.x;
    if (key == brick + 1ull) { start = e.z; end = e.w; return true; }
    if (key == 0ull) return false;
    s = (s + 1u) & mask;
  }
}

// The same look-up with the first kSpec probe positions requested TOGETHER: a miss (most of the 27 bricks around a surface
// tile are empty) ends at the first empty slot of its probe sequence, two or three dependent round trips down the line when
// they are taken one at a time -- and the tile's look-ups last as long as the longest of them.
template <int kSpec>
__device__ __forceinline__ bool find_brick_spec(const BrickSlot* __restrict__ table, uint32_t mask, unsigned long long brick,
                                                uint32_t& start, uint32_t& end) {
  uint32_t s = hash_brick(brick, mask);
  uint4 e[kSpec];
#pragma unroll
  for (int u = 0; u < kSpec; ++u) e[u] = *reinterpret_cast<const uint4*>(&table[(s + (uint32_t)u) & mask]);
#pragma unroll
  for (int u = 0; u < kSpec; ++u) {
    const unsigned long long key = ((unsigned long long)e[u].y << 32) | e[u].x;
    if (key == brick + 1ull) { start = e[u].z; end = e[u].w; return true; }
    if (key == 0ull) return false;
  }
  s = (s + (uint32_t)kSpec) & mask;
  for (;;) {
    const uint4 f = *reinterpret_cast<const uint4*>(&table[s]);
    const unsigned long long key = ((unsigned long long)f.y << 32) | f.x;
    if (key == brick + 1ull) { start = f.z; end = f.w; return true; }
    if (key == 0ull) return false;
    s = (s + 1u) & mask;
  }
}

struct BuildCounts { uint32_t n_valid, n_bricks; };

// counts of indexed points and of occupied bricks (one partial per workgroup, summed by k_sum_counts)
__global__ void __launch_bounds__(kBlock)
k_brick_count(const unsigned long long* __restrict__ keys, uint32_t n, unsigned long long sentinel,
              uint32_t* __restrict__ partial /* [gridDim.x][2] */) {
  __shared__ uint32_t acc[2][kBlock / 64];
  uint32_t valid = 0, bricks = 0;
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
    const unsigned long long k = keys[j];
    if (k == sentinel) continue;
    ++valid;
    if (j == 0 || (keys[j - 1] >> kLocalBits) != (k >> kLocalBits)) ++bricks;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { valid += __shfl_xor(valid, off); bricks += __shfl_xor(bricks, off); }
  if ((threadIdx.x & 63) == 0) { acc[0][threadIdx.x >> 6] = valid; acc[1][threadIdx.x >> 6] = bricks; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = 0, b = 0;
    for (int w = 0; w < kBlock / 64; ++w) { a += acc[0][w]; b += acc[1][w]; }
    partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = b;
  }
}
__global__ void __launch_bounds__(kBlock)
k_sum_counts(const uint32_t* __restrict__ partial, int nparts, BuildCounts* __restrict__ out) {
  __shared__ uint32_t acc[2][kBlock / 64];
  uint32_t a = 0, b = 0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) { a += partial[2 * i]; b += partial[2 * i + 1]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
  if ((threadIdx.x & 63) == 0) { acc[0][threadIdx.x >> 6] = a; acc[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t va = 0, vb = 0;
    for (int w = 0; w < kBlock / 64; ++w) { va += acc[0][w]; vb += acc[1][w]; }
    out->n_valid = va; out->n_bricks = vb;
  }
}
// one compare-and-swap per OCCUPIED BRICK (its first point inserts the start) ...
__global__ void __launch_bounds__(kBlock)
k_brick_insert(const unsigned long long* __restrict__ keys, uint32_t n_valid, BrickSlot* __restrict__ table, uint32_t mask) {
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n_valid; j += gridDim.x * kBlock) {
    const unsigned long long b = keys[j] >> kLocalBits;
    if (j != 0 && (keys[j - 1] >> kLocalBits) == b) continue;
    uint32_t s = hash_brick(b, mask);
    for (;;) {
      const unsigned long long prev = atomicCAS(&table[s].key, 0ull, b + 1ull);
      if (prev == 0ull) { table[s].start = j; break; }
      s = (s + 1u) & mask;   // (brick keys are distinct: prev is another brick)
    }
  }
}
// ... and, in a second launch, the last point of every brick stores the end
__global__ void __launch_bounds__(kBlock)
k_brick_ends(const unsigned long long* __restrict__ keys, uint32_t n_valid, BrickSlot* __restrict__ table, uint32_t mask) {
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n_valid; j += gridDim.x * kBlock) {
    const unsigned long long b = keys[j] >> kLocalBits;
    if (j + 1 != n_valid && (keys[j + 1] >> kLocalBits) == b) continue;
    uint32_t s = hash_brick(b, mask);
    while (table[s].key != b + 1ull) s = (s + 1u) & mask;
    table[s].end = j + 1;
  }
}

// ---- queries -----------------------------------------------------------------------------------------------------------
// Conservative radius for the cell range of a query ball: covers the float rounding of the exact dist^2 <= r^2 test and
// of q -+ rad itself (half an ulp of the coordinate, which matters for |q| of tens of metres and millimetre radii).
__device__ __forceinline__ float cover_radius(float r2, float q) {
  return sqrtf(r2) * 1.0001f + 1e-6f + fabsf(q) * 2.4e-7f;
}

__global__ void __launch_bounds__(kBlock)
k_query_keys(const float* __restrict__ qx, const float* __restrict__ qy, const float* __restrict__ qz, uint32_t nq, Grid g,
             unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < nq; q += gridDim.x * kBlock) {
    // (a NaN coordinate lands in cell 0: such a query has an empty ball anyway)
    const int cx = cell_coord(qx[q], g.min[0], g.cell, g.dim[0]);
    const int cy = cell_coord(qy[q], g.min[1], g.cell, g.dim[1]);
    const int cz = cell_coord(qz[q], g.min[2], g.cell, g.dim[2]);
    keys[q] = brick_index(g, cx >> kBrickShift, cy >> kBrickShift, cz >> kBrickShift);
    vals[q] = q;
  }
}

// (x, y, z, r^2) of the queries in brick order: a tile reads its queries with one coalesced 16-byte load per lane
__global__ void __launch_bounds__(kBlock)
k_gather_queries(const uint32_t* __restrict__ order, const float* __restrict__ qx, const float* __restrict__ qy,
                 const float* __restrict__ qz, const float* __restrict__ qr2, uint32_t nq, float4* __restrict__ qrec) {
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < nq; j += gridDim.x * kBlock) {
    const uint32_t q = order[j];
    qrec[j] = make_float4(qx[q], qy[q], qz[q], qr2[q]);
  }
}

// Tile boundaries over the sorted queries: a tile is a run of <= 64 queries of one brick.  flag[j] = 1 where a tile starts.
constexpr int kTile = 64;
__global__ void __launch_bounds__(kBlock)
k_tile_flags(const unsigned long long* __restrict__ qkeys, uint32_t nq, int shift, uint32_t* __restrict__ flags) {
  // position of the last brick change at or before j, within the workgroup's chunk (chunk starts count as changes)
  __shared__ uint32_t wave_last[kBlock / 64];
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  const bool in = j < nq;
  const bool change = in && (threadIdx.x == 0 || (qkeys[j - 1] >> shift) != (qkeys[j] >> shift));
  uint32_t last = change ? threadIdx.x : 0u;   // inclusive max-scan over the workgroup
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(last, off);
    if (lane >= (uint32_t)off) last = max(last, t);
  }
  if (lane == 63) wave_last[wave] = last;
  __syncthreads();
  for (uint32_t w = 0; w < wave; ++w) last = max(last, wave_last[w]);
  if (in) flags[j] = ((threadIdx.x - last) % kTile == 0) ? 1u : 0u;
}
__global__ void __launch_bounds__(kBlock)
k_tile_starts(const uint32_t* __restrict__ tile_of /* exclusive scan of the flags */, uint32_t nq,
              const uint32_t* __restrict__ total, uint32_t* __restrict__ tile_start) {
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < nq; j += gridDim.x * kBlock) {
    const uint32_t t = tile_of[j];
    const uint32_t next = (j + 1 < nq) ? tile_of[j + 1] : *total;
    if (next != t) tile_start[t] = j;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) tile_start[*total] = nq;
}

// One tile = one workgroup of four wavefronts that share the staged candidate points (the only LDS user, so that 24
// wavefronts are resident per CU: the top-K insertion is a chain of cross-lane operations and needs the latency hiding).
constexpr int kStage = 1536;            // staged points per tile (24 KB): a brick and its 26 neighbours on a surface hold
                                        // 9 bricks x 16 cells x (cell / spacing)^2 points -- 324 at cell = 1.5 x spacing
constexpr unsigned long long kManyBricks = 1ull << 15;   // regions of more bricks than this stream every indexed point

struct QueryArgs {
  uint32_t nq;
  const float4* qrec;          // queries in brick order: x, y, z, r^2
  int K;
  const uint8_t* state;
  uint8_t skip_mask;
  Grid g;
  const BrickSlot* table;
  uint32_t mask;
  const float4* sorted;
  uint32_t n_valid;
  const uint32_t* qorder;      // query ids in brick order
  const uint32_t* tile_start;  // [n_tiles + 1]
  const uint32_t* n_tiles;
  uint32_t* out_idx; float* out_d2; int32_t* out_count;
  unsigned long long* stat;    // optional: [0] tiles, [1] staged candidates, [2] distance tests, [3] results
  const float* self_r2;        // kSelf: per-point r^2 (indexed by point index) times self_factor, or NULL: r^2 = self_factor
  float self_factor;
  // k_query_lanes answers the common case and marks what it cannot (out_count = -1, *redo_flag = 1); k_query_tiles
  // with redo = 1 then answers exactly the marked queries (and returns at once if nothing was marked)
  int redo;
  uint32_t* redo_flag;
  uint8_t* tile_redo;          // per tile (kSelf: per table slot): 1 = some query of it is marked
};

__device__ __forceinline__ bool before(float d2a, uint32_t ia, float d2b, uint32_t ib) {
  return d2a < d2b || (d2a == d2b && ia < ib);
}

// The tile's queries share one set of candidate points: the bricks their balls overlap are looked up once (64 hash
// probes at a time, one per lane) and their points -- contiguous ranges of the key-sorted records -- are staged in LDS
// once (or, when a tile's surroundings hold more than kStage points, streamed from the L2-resident ranges).  Then the
// wavefront answers the queries one after the other: 64 candidates per step (one per lane, conflict-free 16-byte LDS
// reads), the running top-K of the query lives one entry per lane in registers, ordered by (dist^2, index), and is
// updated with ballot / shuffle insertion; the finished row is written out coalesced.  A region of more than 64 bricks
// is handled 64 bricks at a time, the row written so far being the list the next round continues.
// value of lane `src` (uniform) / of the lane below, without a trip through the LDS crossbar
__device__ __forceinline__ float lane_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ uint32_t lane_u(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ int lane_below(int v) {   // DPP wave_shr:1 (lane 0 keeps its own value)
  return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false);
}

// kSelf: the queries are the indexed points themselves (the full-retriangulation pattern of config C5: every surfel asks
// for its own neighbourhood, APP/surfel_meshing.cc:549, 819-823).  Then nothing has to be keyed, sorted or gathered: a
// tile is an occupied slot of the brick table, its queries are that brick's own records.
constexpr int kTileWaves = 4;
template <bool kSelf>
__global__ void __launch_bounds__(64 * kTileWaves)
k_query_tiles(QueryArgs a) {
  __shared__ float4 stage[kStage];
  __shared__ uint32_t seg_end[64], seg_src[64];
  __shared__ uint32_t seg_total;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Grid& g = a.g;
  const uint32_t n_tiles = (kSelf && !a.tile_start) ? a.mask + 1u : *a.n_tiles;
  const int K = a.K;
  if (a.redo && *a.redo_flag == 0u) return;   // (uniform) k_query_lanes answered everything
  for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    uint32_t sub_begin, sub_end;
    if (kSelf && !a.tile_start) {
      const uint4 slot = *reinterpret_cast<const uint4*>(&a.table[t]);
      if ((slot.x | slot.y) == 0u) continue;   // (uniform) empty slot
      sub_begin = slot.z; sub_end = slot.w;
    } else {
      sub_begin = a.tile_start[t]; sub_end = a.tile_start[t + 1];
    }
    if (a.redo && !a.tile_redo[t]) continue;   // (uniform) nothing of this tile was marked
   for (uint32_t j0 = sub_begin; j0 < sub_end; j0 += 64) {   // (general tiles hold <= 64 queries: one pass)
    const uint32_t nt = min(sub_end - j0, 64u);
    bool have = lane < nt;
    uint32_t q = 0;
    float px = 0, py = 0, pz = 0, r2 = -1.0f;
    if (have) {
      if (kSelf) {
        const float4 rec = a.sorted[j0 + lane];
        q = __float_as_uint(rec.w); px = rec.x; py = rec.y; pz = rec.z;
        r2 = a.self_r2 ? a.self_r2[q] * a.self_factor : a.self_factor;
      } else {
        const float4 qr = a.qrec[j0 + lane]; q = a.qorder[j0 + lane]; px = qr.x; py = qr.y; pz = qr.z; r2 = qr.w;
      }
    }
    if (a.redo) {
      // only the queries k_query_lanes marked; the others keep the rows they have
      have = have && a.out_count[q] == -1;
      if (!have) r2 = -1.0f;
      if (!__syncthreads_or(have ? 1 : 0)) continue;   // (uniform)
    }
    // the lane's cell range (clamped to the grid), empty for r^2 < 0 / NaN
    int lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1};
    bool ball = have && (r2 >= 0);
    if (ball) {
      const float qp[3] = {px, py, pz};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float rad = cover_radius(r2, qp[k]);
        const float flo = floorf((qp[k] - rad - g.min[k]) / g.cell), fhi = floorf((qp[k] + rad - g.min[k]) / g.cell);
        // (compared as floats: no integer overflow; NaN compares false and leaves the range empty)
        lo[k] = flo >= 0.0f ? (flo < (float)g.dim[k] ? (int)flo : g.dim[k]) : 0;
        hi[k] = fhi >= 0.0f ? (fhi < (float)g.dim[k] ? (int)fhi : g.dim[k] - 1) : -1;
        if (!(lo[k] <= hi[k])) ball = false;
      }
    }
    if (!ball) r2 = -1.0f;   // (no candidate passes d2 <= r2)
    // the tile's region = bounding box of the lanes' ranges
    int rlo[3], rhi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int mn = ball ? lo[k] : 0x7FFFFFFF, mx = ball ? hi[k] : -1;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off)); mx = max(mx, __shfl_xor(mx, off)); }
      rlo[k] = mn; rhi[k] = mx;
    }
    uint32_t my_count = 0;   // results of the lane's own query so far (its row in the output holds them, sorted)
    unsigned long long n_tests = 0, n_staged = 0;
    if (rhi[0] >= rlo[0]) {   // (uniform) at least one lane has a ball
      const int blo[3] = {rlo[0] >> kBrickShift, rlo[1] >> kBrickShift, rlo[2] >> kBrickShift};
      const int bn[3] = {(rhi[0] >> kBrickShift) - blo[0] + 1, (rhi[1] >> kBrickShift) - blo[1] + 1, (rhi[2] >> kBrickShift) - blo[2] + 1};
      unsigned long long nbricks = (unsigned long long)bn[0] * bn[1] * bn[2];
      const bool everything = nbricks > kManyBricks;   // a ball that spans the map: every indexed point is a candidate
      if (everything) nbricks = 1;
      for (unsigned long long b0 = 0; b0 < nbricks; b0 += 64) {
        // this round's candidate ranges, one per lane
        const unsigned long long b = b0 + lane;
        uint32_t s = 0, e = 0;
        int clo[3] = {0, 0, 0}, chi[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF};   // cells the lane's range covers
        if (everything) {
          if (lane == 0) e = a.n_valid;
        } else if (b < nbricks) {
          const int bx = blo[0] + (int)(b % (unsigned long long)bn[0]);
          const int by = blo[1] + (int)((b / (unsigned long long)bn[0]) % (unsigned long long)bn[1]);
          const int bz = blo[2] + (int)(b / ((unsigned long long)bn[0] * (unsigned long long)bn[1]));
          if (!find_brick(a.table, a.mask, brick_index(g, bx, by, bz), s, e)) { s = 0; e = 0; }
          clo[0] = bx << kBrickShift; clo[1] = by << kBrickShift; clo[2] = bz << kBrickShift;
          chi[0] = clo[0] + kBrickCells - 1; chi[1] = clo[1] + kBrickCells - 1; chi[2] = clo[2] + kBrickCells - 1;
        }
        const uint32_t len = e - s;
        uint32_t incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t tv = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += tv; }
        __syncthreads();   // (the previous round's readers of the segment table and of the stage are done)
        if (wave == 0) {   // (every wave computed the same look-ups; one publishes them)
          seg_end[lane] = incl;               // the range of lane L fills flat positions [seg_end[L] - len, seg_end[L])
          seg_src[lane] = s - (incl - len);   // source index = seg_src[L] + flat position (mod 2^32)
          if (lane == 63) seg_total = incl;
        }
        __syncthreads();
        const uint32_t total = seg_total;
        if (total == 0) continue;   // (uniform)
        const bool staged = total <= (uint32_t)kStage;
        if (staged) {
          // one flat copy by all four waves: the loads of different bricks are independent and go out back to back
          uint32_t cur = 0;
          for (uint32_t k0 = 0; k0 < total; k0 += 2 * 64 * kTileWaves) {
            // (out-of-range lanes re-read the last point: unconditional loads keep the values in registers)
            const uint32_t ka = k0 + threadIdx.x, kb = ka + 64 * kTileWaves;
            uint32_t src0, src1;
            { const uint32_t k = min(ka, total - 1u); while (k >= seg_end[cur]) ++cur; src0 = seg_src[cur] + k; }
            { const uint32_t k = min(kb, total - 1u); while (k >= seg_end[cur]) ++cur; src1 = seg_src[cur] + k; }
            const float4 v0 = a.sorted[src0], v1 = a.sorted[src1];
            if (ka < total) stage[ka] = v0;
            if (kb < total) stage[kb] = v1;
          }
          __syncthreads();
          if (wave == 0) n_staged += total;
        }
        // the queries of the tile: wave w answers queries w, w + 4, ... one after the other
        for (uint32_t ql = wave; ql < nt; ql += kTileWaves) {
          const float qr2 = lane_f(r2, (int)ql);
          if (!(qr2 >= 0)) continue;   // (uniform) no ball
          const float qx = lane_f(px, (int)ql), qy = lane_f(py, (int)ql), qz = lane_f(pz, (int)ql);
          const uint32_t qq = lane_u(q, (int)ql);
          int count = (int)lane_u(my_count, (int)ql);
          // running top-K: lane j holds the j-th best (dist^2, index); continued from the row written by an earlier round
          float my_d2 = __builtin_inff();
          uint32_t my_idx = kInvalid;
          if ((int)lane < count) { my_d2 = a.out_d2[(size_t)qq * K + lane]; my_idx = a.out_idx[(size_t)qq * K + lane]; }
          // only the ranges (bricks) the query's own cell range overlaps: most queries sit inside their brick and need
          // one or two of the nine staged bricks
          const int qlo0 = __builtin_amdgcn_readlane(lo[0], (int)ql), qhi0 = __builtin_amdgcn_readlane(hi[0], (int)ql);
          const int qlo1 = __builtin_amdgcn_readlane(lo[1], (int)ql), qhi1 = __builtin_amdgcn_readlane(hi[1], (int)ql);
          const int qlo2 = __builtin_amdgcn_readlane(lo[2], (int)ql), qhi2 = __builtin_amdgcn_readlane(hi[2], (int)ql);
          unsigned long long segs = __ballot(len > 0 && clo[0] <= qhi0 && chi[0] >= qlo0 && clo[1] <= qhi1 && chi[1] >= qlo1 &&
                                             clo[2] <= qhi2 && chi[2] >= qlo2);
          while (segs) {
            const int sg = __ffsll((long long)segs) - 1;
            segs &= segs - 1;
            const uint32_t seg_len = lane_u(len, sg), seg_s = lane_u(s, sg), seg_flat = lane_u(incl - len, sg);
          n_tests += seg_len;
          for (uint32_t k0 = 0; k0 < seg_len; k0 += 64) {
            const uint32_t k = k0 + lane;
            float d2 = __builtin_inff();
            uint32_t idx = kInvalid;
            bool ok = false;
            if (k < seg_len) {
              const float4 rec = staged ? stage[seg_flat + k] : a.sorted[seg_s + k];
              idx = __float_as_uint(rec.w);
              const float dx = rec.x - qx, dy = rec.y - qy, dz = rec.z - qz;
              d2 = dx * dx + dy * dy + dz * dz;
              ok = d2 <= qr2;
              if (ok && a.state != nullptr && (a.state[idx] & a.skip_mask)) ok = false;
            }
            unsigned long long m = __ballot(ok);
            while (m) {
              const int src = __ffsll((long long)m) - 1;
              m &= m - 1;
              const float cd2 = lane_f(d2, src);
              const uint32_t cidx = lane_u(idx, src);
              // is the list full and the candidate not better than the current K-th?
              if (count == K) {
                const float kth_d2 = lane_f(my_d2, K - 1);
                const uint32_t kth_idx = lane_u(my_idx, K - 1);
                if (!before(cd2, cidx, kth_d2, kth_idx)) continue;
              }
              // insertion position = number of held entries ordered before the candidate
              const bool mine_before = ((int)lane < count) && before(my_d2, my_idx, cd2, cidx);
              const int pos = __popcll(__ballot(mine_before));
              const float up_d2 = __int_as_float(lane_below(__float_as_int(my_d2)));
              const uint32_t up_idx = (uint32_t)lane_below((int)my_idx);
              if ((int)lane > pos) { my_d2 = up_d2; my_idx = up_idx; }
              else if ((int)lane == pos) { my_d2 = cd2; my_idx = cidx; }
              if (count < K) ++count;
            }
          }
          }
          if ((int)lane < count) {
            a.out_idx[(size_t)qq * K + lane] = my_idx;
            a.out_d2[(size_t)qq * K + lane] = my_d2;
          }
          if (lane == ql) my_count = (uint32_t)count;
        }
      }
    }
    const bool mine = have && (lane % kTileWaves) == wave;   // the queries this wave answered
    if (mine) a.out_count[q] = (int32_t)my_count;
    if (a.stat) {
      uint32_t csum = mine ? my_count : 0u;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
      if (lane == 0) {
        if (wave == 0) atomicAdd(&a.stat[0], 1ull);
        atomicAdd(&a.stat[1], n_staged); atomicAdd(&a.stat[2], n_tests);
        atomicAdd(&a.stat[3], (unsigned long long)csum);
      }
    }
    __syncthreads();
   }
  }
}

// One LANE per query (the default, smx_nn_set_query_mode 2).  k_query_tiles spends a wavefront's 64 lanes on one query at
// a time: at the candidate counts of a surfel cloud (81 distance tests and 7 results per query at C5) most of its
// instructions are cross-lane bookkeeping -- readlane broadcasts of the query, ballots, the insertion network.  Here the
// tile's candidates are staged the same way, and then every lane walks ALL of them for its own query: the read of a
// candidate is one broadcast LDS load for the whole wavefront, the test is eight VALU instructions with no cross-lane
// traffic, a match is appended to the lane's private list in LDS (lane-major, odd stride: conflict-free).  The lists are
// sorted by (dist^2, index) lane-privately and written out a row at a time, coalesced.  Same candidates, same
// arithmetic, same total order: the rows are identical to k_query_tiles'.  What does not fit the simple shape --
// more than kLaneCap matches, a region of more than 64 bricks or more than kStageL points -- is marked and answered
// by k_query_tiles afterwards (same launch sequence, no host round trip).
#ifndef SMX_STAGE_L
#define SMX_STAGE_L 256
#endif
constexpr int kStageL = SMX_STAGE_L;     // staged points per tile, after the box filter (a surfel surface at cell = 1.5 x spacing: 60 - 100).
static_assert(kStageL <= 256, "a lane's list holds stage positions as bytes");
// LDS decides how many of these one-wavefront workgroups a CU holds, and the kernel lives on that: 768 entries (17 KB with the
// lists) 2.86 G queries/s at C5, 512: 3.32, 384: 3.87, 288 (9 KB: what the sorted keys needed to park in) 4.26 --
// profiles/r14_c5_stage.txt.  Round 4: the sorted rows leave the registers directly (no parking), the lists hold bytes,
// the staging loop's two small tables lie over the lists: 6.25 KB, 25 workgroups per CU.
constexpr int kLaneCap = 32;     // entries of a lane's private list (positions in the stage: 1 byte each)
constexpr int kLaneStride = kLaneCap + 4;   // bytes: 33 entries (the last one is the spare a full list keeps overwriting), entry 34 is the dump byte of the predicated append; an odd number of 32-bit words per lane

// The lane's matches, sorted in registers by Batcher's odd-even merge network over 64-bit keys (dist^2 bits << 32 |
// index: a non-negative float orders like its bit pattern, so one unsigned 64-bit compare is the (dist^2, index) order),
// then parked lane-major in LDS (`o_key`, stride N + 1 words of 8 bytes) for the coalesced row output.  Every lane of
// the wavefront sorts its own list at the same time: N = 16 costs 63 compare-exchanges for all 64 queries together,
// where ranking the matches of one query after the other costs about as much per QUERY.
template <int N>
__device__ __forceinline__ void sort_lane_matches(const float4* __restrict__ stage, const uint8_t* __restrict__ l_pos, uint32_t lbase,
                                                  uint32_t n, float px, float py, float pz, unsigned long long (&key)[N]) {
#pragma unroll
  for (int e = 0; e < N; ++e) {
    key[e] = ~0ull;
    if ((uint32_t)e < n) {
      const float4 rec = stage[l_pos[lbase + e]];
      const float dx = rec.x - px, dy = rec.y - py, dz = rec.z - pz;
      const float d2 = dx * dx + dy * dy + dz * dz;   // (the same subtraction and sum as in the test that accepted it)
      key[e] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)__float_as_uint(rec.w);
    }
    if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four records in flight at a time: all N of them cost 4 N registers)
  }
#pragma unroll
  for (int p = 1; p < N; p <<= 1)
#pragma unroll
    for (int k = p; k >= 1; k >>= 1)
#pragma unroll
      for (int j = k % p; j + k < N; j += 2 * k)
#pragma unroll
        for (int i = 0; i < k; ++i) {
          if (i + j + k < N && (i + j) / (2 * p) == (i + j + k) / (2 * p)) {
            const unsigned long long a0 = key[i + j], b0 = key[i + j + k];
            const bool sw = b0 < a0;
            key[i + j] = sw ? b0 : a0;
            key[i + j + k] = sw ? a0 : b0;
          }
        }
}
#ifndef SMX_NN_BRANCHFREE
#define SMX_NN_BRANCHFREE 1   // (the walk's append: 1 = predicated, 0 = a branch per candidate as in rounds 3-5; profiles/r6_ab_notes.md)
#endif
// kFilter: the caller passed a state row (skip kFree / kCompleted surfels, APP/octree.cc:330-335) -- a template parameter,
// because the test sits in the innermost loop: as a run-time (uniform) condition it cost every candidate an exec-mask
// save, a branch and a restore whether a state row was there or not.
#ifndef SMX_NN_STAGE_MLP
#define SMX_NN_STAGE_MLP 6   // (1: 5.67, 2: 6.05, 4: 6.25 G queries/s at C5 with the table-order tiles; 6 on top of the key-order tiles: 6.86 -> 7.05; profiles/r6_ab_notes.md)
#endif
#ifndef SMX_NN_SELF_SORTED
#define SMX_NN_SELF_SORTED 1
#endif
#ifndef SMX_NN_XCD_MAP
#define SMX_NN_XCD_MAP 0   // (measured: 6.82 - 6.92 against 6.89 - 6.95 G queries/s without -- the kernel is not bound by its bytes; profiles/r6_ab_notes.md)
#endif
#ifndef SMX_NN_PROBE_SPEC
#define SMX_NN_PROBE_SPEC 1
#endif
#ifndef SMX_NN_LANES_WAVES
#define SMX_NN_LANES_WAVES 1   // (__launch_bounds__'s second argument: 1 = let the compiler have the registers it likes)
#endif
template <bool kSelf, bool kFilter>
__global__ void __launch_bounds__(64, SMX_NN_LANES_WAVES)
k_query_lanes(QueryArgs a) {
  // one block: the stage, then the lanes' lists; the sorted keys later lie over both (neither is needed any more once
  // every lane holds its keys in registers).  LDS is what limits the wavefronts per CU here.
  __shared__ __align__(16) unsigned char lds_block[sizeof(float4) * kStageL + 64 * kLaneStride];
  float4* stage = reinterpret_cast<float4*>(lds_block);
  uint8_t* l_pos = lds_block + sizeof(float4) * kStageL;
  // (the staging loop's tables lie over the lists: the lists are written by the walk, which follows the staging)
  static_assert(2 * 64 * sizeof(uint32_t) <= 64 * kLaneStride, "seg_end / seg_src fit the list area");
  uint32_t* seg_end = reinterpret_cast<uint32_t*>(l_pos);
  uint32_t* seg_src = seg_end + 64;
  const uint32_t lane = threadIdx.x;
  const Grid& g = a.g;
  const uint32_t n_tiles = (kSelf && !a.tile_start) ? a.mask + 1u : *a.n_tiles;
  const int K = a.K;
#if SMX_NN_XCD_MAP
  // Workgroup b runs on XCD b % 8, and every XCD has an L2 of its own: with tile = blockIdx + k * gridDim the eight
  // neighbours of a tile in key order are staged by eight DIFFERENT L2s and each brick is fetched from memory once per XCD
  // that needs it.  Here every XCD walks a contiguous eighth of the tiles, its workgroups side by side: what a tile stages,
  // the tiles next to it (same row of bricks) and a row further (a few hundred tiles on) find in that XCD's L2.
  const uint32_t xcd = blockIdx.x & 7u, wi = blockIdx.x >> 3, per_xcd = (n_tiles + 7u) / 8u, wgs_per_xcd = gridDim.x >> 3;
  for (uint32_t ti = wi; ti < per_xcd; ti += wgs_per_xcd) {
    const uint32_t t = xcd * per_xcd + ti;
    if (t >= n_tiles) break;
#else
  for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
#endif
    uint32_t sub_begin, sub_end;
    if (kSelf && !a.tile_start) {
      const uint4 slot = *reinterpret_cast<const uint4*>(&a.table[t]);
      if ((slot.x | slot.y) == 0u) continue;   // (uniform) empty slot
      sub_begin = slot.z; sub_end = slot.w;
    } else {
      sub_begin = a.tile_start[t]; sub_end = a.tile_start[t + 1];
    }
    bool tile_marked = false;
    for (uint32_t j0 = sub_begin; j0 < sub_end; j0 += 64) {
      const uint32_t nt = min(sub_end - j0, 64u);
      const bool have = lane < nt;
      uint32_t q = 0;
      float px = 0, py = 0, pz = 0, r2 = -1.0f;
      if (have) {
        if (kSelf) {
          const float4 rec = a.sorted[j0 + lane];
          q = __float_as_uint(rec.w); px = rec.x; py = rec.y; pz = rec.z;
          r2 = a.self_r2 ? a.self_r2[q] * a.self_factor : a.self_factor;
        } else {
          const float4 qr = a.qrec[j0 + lane]; q = a.qorder[j0 + lane]; px = qr.x; py = qr.y; pz = qr.z; r2 = qr.w;
        }
      }
      // the lane's cell range (clamped to the grid), empty for r^2 < 0 / NaN -- as in k_query_tiles
      int lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1};
      bool ball = have && (r2 >= 0);
      if (ball) {
        const float qp[3] = {px, py, pz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float rad = cover_radius(r2, qp[k]);
          const float flo = floorf((qp[k] - rad - g.min[k]) / g.cell), fhi = floorf((qp[k] + rad - g.min[k]) / g.cell);
          lo[k] = flo >= 0.0f ? (flo < (float)g.dim[k] ? (int)flo : g.dim[k]) : 0;
          hi[k] = fhi >= 0.0f ? (fhi < (float)g.dim[k] ? (int)fhi : g.dim[k] - 1) : -1;
          if (!(lo[k] <= hi[k])) ball = false;
        }
      }
      if (!ball) r2 = -1.0f;
      int rlo[3], rhi[3];
      // ... and the box of the tile's balls in coordinates (same conservative radius): what the stage keeps
      float blo[3], bhi[3];
      {
        const float qp[3] = {px, py, pz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          int mn = ball ? lo[k] : 0x7FFFFFFF, mx = ball ? hi[k] : -1;
          const float rad = ball ? cover_radius(r2, qp[k]) : 0.0f;
          float fmn = ball ? qp[k] - rad : __builtin_inff(), fmx = ball ? qp[k] + rad : -__builtin_inff();
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) {
            mn = min(mn, __shfl_xor(mn, off)); mx = max(mx, __shfl_xor(mx, off));
            fmn = fminf(fmn, __shfl_xor(fmn, off)); fmx = fmaxf(fmx, __shfl_xor(fmx, off));
          }
          rlo[k] = mn; rhi[k] = mx; blo[k] = fmn; bhi[k] = fmx;
        }
      }
      bool redo = false;      // the lane's query goes to k_query_tiles
      uint32_t cnt = 0;       // matches of the lane's query (the first kLaneCap of them are in its list)
      unsigned long long n_tests = 0, n_staged = 0;
      if (rhi[0] >= rlo[0]) {   // (uniform) at least one lane has a ball
        const int kb[3] = {rlo[0] >> kBrickShift, rlo[1] >> kBrickShift, rlo[2] >> kBrickShift};
        const int bn[3] = {(rhi[0] >> kBrickShift) - kb[0] + 1, (rhi[1] >> kBrickShift) - kb[1] + 1, (rhi[2] >> kBrickShift) - kb[2] + 1};
        const unsigned long long nbricks = (unsigned long long)bn[0] * bn[1] * bn[2];
        if (nbricks > 64ull) {
          redo = ball;
        } else {
          uint32_t s = 0, e = 0;
          if ((unsigned long long)lane < nbricks) {
            const int bx = kb[0] + (int)(lane % (uint32_t)bn[0]);
            const int by = kb[1] + (int)((lane / (uint32_t)bn[0]) % (uint32_t)bn[1]);
            const int bz = kb[2] + (int)(lane / ((uint32_t)bn[0] * (uint32_t)bn[1]));
#if SMX_NN_PROBE_SPEC > 1
            if (!find_brick_spec<SMX_NN_PROBE_SPEC>(a.table, a.mask, brick_index(g, bx, by, bz), s, e)) { s = 0; e = 0; }
#else
            if (!find_brick(a.table, a.mask, brick_index(g, bx, by, bz), s, e)) { s = 0; e = 0; }
#endif
          }
          const uint32_t len = e - s;
          uint32_t incl = len;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) { const uint32_t tv = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += tv; }
          const uint32_t total = lane_u(incl, 63);
          if (total > 0) {
            __syncthreads();   // (the previous tile's readers are done)
            seg_end[lane] = incl;               // the range of lane L fills flat positions [seg_end[L] - len, seg_end[L])
            seg_src[lane] = s - (incl - len);   // source index = seg_src[L] + flat position (mod 2^32)
            __syncthreads();
            // The stage keeps the records inside the box of the tile's balls, compacted (ballot + popcount): the bricks
            // around a tile hold 316 points on a surfel surface at cell = 1.5 x spacing, the box 80-100, and every
            // candidate in the stage costs the whole wavefront a dozen instructions in the walk below, whichever lanes
            // still care -- the walk was three quarters of this kernel's instructions.  (A record outside the box fails
            // d2 <= r2 for every query of the tile: cover_radius bounds |dx| of an accepted pair.)
            uint32_t cur = 0, kept = 0;
#if SMX_NN_STAGE_MLP > 1
            // (round 6: SMX_NN_STAGE_MLP records per lane requested before the first is looked at -- the loop was a chain of
            // dependent round trips, one per 64 records, five per tile at C5, in a kernel that is bound by exactly those)
            for (uint32_t k0 = 0; k0 < total; k0 += 64 * SMX_NN_STAGE_MLP) {
              float4 v[SMX_NN_STAGE_MLP];
#pragma unroll
              for (int u = 0; u < SMX_NN_STAGE_MLP; ++u) {
                const uint32_t k = min(k0 + 64u * u + lane, total - 1u);
                while (k >= seg_end[cur]) ++cur;
                v[u] = a.sorted[seg_src[cur] + k];
              }
#pragma unroll
              for (int u = 0; u < SMX_NN_STAGE_MLP; ++u) {
                const bool in = k0 + 64u * u + lane < total && v[u].x >= blo[0] && v[u].x <= bhi[0] && v[u].y >= blo[1] && v[u].y <= bhi[1] &&
                                v[u].z >= blo[2] && v[u].z <= bhi[2];
                const unsigned long long m = __ballot(in);
                const uint32_t pos = kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (in && pos < (uint32_t)kStageL) stage[pos] = v[u];
                kept += (uint32_t)__popcll(m);
              }
            }
#else
            for (uint32_t k0 = 0; k0 < total; k0 += 64) {
              const uint32_t k = min(k0 + lane, total - 1u);
              while (k >= seg_end[cur]) ++cur;
              const float4 v = a.sorted[seg_src[cur] + k];
              const bool in = k0 + lane < total && v.x >= blo[0] && v.x <= bhi[0] && v.y >= blo[1] && v.y <= bhi[1] &&
                              v.z >= blo[2] && v.z <= bhi[2];
              const unsigned long long m = __ballot(in);
              const uint32_t pos = kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
              if (in && pos < (uint32_t)kStageL) stage[pos] = v;
              kept += (uint32_t)__popcll(m);
            }
#endif
            __syncthreads();
            if (kept > (uint32_t)kStageL) { redo = ball; kept = 0; }   // (more than the stage holds: the other kernel)
            n_staged = total;   // (statistics: records read from the index; n_tests below counts the tests on the kept ones)
            n_tests += (unsigned long long)__popcll(__ballot(ball)) * kept;
            // Four candidates per step, all four LDS reads (one address for the whole wavefront: broadcasts) issued
            // before the first test; a match is appended without a branch (a list that is full keeps overwriting its
            // spare last entry; cnt keeps counting and marks the query for the other kernel afterwards).
            const uint32_t lbase = lane * kLaneStride;
#if SMX_NN_BRANCHFREE
            // Round 6: no branch in the walk.  A lane without a ball carries r^2 = -1 (no d^2 passes), so the test is the
            // compare alone; the append is predicated -- a match goes to entry min(cnt, kLaneCap) of the lane's list, a miss
            // to the list's dump byte (entry kLaneCap + 1; the stride leaves room) -- one byte store per candidate whatever
            // the outcome, no exec-mask juggling (rounds 3-5: two branches and a dozen scalar instructions per candidate,
            // as many as the arithmetic).  Whole groups of four first, the tail one at a time.
            auto test_one = [&](const float4& rc, uint32_t k) {
              const float dx = rc.x - px, dy = rc.y - py, dz = rc.z - pz;
              const float d2 = dx * dx + dy * dy + dz * dz;
              bool ok = d2 <= r2;
              if (kFilter) { if (ok && (a.state[__float_as_uint(rc.w)] & a.skip_mask)) ok = false; }
              const uint32_t slot = ok ? min(cnt, (uint32_t)kLaneCap) : (uint32_t)kLaneCap + 1u;
              l_pos[lbase + slot] = (uint8_t)k;
              cnt += ok ? 1u : 0u;
            };
            uint32_t k0 = 0;
            for (; k0 + 4 <= kept; k0 += 4) {
              float4 rec[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) rec[u] = stage[k0 + u];
#pragma unroll
              for (int u = 0; u < 4; ++u) test_one(rec[u], k0 + u);
            }
            for (; k0 < kept; ++k0) test_one(stage[k0], k0);
#else
            for (uint32_t k0 = 0; k0 < kept; k0 += 4) {
              float4 rec[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) rec[u] = stage[min(k0 + u, kept - 1u)];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float dx = rec[u].x - px, dy = rec[u].y - py, dz = rec[u].z - pz;
                const float d2 = dx * dx + dy * dy + dz * dz;
                bool ok = ball && (k0 + u < kept) && d2 <= r2;
                if (kFilter) {
                  if (ok && (a.state[__float_as_uint(rec[u].w)] & a.skip_mask)) ok = false;
                }
                if (ok) {
                  l_pos[lbase + min(cnt, (uint32_t)kLaneCap)] = (uint8_t)(k0 + u);
                  ++cnt;
                }
              }
            }
#endif
            if (cnt > (uint32_t)kLaneCap) redo = true;
          }
        }
      }
      const uint32_t n = redo ? 0u : cnt;
      const uint32_t n_out = min(n, (uint32_t)K);
      uint32_t n_max = n;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, off));
      __syncthreads();
      if (n_max <= 16u) {   // (uniform) the usual case: every lane sorts its own matches in registers
        // ... and writes its own row from them: 16-byte stores while whole groups of four entries lie inside the row (the
        // rows are 16-byte aligned when K is a multiple of 4), single entries for the rest.  (Rounds 2-3 parked the keys
        // lane-major in LDS and wrote the rows one query at a time, entry e from lane e: 8.7 KB of LDS -- what capped the
        // workgroups per CU -- and a loop of 34 broadcasts + stores per tile.)
        auto rows_out = [&](auto& key, auto n_const) {
          constexpr int N = decltype(n_const)::value;
          if (have && n_out) {
            uint32_t* oi = a.out_idx + (size_t)q * K;
            float* od = a.out_d2 + (size_t)q * K;
            const bool wide = (K & 3) == 0;
#pragma unroll
            for (int e = 0; e < N; e += 4) {
              if ((uint32_t)e >= n_out) break;
              if (wide && (uint32_t)(e + 4) <= n_out) {
                *reinterpret_cast<uint4*>(oi + e) = make_uint4((uint32_t)key[e], (uint32_t)key[e + 1], (uint32_t)key[e + 2], (uint32_t)key[e + 3]);
                *reinterpret_cast<float4*>(od + e) = make_float4(__uint_as_float((uint32_t)(key[e] >> 32)), __uint_as_float((uint32_t)(key[e + 1] >> 32)),
                                                                 __uint_as_float((uint32_t)(key[e + 2] >> 32)), __uint_as_float((uint32_t)(key[e + 3] >> 32)));
              } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                  if ((uint32_t)(e + u) < n_out) { oi[e + u] = (uint32_t)key[e + u]; od[e + u] = __uint_as_float((uint32_t)(key[e + u] >> 32)); }
              }
            }
          }
        };
        if (n_max <= 8u) {
          unsigned long long key[8];
          sort_lane_matches<8>(stage, l_pos, lane * kLaneStride, n, px, py, pz, key);
          rows_out(key, std::integral_constant<int, 8>());
        } else {
          unsigned long long key[16];
          sort_lane_matches<16>(stage, l_pos, lane * kLaneStride, n, px, py, pz, key);
          rows_out(key, std::integral_constant<int, 16>());
        }
      } else {
      // Rows out, one query at a time: lane e takes the query's e-th match (recomputed from the stage: the same
      // subtraction and sum as in the test above), finds its rank in the (dist^2, index) order by comparing with all
      // matches (broadcasts from registers, no sorting network, no data movement) and stores it at that position of the
      // output row -- the row's entries leave the wavefront together.
      for (uint32_t ql = 0; ql < nt; ++ql) {
        const uint32_t nm = lane_u(n, (int)ql);
        if (nm == 0) continue;   // (uniform)
        const uint32_t qq = lane_u(q, (int)ql);
        const float qx = lane_f(px, (int)ql), qy = lane_f(py, (int)ql), qz = lane_f(pz, (int)ql);
        float my_d2 = __builtin_inff();
        uint32_t my_idx = kInvalid;
        if (lane < nm) {
          const float4 rec = stage[l_pos[ql * kLaneStride + lane]];
          const float dx = rec.x - qx, dy = rec.y - qy, dz = rec.z - qz;
          my_d2 = dx * dx + dy * dy + dz * dz;
          my_idx = __float_as_uint(rec.w);
        }
        uint32_t rank = 0;
        for (uint32_t e = 0; e < nm; ++e) {
          const float ed2 = lane_f(my_d2, (int)e);
          const uint32_t eidx = lane_u(my_idx, (int)e);
          rank += before(ed2, eidx, my_d2, my_idx) ? 1u : 0u;
        }
        if (lane < nm && rank < (uint32_t)K) {
          a.out_idx[(size_t)qq * K + rank] = my_idx;
          a.out_d2[(size_t)qq * K + rank] = my_d2;
        }
      }
      }
      if (have) a.out_count[q] = redo ? -1 : (int32_t)n_out;
      if (__ballot(redo)) { tile_marked = true; if (lane == 0) *a.redo_flag = 1u; }
      if (a.stat) {
        uint32_t csum = have ? n_out : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
        if (lane == 0) {
          atomicAdd(&a.stat[0], 1ull);
          atomicAdd(&a.stat[1], n_staged); atomicAdd(&a.stat[2], n_tests);
          atomicAdd(&a.stat[3], (unsigned long long)csum);
        }
      }
      __syncthreads();
    }
    if (lane == 0) a.tile_redo[t] = tile_marked ? 1 : 0;
  }
}

// The same search without the LDS stage (A/B partner of k_query_tiles, smx_nn_set_query_mode): one wavefront per query,
// queries still in brick order, so the bricks a query needs were read by its neighbours a moment ago and come from
// L1 / L2.  No shared memory, no barriers, 8 wavefronts per SIMD.
__global__ void __launch_bounds__(kBlock)
k_query_stream(QueryArgs a) {
  const uint32_t lane = threadIdx.x & 63;
  const Grid& g = a.g;
  const int K = a.K;
  const uint32_t waves = gridDim.x * (kBlock / 64);
  for (uint32_t j = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); j < a.nq; j += waves) {
    const float4 qr = a.qrec[j];
    const uint32_t qq = a.qorder[j];
    const float qx = qr.x, qy = qr.y, qz = qr.z, qr2 = qr.w;
    int lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1};
    bool ball = qr2 >= 0;
    if (ball) {
      const float qp[3] = {qx, qy, qz};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float rad = cover_radius(qr2, qp[k]);
        const float flo = floorf((qp[k] - rad - g.min[k]) / g.cell), fhi = floorf((qp[k] + rad - g.min[k]) / g.cell);
        lo[k] = flo >= 0.0f ? (flo < (float)g.dim[k] ? (int)flo : g.dim[k]) : 0;
        hi[k] = fhi >= 0.0f ? (fhi < (float)g.dim[k] ? (int)fhi : g.dim[k] - 1) : -1;
        if (!(lo[k] <= hi[k])) ball = false;
      }
    }
    int count = 0;
    float my_d2 = __builtin_inff();
    uint32_t my_idx = kInvalid;
    unsigned long long n_tests = 0;
    if (ball) {   // (uniform: one query per wave)
      const int blo[3] = {lo[0] >> kBrickShift, lo[1] >> kBrickShift, lo[2] >> kBrickShift};
      const int bn[3] = {(hi[0] >> kBrickShift) - blo[0] + 1, (hi[1] >> kBrickShift) - blo[1] + 1, (hi[2] >> kBrickShift) - blo[2] + 1};
      unsigned long long nbricks = (unsigned long long)bn[0] * bn[1] * bn[2];
      const bool everything = nbricks > kManyBricks;
      if (everything) nbricks = 1;
      for (unsigned long long b0 = 0; b0 < nbricks; b0 += 64) {
        const unsigned long long b = b0 + lane;
        uint32_t s = 0, e = 0;
        if (everything) {
          if (lane == 0) e = a.n_valid;
        } else if (b < nbricks) {
          const int bx = blo[0] + (int)(b % (unsigned long long)bn[0]);
          const int by = blo[1] + (int)((b / (unsigned long long)bn[0]) % (unsigned long long)bn[1]);
          const int bz = blo[2] + (int)(b / ((unsigned long long)bn[0] * (unsigned long long)bn[1]));
          if (!find_brick(a.table, a.mask, brick_index(g, bx, by, bz), s, e)) { s = 0; e = 0; }
        }
        unsigned long long segs = __ballot(e > s);
        while (segs) {
          const int sg = __ffsll((long long)segs) - 1;
          segs &= segs - 1;
          const uint32_t seg_s = lane_u(s, sg), seg_len = lane_u(e, sg) - seg_s;
          n_tests += seg_len;
          for (uint32_t k0 = 0; k0 < seg_len; k0 += 64) {
            const uint32_t k = k0 + lane;
            float d2 = __builtin_inff();
            uint32_t idx = kInvalid;
            bool ok = false;
            if (k < seg_len) {
              const float4 rec = a.sorted[seg_s + k];
              idx = __float_as_uint(rec.w);
              const float dx = rec.x - qx, dy = rec.y - qy, dz = rec.z - qz;
              d2 = dx * dx + dy * dy + dz * dz;
              ok = d2 <= qr2;
              if (ok && a.state != nullptr && (a.state[idx] & a.skip_mask)) ok = false;
            }
            unsigned long long m = __ballot(ok);
            while (m) {
              const int src = __ffsll((long long)m) - 1;
              m &= m - 1;
              const float cd2 = lane_f(d2, src);
              const uint32_t cidx = lane_u(idx, src);
              if (count == K) {
                const float kth_d2 = lane_f(my_d2, K - 1);
                const uint32_t kth_idx = lane_u(my_idx, K - 1);
                if (!before(cd2, cidx, kth_d2, kth_idx)) continue;
              }
              const bool mine_before = ((int)lane < count) && before(my_d2, my_idx, cd2, cidx);
              const int pos = __popcll(__ballot(mine_before));
              const float up_d2 = __int_as_float(lane_below(__float_as_int(my_d2)));
              const uint32_t up_idx = (uint32_t)lane_below((int)my_idx);
              if ((int)lane > pos) { my_d2 = up_d2; my_idx = up_idx; }
              else if ((int)lane == pos) { my_d2 = cd2; my_idx = cidx; }
              if (count < K) ++count;
            }
          }
        }
      }
    }
    if ((int)lane < count) {
      a.out_idx[(size_t)qq * K + lane] = my_idx;
      a.out_d2[(size_t)qq * K + lane] = my_d2;
    }
    if (lane == 0) {
      a.out_count[qq] = count;
      if (a.stat) { atomicAdd(&a.stat[2], n_tests); atomicAdd(&a.stat[3], (unsigned long long)count); }
    }
  }
}

}  // namespace

struct smx_nn_s {
  int device;
  uint32_t n;          // points given to the last build
  uint32_t n_valid;    // indexed points (finite coordinates)
  uint32_t n_bricks;
  Grid grid;
  // index (owned, grown on demand, reused by the next build)
  size_t cap_points;
  unsigned long long* keys[2];
  uint32_t* vals[2];
  float* rows;          // upload target for host rows [3][cap]
  float4* sorted;
  uint32_t* hist;       // [256][tiles] + scan workspaces behind it
  size_t hist_elems;
  BrickSlot* table;
  size_t table_slots;   // power of two
  uint32_t* bbox;       // 6 order keys
  uint32_t* partial;    // count partials
  BuildCounts* counts;
  // query workspace
  size_t cap_queries;
  unsigned long long* qkeys[2];
  uint32_t* qvals[2];
  uint32_t *qflags, *qtile_start;
  uint8_t *tile_redo_q, *tile_redo_self;   // k_query_lanes' per-tile marks: one per query tile / per self tile (or table slot)
  uint32_t* self_tile_start;    // the self queries' tiles in KEY order: tile t = points [start[t], start[t + 1]) of nn->sorted,
  uint32_t* self_n_tiles;       // <= 64 points of one brick each (k_tile_flags over the point keys); their number (device word)
  size_t cap_self_tiles, cap_tile_redo_self;
  float* qrows;         // upload target for host queries [4][cap]
  float4* qrec;         // queries in brick order
  uint8_t* dstate; size_t cap_state;
  uint32_t* didx; float* dd2; int32_t* dcnt; size_t cap_out;   // staging for host outputs [nq * k]
  unsigned long long* stat;   // 4 counters (device), filled while stats_enabled
  int stats_enabled;
  int query_mode;       // 2 = one lane per query (k_query_lanes; default), 0 = one wavefront per query over LDS-staged brick tiles
                        // (k_query_tiles), 1 = one wavefront per query from L1 / L2 (k_query_stream)
  int grid_blocks;      // persistent grid of the tile kernel
};

namespace {

void nn_free(smx_nn nn) {
  void* ptrs[] = {nn->keys[0], nn->keys[1], nn->vals[0], nn->vals[1], nn->rows, nn->sorted, nn->hist, nn->table, nn->bbox,
                  nn->partial, nn->counts, nn->qkeys[0], nn->qkeys[1], nn->qvals[0], nn->qvals[1], nn->qflags, nn->tile_redo_q, nn->tile_redo_self, nn->self_tile_start, nn->self_n_tiles,
                  nn->qtile_start, nn->qrows, nn->qrec, nn->dstate, nn->didx, nn->dd2, nn->dcnt, nn->stat};
  for (void* p : ptrs) if (p) (void)hipFree(p);
}

template <typename T>
int grow(T** p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)));
  return SMX_OK;
}

size_t sort_hist_elems(size_t n) { return (size_t)kRadix * ((n + kSortTile - 1) / kSortTile); }

// Workspace is only ever grown, and only when it is too small: the device is synchronised first (earlier calls may
// still be using the old buffers).  Steady-state builds and queries allocate nothing.
int ensure_points(smx_nn nn, size_t n) {
  if (n <= nn->cap_points) return SMX_OK;
  SMX_HIP(hipDeviceSynchronize());
  const size_t cap = n + n / 8 + 1024;
  int rc = SMX_OK;
  for (int k = 0; k < 2 && rc == SMX_OK; ++k) { rc = grow(&nn->keys[k], cap); if (rc == SMX_OK) rc = grow(&nn->vals[k], cap); }
  if (rc == SMX_OK) rc = grow(&nn->rows, 3 * cap);
  if (rc == SMX_OK) rc = grow(&nn->sorted, cap);
  nn->cap_points = rc == SMX_OK ? cap : 0;
  return rc;
}
// hist layout: [sort histograms for max_n][their scan levels][scan levels of a flag array of max_n]
size_t hist_need(size_t max_n) {
  return sort_hist_elems(max_n) + scan_workspace_elems(sort_hist_elems(max_n)) + scan_workspace_elems(max_n) + 16;
}
int ensure_hist(smx_nn nn, size_t max_n) {
  const size_t need = hist_need(max_n);
  if (need <= nn->hist_elems) return SMX_OK;
  SMX_HIP(hipDeviceSynchronize());
  const size_t cap = need + need / 8;
  const int rc = grow(&nn->hist, cap);
  nn->hist_elems = rc == SMX_OK ? cap : 0;
  return rc;
}
int ensure_queries(smx_nn nn, size_t nq) {
  if (nq <= nn->cap_queries) return SMX_OK;
  SMX_HIP(hipDeviceSynchronize());
  const size_t cap = nq + nq / 8 + 1024;
  int rc = SMX_OK;
  for (int k = 0; k < 2 && rc == SMX_OK; ++k) { rc = grow(&nn->qkeys[k], cap); if (rc == SMX_OK) rc = grow(&nn->qvals[k], cap); }
  if (rc == SMX_OK) rc = grow(&nn->qflags, cap + 1);
  if (rc == SMX_OK) rc = grow(&nn->tile_redo_q, cap + 1);
  if (rc == SMX_OK) rc = grow(&nn->qtile_start, cap + 2);
  if (rc == SMX_OK) rc = grow(&nn->qrows, 4 * cap);
  if (rc == SMX_OK) rc = grow(&nn->qrec, cap);
  nn->cap_queries = rc == SMX_OK ? cap : 0;
  return rc;
}

// Sorts (keys[0], vals[0]) by the low `bits` bits; returns the index (0 / 1) of the buffers that hold the result.
int radix_sort(unsigned long long* const keys[2], uint32_t* const vals[2], uint32_t n, int bits, uint32_t* hist,
               hipStream_t st) {
  const uint32_t tiles = (uint32_t)((n + kSortTile - 1) / kSortTile);
  const size_t hn = (size_t)kRadix * tiles;
  uint32_t* scan_ws = hist + hn;
  int cur = 0;
  for (int shift = 0; shift < bits; shift += 8) {
    hipLaunchKernelGGL(k_rs_hist, dim3(tiles), dim3(kBlock), 0, st, keys[cur], n, shift, hist, tiles);
    exclusive_scan_inplace(hist, hn, scan_ws, st, nullptr);
    hipLaunchKernelGGL(k_rs_scatter, dim3(tiles), dim3(kBlock), 0, st, keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n,
                       shift, hist, tiles);
    cur ^= 1;
  }
  return cur;
}

int bit_length(unsigned long long v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

}  // namespace

extern "C" {

int smx_nn_create(int32_t device_id, smx_nn* out) {
  SMX_CHECK_ARG(out != nullptr);
  int device = 0;
  { const int rcd = resolve_device(device_id, &device); if (rcd != SMX_OK) return rcd; }
  SMX_ON_DEVICE(device);
  smx_nn_s* nn = new smx_nn_s();
  memset(nn, 0, sizeof(*nn));
  nn->device = device;
  hipDeviceProp_t prop;
  SMX_HIP(hipGetDeviceProperties(&prop, device));
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  nn->query_mode = 2;           // lane per query, k_query_tiles for what it marks
  nn->grid_blocks = cus * 12;   // persistent grid of the tile kernel: 6 workgroups (24 wavefronts) fit a CU, two rounds
  int rc = grow(&nn->bbox, 6);
  if (rc == SMX_OK) rc = grow(&nn->partial, 6 * 2048);
  if (rc == SMX_OK) rc = grow(&nn->counts, 1);
  if (rc == SMX_OK) rc = grow(&nn->stat, 5);   // [4]: k_query_lanes' redo flag
  if (rc != SMX_OK) { nn_free(nn); delete nn; return rc; }
  SMX_HIP(hipMemset(nn->stat, 0, 5 * sizeof(unsigned long long)));
  *out = nn;
  return SMX_OK;
}

int smx_nn_destroy(smx_nn nn) {
  if (!nn) return SMX_OK;
  SMX_ON_DEVICE(nn->device);
  (void)hipDeviceSynchronize();
  nn_free(nn);
  delete nn;
  return SMX_OK;
}

int smx_nn_build(smx_nn nn, smx_stream s, const float* x, const float* y, const float* z, uint32_t n,
                 float cell_size, int32_t rows_on_device) {
  SMX_CHECK_ARG(nn != nullptr && cell_size > 0 && (n == 0 || (x && y && z)));
  SMX_ON_DEVICE(nn->device);
  hipStream_t st = (hipStream_t)s;
  nn->n = n; nn->n_valid = 0; nn->n_bricks = 0;
  if (n == 0) return SMX_OK;
  int rc = ensure_points(nn, n);
  const size_t hist_max_n = std::max<size_t>(n, nn->cap_queries);
  if (rc == SMX_OK) rc = ensure_hist(nn, hist_max_n);
  if (rc != SMX_OK) return rc;
  const float *dx = x, *dy = y, *dz = z;
  if (!rows_on_device) {
    const size_t cap = nn->cap_points;
    SMX_HIP(hipMemcpyAsync(nn->rows, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    SMX_HIP(hipMemcpyAsync(nn->rows + cap, y, (size_t)n * 4, hipMemcpyHostToDevice, st));
    SMX_HIP(hipMemcpyAsync(nn->rows + 2 * cap, z, (size_t)n * 4, hipMemcpyHostToDevice, st));
    dx = nn->rows; dy = nn->rows + cap; dz = nn->rows + 2 * cap;
  }
  const int grid = 2048;
  hipLaunchKernelGGL(k_bbox, dim3(grid), dim3(kBlock), 0, st, dx, dy, dz, n, nn->partial);
  hipLaunchKernelGGL(k_bbox_finish, dim3(1), dim3(kBlock), 0, st, nn->partial, grid, nn->bbox);
  uint32_t bb[6];
  SMX_HIP(hipMemcpyAsync(bb, nn->bbox, sizeof(bb), hipMemcpyDeviceToHost, st));
  SMX_HIP(hipStreamSynchronize(st));   // read-back 1 of 2: the grid dimensions decide the number of sort passes
  if (bb[0] > bb[3]) return SMX_OK;    // no indexable point
  float mn[3], mx[3];
  for (int a = 0; a < 3; ++a) { mn[a] = order_unkey(bb[a]); mx[a] = order_unkey(bb[3 + a]); }
  // The grid is sparse: only the key width limits it (2^21 cells per axis, 56 brick bits).  A cell size that would
  // exceed that is doubled until it fits -- at 1 mm cells that happens beyond 2 km of extent.
  Grid& g = nn->grid;
  float cell = cell_size;
  for (;;) {
    bool ok = true;
    unsigned long long bricks = 1;
    for (int a = 0; a < 3; ++a) {
      const double d = floor(((double)mx[a] - (double)mn[a]) / cell) + 1.0;
      if (!(d <= 2097152.0)) { ok = false; break; }
      g.dim[a] = (int)d < 1 ? 1 : (int)d;
      g.bdim[a] = (g.dim[a] + kBrickCells - 1) >> kBrickShift;
      bricks *= (unsigned long long)g.bdim[a];
    }
#if SMX_NN_MORTON
    // (Z-order brick numbers: three interleaved fields as wide as the longest axis needs)
    int axis_bits = 1;
    for (int a = 0; a < 3; ++a) axis_bits = std::max(axis_bits, bit_length((unsigned long long)(g.bdim[a] - 1)));
    if (ok) bricks = 1ull << (3 * axis_bits);
    if (ok && 3 * axis_bits > 56) ok = false;
#endif
    if (ok && bit_length(bricks) <= (SMX_NN_MORTON ? 57 : 56)) {
      g.sentinel = bricks << kLocalBits;
      g.key_bits = bit_length(g.sentinel);
      g.brick_bits = bit_length(bricks - 1);
      break;
    }
    cell *= 2.0f;
  }
  g.cell = cell;
  for (int a = 0; a < 3; ++a) g.min[a] = mn[a];

  hipLaunchKernelGGL(k_point_keys, dim3(grid), dim3(kBlock), 0, st, dx, dy, dz, n, g, nn->keys[0], nn->vals[0]);
  const int cur = radix_sort(nn->keys, nn->vals, n, g.key_bits, nn->hist, st);
  if (cur != 0) { std::swap(nn->keys[0], nn->keys[1]); std::swap(nn->vals[0], nn->vals[1]); }   // result in [0]
  hipLaunchKernelGGL(k_gather_records, dim3(grid), dim3(kBlock), 0, st, nn->vals[0], dx, dy, dz, n, nn->sorted);
  hipLaunchKernelGGL(k_brick_count, dim3(grid), dim3(kBlock), 0, st, nn->keys[0], n, g.sentinel, nn->partial);
  hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(kBlock), 0, st, nn->partial, grid, nn->counts);
  BuildCounts h;
  SMX_HIP(hipMemcpyAsync(&h, nn->counts, sizeof(h), hipMemcpyDeviceToHost, st));
  SMX_HIP(hipStreamSynchronize(st));   // read-back 2 of 2: the hash table is sized to the occupied bricks
  size_t slots = 1024;
  while (slots < (size_t)h.n_bricks * 2) slots <<= 1;
  if (slots > nn->table_slots) {
    rc = grow(&nn->table, slots);
    if (rc == SMX_OK && slots > nn->cap_tile_redo_self) { rc = grow(&nn->tile_redo_self, slots); nn->cap_tile_redo_self = rc == SMX_OK ? slots : 0; }
    nn->table_slots = rc == SMX_OK ? slots : 0;
    if (rc != SMX_OK) return rc;
  }
  slots = nn->table_slots;   // (a larger table left from an earlier build is simply sparser)
  SMX_HIP(hipMemsetAsync(nn->table, 0, slots * sizeof(BrickSlot), st));
  hipLaunchKernelGGL(k_brick_insert, dim3(grid), dim3(kBlock), 0, st, nn->keys[0], h.n_valid, nn->table, (uint32_t)(slots - 1));
  hipLaunchKernelGGL(k_brick_ends, dim3(grid), dim3(kBlock), 0, st, nn->keys[0], h.n_valid, nn->table, (uint32_t)(slots - 1));
#if SMX_NN_SELF_SORTED
  // The self queries' tiles in KEY order (round 6).  Rounds 2-5 walked the brick TABLE: tile = hash slot -- half of the
  // slots empty (a round trip each to find out) and neighbouring bricks a random distance apart in the walk, so that each
  // tile's 27 bricks came from memory anew (PMC traffic 1.78 x the algorithmic bytes).  In key order (brick index, row-major)
  // the tiles in flight at any moment are a few adjacent rows of bricks, and what one stages its neighbours find in the L2.
  if (h.n_valid > 0) {
    const size_t max_tiles = (size_t)h.n_bricks + (size_t)h.n_valid / kTile + 2;
    if (max_tiles + 1 > nn->cap_self_tiles) {
      SMX_HIP(hipDeviceSynchronize());
      rc = grow(&nn->self_tile_start, max_tiles + 1 + max_tiles / 8);
      if (rc == SMX_OK && !nn->self_n_tiles) rc = grow(&nn->self_n_tiles, 4);
      nn->cap_self_tiles = rc == SMX_OK ? max_tiles + 1 + max_tiles / 8 : 0;
      if (rc != SMX_OK) return rc;
    }
    if (max_tiles > nn->cap_tile_redo_self) {
      SMX_HIP(hipDeviceSynchronize());
      rc = grow(&nn->tile_redo_self, std::max(max_tiles + max_tiles / 8, slots));
      nn->cap_tile_redo_self = rc == SMX_OK ? std::max(max_tiles + max_tiles / 8, slots) : 0;
      if (rc != SMX_OK) return rc;
    }
    uint32_t* flags = nn->vals[1];   // (scratch of the sort: the order lives in vals[0])
    hipLaunchKernelGGL(k_tile_flags, dim3((h.n_valid + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nn->keys[0], h.n_valid, kLocalBits, flags);
    uint32_t* total = nullptr;
    uint32_t* flag_ws = nn->hist + sort_hist_elems(hist_max_n) + scan_workspace_elems(sort_hist_elems(hist_max_n));
    exclusive_scan_inplace(flags, h.n_valid, flag_ws, st, &total);
    hipLaunchKernelGGL(k_tile_starts, dim3(grid), dim3(kBlock), 0, st, flags, h.n_valid, total, nn->self_tile_start);
    SMX_HIP(hipMemcpyAsync(nn->self_n_tiles, total, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  }
#endif
  SMX_LAUNCH_CHECK();
  nn->n_valid = h.n_valid; nn->n_bricks = h.n_bricks;
  return SMX_OK;
}

int smx_nn_query_batch(smx_nn nn, smx_stream s, uint32_t nq, const float* qx, const float* qy, const float* qz,
                       const float* r2, int32_t k, const uint8_t* state, uint8_t skip_mask, int32_t queries_on_device,
                       uint32_t* out_idx, float* out_d2, int32_t* out_count, int32_t outputs_on_device) {
  SMX_CHECK_ARG(nn != nullptr && k >= 1 && k <= 64);
  SMX_ON_DEVICE(nn->device);
  SMX_CHECK_ARG(nq == 0 || (qx && qy && qz && r2 && out_idx && out_d2 && out_count));
  if (nq == 0) return SMX_OK;
  hipStream_t st = (hipStream_t)s;
  if (nn->n_valid == 0) {
    if (outputs_on_device) SMX_HIP(hipMemsetAsync(out_count, 0, (size_t)nq * 4, st));
    else memset(out_count, 0, (size_t)nq * 4);
    return SMX_OK;
  }
  int rc = ensure_queries(nn, nq);
  const size_t max_n = std::max<size_t>(nn->cap_queries, nn->n);
  if (rc == SMX_OK) rc = ensure_hist(nn, max_n);
  if (rc != SMX_OK) return rc;
  QueryArgs a;
  memset(&a, 0, sizeof(a));
  const float *dqx, *dqy, *dqz, *dqr2;
  a.nq = nq; a.K = k; a.skip_mask = skip_mask; a.g = nn->grid; a.table = nn->table; a.mask = (uint32_t)(nn->table_slots - 1);
  a.sorted = nn->sorted; a.n_valid = nn->n_valid;
  if (queries_on_device) {
    dqx = qx; dqy = qy; dqz = qz; dqr2 = r2; a.state = state;
  } else {
    const size_t cap = nn->cap_queries;
    const float* src[4] = {qx, qy, qz, r2};
    for (int c = 0; c < 4; ++c) SMX_HIP(hipMemcpyAsync(nn->qrows + c * cap, src[c], (size_t)nq * 4, hipMemcpyHostToDevice, st));
    dqx = nn->qrows; dqy = nn->qrows + cap; dqz = nn->qrows + 2 * cap; dqr2 = nn->qrows + 3 * cap;
    if (state) {
      if (nn->n > nn->cap_state) {
        SMX_HIP(hipDeviceSynchronize());
        const size_t cap_s = (size_t)nn->n + nn->n / 8;
        rc = grow(&nn->dstate, cap_s);
        nn->cap_state = rc == SMX_OK ? cap_s : 0;
        if (rc != SMX_OK) return rc;
      }
      SMX_HIP(hipMemcpyAsync(nn->dstate, state, nn->n, hipMemcpyHostToDevice, st));
      a.state = nn->dstate;
    }
  }
  if (outputs_on_device) {
    a.out_idx = out_idx; a.out_d2 = out_d2; a.out_count = out_count;
  } else {
    const size_t need = (size_t)nq * (size_t)k;
    if (need > nn->cap_out) {
      SMX_HIP(hipDeviceSynchronize());
      const size_t cap_o = need + need / 8;
      rc = grow(&nn->didx, cap_o);
      if (rc == SMX_OK) rc = grow(&nn->dd2, cap_o);
      if (rc == SMX_OK) rc = grow(&nn->dcnt, cap_o);
      nn->cap_out = rc == SMX_OK ? cap_o : 0;
      if (rc != SMX_OK) return rc;
    }
    a.out_idx = nn->didx; a.out_d2 = nn->dd2; a.out_count = nn->dcnt;
  }
  const int grid = 2048;
  hipLaunchKernelGGL(k_query_keys, dim3(grid), dim3(kBlock), 0, st, dqx, dqy, dqz, nq, nn->grid, nn->qkeys[0], nn->qvals[0]);
  const int cur = radix_sort(nn->qkeys, nn->qvals, nq, nn->grid.brick_bits, nn->hist, st);
  hipLaunchKernelGGL(k_gather_queries, dim3(grid), dim3(kBlock), 0, st, nn->qvals[cur], dqx, dqy, dqz, dqr2, nq, nn->qrec);
  a.qrec = nn->qrec;
  hipLaunchKernelGGL(k_tile_flags, dim3((nq + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nn->qkeys[cur], nq, 0, nn->qflags);
  uint32_t* total = nullptr;
  uint32_t* flag_ws = nn->hist + sort_hist_elems(max_n) + scan_workspace_elems(sort_hist_elems(max_n));
  exclusive_scan_inplace(nn->qflags, nq, flag_ws, st, &total);
  hipLaunchKernelGGL(k_tile_starts, dim3(grid), dim3(kBlock), 0, st, nn->qflags, nq, total, nn->qtile_start);
  a.qorder = nn->qvals[cur]; a.tile_start = nn->qtile_start; a.n_tiles = total;
  a.stat = nn->stats_enabled ? nn->stat : nullptr;
  const unsigned blocks = (unsigned)std::min<size_t>((size_t)nn->grid_blocks, ((size_t)nq + 15) / 16 + 1);
  if (nn->query_mode == 1) {
    const unsigned sb = (unsigned)std::min<size_t>(((size_t)nq + 3) / 4, 65536);
    hipLaunchKernelGGL(k_query_stream, dim3(sb), dim3(kBlock), 0, st, a);
  } else if (nn->query_mode == 0) {
    hipLaunchKernelGGL(k_query_tiles<false>, dim3(blocks), dim3(64 * kTileWaves), 0, st, a);
  } else {
    a.redo_flag = reinterpret_cast<uint32_t*>(nn->stat + 4);
    a.tile_redo = nn->tile_redo_q;
    SMX_HIP(hipMemsetAsync(a.redo_flag, 0, 4, st));
    const unsigned lb = std::max(8u, (unsigned)std::min<size_t>((size_t)nn->grid_blocks * 4, (size_t)nq + 1) & ~7u);   // (a multiple of 8: SMX_NN_XCD_MAP)
    if (a.state) hipLaunchKernelGGL((k_query_lanes<false, true>), dim3(lb), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_query_lanes<false, false>), dim3(lb), dim3(64), 0, st, a);
    a.redo = 1;
    hipLaunchKernelGGL(k_query_tiles<false>, dim3(blocks), dim3(64 * kTileWaves), 0, st, a);
  }
  SMX_LAUNCH_CHECK();
  if (!outputs_on_device) {
    SMX_HIP(hipMemcpyAsync(out_idx, a.out_idx, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    SMX_HIP(hipMemcpyAsync(out_d2, a.out_d2, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    SMX_HIP(hipMemcpyAsync(out_count, a.out_count, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
  }
  if (!queries_on_device || !outputs_on_device) SMX_HIP(hipStreamSynchronize(st));   // host memory was read / written
  return SMX_OK;
}

int smx_nn_query_self(smx_nn nn, smx_stream s, const float* radius_squared, float factor, int32_t k, const uint8_t* state,
                      uint8_t skip_mask, uint32_t* out_idx, float* out_d2, int32_t* out_count) {
  SMX_CHECK_ARG(nn != nullptr && k >= 1 && k <= 64 && factor >= 0);
  SMX_ON_DEVICE(nn->device);
  if (nn->n == 0) return SMX_OK;
  SMX_CHECK_ARG(out_idx && out_d2 && out_count);
  hipStream_t st = (hipStream_t)s;
  // points that are not indexed (non-finite coordinates) get no row: their count is 0
  SMX_HIP(hipMemsetAsync(out_count, 0, (size_t)nn->n * 4, st));
  if (nn->n_valid == 0) return SMX_OK;
  QueryArgs a;
  memset(&a, 0, sizeof(a));
  a.nq = nn->n_valid; a.K = k; a.state = state; a.skip_mask = skip_mask; a.g = nn->grid; a.table = nn->table;
  a.mask = (uint32_t)(nn->table_slots - 1); a.sorted = nn->sorted; a.n_valid = nn->n_valid;
  a.out_idx = out_idx; a.out_d2 = out_d2; a.out_count = out_count;
  a.self_r2 = radius_squared; a.self_factor = factor;
  a.stat = nn->stats_enabled ? nn->stat : nullptr;
#if SMX_NN_SELF_SORTED
  a.tile_start = nn->self_tile_start; a.n_tiles = nn->self_n_tiles;   // tiles in key order (smx_nn_build)
  const size_t tiles_bound = (size_t)nn->n_bricks + (size_t)nn->n_valid / kTile + 2;
#else
  const size_t tiles_bound = nn->table_slots;
#endif
  const unsigned blocks = (unsigned)std::min<size_t>((size_t)nn->grid_blocks, tiles_bound);
  if (nn->query_mode == 2) {
    a.redo_flag = reinterpret_cast<uint32_t*>(nn->stat + 4);
    a.tile_redo = nn->tile_redo_self;
    SMX_HIP(hipMemsetAsync(a.redo_flag, 0, 4, st));
    const unsigned lb = std::max(8u, (unsigned)std::min<size_t>((size_t)nn->grid_blocks * 4, tiles_bound) & ~7u);   // (a multiple of 8: SMX_NN_XCD_MAP)
    if (a.state) hipLaunchKernelGGL((k_query_lanes<true, true>), dim3(lb), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_query_lanes<true, false>), dim3(lb), dim3(64), 0, st, a);
    a.redo = 1;
  }
  hipLaunchKernelGGL(k_query_tiles<true>, dim3(blocks), dim3(64 * kTileWaves), 0, st, a);
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_nn_set_query_mode(smx_nn nn, int32_t mode) {
  SMX_CHECK_ARG(nn != nullptr && mode >= 0 && mode <= 2);
  nn->query_mode = mode;
  return SMX_OK;
}

int smx_nn_set_stats_enabled(smx_nn nn, smx_stream s, int32_t enabled) {
  SMX_CHECK_ARG(nn != nullptr);
  SMX_ON_DEVICE(nn->device);
  nn->stats_enabled = enabled ? 1 : 0;
  SMX_HIP(hipMemsetAsync(nn->stat, 0, 4 * sizeof(unsigned long long), (hipStream_t)s));
  return SMX_OK;
}

int smx_nn_get_stats(smx_nn nn, smx_stream s, smx_nn_stats* out) {
  SMX_CHECK_ARG(nn != nullptr && out != nullptr);
  SMX_ON_DEVICE(nn->device);
  unsigned long long h[4];
  SMX_HIP(hipMemcpyAsync(h, nn->stat, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)s));
  SMX_HIP(hipStreamSynchronize((hipStream_t)s));
  out->n_points = nn->n; out->n_indexed = nn->n_valid; out->n_bricks = nn->n_bricks;
  out->cell_size = nn->grid.cell;
  for (int a = 0; a < 3; ++a) out->dim[a] = nn->grid.dim[a];
  out->key_bits = nn->grid.key_bits;
  out->tiles = h[0]; out->staged_candidates = h[1]; out->distance_tests = h[2]; out->results = h[3];
  return SMX_OK;
}

}  // extern "C"
