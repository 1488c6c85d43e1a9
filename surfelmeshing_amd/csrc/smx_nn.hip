// smx_nn.hip -- batched radius-neighbor search for the mesher (gfx950).
//
// Replaces, for batched queries, CompressedOctree::FindNearestSurfelsWithinRadius
// (APP/octree.cc:313-470): up to K nearest points with dist^2 <= r^2 in ascending
// order, optional exclusion by a per-point state byte (octree.cc:330-335).  The
// pointer-chasing octree is replaced by a uniform grid rebuilt from the position rows:
//   build:  bounding box -> cell histogram (atomics) -> exclusive scan -> scatter of
//           (x, y, z, index) records sorted by cell;
//   query:  one wavefront per query; the cells overlapping the query ball are streamed
//           64 candidates at a time (one per lane, coalesced 16-byte records) and the
//           running top-K (K <= 64) is kept one entry per lane, ordered by (dist^2, index).
#include <math.h>

#include <algorithm>

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
constexpr uint32_t kNoCell = 0xFFFFFFFFu;

__global__ void __launch_bounds__(kBlock)
k_bbox(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
       uint32_t* __restrict__ bb /* min xyz, max xyz as order keys */) {
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
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { atomicMin(&bb[a], mn[a]); atomicMax(&bb[3 + a], mx[a]); }
  }
}

struct Grid {
  float min[3];
  float cell;
  int dim[3];
};

__device__ __forceinline__ int cell_coord(float p, float mn, float cell, int dim) {
  int c = (int)floorf((p - mn) / cell);
  return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

__global__ void __launch_bounds__(kBlock)
k_count_cells(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
              Grid g, uint32_t* __restrict__ cell_of, uint32_t* __restrict__ counts) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    if (!indexable(x[i], y[i], z[i])) { cell_of[i] = kNoCell; continue; }
    const int cx = cell_coord(x[i], g.min[0], g.cell, g.dim[0]);
    const int cy = cell_coord(y[i], g.min[1], g.cell, g.dim[1]);
    const int cz = cell_coord(z[i], g.min[2], g.cell, g.dim[2]);
    const uint32_t c = (uint32_t)(((size_t)cz * g.dim[1] + cy) * g.dim[0] + cx);
    cell_of[i] = c;
    atomicAdd(&counts[c], 1u);
  }
}

// in-place exclusive scan of 2048 elements per block; block totals to `sums`
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

__global__ void __launch_bounds__(kBlock)
k_scatter(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
          const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ start, uint32_t* __restrict__ fill,
          float4* __restrict__ sorted) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const uint32_t c = cell_of[i];
    if (c == kNoCell) continue;
    const uint32_t pos = start[c] + atomicAdd(&fill[c], 1u);
    sorted[pos] = make_float4(x[i], y[i], z[i], __uint_as_float(i));
  }
}

__device__ __forceinline__ bool before(float d2a, uint32_t ia, float d2b, uint32_t ib) {
  return d2a < d2b || (d2a == d2b && ia < ib);
}

// One wavefront per query.
__global__ void __launch_bounds__(kBlock)
k_query(uint32_t nq, const float* __restrict__ qx, const float* __restrict__ qy, const float* __restrict__ qz,
        const float* __restrict__ qr2, int K, const uint8_t* __restrict__ state, uint8_t skip_mask, Grid g,
        const uint32_t* __restrict__ start, const float4* __restrict__ sorted,
        uint32_t* __restrict__ out_idx, float* __restrict__ out_d2, int32_t* __restrict__ out_count) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t waves_per_block = kBlock / 64;
  for (uint32_t q = blockIdx.x * waves_per_block + (threadIdx.x >> 6); q < nq; q += gridDim.x * waves_per_block) {
    const float px = qx[q], py = qy[q], pz = qz[q], r2 = qr2[q];
    const float rad = sqrtf(r2) * 1.0001f + 1e-6f;  // conservative cell range; the dist^2 test is exact
    int lo[3], hi[3];
    const float qp[3] = {px, py, pz};
    bool empty = !(r2 >= 0);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = (int)floorf((qp[a] - rad - g.min[a]) / g.cell);
      hi[a] = (int)floorf((qp[a] + rad - g.min[a]) / g.cell);
      if (lo[a] < 0) lo[a] = 0;
      if (hi[a] >= g.dim[a]) hi[a] = g.dim[a] - 1;
      if (lo[a] > hi[a]) empty = true;
    }
    // running top-K: lane j holds the j-th best (dist^2, index)
    float my_d2 = __builtin_inff();
    uint32_t my_idx = kInvalid;
    int count = 0;
    if (!empty) {
      for (int cz = lo[2]; cz <= hi[2]; ++cz)
        for (int cy = lo[1]; cy <= hi[1]; ++cy) {
          // cells lo[0]..hi[0] of one x-row are contiguous in the sorted array
          const size_t c0 = ((size_t)cz * g.dim[1] + cy) * g.dim[0];
          const uint32_t s0 = start[c0 + lo[0]], s1 = start[c0 + hi[0] + 1];
          for (uint32_t s = s0; s < s1; s += 64) {
            const uint32_t t = s + lane;
            float d2 = __builtin_inff();
            uint32_t idx = kInvalid;
            bool ok = false;
            if (t < s1) {
              const float4 rec = sorted[t];
              idx = __float_as_uint(rec.w);
              const float dx = rec.x - px, dy = rec.y - py, dz = rec.z - pz;
              d2 = dx * dx + dy * dy + dz * dz;
              ok = d2 <= r2;
              if (ok && state != nullptr && (state[idx] & skip_mask)) ok = false;
            }
            unsigned long long m = __ballot(ok);
            while (m) {
              const int src = __ffsll((long long)m) - 1;
              m &= m - 1;
              const float cd2 = __shfl(d2, src);
              const uint32_t cidx = (uint32_t)__shfl((int)idx, src);
              // is the list full and the candidate not better than the current K-th?
              const float kth_d2 = __shfl(my_d2, K - 1);
              const uint32_t kth_idx = (uint32_t)__shfl((int)my_idx, K - 1);
              if (count == K && !before(cd2, cidx, kth_d2, kth_idx)) continue;
              // insertion position = number of held entries ordered before the candidate
              const bool mine_before = ((int)lane < count) && before(my_d2, my_idx, cd2, cidx);
              const int pos = __popcll(__ballot(mine_before));
              const float up_d2 = __shfl_up(my_d2, 1);
              const uint32_t up_idx = (uint32_t)__shfl_up((int)my_idx, 1);
              if ((int)lane > pos) { my_d2 = up_d2; my_idx = up_idx; }
              else if ((int)lane == pos) { my_d2 = cd2; my_idx = cidx; }
              if (count < K) ++count;
            }
          }
        }
    }
    if ((int)lane < count) {
      out_idx[(size_t)q * K + lane] = my_idx;
      out_d2[(size_t)q * K + lane] = my_d2;
    }
    if (lane == 0) out_count[q] = count;
  }
}

}  // namespace

struct smx_nn_s {
  int device;
  uint32_t n;
  Grid grid;
  size_t ncell;
  float *x, *y, *z;  // device copies (owned)
  uint32_t* cell_start;  // [ncell + 1]
  float4* sorted;
  uint32_t* bbox;
};

namespace {

int exclusive_scan_inplace(uint32_t* data, size_t n, hipStream_t st) {
  const size_t nblocks = (n + kScanPerBlock - 1) / kScanPerBlock;
  uint32_t* sums = nullptr;
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&sums), nblocks * sizeof(uint32_t)));
  hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nblocks), dim3(kBlock), 0, st, data, n, sums);
  int rc = SMX_OK;
  if (nblocks > 1) {
    rc = exclusive_scan_inplace(sums, nblocks, st);
    if (rc == SMX_OK) hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nblocks), dim3(kBlock), 0, st, data, n, sums);
  }
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(sums);
  if (e != hipSuccess) { set_error("scan failed: %s", hipGetErrorString(e)); return SMX_ERR_HIP; }
  return rc;
}

void nn_free(smx_nn nn) {
  void* ptrs[] = {nn->x, nn->y, nn->z, nn->cell_start, nn->sorted, nn->bbox};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  nn->x = nn->y = nn->z = nullptr; nn->cell_start = nullptr; nn->sorted = nullptr; nn->bbox = nullptr;
  nn->n = 0;
}

}  // namespace

extern "C" {

int smx_nn_create(int32_t device_id, smx_nn* out) {
  SMX_CHECK_ARG(out != nullptr);
  int device = 0;
  { const int rcd = resolve_device(device_id, &device); if (rcd != SMX_OK) return rcd; }
  smx_nn_s* nn = new smx_nn_s();
  memset(nn, 0, sizeof(*nn));
  nn->device = device;
  *out = nn;
  return SMX_OK;
}

int smx_nn_destroy(smx_nn nn) {
  if (!nn) return SMX_OK;
  SMX_ON_DEVICE(nn->device);
  nn_free(nn);
  delete nn;
  return SMX_OK;
}

int smx_nn_build(smx_nn nn, smx_stream s, const float* x, const float* y, const float* z, uint32_t n,
                 float cell_size, int32_t rows_on_device) {
  SMX_CHECK_ARG(nn != nullptr && cell_size > 0 && (n == 0 || (x && y && z)));
  SMX_ON_DEVICE(nn->device);
  hipStream_t st = (hipStream_t)s;
  nn_free(nn);
  nn->n = n;
  if (n == 0) return SMX_OK;
  const hipMemcpyKind kind = rows_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&nn->x), (size_t)n * 4));
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&nn->y), (size_t)n * 4));
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&nn->z), (size_t)n * 4));
  SMX_HIP(hipMemcpyAsync(nn->x, x, (size_t)n * 4, kind, st));
  SMX_HIP(hipMemcpyAsync(nn->y, y, (size_t)n * 4, kind, st));
  SMX_HIP(hipMemcpyAsync(nn->z, z, (size_t)n * 4, kind, st));
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&nn->bbox), 6 * sizeof(uint32_t)));
  const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0};
  SMX_HIP(hipMemcpyAsync(nn->bbox, init, sizeof(init), hipMemcpyHostToDevice, st));
  const int grid = 2048;
  hipLaunchKernelGGL(k_bbox, dim3(grid), dim3(kBlock), 0, st, nn->x, nn->y, nn->z, n, nn->bbox);
  uint32_t bb[6];
  SMX_HIP(hipMemcpyAsync(bb, nn->bbox, sizeof(bb), hipMemcpyDeviceToHost, st));
  SMX_HIP(hipStreamSynchronize(st));
  if (bb[0] > bb[3]) {  // no indexable point
    nn_free(nn);
    return SMX_OK;
  }
  float mn[3], mx[3];
  for (int a = 0; a < 3; ++a) { mn[a] = order_unkey(bb[a]); mx[a] = order_unkey(bb[3 + a]); }
  // grow the cell until the dense grid fits (queries stay exact: they visit every overlapped cell)
  float cell = cell_size;
  const size_t kMaxCells = (size_t)1 << 28;
  for (;;) {
    size_t total = 1;
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
      const double d = floor(((double)mx[a] - (double)mn[a]) / cell) + 1.0;
      if (!(d < 2.0e9)) { ok = false; break; }
      nn->grid.dim[a] = (int)d < 1 ? 1 : (int)d;
      total *= (size_t)nn->grid.dim[a];
      if (total > kMaxCells) { ok = false; break; }
    }
    if (ok) { nn->ncell = total; break; }
    cell *= 2.0f;
  }
  nn->grid.cell = cell;
  for (int a = 0; a < 3; ++a) nn->grid.min[a] = mn[a];

  uint32_t *cell_of = nullptr, *fill = nullptr;
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&cell_of), (size_t)n * 4));
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&nn->cell_start), (nn->ncell + 1) * 4));
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&fill), nn->ncell * 4));
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&nn->sorted), (size_t)n * sizeof(float4)));
  SMX_HIP(hipMemsetAsync(nn->cell_start, 0, (nn->ncell + 1) * 4, st));
  SMX_HIP(hipMemsetAsync(fill, 0, nn->ncell * 4, st));
  hipLaunchKernelGGL(k_count_cells, dim3(grid), dim3(kBlock), 0, st, nn->x, nn->y, nn->z, n, nn->grid, cell_of, nn->cell_start);
  int rc = exclusive_scan_inplace(nn->cell_start, nn->ncell + 1, st);
  if (rc == SMX_OK) {
    hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(kBlock), 0, st, nn->x, nn->y, nn->z, n, cell_of, nn->cell_start, fill, nn->sorted);
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("nn build failed: %s", hipGetErrorString(e)); rc = SMX_ERR_HIP; }
  }
  (void)hipFree(cell_of);
  (void)hipFree(fill);
  return rc;
}

int smx_nn_query_batch(smx_nn nn, smx_stream s, uint32_t nq, const float* qx, const float* qy, const float* qz,
                       const float* r2, int32_t k, const uint8_t* state, uint8_t skip_mask, int32_t queries_on_device,
                       uint32_t* out_idx, float* out_d2, int32_t* out_count, int32_t outputs_on_device) {
  SMX_CHECK_ARG(nn != nullptr && k >= 1 && k <= 64);
  SMX_ON_DEVICE(nn->device);
  SMX_CHECK_ARG(nq == 0 || (qx && qy && qz && r2 && out_idx && out_d2 && out_count));
  if (nq == 0) return SMX_OK;
  hipStream_t st = (hipStream_t)s;
  if (nn->n == 0) {
    if (outputs_on_device) SMX_HIP(hipMemsetAsync(out_count, 0, (size_t)nq * 4, st));
    else memset(out_count, 0, (size_t)nq * 4);
    return SMX_OK;
  }
  float *dq[4] = {nullptr, nullptr, nullptr, nullptr};
  const float* src[4] = {qx, qy, qz, r2};
  uint8_t* dstate = nullptr;
  uint32_t* didx = out_idx; float* dd2 = out_d2; int32_t* dcnt = out_count;
  int rc = SMX_OK;
  hipError_t e = hipSuccess;
  auto fail = [&](hipError_t err) { set_error("nn query failed: %s", hipGetErrorString(err)); rc = SMX_ERR_HIP; };
  if (!queries_on_device) {
    for (int a = 0; a < 4 && rc == SMX_OK; ++a) {
      if ((e = hipMalloc(reinterpret_cast<void**>(&dq[a]), (size_t)nq * 4)) != hipSuccess) { fail(e); break; }
      if ((e = hipMemcpyAsync(dq[a], src[a], (size_t)nq * 4, hipMemcpyHostToDevice, st)) != hipSuccess) fail(e);
    }
    if (rc == SMX_OK && state) {
      if ((e = hipMalloc(reinterpret_cast<void**>(&dstate), nn->n)) != hipSuccess) fail(e);
      else if ((e = hipMemcpyAsync(dstate, state, nn->n, hipMemcpyHostToDevice, st)) != hipSuccess) fail(e);
    }
  }
  if (rc == SMX_OK && !outputs_on_device) {
    if ((e = hipMalloc(reinterpret_cast<void**>(&didx), (size_t)nq * k * 4)) != hipSuccess) fail(e);
    else if ((e = hipMalloc(reinterpret_cast<void**>(&dd2), (size_t)nq * k * 4)) != hipSuccess) fail(e);
    else if ((e = hipMalloc(reinterpret_cast<void**>(&dcnt), (size_t)nq * 4)) != hipSuccess) fail(e);
  }
  if (rc == SMX_OK) {
    const float* a0 = queries_on_device ? qx : dq[0];
    const float* a1 = queries_on_device ? qy : dq[1];
    const float* a2 = queries_on_device ? qz : dq[2];
    const float* a3 = queries_on_device ? r2 : dq[3];
    const uint8_t* stp = queries_on_device ? state : dstate;
    const unsigned blocks = (unsigned)std::min<size_t>(((size_t)nq + 3) / 4, 65536);
    hipLaunchKernelGGL(k_query, dim3(blocks), dim3(kBlock), 0, st, nq, a0, a1, a2, a3, (int)k, stp, skip_mask,
                       nn->grid, nn->cell_start, nn->sorted, didx, dd2, dcnt);
    if ((e = hipGetLastError()) != hipSuccess) fail(e);
  }
  if (rc == SMX_OK && !outputs_on_device) {
    if ((e = hipMemcpyAsync(out_idx, didx, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) fail(e);
    else if ((e = hipMemcpyAsync(out_d2, dd2, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) fail(e);
    else if ((e = hipMemcpyAsync(out_count, dcnt, (size_t)nq * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) fail(e);
  }
  if (!queries_on_device || !outputs_on_device) {
    if ((e = hipStreamSynchronize(st)) != hipSuccess && rc == SMX_OK) fail(e);
    for (int a = 0; a < 4; ++a) if (dq[a]) (void)hipFree(dq[a]);
    if (dstate) (void)hipFree(dstate);
    if (!outputs_on_device) { if (didx) (void)hipFree(didx); if (dd2) (void)hipFree(dd2); if (dcnt) (void)hipFree(dcnt); }
  }
  return rc;
}

}  // extern "C"
