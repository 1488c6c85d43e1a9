/*
 * smx_oracle_nn.c -- CPU oracle for the radius-neighbor search that feeds the
 * mesher.  TEST INFRASTRUCTURE ONLY (see smx_oracle.h).
 *
 * Contract restated from the reference:
 *   CompressedOctree::FindNearestSurfelsWithinRadius  APP/octree.cc:313-470
 *     up to K nearest points with dist^2 <= radius^2, ascending dist^2,
 *     optional exclusion by meshing state (octree.cc:330-335);
 *   FindNearestSurfelsWithinRadiusBruteForce          APP/test/test_octree.cc:116-143
 *     the reference's own oracle for it (the octree tests demand exact
 *     equality with this brute force, test_octree.cc:369-495).
 * The reference leaves the order of equal-distance results unspecified
 * (std::sort on distance only / visit order); here ties are ordered by index.
 */
#include "smx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  /* Eigen (a - b).squaredNorm() for Vec3f: (dx*dx + dy*dy) + dz*dz */
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return dx * dx + dy * dy + dz * dz;
}

static inline int before(float d2a, uint32_t ia, float d2b, uint32_t ib) {
  return d2a < d2b || (d2a == d2b && ia < ib);
}

/* insertion into a (dist^2, index)-sorted top-K list, octree.cc:336-356 */
static inline void topk_insert(float d2, uint32_t idx, int k, float* od2, uint32_t* oidx, int* count) {
  if (*count == k) {
    if (!before(d2, idx, od2[k - 1], oidx[k - 1])) return;
  } else {
    ++*count;
  }
  int i;
  for (i = *count - 1; i > 0; --i) {
    if (before(d2, idx, od2[i - 1], oidx[i - 1])) { od2[i] = od2[i - 1]; oidx[i] = oidx[i - 1]; }
    else break;
  }
  od2[i] = d2; oidx[i] = idx;
}

int orc_nn_bruteforce(const float* px, const float* py, const float* pz, uint32_t n,
                      float qx, float qy, float qz, float radius_sq, int k,
                      const uint8_t* state, uint8_t skip_mask,
                      float* out_d2, uint32_t* out_idx) {
  int count = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (state && (state[i] & skip_mask)) continue;
    const float d2 = dist2(px[i], py[i], pz[i], qx, qy, qz);
    if (!(d2 <= radius_sq)) continue;
    topk_insert(d2, i, k, out_d2, out_idx, &count);
  }
  return count;
}

/* Uniform-grid version (same answers); used as the CPU baseline at sizes where
 * brute force is quadratic.  Grid = counting sort of points by cell. */
void orc_nn_grid_batch(const float* px, const float* py, const float* pz, uint32_t n,
                       float cell, const float* qx, const float* qy, const float* qz, const float* qr2,
                       uint32_t nq, int k, float* out_d2, uint32_t* out_idx, int32_t* out_count) {
  if (n == 0) { for (uint32_t q = 0; q < nq; ++q) out_count[q] = 0; return; }
  float mn[3] = {px[0], py[0], pz[0]}, mx[3] = {px[0], py[0], pz[0]};
  for (uint32_t i = 1; i < n; ++i) {
    if (px[i] < mn[0]) mn[0] = px[i]; if (px[i] > mx[0]) mx[0] = px[i];
    if (py[i] < mn[1]) mn[1] = py[i]; if (py[i] > mx[1]) mx[1] = py[i];
    if (pz[i] < mn[2]) mn[2] = pz[i]; if (pz[i] > mx[2]) mx[2] = pz[i];
  }
  int dim[3];
  for (int a = 0; a < 3; ++a) { dim[a] = (int)floorf((mx[a] - mn[a]) / cell) + 1; if (dim[a] < 1) dim[a] = 1; }
  const size_t ncell = (size_t)dim[0] * dim[1] * dim[2];
  uint32_t* start = (uint32_t*)calloc(ncell + 1, 4);
  uint32_t* cell_of = (uint32_t*)malloc((size_t)n * 4);
  uint32_t* order = (uint32_t*)malloc((size_t)n * 4);
  for (uint32_t i = 0; i < n; ++i) {
    int c[3] = {(int)floorf((px[i] - mn[0]) / cell), (int)floorf((py[i] - mn[1]) / cell), (int)floorf((pz[i] - mn[2]) / cell)};
    for (int a = 0; a < 3; ++a) { if (c[a] < 0) c[a] = 0; if (c[a] >= dim[a]) c[a] = dim[a] - 1; }
    cell_of[i] = (uint32_t)(((size_t)c[2] * dim[1] + c[1]) * dim[0] + c[0]);
    start[cell_of[i] + 1]++;
  }
  for (size_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
  uint32_t* fill = (uint32_t*)malloc(ncell * 4);
  memcpy(fill, start, ncell * 4);
  for (uint32_t i = 0; i < n; ++i) order[fill[cell_of[i]]++] = i;

  for (uint32_t q = 0; q < nq; ++q) {
    float* od2 = out_d2 + (size_t)q * k; uint32_t* oidx = out_idx + (size_t)q * k;
    int count = 0;
    const float rad = sqrtf(qr2[q]) * 1.0001f + 1e-6f;   /* conservative cell range; the d2 test is exact */
    int lo[3], hi[3];
    const float qp[3] = {qx[q], qy[q], qz[q]};
    int empty = 0;
    for (int a = 0; a < 3; ++a) {
      lo[a] = (int)floorf((qp[a] - rad - mn[a]) / cell); hi[a] = (int)floorf((qp[a] + rad - mn[a]) / cell);
      if (lo[a] < 0) lo[a] = 0; if (hi[a] >= dim[a]) hi[a] = dim[a] - 1;
      if (lo[a] > hi[a]) empty = 1;
    }
    if (!empty)
      for (int cz = lo[2]; cz <= hi[2]; ++cz)
        for (int cy = lo[1]; cy <= hi[1]; ++cy)
          for (int cx = lo[0]; cx <= hi[0]; ++cx) {
            const size_t c = ((size_t)cz * dim[1] + cy) * dim[0] + cx;
            for (uint32_t s = start[c]; s < start[c + 1]; ++s) {
              const uint32_t i = order[s];
              const float d2 = dist2(px[i], py[i], pz[i], qp[0], qp[1], qp[2]);
              if (!(d2 <= qr2[q])) continue;
              topk_insert(d2, i, k, od2, oidx, &count);
            }
          }
    out_count[q] = count;
  }
  free(start); free(cell_of); free(order); free(fill);
}

/* ---- CheckRemeshing's per-triangle tests (APP/surfel_meshing.cc:590-650) ---- */
static float sqn3(const float a[3]) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }
static float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

void orc_check_triangles(const float* x, const float* y, const float* z, const float* radius_squared,
                         const float* nx, const float* ny, const float* nz, uint32_t n,
                         const uint32_t* triangles, uint32_t n_triangles, float long_edge_total_factor_squared,
                         uint8_t* flags) {
  for (uint32_t t = 0; t < n_triangles; ++t) {
    const uint32_t* v = triangles + 3 * (size_t)t;
    if (v[0] >= n || v[1] >= n || v[2] >= n) { flags[t] = 16; continue; }
    float p[3][3], nrm[3][3], maxsq[3];
    uint8_t f = 0;
    for (int k = 0; k < 3; ++k) {
      p[k][0] = x[v[k]]; p[k][1] = y[v[k]]; p[k][2] = z[v[k]];
      nrm[k][0] = nx[v[k]]; nrm[k][1] = ny[v[k]]; nrm[k][2] = nz[v[k]];
      maxsq[k] = long_edge_total_factor_squared * radius_squared[v[k]];   /* :556-557, 593-596 */
      if (radius_squared[v[k]] < 0) f |= 16;                               /* :559 */
    }
    /* edge e[k] joins vertex k and vertex k+1 (a difference and its negation have the same squared norm) */
    float e[3];
    for (int k = 0; k < 3; ++k) {
      const int a = k, b = (k + 1) % 3;
      const float d[3] = {p[b][0] - p[a][0], p[b][1] - p[a][1], p[b][2] - p[a][2]};
      e[k] = sqn3(d);
    }
    /* :605-617 -- an edge longer than both of its ends allow, and the opposite vertex has an over-long edge too */
    for (int k = 0; k < 3; ++k) {
      const int a = k, b = (k + 1) % 3, c = (k + 2) % 3;     /* edge a-b = e[k]; c's edges: e[b] (b-c), e[c] (c-a) */
      if (e[k] > maxsq[a] && e[k] > maxsq[b] && (e[b] > maxsq[c] || e[c] > maxsq[c])) f |= 1;
    }
    /* :632-635 with pivot k: right = next vertex, left = previous (:576-589) */
    for (int k = 0; k < 3; ++k) {
      const int r = (k + 1) % 3, l = (k + 2) % 3;
      const float sr[3] = {p[r][0] - p[k][0], p[r][1] - p[k][1], p[r][2] - p[k][2]};
      const float sl[3] = {p[l][0] - p[k][0], p[l][1] - p[k][1], p[l][2] - p[k][2]};
      const float c[3] = {sr[1] * sl[2] - sr[2] * sl[1], sr[2] * sl[0] - sr[0] * sl[2], sr[0] * sl[1] - sr[1] * sl[0]};
      if (dot3(c, nrm[k]) <= 0 && dot3(c, nrm[r]) <= 0 && dot3(c, nrm[l]) <= 0) f |= (uint8_t)(2 << k);
    }
    flags[t] = f;
  }
}
