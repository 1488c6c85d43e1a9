/*
 * smx_oracle_recon.c -- CPU oracle, CUDASurfelReconstruction::Integrate & co.
 * TEST INFRASTRUCTURE ONLY (see smx_oracle.h).  Restates
 * APP/cuda_surfel_reconstruction.cc:112-359 and the kernels of
 * APP/cuda_surfel_reconstruction_kernels.cu as single-threaded loops.
 * Build with -ffp-contract=off.
 */
#include "smx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SURF(r, row, i) ((r)->surfels[(size_t)(row) * (r)->max_surfels + (i)])

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#define SURF_U32(r, row, i) f2u(SURF(r, row, i))
#define SET_SURF_U32(r, row, i, v) (SURF(r, row, i) = u2f(v))

static inline uint16_t f2u16(float v) {
  if (!(v > 0.0f)) return 0;
  if (v >= 65535.0f) return 65535;
  return (uint16_t)(int32_t)v;
}

/* 2^-32 fixed point: exact for every float >= 2^-8 in magnitude. */
static inline int64_t q_from_float(float v) { return (int64_t)((double)v * 4294967296.0); }
static inline float q_to_float(int64_t s) { return (float)((double)s * (1.0 / 4294967296.0)); }
/* Regulariser gradient terms: 2^-22 fixed point (0.24 um), clamped to +-16 (NaN -> lower bound), so that the
 * HIP side can keep two of them in one 64-bit accumulator word; the sums are exact integer sums either way. */
static inline int32_t q22_from_float(float v) {
  double d = (double)v * 4194304.0;
  if (!(d > -67108864.0)) d = -67108864.0;
  if (d > 67108864.0) d = 67108864.0;
  return (int32_t)d;
}
static inline float q22_to_float(int64_t s) { return (float)((double)s * (1.0 / 4194304.0)); }

/* CUDAMatrix3x4::operator*, VIS/cuda/cuda_matrix.cuh:88-94 (left-to-right adds) */
static inline void mat_point(const float* m, const float* p, float* o) {
  o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
  o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
  o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
}
/* CUDAMatrix3x4::Rotate, cuda_matrix.cuh:106-112 */
static inline void mat_rotate(const float* m, const float* p, float* o) {
  o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2];
  o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2];
  o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2];
}
/* SE3f::inverse() as R^T, -(R^T t); the reference gets it from Sophus/Eigen
 * (cuda_surfel_reconstruction.cc:144,156,181,251), version unpinned. */
static void se3_inverse(const float* m, float* o) {
  for (int i = 0; i < 3; ++i) {
    o[4 * i + 0] = m[0 + i]; o[4 * i + 1] = m[4 + i]; o[4 * i + 2] = m[8 + i];
    o[4 * i + 3] = -(o[4 * i + 0] * m[3] + o[4 * i + 1] * m[7] + o[4 * i + 2] * m[11]);
  }
}

typedef struct {
  float fx, fy, cx, cy;                 /* projection (pixel corner)  */
  float fx_inv, fy_inv, cx_inv, cy_inv; /* unprojection (pixel centre), kernels.cc:69-74 */
  float L[12];                          /* local_T_global */
  float G[12];                          /* global_T_local */
  float inv_depth_scaling;              /* 1 / depth_scaling */
  float sensor_noise_factor;
  float cos_normal_compat;              /* kernels.cc:261,428,489 */
  float rf2;                            /* radius factor squared, kernels.cc:145,320 */
  float max_conf;
  int window;
  uint32_t frame;
} frame_ctx;

typedef struct { float l[3]; float g[3]; float u, v; int px, py; } proj_t;

/* IsSurfelActiveForIntegration, kernels.cu:77-87 */
static inline int is_active(const orc_recon* r, uint32_t i, const frame_ctx* c) {
  int32_t bound = (int32_t)((uint32_t)c->frame - (uint32_t)c->window);
  return (int32_t)SURF_U32(r, ORC_LAST_UPDATE_STAMP, i) > bound;
}

/* Shared projection: kernels.cu:1481-1500 == 1722-1741 == 2018-2031 == 1023-1048.
 * The float range test is equivalent to the reference's (u<0|v<0|px<0|py<0|
 * px>=W|py>=H) for every non-NaN input and keeps the int cast in range. */
static int project(const orc_recon* r, uint32_t i, const frame_ctx* c, proj_t* o) {
  o->g[0] = SURF(r, ORC_X, i); o->g[1] = SURF(r, ORC_Y, i); o->g[2] = SURF(r, ORC_Z, i);
  mat_point(c->L, o->g, o->l);
  if (!(o->l[2] > 0)) return 0;
  o->u = c->fx * (o->l[0] / o->l[2]) + c->cx;
  o->v = c->fy * (o->l[1] / o->l[2]) + c->cy;
  if (!(o->u >= 0 && o->v >= 0 && o->u < (float)r->width && o->v < (float)r->height)) return 0;
  o->px = (int)o->u; o->py = (int)o->v;
  return 1;
}

/* The "triangle quadrant" neighbour, kernels.cu:1078-1120 == 1506-1549 == 1752-1795
 * (including the px > 1 [sic] test for the left neighbour). */
static int quadrant(const orc_recon* r, const proj_t* p, int* ox, int* oy) {
  const float xf = p->u - (float)p->px, yf = p->v - (float)p->py;
  if (xf < yf) {
    if (xf < 1 - yf) { if (p->px > 1) { *ox = p->px - 1; *oy = p->py; return 1; } return 0; }
    else { if (p->py < r->height - 1) { *ox = p->px; *oy = p->py + 1; return 1; } return 0; }
  } else {
    if (xf < 1 - yf) { if (p->py > 0) { *ox = p->px; *oy = p->py - 1; return 1; } return 0; }
    else { if (p->px < r->width - 1) { *ox = p->px + 1; *oy = p->py; return 1; } return 0; }
  }
}

static inline float meas_normal_z(float nx, float ny) {
  /* kernels.cu:172, 811, 1656 */
  float t = 1 - nx * nx - ny * ny;
  return -sqrtf(t > 0.f ? t : 0.f);
}

/* ------------------------------------------------------------------------- */
orc_recon* orc_recon_create(uint32_t max_surfels, int width, int height,
                            float fx, float fy, float cx, float cy, int sum_mode) {
  orc_recon* r = (orc_recon*)calloc(1, sizeof(orc_recon));
  const size_t P = (size_t)width * height;
  r->width = width; r->height = height; r->fx = fx; r->fy = fy; r->cx = cx; r->cy = cy;
  r->max_surfels = max_surfels; r->sum_mode = sum_mode;
  r->surfels = (float*)calloc((size_t)ORC_ROWS * max_surfels, sizeof(float));
  r->grad_acc = (int64_t*)calloc((size_t)4 * max_surfels, sizeof(int64_t));
  r->supporting = (uint32_t*)calloc(P, 4);
  r->support_counts = (uint32_t*)calloc(P, 4);
  r->depth_sums_f = (float*)calloc(P, 4);
  r->depth_sums_q = (int64_t*)calloc(P, 8);
  r->conflicting = (uint32_t*)calloc(P, 4);
  r->conflicting_key = (uint32_t*)calloc(P, 4);
  r->first_depth = (float*)calloc(P, 4);
  r->distance_map = (uint8_t*)calloc(P, 1);
  r->new_distance_map = (uint8_t*)calloc(P, 1);
  r->deltas = (float*)calloc(P, 4);
  r->new_deltas = (float*)calloc(P, 4);
  r->new_flags = (uint8_t*)calloc(P, 1);
  r->new_indices = (uint32_t*)calloc(P, 4);
  r->merge_decision = (uint8_t*)calloc(max_surfels, 1);
  return r;
}

void orc_recon_destroy(orc_recon* r) {
  if (!r) return;
  free(r->surfels); free(r->grad_acc); free(r->supporting); free(r->support_counts);
  free(r->depth_sums_f); free(r->depth_sums_q); free(r->conflicting); free(r->conflicting_key);
  free(r->first_depth); free(r->distance_map); free(r->new_distance_map); free(r->deltas);
  free(r->new_deltas); free(r->new_flags); free(r->new_indices); free(r->merge_decision);
  free(r);
}

/* ---- race-outcome overrides (pinning against the reference's own kernels, tests/test_gpu_reference_pin.py) ----
 * The reference decides two things by races: the supporting surfel of a pixel (first atomicCAS wins, :1688) and
 * the conflicting surfel (last plain store wins, :1615 / :1887).  This oracle replaces each race by a fixed legal
 * rule.  To compare it with a run of the reference's kernels, the outcome THAT run produced can be imposed: a value
 * is accepted only if this oracle saw the same surfel qualify for the same pixel in the same phase (so only legal
 * outcomes pass; the rest is counted as rejected), and everything downstream is then computed from it. */
static const uint32_t* g_ovr_sup = NULL;
static const uint32_t* g_ovr_conf = NULL;
static uint8_t* g_ovr_seen = NULL;   /* bit 0: supporting candidate, bit 1 / 2: conflict writer in the associate / merge phase */
static uint32_t g_ovr_stats[4];      /* applied supporting, rejected supporting, applied conflicting, rejected conflicting */
void orc_set_race_overrides(const uint32_t* supporting, const uint32_t* conflicting, size_t pixels) {
  free(g_ovr_seen);
  g_ovr_seen = NULL;
  g_ovr_sup = supporting; g_ovr_conf = conflicting;
  if (supporting || conflicting) g_ovr_seen = (uint8_t*)calloc(pixels, 1);
  memset(g_ovr_stats, 0, sizeof(g_ovr_stats));
}
void orc_get_race_override_stats(uint32_t out[4]) { memcpy(out, g_ovr_stats, sizeof(g_ovr_stats)); }

/* ---- stage: 5 clears, cuda_surfel_reconstruction.cc:134-138 ---- */
static void stage_clear(orc_recon* r) {
  const size_t P = (size_t)r->width * r->height;
  for (size_t k = 0; k < P; ++k) {
    r->supporting[k] = ORC_INVALID; r->support_counts[k] = 0;
    r->depth_sums_f[k] = 0; r->depth_sums_q[k] = 0;
    r->conflicting_key[k] = ORC_INVALID; r->first_depth[k] = INFINITY;
  }
}

/* ---- stage: RenderMinDepthCUDAKernel, kernels.cu:1466-1557 ---- */
static inline void min_depth_at(orc_recon* r, int x, int y, float z) {
  /* atomicMin on the int bit pattern (:1463); positive floats order like ints */
  float* f = &r->first_depth[(size_t)y * r->width + x];
  if ((int32_t)f2u(z) < (int32_t)f2u(*f)) *f = z;
}
static void stage_min_depth(orc_recon* r, const frame_ctx* c) {
  uint32_t nvis = 0;
  for (uint32_t i = 0; i < r->surfel_count; ++i) {
    if (!is_active(r, i, c)) continue;
    proj_t p;
    if (!project(r, i, c, &p)) continue;
    ++nvis;
    min_depth_at(r, p.px, p.py, p.l[2]);
    int ox, oy;
    if (quadrant(r, &p, &ox, &oy)) min_depth_at(r, ox, oy, p.l[2]);
  }
  r->last_n_visible = nvis;
}

/* ---- stage: AssociateSurfelsCUDAKernel, kernels.cu:1586-1808 ---- */
static void associate_at(orc_recon* r, const frame_ctx* c, const uint16_t* depth, const float* normals,
                         int x, int y, const proj_t* p, uint32_t i) {
  const size_t k = (size_t)y * r->width + x;
  const float measurement_depth = c->inv_depth_scaling * (float)depth[k];
  if (measurement_depth <= 0) return;
  const float first = r->first_depth[k];
  if (first < (1 - c->sensor_noise_factor) * measurement_depth) {
    if (first == p->l[2]) {
      /* :1615 plain store; deterministic rule: class-1 key, lowest index wins */
      const uint32_t key = 0x80000000u | i;
      if (key < r->conflicting_key[k]) r->conflicting_key[k] = key;
      if (g_ovr_conf && g_ovr_conf[k] == i) g_ovr_seen[k] |= 2;
    }
    return;
  }
  const float occlusion_depth = (1 + c->sensor_noise_factor) * measurement_depth;
  if (p->l[2] > occlusion_depth) return;

  const float surfel_distance = sqrtf(p->l[0] * p->l[0] + p->l[1] * p->l[1] + p->l[2] * p->l[2]);
  const float gn[3] = {SURF(r, ORC_NORMAL_X, i), SURF(r, ORC_NORMAL_Y, i), SURF(r, ORC_NORMAL_Z, i)};
  float ln[3];
  mat_rotate(c->L, gn, ln);
  const float dot_angle = (1.0f / surfel_distance) * (p->l[0] * ln[0] + p->l[1] * ln[1] + p->l[2] * ln[2]);
  if (dot_angle > 0) return;                     /* kSurfelNormalToViewingDirThreshold = 0 */

  if (measurement_depth < p->l[2]) {
    const float nx = normals[2 * k], ny = normals[2 * k + 1];
    const float nz = meas_normal_z(nx, ny);
    const float d = ln[0] * nx + ln[1] * ny + ln[2] * nz;
    if (d < c->cos_normal_compat) return;
  }
  if (SURF(r, ORC_RADIUS_SQ, i) <= 0) return;     /* :1674 */

  /* :1688 atomicCAS first-wins -> lowest index; ascending loop == first wins */
  if (i < r->supporting[k]) r->supporting[k] = i;
  if (g_ovr_sup && g_ovr_sup[k] == i) g_ovr_seen[k] |= 1;
  r->support_counts[k] += 1;
  r->depth_sums_f[k] += p->l[2];
  r->depth_sums_q[k] += q_from_float(p->l[2]);
}
static void stage_associate(orc_recon* r, const frame_ctx* c, const uint16_t* depth, const float* normals) {
  for (uint32_t i = 0; i < r->surfel_count; ++i) {
    if (!is_active(r, i, c)) continue;
    proj_t p;
    if (!project(r, i, c, &p)) continue;
    associate_at(r, c, depth, normals, p.px, p.py, &p, i);
    int ox, oy;
    if (quadrant(r, &p, &ox, &oy)) associate_at(r, c, depth, normals, ox, oy, &p, i);
  }
  if (g_ovr_sup) {  /* impose the supporting surfels of a reference run where they are legal (see above) */
    const size_t P = (size_t)r->width * r->height;
    for (size_t k = 0; k < P; ++k) {
      const uint32_t o = g_ovr_sup[k];
      if (o == r->supporting[k]) continue;
      if (o != ORC_INVALID && (g_ovr_seen[k] & 1)) { r->supporting[k] = o; g_ovr_stats[0]++; }
      else g_ovr_stats[1]++;
    }
  }
}

/* ---- stage: MergeSurfelsCUDAKernel, kernels.cu:1857-2052 (snapshot semantics) ---- */
static int merge_decide(orc_recon* r, const frame_ctx* c, const uint16_t* depth, const float* normals,
                        const proj_t* p, uint32_t i) {
  const int x = p->px, y = p->py;
  const size_t k = (size_t)y * r->width + x;
  const float measurement_depth = c->inv_depth_scaling * (float)depth[k];
  if (measurement_depth <= 0) return 0;
  const float first = r->first_depth[k];
  if (first < (1 - c->sensor_noise_factor) * measurement_depth) {
    if (first == p->l[2]) {
      /* :1887 plain store after the associate kernel finished: merge-phase
       * writers override associate-phase writers -> class-0 key */
      if (i < r->conflicting_key[k]) r->conflicting_key[k] = i;
      if (g_ovr_conf && g_ovr_conf[k] == i) g_ovr_seen[k] |= 4;
    }
    return 0;
  }
  const float occlusion_depth = (1 + c->sensor_noise_factor) * measurement_depth;
  if (p->l[2] > occlusion_depth) return 0;

  const float surfel_distance = sqrtf(p->l[0] * p->l[0] + p->l[1] * p->l[1] + p->l[2] * p->l[2]);
  const float gn[3] = {SURF(r, ORC_NORMAL_X, i), SURF(r, ORC_NORMAL_Y, i), SURF(r, ORC_NORMAL_Z, i)};
  float ln[3];
  mat_rotate(c->L, gn, ln);
  float dot_angle = (1.0f / surfel_distance) * (p->l[0] * ln[0] + p->l[1] * ln[1] + p->l[2] * ln[2]);
  if (dot_angle > 0) return 0;
  if (measurement_depth < p->l[2]) {
    const float nx = normals[2 * k], ny = normals[2 * k + 1];
    const float nz = meas_normal_z(nx, ny);
    const float d = ln[0] * nx + ln[1] * ny + ln[2] * nz;
    if (d < c->cos_normal_compat) return 0;
  }
  const float r2 = SURF(r, ORC_RADIUS_SQ, i);
  const uint32_t s = r->supporting[k];
  if (s == i || s == ORC_INVALID) return 0;       /* :1950-1953 */

  const float other_r2 = SURF(r, ORC_RADIUS_SQ, s);
  const float radius_diff = r2 / other_r2;
  const float kT = 1.2f * 1.2f;
  if (radius_diff > kT || radius_diff < 1 / kT) return 0;

  const float dx = p->g[0] - SURF(r, ORC_X, s), dy = p->g[1] - SURF(r, ORC_Y, s), dz = p->g[2] - SURF(r, ORC_Z, s);
  const float d2 = dx * dx + dy * dy + dz * dz;
  const float kDist = 0.5f * (0.25f * 0.25f);
  if (d2 > kDist * (r2 + other_r2)) return 0;

  dot_angle = gn[0] * SURF(r, ORC_NORMAL_X, s) + gn[1] * SURF(r, ORC_NORMAL_Y, s) + gn[2] * SURF(r, ORC_NORMAL_Z, s);
  if (dot_angle < 0.93969f) return 0;
  return 1;
}
static void stage_merge(orc_recon* r, const frame_ctx* c, const uint16_t* depth, const float* normals) {
  uint32_t n = 0;
  for (uint32_t i = 0; i < r->surfel_count; ++i) {
    r->merge_decision[i] = 0;
    if (!(SURF(r, ORC_RADIUS_SQ, i) >= 0)) continue;   /* :2017 */
    proj_t p;
    if (!project(r, i, c, &p)) continue;
    r->merge_decision[i] = (uint8_t)merge_decide(r, c, depth, normals, &p, i);
  }
  for (uint32_t i = 0; i < r->surfel_count; ++i) {
    if (!r->merge_decision[i]) continue;
    SET_SURF_U32(r, ORC_LAST_UPDATE_STAMP, i, 0);            /* :1987-1989 */
    SURF(r, ORC_RADIUS_SQ, i) = -1;
    SET_SURF_U32(r, ORC_COLOR, i, (SURF_U32(r, ORC_COLOR, i) & 0x00FFFFFFu) | 0x01000000u);
    ++n;
  }
  r->merge_count += n;
  r->last_n_merged = n;
  /* decode the conflicting keys for the consumers */
  const size_t P = (size_t)r->width * r->height;
  if (g_ovr_conf) {
    for (size_t k = 0; k < P; ++k) {
      const uint32_t cur = r->conflicting_key[k];
      const uint32_t cur_index = (cur == ORC_INVALID) ? ORC_INVALID : (cur & 0x7FFFFFFFu);
      const uint32_t o = g_ovr_conf[k];
      if (o == cur_index) continue;
      /* legal: a writer of the phase whose stores land last (merge phase if it wrote at all) */
      const int merge_phase = (cur != ORC_INVALID) && ((cur >> 31) == 0);
      if (cur != ORC_INVALID && o != ORC_INVALID && (g_ovr_seen[k] & (merge_phase ? 4 : 2))) {
        r->conflicting_key[k] = (merge_phase ? 0u : 0x80000000u) | o;
        g_ovr_stats[2]++;
      } else {
        g_ovr_stats[3]++;
      }
    }
  }
  for (size_t k = 0; k < P; ++k)
    r->conflicting[k] = (r->conflicting_key[k] == ORC_INVALID) ? ORC_INVALID : (r->conflicting_key[k] & 0x7FFFFFFFu);
}

/* ---- stage: BlendMeasurementsCUDA, kernels.cc:148-205, kernels.cu:563-708 ---- */
static inline float depth_sum_avg(const orc_recon* r, size_t k) {
  const float sum = (r->sum_mode == ORC_SUM_EXACT) ? q_to_float(r->depth_sums_q[k]) : r->depth_sums_f[k];
  return sum / (float)r->support_counts[k];
}
static void stage_blend(orc_recon* r, int radius, float depth_correction_factor, uint16_t* depth) {
  const int W = r->width, H = r->height;
  const size_t P = (size_t)W * H;
  const float ds = 1.0f / depth_correction_factor;            /* kernels.cc:179 */
  memset(r->distance_map, 0, P);
  memset(r->new_distance_map, 0, P);
  /* start kernel :563-615.  Reads of neighbouring depth only test "== 0", which
   * the in-kernel writes cannot produce for depth averages >= 1 unit; to make
   * the restatement independent of loop order the tests use a snapshot. */
  uint16_t* snap = (uint16_t*)malloc(P * 2);
  memcpy(snap, depth, P * 2);
  for (int y = 1; y < H - 1; ++y) {
    for (int x = 1; x < W - 1; ++x) {
      const size_t k = (size_t)y * W + x;
      if (snap[k] == 0 || r->supporting[k] == ORC_INVALID) continue;
      int measurement_border = 0, surfel_border = 0;
      for (int wy = y - 1; wy <= y + 1; ++wy)
        for (int wx = x - 1; wx <= x + 1; ++wx) {
          const size_t kk = (size_t)wy * W + wx;
          if (snap[kk] == 0) measurement_border = 1;
          else if (r->supporting[kk] == ORC_INVALID) surfel_border = 1;
        }
      if (surfel_border) {
        r->new_distance_map[k] = 1;
        const float avg = depth_sum_avg(r, k);
        r->new_deltas[k] = avg - (float)snap[k] / ds;
      }
      if (measurement_border) {
        r->distance_map[k] = 1;
        const float avg = depth_sum_avg(r, k);
        r->deltas[k] = avg - (float)snap[k] / ds;
        depth[k] = f2u16(ds * avg + 0.5f);                  /* :610 */
      } else {
        r->distance_map[k] = 255;
      }
    }
  }
  free(snap);
  /* iteration kernel :647-708, iteration = 2 .. radius-1 (kernels.cc:190) */
  const float term = 1.0f / ((float)radius - 1.0f);           /* kernels.cc:196 */
  for (int it = 2; it < radius; ++it) {
    for (int y = 1; y < H - 1; ++y) {
      for (int x = 1; x < W - 1; ++x) {
        const size_t k = (size_t)y * W + x;
        if (r->distance_map[k] == 255) {
          float delta_sum = 0; int count = 0;
          for (int wy = y - 1; wy <= y + 1; ++wy)
            for (int wx = x - 1; wx <= x + 1; ++wx) {
              const size_t kk = (size_t)wy * W + wx;
              if (r->distance_map[kk] == it - 1) { delta_sum += r->deltas[kk]; ++count; }
            }
          if (count > 0) {
            r->distance_map[k] = (uint8_t)it;
            const float avg = delta_sum / (float)count;
            r->deltas[k] = avg;
            const float f = (float)(it - 1) * term;
            depth[k] = f2u16((float)depth[k] + (ds * (1 - f) * avg + 0.5f));   /* :681 */
          }
        }
        if (depth[k] != 0 && r->supporting[k] == ORC_INVALID && r->new_distance_map[k] == 0) {
          float delta_sum = 0; int count = 0;
          for (int wy = y - 1; wy <= y + 1; ++wy)
            for (int wx = x - 1; wx <= x + 1; ++wx) {
              const size_t kk = (size_t)wy * W + wx;
              if (r->new_distance_map[kk] == it - 1) { delta_sum += r->new_deltas[kk]; ++count; }
            }
          if (count > 0) {
            r->new_distance_map[k] = (uint8_t)it;
            const float avg = delta_sum / (float)count;
            r->new_deltas[k] = avg;
            const float f = (float)(it - 1) * term;
            depth[k] = f2u16((float)depth[k] + (ds * (1 - f) * avg + 0.5f));   /* :704 */
          }
        }
      }
    }
  }
}

/* ---- stage: IntegrateMeasurementsCUDAKernel, kernels.cu:741-1142 ---- */
typedef struct { const uint16_t* depth; const float* normals; const float* radius; const uint8_t* color; } frame_in;

static void integrate_or_conflict(orc_recon* r, const frame_ctx* c, const frame_in* in,
                                  int integrate, int x, int y, const float* cam, uint32_t i) {
  /* per-thread restatement of :741-982; the block-wide __syncthreads_or exits
   * are pure optimisations */
  if (!integrate) return;
  const size_t k = (size_t)y * r->width + x;
  const float measurement_depth = c->inv_depth_scaling * (float)in->depth[k];
  if (measurement_depth <= 0) return;

  int conflicting = 0;
  const float first = r->first_depth[k];
  if (first < (1 - c->sensor_noise_factor) * measurement_depth) {
    if (first == cam[2]) {
      if (r->conflicting[k] == i) conflicting = 1;
    }
    integrate = 0;
  }
  if (!integrate && !conflicting) return;

  const float occlusion_depth = (1 + c->sensor_noise_factor) * measurement_depth;
  if (cam[2] > occlusion_depth) integrate = 0;
  if (!integrate && !conflicting) return;

  const float depth = measurement_depth;                       /* :805, same expression */
  const float lp[3] = {depth * (c->fx_inv * (float)x + c->cx_inv), depth * (c->fy_inv * (float)y + c->cy_inv), depth};
  float gp[3];
  mat_point(c->G, lp, gp);
  const float nx = in->normals[2 * k], ny = in->normals[2 * k + 1];
  const float mn[3] = {nx, ny, meas_normal_z(nx, ny)};
  float gn[3];
  mat_rotate(c->G, mn, gn);
  const uint8_t* col = &in->color[3 * k];

  if (conflicting) {                                           /* :816-868 */
    r->last_n_conflict_hits++;
    float confidence = SURF(r, ORC_CONFIDENCE, i);
    confidence -= 1;
    if (confidence <= 0) {
      r->last_n_replaced++;
      SURF(r, ORC_X, i) = gp[0]; SURF(r, ORC_Y, i) = gp[1]; SURF(r, ORC_Z, i) = gp[2];
      SURF(r, ORC_SMOOTH_X, i) = gp[0]; SURF(r, ORC_SMOOTH_Y, i) = gp[1]; SURF(r, ORC_SMOOTH_Z, i) = gp[2];
      SURF(r, ORC_NORMAL_X, i) = gn[0]; SURF(r, ORC_NORMAL_Y, i) = gn[1]; SURF(r, ORC_NORMAL_Z, i) = gn[2];
      SET_SURF_U32(r, ORC_COLOR, i, (uint32_t)col[0] | ((uint32_t)col[1] << 8) | ((uint32_t)col[2] << 16) | (1u << 24));
      SURF(r, ORC_RADIUS_SQ, i) = in->radius[k];
      for (int n = 0; n < 4; ++n) SET_SURF_U32(r, ORC_NEIGHBOR0 + n, i, ORC_INVALID);
      SURF(r, ORC_CONFIDENCE, i) = 1;
      SET_SURF_U32(r, ORC_CREATION_STAMP, i, c->frame);
      SET_SURF_U32(r, ORC_LAST_UPDATE_STAMP, i, c->frame);
    } else {
      SURF(r, ORC_CONFIDENCE, i) = confidence;
    }
  }
  if (!integrate) return;

  const float surfel_distance = sqrtf(cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2]);
  const float sn[3] = {SURF(r, ORC_NORMAL_X, i), SURF(r, ORC_NORMAL_Y, i), SURF(r, ORC_NORMAL_Z, i)};
  float ln[3];
  mat_rotate(c->L, sn, ln);
  const float dot_angle = (1.0f / surfel_distance) * (cam[0] * ln[0] + cam[1] * ln[1] + cam[2] * ln[2]);
  if (dot_angle > 0) return;
  if (measurement_depth < cam[2]) {
    const float d = sn[0] * gn[0] + sn[1] * gn[1] + sn[2] * gn[2];   /* :898-903, global frame */
    if (d < c->cos_normal_compat) integrate = 0;
  }
  if (SURF(r, ORC_RADIUS_SQ, i) < 0) integrate = 0;                 /* :907-910 */
  if (!integrate) return;

  /* :925-981 */
  uint32_t cnt = r->support_counts[k];
  if (cnt < 1) cnt = 1;
  const float weight = 1.0f / (float)cnt;
  if (SURF_U32(r, ORC_CREATION_STAMP, i) < c->frame) {
    r->last_n_integrated++;
    const float confidence = SURF(r, ORC_CONFIDENCE, i);
    SURF(r, ORC_CONFIDENCE, i) = (confidence + weight < c->max_conf) ? (confidence + weight) : c->max_conf;
    const float nf = 1.0f / (confidence + weight);
    SURF(r, ORC_X, i) = (confidence * SURF(r, ORC_X, i) + weight * gp[0]) * nf;
    SURF(r, ORC_Y, i) = (confidence * SURF(r, ORC_Y, i) + weight * gp[1]) * nf;
    SURF(r, ORC_Z, i) = (confidence * SURF(r, ORC_Z, i) + weight * gp[2]) * nf;
    const float nn[3] = {confidence * SURF(r, ORC_NORMAL_X, i) + weight * gn[0],
                         confidence * SURF(r, ORC_NORMAL_Y, i) + weight * gn[1],
                         confidence * SURF(r, ORC_NORMAL_Z, i) + weight * gn[2]};
    const float inv = 1.0f / sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
    SURF(r, ORC_NORMAL_X, i) = inv * nn[0]; SURF(r, ORC_NORMAL_Y, i) = inv * nn[1]; SURF(r, ORC_NORMAL_Z, i) = inv * nn[2];
    SURF(r, ORC_RADIUS_SQ, i) = fminf(SURF(r, ORC_RADIUS_SQ, i), in->radius[k]);
    const uint32_t oc = SURF_U32(r, ORC_COLOR, i);
    const uint32_t c0 = (uint32_t)(uint8_t)((confidence * (float)(oc & 255u) + weight * (float)col[0]) * nf + 0.5f);
    const uint32_t c1 = (uint32_t)(uint8_t)((confidence * (float)((oc >> 8) & 255u) + weight * (float)col[1]) * nf + 0.5f);
    const uint32_t c2 = (uint32_t)(uint8_t)((confidence * (float)((oc >> 16) & 255u) + weight * (float)col[2]) * nf + 0.5f);
    SET_SURF_U32(r, ORC_COLOR, i, c0 | (c1 << 8) | (c2 << 16));       /* w = 0 */
    SET_SURF_U32(r, ORC_LAST_UPDATE_STAMP, i, c->frame);
  }
}
static void stage_integrate(orc_recon* r, const frame_ctx* c, const frame_in* in) {
  r->last_n_integrated = r->last_n_replaced = r->last_n_conflict_hits = 0;
  for (uint32_t i = 0; i < r->surfel_count; ++i) {
    if (!is_active(r, i, c)) continue;
    proj_t p;
    if (!project(r, i, c, &p)) continue;
    if (SURF(r, ORC_RADIUS_SQ, i) < 0) continue;                    /* :1050-1052 */
    integrate_or_conflict(r, c, in, 1, p.px, p.py, p.l, i);
    int ox = 0, oy = 0;
    const int second = quadrant(r, &p, &ox, &oy);
    integrate_or_conflict(r, c, in, second, ox, oy, p.l, i);
  }
}

/* ---- stage: UpdateNeighborsCUDAKernel (+RemoveReplaced), kernels.cu:1197-1437 ---- */
static void stage_update_neighbors(orc_recon* r, const frame_ctx* c, const frame_in* in) {
  static const int kDX[4] = {-1, 1, 0, 0}, kDY[4] = {0, 0, -1, 1};
  const int W = r->width, H = r->height;
  for (uint32_t i = 0; i < r->surfel_count; ++i) {
    if (!is_active(r, i, c)) continue;
    const float g[3] = {SURF(r, ORC_X, i), SURF(r, ORC_Y, i), SURF(r, ORC_Z, i)};
    float cam[3];
    mat_point(c->L, g, cam);
    if (!(cam[2] > 0)) continue;
    const float u = c->fx * (cam[0] / cam[2]) + c->cx, v = c->fy * (cam[1] / cam[2]) + c->cy;
    /* :1232-1240 int truncation then 1 px border; float form keeps the cast in range */
    if (!(u >= 1.0f && v >= 1.0f && u < (float)(W - 1) && v < (float)(H - 1))) continue;
    const int x = (int)u, y = (int)v;
    const size_t k = (size_t)y * W + x;
    const float measurement_depth = c->inv_depth_scaling * (float)in->depth[k];
    const float occlusion_depth = (1 + c->sensor_noise_factor) * measurement_depth;
    if (cam[2] > occlusion_depth) continue;
    const float surfel_distance = sqrtf(cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2]);
    const float gn[3] = {SURF(r, ORC_NORMAL_X, i), SURF(r, ORC_NORMAL_Y, i), SURF(r, ORC_NORMAL_Z, i)};
    float ln[3];
    mat_rotate(c->L, gn, ln);
    const float dot_angle = (1.0f / surfel_distance) * (cam[0] * ln[0] + cam[1] * ln[1] + cam[2] * ln[2]);
    if (dot_angle > 0) continue;
    const float r2 = SURF(r, ORC_RADIUS_SQ, i);
    if (r2 < 0) continue;
    if (in->radius[k] / r2 > 1.5f * 1.5f) continue;                 /* :1287-1291 */

    float nd2[4]; uint32_t ni[4];
    for (int n = 0; n < 4; ++n) {
      ni[n] = SURF_U32(r, ORC_NEIGHBOR0 + n, i);
      if (ni[n] == ORC_INVALID) nd2[n] = INFINITY;
      else {
        const float dx = g[0] - SURF(r, ORC_X, ni[n]), dy = g[1] - SURF(r, ORC_Y, ni[n]), dz = g[2] - SURF(r, ORC_Z, ni[n]);
        nd2[n] = dx * dx + dy * dy + dz * dz;
      }
    }
    for (int d = 0; d < 4; ++d) {
      uint32_t nb = r->supporting[(size_t)(y + kDY[d]) * W + (x + kDX[d])];
      if (nb == ORC_INVALID || nb == i) continue;
      const float dx = SURF(r, ORC_X, nb) - g[0], dy = SURF(r, ORC_Y, nb) - g[1], dz = SURF(r, ORC_Z, nb) - g[2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 > c->rf2 * r2) continue;
      const float nd = gn[0] * SURF(r, ORC_NORMAL_X, nb) + gn[1] * SURF(r, ORC_NORMAL_Y, nb) + gn[2] * SURF(r, ORC_NORMAL_Z, nb);
      if (nd <= 0) continue;
      int best_n = -1; float best_d2 = -1;
      for (int n = 0; n < 4; ++n) {
        if (nb == ni[n]) { best_n = -1; break; }
        else if (nd2[n] > best_d2) { best_n = n; best_d2 = nd2[n]; }
      }
      if (best_n >= 0 && d2 < best_d2) { ni[best_n] = nb; nd2[best_n] = d2; }
    }
    for (int n = 0; n < 4; ++n) SET_SURF_U32(r, ORC_NEIGHBOR0 + n, i, ni[n]);
  }
  /* RemoveReplacedNeighbors :1420-1437 */
  for (uint32_t i = 0; i < r->surfel_count; ++i)
    for (int n = 0; n < 4; ++n) {
      const uint32_t nb = SURF_U32(r, ORC_NEIGHBOR0 + n, i);
      if (nb != ORC_INVALID && ((SURF_U32(r, ORC_COLOR, nb) >> 24) & 255u) == 1)
        SET_SURF_U32(r, ORC_NEIGHBOR0 + n, i, ORC_INVALID);
    }
}

/* ---- stage: CreateNewSurfelsCUDA, kernels.cc:37-146, kernels.cu:90-231 ---- */
static void stage_create(orc_recon* r, const frame_ctx* c, const frame_in* in) {
  static const int kDX[4] = {-1, 1, 0, 0}, kDY[4] = {0, 0, -1, 1};
  const int W = r->width, H = r->height;
  const size_t P = (size_t)W * H;
  uint32_t run = 0;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const size_t k = (size_t)y * W + x;
      const int f = x >= 1 && y >= 1 && x < W - 1 && y < H - 1 && in->depth[k] > 0 &&
                    r->supporting[k] == ORC_INVALID && r->conflicting[k] == ORC_INVALID;
      r->new_flags[k] = (uint8_t)f;
      r->new_indices[k] = run;                                       /* exclusive sum */
      run += (uint32_t)f;
    }
  (void)P;
  /* Capacity rule (reference: unchecked overflow, cc:291): clamp at the cap,
   * dropping the highest ranks. */
  const uint32_t room = r->max_surfels - r->surfel_count;
  const uint32_t created = run < room ? run : room;
  const uint32_t base = r->surfel_count;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const size_t k = (size_t)y * W + x;
      if (r->new_flags[k] != 1 || r->new_indices[k] >= created) continue;
      const uint32_t i = base + r->new_indices[k];
      const float depth = c->inv_depth_scaling * (float)in->depth[k];
      const float lp[3] = {depth * (c->fx_inv * (float)x + c->cx_inv), depth * (c->fy_inv * (float)y + c->cy_inv), depth};
      float gp[3];
      mat_point(c->G, lp, gp);
      SURF(r, ORC_X, i) = gp[0]; SURF(r, ORC_Y, i) = gp[1]; SURF(r, ORC_Z, i) = gp[2];
      const float nx = in->normals[2 * k], ny = in->normals[2 * k + 1];
      const float mn[3] = {nx, ny, meas_normal_z(nx, ny)};
      float gn[3];
      mat_rotate(c->G, mn, gn);
      SURF(r, ORC_NORMAL_X, i) = gn[0]; SURF(r, ORC_NORMAL_Y, i) = gn[1]; SURF(r, ORC_NORMAL_Z, i) = gn[2];
      const uint8_t* col = &in->color[3 * k];
      SET_SURF_U32(r, ORC_COLOR, i, (uint32_t)col[0] | ((uint32_t)col[1] << 8) | ((uint32_t)col[2] << 16));
      SURF(r, ORC_CONFIDENCE, i) = 1;
      SET_SURF_U32(r, ORC_CREATION_STAMP, i, c->frame);
      SET_SURF_U32(r, ORC_LAST_UPDATE_STAMP, i, c->frame);
      const float r2 = in->radius[k];
      SURF(r, ORC_RADIUS_SQ, i) = r2;

      float sum[3] = {0, 0, 0};
      int count_plus_1 = 1;
      for (int d = 0; d < 4; ++d) {
        const size_t kk = (size_t)(y + kDY[d]) * W + (x + kDX[d]);
        uint32_t nb = r->supporting[kk];
        if (nb != ORC_INVALID) {
          const float dx = SURF(r, ORC_X, nb) - gp[0], dy = SURF(r, ORC_Y, nb) - gp[1], dz = SURF(r, ORC_Z, nb) - gp[2];
          const float d2 = dx * dx + dy * dy + dz * dz;
          if (d2 > c->rf2 * r2) nb = ORC_INVALID;
          else {
            sum[0] = sum[0] + SURF(r, ORC_SMOOTH_X, nb);
            sum[1] = sum[1] + SURF(r, ORC_SMOOTH_Y, nb);
            sum[2] = sum[2] + SURF(r, ORC_SMOOTH_Z, nb);
            ++count_plus_1;
          }
        } else if (r->new_flags[kk] == 1 && r->new_indices[kk] < created) {
          const float od = c->inv_depth_scaling * (float)in->depth[kk];
          const float ad2 = (depth - od) * (depth - od);
          if (ad2 <= c->rf2 * r2) nb = base + r->new_indices[kk];
        }
        SET_SURF_U32(r, ORC_NEIGHBOR0 + d, i, nb);
      }
      SURF(r, ORC_SMOOTH_X, i) = (gp[0] + sum[0]) / (float)count_plus_1;   /* :227-229 */
      SURF(r, ORC_SMOOTH_Y, i) = (gp[1] + sum[1]) / (float)count_plus_1;
      SURF(r, ORC_SMOOTH_Z, i) = (gp[2] + sum[2]) / (float)count_plus_1;
    }
  r->surfel_count += created;                                         /* cc:291 */
  r->last_n_new = created;
}

/* ---- RegularizeSurfelsCUDA, kernels.cu:2099-2410 ---- */
static inline int stamp_outside_window(const orc_recon* r, uint32_t i, uint32_t frame, int window) {
  /* :2132 -- u32 subtraction, then both sides cast to int */
  return (int32_t)SURF_U32(r, ORC_LAST_UPDATE_STAMP, i) < (int32_t)(frame - (uint32_t)window);
}
static void regularize_once(orc_recon* r, uint32_t frame, float rf, float weight, int window, int copy_only) {
  const uint32_t N = r->surfel_count;
  if (N == 0) return;
  if (copy_only) {                                                   /* :2310-2327 */
    r->last_n_recent = 0; r->last_n_edges = 0;
    for (uint32_t i = 0; i < N; ++i) {
      if (stamp_outside_window(r, i, frame, window)) continue;
      r->last_n_recent++;
      SURF(r, ORC_SMOOTH_X, i) = SURF(r, ORC_X, i);
      SURF(r, ORC_SMOOTH_Y, i) = SURF(r, ORC_Y, i);
      SURF(r, ORC_SMOOTH_Z, i) = SURF(r, ORC_Z, i);
    }
    return;
  }
  const float rf2 = rf * rf;
  const int exact = (r->sum_mode == ORC_SUM_EXACT);
  /* clear :2099-2113 */
  for (uint32_t i = 0; i < N; ++i) {
    SURF(r, ORC_GRAD_X, i) = 0; SURF(r, ORC_GRAD_Y, i) = 0; SURF(r, ORC_GRAD_Z, i) = 0; SURF(r, ORC_GRAD_COUNT, i) = 0;
    r->grad_acc[4 * (size_t)i] = r->grad_acc[4 * (size_t)i + 1] = r->grad_acc[4 * (size_t)i + 2] = r->grad_acc[4 * (size_t)i + 3] = 0;
  }
  /* accumulate :2115-2195 */
  uint32_t nedges = 0;
  for (uint32_t i = 0; i < N; ++i) {
    int neighbor_count = 0;
    for (int n = 0; n < 4; ++n) {
      const uint32_t nb = SURF_U32(r, ORC_NEIGHBOR0 + n, i);
      if (nb == ORC_INVALID) continue;
      ++nedges;
      if (stamp_outside_window(r, nb, frame, window)) continue;
      ++neighbor_count;
    }
    if (neighbor_count == 0) continue;
    const float sp[3] = {SURF(r, ORC_SMOOTH_X, i), SURF(r, ORC_SMOOTH_Y, i), SURF(r, ORC_SMOOTH_Z, i)};
    const float nrm[3] = {SURF(r, ORC_NORMAL_X, i), SURF(r, ORC_NORMAL_Y, i), SURF(r, ORC_NORMAL_Z, i)};
    const float r2 = SURF(r, ORC_RADIUS_SQ, i);
    const float factor = 2 * weight / (float)neighbor_count;        /* :2153 */
    const float wk = weight / (float)neighbor_count;                /* :2182 */
    for (int n = 0; n < 4; ++n) {
      const uint32_t nb = SURF_U32(r, ORC_NEIGHBOR0 + n, i);
      if (nb == ORC_INVALID) continue;
      if (stamp_outside_window(r, nb, frame, window)) continue;
      const float t[3] = {SURF(r, ORC_SMOOTH_X, nb) - sp[0], SURF(r, ORC_SMOOTH_Y, nb) - sp[1], SURF(r, ORC_SMOOTH_Z, nb) - sp[2]};
      const float f = factor * (nrm[0] * t[0] + nrm[1] * t[1] + nrm[2] * t[2]);
      const float gt[3] = {f * nrm[0], f * nrm[1], f * nrm[2]};
      if (exact) {
        int64_t* a = &r->grad_acc[4 * (size_t)nb];
        a[0] += q22_from_float(gt[0]); a[1] += q22_from_float(gt[1]); a[2] += q22_from_float(gt[2]); a[3] += q_from_float(wk);
      } else {
        SURF(r, ORC_GRAD_X, nb) += gt[0]; SURF(r, ORC_GRAD_Y, nb) += gt[1]; SURF(r, ORC_GRAD_Z, nb) += gt[2];
        SURF(r, ORC_GRAD_COUNT, nb) += wk;
      }
      const float d2 = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
      if (d2 > rf2 * r2) SET_SURF_U32(r, ORC_NEIGHBOR0 + n, i, ORC_INVALID);   /* :2190-2192 */
    }
  }
  r->last_n_edges = nedges;
  /* step :2197-2290 */
  uint32_t nrecent = 0;
  for (uint32_t i = 0; i < N; ++i) {
    if (stamp_outside_window(r, i, frame, window)) continue;
    ++nrecent;
    const float mp[3] = {SURF(r, ORC_X, i), SURF(r, ORC_Y, i), SURF(r, ORC_Z, i)};
    const float sp[3] = {SURF(r, ORC_SMOOTH_X, i), SURF(r, ORC_SMOOTH_Y, i), SURF(r, ORC_SMOOTH_Z, i)};
    const float nrm[3] = {SURF(r, ORC_NORMAL_X, i), SURF(r, ORC_NORMAL_Y, i), SURF(r, ORC_NORMAL_Z, i)};
    float acc[4];
    if (exact) { for (int q = 0; q < 3; ++q) acc[q] = q22_to_float(r->grad_acc[4 * (size_t)i + q]); acc[3] = q_to_float(r->grad_acc[4 * (size_t)i + 3]); }
    else { acc[0] = SURF(r, ORC_GRAD_X, i); acc[1] = SURF(r, ORC_GRAD_Y, i); acc[2] = SURF(r, ORC_GRAD_Z, i); acc[3] = SURF(r, ORC_GRAD_COUNT, i); }
    float grad[3] = {2 * (sp[0] - mp[0]) + acc[0], 2 * (sp[1] - mp[1]) + acc[1], 2 * (sp[2] - mp[2]) + acc[2]};
    int neighbor_count = 0;
    float rg[3] = {0, 0, 0};
    for (int n = 0; n < 4; ++n) {
      const uint32_t nb = SURF_U32(r, ORC_NEIGHBOR0 + n, i);
      if (nb == ORC_INVALID) continue;
      ++neighbor_count;
      const float t[3] = {SURF(r, ORC_SMOOTH_X, nb) - sp[0], SURF(r, ORC_SMOOTH_Y, nb) - sp[1], SURF(r, ORC_SMOOTH_Z, nb) - sp[2]};
      const float nd = nrm[0] * t[0] + nrm[1] * t[1] + nrm[2] * t[2];
      rg[0] = rg[0] - nd * nrm[0]; rg[1] = rg[1] - nd * nrm[1]; rg[2] = rg[2] - nd * nrm[2];
    }
    if (neighbor_count > 0) {
      const float factor = 2 * weight / (float)neighbor_count;
      grad[0] = grad[0] + factor * rg[0]; grad[1] = grad[1] + factor * rg[1]; grad[2] = grad[2] + factor * rg[2];
    }
    const float wsum = 1 + weight + acc[3];                          /* :2267 */
    const float kStep = 0.5f / wsum;
    const float max_step = 1.0f * sqrtf(SURF(r, ORC_RADIUS_SQ, i));
    const float step_len = kStep * sqrtf(grad[0] * grad[0] + grad[1] * grad[1] + grad[2] * grad[2]);
    float step = kStep;
    if (step_len > max_step) step = max_step / step_len * kStep;
    SURF(r, ORC_GRAD_X, i) = sp[0] - step * grad[0];
    SURF(r, ORC_GRAD_Y, i) = sp[1] - step * grad[1];
    SURF(r, ORC_GRAD_Z, i) = sp[2] - step * grad[2];
  }
  r->last_n_recent = nrecent;
  /* update :2292-2308 */
  for (uint32_t i = 0; i < N; ++i) {
    if (stamp_outside_window(r, i, frame, window)) continue;
    SURF(r, ORC_SMOOTH_X, i) = SURF(r, ORC_GRAD_X, i);
    SURF(r, ORC_SMOOTH_Y, i) = SURF(r, ORC_GRAD_Y, i);
    SURF(r, ORC_SMOOTH_Z, i) = SURF(r, ORC_GRAD_Z, i);
  }
}

void orc_recon_regularize(orc_recon* r, uint32_t frame_index, float regularizer_weight,
                          float radius_factor_for_regularization_neighbors,
                          int regularization_frame_window_size) {
  /* cuda_surfel_reconstruction.cc:322-337 */
  regularize_once(r, frame_index, radius_factor_for_regularization_neighbors, regularizer_weight,
                  regularization_frame_window_size, 0);
}

/* ---- CUDASurfelReconstruction::Integrate, cuda_surfel_reconstruction.cc:112-320 ---- */
void orc_recon_integrate(orc_recon* r, uint32_t frame_index, float depth_scaling,
                         uint16_t* depth, const float* normals, const float* radius, const uint8_t* color,
                         const float global_T_local[12], const orc_integrate_params* p) {
  frame_ctx c;
  c.fx = r->fx; c.fy = r->fy; c.cx = r->cx; c.cy = r->cy;
  c.fx_inv = 1.0f / r->fx; c.fy_inv = 1.0f / r->fy;                    /* kernels.cc:69-74 */
  c.cx_inv = -(r->cx - 0.5f) / r->fx; c.cy_inv = -(r->cy - 0.5f) / r->fy;
  memcpy(c.G, global_T_local, sizeof(c.G));
  se3_inverse(global_T_local, c.L);
  c.inv_depth_scaling = 1.0f / depth_scaling;
  c.sensor_noise_factor = p->sensor_noise_factor;
  c.cos_normal_compat = cosf((float)(M_PI / 180.0f * p->normal_compatibility_threshold_deg));
  c.rf2 = p->radius_factor_for_regularization_neighbors * p->radius_factor_for_regularization_neighbors;
  c.max_conf = p->max_surfel_confidence;
  c.window = p->surfel_integration_active_window_size;
  c.frame = frame_index;
  frame_in in = {depth, normals, radius, color};

  stage_clear(r);
  stage_min_depth(r, &c);
  stage_associate(r, &c, depth, normals);
  stage_merge(r, &c, depth, normals);
  if (p->do_blending) stage_blend(r, p->measurement_blending_radius, c.inv_depth_scaling, depth);
  stage_integrate(r, &c, &in);
  stage_update_neighbors(r, &c, &in);
  stage_create(r, &c, &in);
  if (p->regularization_iterations_per_integration_iteration == 0) {
    regularize_once(r, frame_index, p->radius_factor_for_regularization_neighbors, p->regularizer_weight,
                    p->regularization_frame_window_size, 1);
  } else {
    for (int k = 0; k < p->regularization_iterations_per_integration_iteration; ++k)
      regularize_once(r, frame_index, p->radius_factor_for_regularization_neighbors, p->regularizer_weight,
                      p->regularization_frame_window_size, 0);
  }
}

void orc_recon_transfer_all(const orc_recon* r, float* x, float* y, float* z, float* radius_sq,
                            float* nx, float* ny, float* nz, uint32_t* last_update_stamp) {
  const size_t n = r->surfel_count, M = r->max_surfels;
  memcpy(x, r->surfels + ORC_SMOOTH_X * M, n * 4);
  memcpy(y, r->surfels + ORC_SMOOTH_Y * M, n * 4);
  memcpy(z, r->surfels + ORC_SMOOTH_Z * M, n * 4);
  memcpy(radius_sq, r->surfels + ORC_RADIUS_SQ * M, n * 4);
  memcpy(nx, r->surfels + ORC_NORMAL_X * M, n * 4);
  memcpy(ny, r->surfels + ORC_NORMAL_Y * M, n * 4);
  memcpy(nz, r->surfels + ORC_NORMAL_Z * M, n * 4);
  memcpy(last_update_stamp, r->surfels + ORC_LAST_UPDATE_STAMP * M, n * 4);
}

void orc_recon_export_vertices(const orc_recon* r, float* positions, uint8_t* colors) {
  for (uint32_t i = 0; i < r->surfel_count; ++i) {
    const int merged = SURF(r, ORC_RADIUS_SQ, i) < 0;
    positions[3 * (size_t)i + 0] = merged ? NAN : SURF(r, ORC_SMOOTH_X, i);
    positions[3 * (size_t)i + 1] = merged ? NAN : SURF(r, ORC_SMOOTH_Y, i);
    positions[3 * (size_t)i + 2] = merged ? NAN : SURF(r, ORC_SMOOTH_Z, i);
    const uint32_t c = SURF_U32(r, ORC_COLOR, i);
    colors[3 * (size_t)i + 0] = (uint8_t)(c & 255u);
    colors[3 * (size_t)i + 1] = (uint8_t)((c >> 8) & 255u);
    colors[3 * (size_t)i + 2] = (uint8_t)((c >> 16) & 255u);
  }
}

/* ---- the loop-closure hook the reference describes but does not ship (README.md:152-176, main.cc:1194-1200) ----
 * Every live surfel created at frame c < n_frames moves by the rigid correction frame_T[c] (row-major 3x4,
 * new_global_T_old_global): offset = T * (X,Y,Z) - (X,Y,Z) is added to the raw and to the smooth position
 * (README.md:160-165), the normal becomes R * normal (:166-168); where reactivate[c] != 0 the surfel's
 * LastUpdateStamp is set to frame_index so that it takes part in integration again (README.md:172-174). */
void orc_recon_deform_by_creation_frame(orc_recon* r, const float* frame_T, uint32_t n_frames,
                                        const uint8_t* reactivate, uint32_t frame_index) {
  for (uint32_t i = 0; i < r->surfel_count; ++i) {
    if (SURF(r, ORC_RADIUS_SQ, i) < 0) continue;                 /* merged zombie */
    const uint32_t c = SURF_U32(r, ORC_CREATION_STAMP, i);
    if (c >= n_frames) continue;
    const float* T = frame_T + 12 * (size_t)c;
    const float p[3] = {SURF(r, ORC_X, i), SURF(r, ORC_Y, i), SURF(r, ORC_Z, i)};
    float q[3], nn[3];
    mat_point(T, p, q);
    const float off[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
    const float n[3] = {SURF(r, ORC_NORMAL_X, i), SURF(r, ORC_NORMAL_Y, i), SURF(r, ORC_NORMAL_Z, i)};
    mat_rotate(T, n, nn);
    const int restamp = reactivate && reactivate[c] && SURF_U32(r, ORC_LAST_UPDATE_STAMP, i) != frame_index;
    /* a correction that leaves the slot as it is (identity rows) does not touch it */
    if (off[0] == 0 && off[1] == 0 && off[2] == 0 && nn[0] == n[0] && nn[1] == n[1] && nn[2] == n[2] && !restamp) continue;
    SURF(r, ORC_X, i) = p[0] + off[0];
    SURF(r, ORC_Y, i) = p[1] + off[1];
    SURF(r, ORC_Z, i) = p[2] + off[2];
    SURF(r, ORC_SMOOTH_X, i) = SURF(r, ORC_SMOOTH_X, i) + off[0];
    SURF(r, ORC_SMOOTH_Y, i) = SURF(r, ORC_SMOOTH_Y, i) + off[1];
    SURF(r, ORC_SMOOTH_Z, i) = SURF(r, ORC_SMOOTH_Z, i) + off[2];
    SURF(r, ORC_NORMAL_X, i) = nn[0];
    SURF(r, ORC_NORMAL_Y, i) = nn[1];
    SURF(r, ORC_NORMAL_Z, i) = nn[2];
    if (restamp) SET_SURF_U32(r, ORC_LAST_UPDATE_STAMP, i, frame_index);
  }
}
