/*
 * smx_oracle_depth.c -- CPU oracle, depth preprocessing stages.
 * TEST INFRASTRUCTURE ONLY (see smx_oracle.h).  Restates
 * APP/cuda_depth_processing.cu of the reference, one plain loop per kernel.
 * Build with -ffp-contract=off: every a*b+c below is a separate IEEE multiply
 * and add unless fmaf() is written explicitly.
 */
#include "smx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Deterministic expf: Cody-Waite reduction + degree-6 polynomial (the classic
 * cephes expf scheme), written with explicit fmaf so that the GPU kernel can
 * reproduce it bit for bit.  Replaces CUDA's exp()/__expf at
 * cuda_depth_processing.cu:110 (fast-math there, so no bit-parity with a real
 * CUDA run is defined anyway; max error here is < 1 ulp on [-86, 0]).        */
float orc_expf(float x) {
  if (x < -86.0f) return 0.0f;
  if (x > 88.0f) return INFINITY;
  float t = x * 1.44269504088896341f;
  float n = rintf(t);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = fmaf(p, r2, r);
  y = y + 1.0f;
  int32_t ni = (int32_t)n;
  uint32_t bits = (uint32_t)(ni + 127) << 23;
  float s;
  memcpy(&s, &bits, 4);
  return y * s;
}

/* Row range of the calling thread for the per-pixel stages below (each output pixel depends on input images only, so
 * any partition of the rows gives the same images): the all-host-cores CPU baseline of bench.py runs one thread per
 * row band, each with its own range (thread-local), over the same loops.  Default: all rows. */
static __thread int g_row_lo = 0, g_row_hi = 0x7FFFFFFF;
void orc_set_row_range(int lo, int hi) { g_row_lo = lo < 0 ? 0 : lo; g_row_hi = hi; }
#define ORC_ROWS_BEGIN(height) (g_row_lo < (height) ? g_row_lo : (height))
#define ORC_ROWS_END(height) (g_row_hi < (height) ? g_row_hi : (height))

static inline uint16_t f2u16(float v) {
  /* float -> u16 store: truncate toward zero, saturating (CUDA cvt.rzi.u16.f32
   * saturates; C leaves out-of-range undefined, so it is pinned here). */
  if (!(v > 0.0f)) return 0;
  if (v >= 65535.0f) return 65535;
  return (uint16_t)(int32_t)v;
}

static inline uint16_t rd(const uint16_t* img, int w, int h, int y, int x) {
  /* The reference reads out of bounds at the image border when the caller has
   * not zeroed it (cu:659-662, 795-797); out-of-image reads are defined as 0. */
  if (x < 0 || y < 0 || x >= w || y >= h) return 0;
  return img[(size_t)y * w + x];
}

/* cuda_depth_processing.cu:50-118 (kernel), :120-158 (host wrapper) */
void orc_bilateral_filter_and_cutoff(
    float sigma_xy, float sigma_value_factor, uint16_t value_to_ignore,
    float radius_factor, uint16_t max_depth, float depth_valid_region_radius,
    int width, int height, const uint16_t* in, uint16_t* out) {
  const int radius = (int)(radius_factor * sigma_xy + 0.5f);            /* :135 */
  const int radius_squared = radius * radius;
  const float denom_xy = 2.0f * sigma_xy * sigma_xy;                     /* :145 */
  const float region_r2 = depth_valid_region_radius * depth_valid_region_radius;
  const unsigned half_w = (unsigned)(width / 2), half_h = (unsigned)(height / 2);

  for (int y = ORC_ROWS_BEGIN(height); y < ORC_ROWS_END(height); ++y) {
    for (int x = 0; x < width; ++x) {
      uint16_t* o = &out[(size_t)y * width + x];
      /* :64-72 unsigned arithmetic, then one conversion to float */
      unsigned dxc = (unsigned)x - half_w, dyc = (unsigned)y - half_h;
      float center_distance_squared = (float)(dxc * dxc + dyc * dyc);
      if (center_distance_squared > region_r2) { *o = value_to_ignore; continue; }

      uint16_t center_value = in[(size_t)y * width + x];
      if (center_value == value_to_ignore || center_value > max_depth) {
        *o = value_to_ignore; continue;
      }
      const float adapted_sigma_value = (float)center_value * sigma_value_factor;
      const float adapted_denom_value = 2.0f * adapted_sigma_value * adapted_sigma_value;
      /* the reference is built with -use_fast_math (a / b == a * rcp(b)); here: one correctly rounded
       * reciprocal per pixel, one multiplication per tap -- same on the CPU and on the GPU */
      const float inv_denom_value = 1.0f / adapted_denom_value;

      float sum = 0, weight = 0;
      const int min_y = (y - radius) > 0 ? (y - radius) : 0;
      const int max_y = (y + radius) < (height - 1) ? (y + radius) : (height - 1);
      const int min_x = (x - radius) > 0 ? (x - radius) : 0;
      const int max_x = (x + radius) < (width - 1) ? (x + radius) : (width - 1);
      for (int sy = min_y; sy <= max_y; ++sy) {
        const int dy = sy - y;
        for (int sx = min_x; sx <= max_x; ++sx) {
          const int dx = sx - x;
          const int g2 = dx * dx + dy * dy;
          if (g2 > radius_squared) continue;
          uint16_t sample = in[(size_t)sy * width + sx];
          if (sample == value_to_ignore) continue;
          float vd = (float)((int)center_value - (int)sample);
          vd *= vd;
          float w = orc_expf((float)(-g2) / denom_xy + (-vd) * inv_denom_value);  /* :110 */
          sum += w * (float)sample;
          weight += w;
        }
      }
      *o = (weight == 0) ? value_to_ignore : f2u16(sum / weight + 0.5f);  /* :116 */
    }
  }
}

/* cuda_depth_processing.cu:168-227 (all must agree), :337-397 (counting) */
void orc_outlier_depth_map_fusion(
    int other_count, int required_count, float tolerance,
    int width, int height, const uint16_t* in,
    float fx, float fy, float cx, float cy,
    const uint16_t* const* others, const float* T,
    uint16_t* out) {
  const float max_tol = 1 + tolerance, min_tol = 1 - tolerance;          /* :255-256 */
  const float fx_inv = 1.0f / fx, fy_inv = 1.0f / fy;                    /* :259-264 */
  const float cx_pc = cx - 0.5f, cy_pc = cy - 0.5f;
  const float cx_inv = -cx_pc / fx, cy_inv = -cy_pc / fy;

  for (int y = ORC_ROWS_BEGIN(height); y < ORC_ROWS_END(height); ++y) {
    for (int x = 0; x < width; ++x) {
      uint16_t d = in[(size_t)y * width + x];
      uint16_t* o = &out[(size_t)y * width + x];
      if (d == 0) { *o = 0; continue; }
      const float fd = (float)d;
      const float rx = fd * (fx_inv * (float)x + cx_inv);
      const float ry = fd * (fy_inv * (float)y + cy_inv);
      const float rz = fd;
      int ok_count = 0; int ok = 1;
      for (int k = 0; k < other_count; ++k) {
        const float* m = T + 12 * k;
        const float ox = m[0] * rx + m[1] * ry + m[2] * rz + m[3];
        const float oy = m[4] * rx + m[5] * ry + m[6] * rz + m[7];
        const float oz = m[8] * rx + m[9] * ry + m[10] * rz + m[11];
        int good = 0;
        if (oz > 0) {
          const float u = fx * (ox / oz) + cx, v = fy * (oy / oz) + cy;
          /* :207-214: int truncation, so u in (-1,0) lands on pixel 0 */
          if (u > -1.0f && v > -1.0f && u < (float)width && v < (float)height) {
            const int px = (int)u, py = (int)v;
            const uint16_t od = others[k][(size_t)py * width + px];
            const float fod = (float)od;
            if (!(od == 0 || fod > max_tol * oz || fod < min_tol * oz)) good = 1;
          }
        }
        if (good) ++ok_count;
        else if (required_count < 0) { ok = 0; break; }
      }
      if (required_count < 0) *o = ok ? d : 0;
      else *o = (ok_count >= required_count) ? d : 0;
    }
  }
}

/* cuda_depth_processing.cu:514-538 */
void orc_erode_depth_map(int radius, int width, int height, const uint16_t* in, uint16_t* out) {
  for (int y = ORC_ROWS_BEGIN(height); y < ORC_ROWS_END(height); ++y) {
    for (int x = 0; x < width; ++x) {
      uint16_t* o = &out[(size_t)y * width + x];
      if (x < radius || y < radius || x >= width - radius || y >= height - radius) { *o = 0; continue; }
      int all_valid = 1;
      for (int dy = y - radius; dy <= y + radius; ++dy)
        for (int dx = x - radius; dx <= x + radius; ++dx)
          if (in[(size_t)dy * width + dx] == 0) all_valid = 0;
      *o = all_valid ? in[(size_t)y * width + x] : 0;
    }
  }
}

/* MedianFilterAndDensifyDepthMap, APP/main.cc:206-252 (a CPU function of the reference's main.cc; its std::sort +
 * size() / 2 indexing restated on a 9-element array) */
void orc_median_filter_and_densify(int width, int height, const uint16_t* in, uint16_t* out) {
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      uint16_t v[9];
      int n = 0;
      const int y0 = y - 1 > 0 ? y - 1 : 0, y1 = y + 1 < height - 1 ? y + 1 : height - 1;
      const int x0 = x - 1 > 0 ? x - 1 : 0, x1 = x + 1 < width - 1 ? x + 1 : width - 1;
      for (int dy = y0; dy <= y1; ++dy)
        for (int dx = x0; dx <= x1; ++dx)
          if (in[(size_t)dy * width + dx] != 0) v[n++] = in[(size_t)dy * width + dx];
      if (n >= 2) {                                               /* kMinNeighbors */
        for (int i = 1; i < n; ++i) {                             /* std::sort */
          const uint16_t t = v[i];
          int k = i;
          while (k > 0 && v[k - 1] > t) { v[k] = v[k - 1]; --k; }
          v[k] = t;
        }
        if (n % 2 == 0) {
          float sum = 0;
          for (int i = 0; i < n; ++i) sum += v[i];
          const float average = sum / n;
          const float prev_diff = fabsf(v[n / 2 - 1] - average), next_diff = fabsf(v[n / 2] - average);
          out[(size_t)y * width + x] = (prev_diff < next_diff) ? v[n / 2 - 1] : v[n / 2];
        } else {
          out[(size_t)y * width + x] = v[n / 2];
        }
      } else {
        out[(size_t)y * width + x] = in[(size_t)y * width + x];
      }
    }
}

/* Image<T>::DownscaleUsingMedianWhileExcluding, VIS/image.h:1003-1053 (u16) */
void orc_downscale_using_median_while_excluding(uint16_t value_to_ignore, int width, int height, const uint16_t* in,
                                                int out_width, int out_height, uint16_t* out) {
  uint16_t* v = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)width * height);
  for (uint32_t y = 0; y < (uint32_t)out_height; ++y)
    for (uint32_t x = 0; x < (uint32_t)out_width; ++x) {
      const uint32_t sx = ((uint32_t)width * x) / out_width, ex = ((uint32_t)width * (x + 1)) / out_width;
      const uint32_t sy = ((uint32_t)height * y) / out_height, ey = ((uint32_t)height * (y + 1)) / out_height;
      uint32_t n = 0;
      float sum = 0;
      for (uint32_t oy = sy; oy < ey; ++oy)
        for (uint32_t ox = sx; ox < ex; ++ox) {
          const uint16_t d = in[(size_t)oy * width + ox];
          if (d != value_to_ignore) { v[n++] = d; sum += d; }
        }
      uint16_t r = value_to_ignore;
      if (n > 0) {
        const float average = sum / n;
        for (uint32_t i = 1; i < n; ++i) {                         /* std::sort */
          const uint16_t t = v[i];
          uint32_t k = i;
          while (k > 0 && v[k - 1] > t) { v[k] = v[k - 1]; --k; }
          v[k] = t;
        }
        if (n % 2 == 1) r = v[n / 2];
        else r = (fabsf(average - v[n / 2 - 1]) < fabsf(average - v[n / 2])) ? v[n / 2 - 1] : v[n / 2];
      }
      out[(size_t)y * out_width + x] = r;
    }
  free(v);
}

/* cuda_depth_processing.cu:589-607 */
void orc_copy_without_border(int width, int height, const uint16_t* in, uint16_t* out) {
  for (int y = ORC_ROWS_BEGIN(height); y < ORC_ROWS_END(height); ++y)
    for (int x = 0; x < width; ++x)
      out[(size_t)y * width + x] =
          (x < 1 || y < 1 || x >= width - 1 || y >= height - 1) ? 0 : in[(size_t)y * width + x];
}

static inline void unproject(int x, int y, float depth, float fx_inv, float fy_inv,
                             float cx_inv, float cy_inv, float* p) {
  /* APP/cuda_util.cuh:61-69 */
  p[0] = depth * (fx_inv * (float)x + cx_inv);
  p[1] = depth * (fy_inv * (float)y + cy_inv);
  p[2] = depth;
}

/* cuda_depth_processing.cu:642-718 (kernel), :720-762 (host wrapper) */
void orc_compute_normals_and_drop_bad_pixels(
    float observation_angle_threshold_deg, float depth_scaling,
    float fx, float fy, float cx, float cy,
    int width, int height, const uint16_t* in, uint16_t* out, float* out_normals) {
  const float normal_dot_threshold =
      -1 * cosf((float)(M_PI / 180.f * observation_angle_threshold_deg));   /* :752 */
  const float inv_depth_scaling = 1.0f / depth_scaling;
  const float fx_inv = 1.0f / fx, fy_inv = 1.0f / fy;
  const float cx_inv = -(cx - 0.5f) / fx, cy_inv = -(cy - 0.5f) / fy;

  for (int y = ORC_ROWS_BEGIN(height); y < ORC_ROWS_END(height); ++y) {
    for (int x = 0; x < width; ++x) {
      const size_t idx = (size_t)y * width + x;
      const uint16_t c = in[idx];
      const uint16_t right = rd(in, width, height, y, x + 1), left = rd(in, width, height, y, x - 1);
      const uint16_t bottom = rd(in, width, height, y + 1, x), top = rd(in, width, height, y - 1, x);
      if (c == 0 || right == 0 || left == 0 || bottom == 0 || top == 0) {
        out[idx] = 0; out_normals[2 * idx] = 0; out_normals[2 * idx + 1] = 0; continue;
      }
      float lp[3], tp[3], rp[3], bp[3];
      unproject(x - 1, y, inv_depth_scaling * (float)left, fx_inv, fy_inv, cx_inv, cy_inv, lp);
      unproject(x, y - 1, inv_depth_scaling * (float)top, fx_inv, fy_inv, cx_inv, cy_inv, tp);
      unproject(x + 1, y, inv_depth_scaling * (float)right, fx_inv, fy_inv, cx_inv, cy_inv, rp);
      unproject(x, y + 1, inv_depth_scaling * (float)bottom, fx_inv, fy_inv, cx_inv, cy_inv, bp);
      const float a[3] = {rp[0] - lp[0], rp[1] - lp[1], rp[2] - lp[2]};   /* left_to_right */
      const float b[3] = {tp[0] - bp[0], tp[1] - bp[1], tp[2] - bp[2]};   /* bottom_to_top */
      /* CrossProduct, APP/cuda_util.cuh:71-75 */
      float n[3] = {a[1] * b[2] - b[1] * a[2], b[0] * a[2] - a[0] * b[2], a[0] * b[1] - b[0] * a[1]};
      const float length = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (!(length > 1e-6f)) {
        n[0] = 0; n[1] = 0; n[2] = -1;
      } else {
        const float inv_length = ((fy_inv < 0) ? -1.0f : 1.0f) / length;
        n[0] *= inv_length; n[1] *= inv_length; n[2] *= inv_length;
      }
      out_normals[2 * idx] = n[0];
      out_normals[2 * idx + 1] = n[1];

      float vd[3] = {fx_inv * (float)x + cx_inv, fy_inv * (float)y + cy_inv, 1};
      const float inv_dir_length = 1.0f / sqrtf(vd[0] * vd[0] + vd[1] * vd[1] + vd[2] * vd[2]);
      vd[0] = inv_dir_length * vd[0]; vd[1] = inv_dir_length * vd[1]; vd[2] = inv_dir_length * vd[2];
      const float dot = vd[0] * n[0] + vd[1] * n[1] + vd[2] * n[2];
      out[idx] = (dot >= normal_dot_threshold) ? 0 : c;
    }
  }
}

/* cuda_depth_processing.cu:765-837 (kernel), :839-883 (host wrapper) */
void orc_compute_point_radii_and_remove_isolated_pixels(
    float point_radius_extension_factor, float point_radius_clamp_factor, float depth_scaling,
    float fx, float fy, float cx, float cy,
    int width, int height, const uint16_t* in, float* out_radius, uint16_t* out) {
  const float ext2 = point_radius_extension_factor * point_radius_extension_factor;
  const float clamp_term =
      point_radius_clamp_factor * point_radius_clamp_factor * sqrtf(2) * sqrtf(2);  /* :873 */
  const float inv_depth_scaling = 1.0f / depth_scaling;
  const float fx_inv = 1.0f / fx, fy_inv = 1.0f / fy;
  const float cx_inv = -(cx - 0.5f) / fx, cy_inv = -(cy - 0.5f) / fy;

  for (int y = ORC_ROWS_BEGIN(height); y < ORC_ROWS_END(height); ++y) {
    for (int x = 0; x < width; ++x) {
      const size_t idx = (size_t)y * width + x;
      if (in[idx] == 0) { out[idx] = 0; continue; }   /* radius left untouched, :777-780 */
      const float depth = inv_depth_scaling * (float)in[idx];
      const float lp[3] = {depth * (fx_inv * (float)x + cx_inv), depth * (fy_inv * (float)y + cy_inv), depth};
      int neighbor_count = 0;
      float radius_squared = 0;
      float min_d2 = INFINITY;
      for (int dy = y - 1; dy < y + 2; ++dy) {
        for (int dx = x - 1; dx < x + 2; ++dx) {
          const float dd = inv_depth_scaling * (float)rd(in, width, height, dy, dx);
          if ((dx == x && dy == y) || dd <= 0) continue;
          ++neighbor_count;
          const float op[3] = {dd * (fx_inv * (float)dx + cx_inv), dd * (fy_inv * (float)dy + cy_inv), dd};
          const float v[3] = {op[0] - lp[0], op[1] - lp[1], op[2] - lp[2]};
          const float d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
          if (d2 > radius_squared) radius_squared = d2;
          if (d2 < min_d2) min_d2 = d2;
        }
      }
      radius_squared *= ext2;
      const float clamp = clamp_term * min_d2;
      if (radius_squared > clamp) radius_squared = clamp;
      out_radius[idx] = radius_squared;
      out[idx] = (neighbor_count < 8) ? 0 : in[idx];
    }
  }
}

/* ImagePyramid over Image<Vec3u8>::DownscaleToHalfSize (VIS/image.h:929-948, VIS/image_cache.h:203-275): one
 * halving per level, each written out like the reference does. */
void orc_color_image_pyramid(int width, int height, const uint8_t* in, int level, uint8_t* out) {
  const uint8_t* src = in;
  uint8_t* tmp = NULL;
  int w = width, h = height;
  for (int l = 0; l < level; ++l) {
    const int ow = w / 2, oh = h / 2;
    uint8_t* dst = (l == level - 1) ? out : (uint8_t*)malloc((size_t)ow * oh * 3);
    for (int y = 0; y < oh; ++y)
      for (int x = 0; x < ow; ++x)
        for (int c = 0; c < 3; ++c) {
          const uint8_t a = src[((size_t)(2 * y) * w + 2 * x) * 3 + c], b = src[((size_t)(2 * y) * w + 2 * x + 1) * 3 + c];
          const uint8_t d = src[((size_t)(2 * y + 1) * w + 2 * x) * 3 + c], e = src[((size_t)(2 * y + 1) * w + 2 * x + 1) * 3 + c];
          dst[((size_t)y * ow + x) * 3 + c] = (uint8_t)(a / 4 + b / 4 + d / 4 + e / 4);
        }
    free(tmp);
    tmp = (l == level - 1) ? NULL : dst;
    src = dst;
    w = ow; h = oh;
  }
}
