// smx_depth.hip -- depth preprocessing kernels for gfx950.
// Behaviour restated from APP/cuda_depth_processing.cu of the reference (cited
// per kernel); the implementation is new: 64-lane-wide row tiles, LDS-staged
// stencils, coalesced u16 row loads.
#include <math.h>
#include <stdlib.h>

#include <atomic>

#include "smx_common.hpp"
#include <hip/hip_ext.h>

using namespace smx;

namespace {

constexpr int kTileW = 64;   // one wavefront per image row segment
constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------
// Bilateral filter + depth cutoff.  Reference: BilateralFilteringAndDepthCutoffCUDAKernel,
// cuda_depth_processing.cu:50-118.  One pixel per lane, 32x8 pixel tiles (1200 workgroups at 640x480, so the
// 256 CUs stay evenly loaded); the tile plus its `radius` halo is staged in LDS with coalesced u16 row
// loads; the spatial exponent term -(dx^2+dy^2)/(2 sigma_xy^2) depends only on the tap offset and is
// tabulated once per workgroup (same IEEE division, so the results are unchanged).  Out-of-image taps are
// masked by coordinates, exactly like the reference's clamped loop bounds.
// (per device: one process may drive several GPUs, one object and one host thread each)
static int device_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  if (dev >= 0 && dev < 64) { const int c = cache[dev].load(std::memory_order_relaxed); if (c > 0) return c; }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
  if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

constexpr int kMaxBilateralRadius = 8;
constexpr int kBilTileW = 32, kBilTileH = 8;
__global__ void __launch_bounds__(kThreads)
k_bilateral(float denom_xy, float sigma_value_factor, int radius, int radius_squared,
            uint16_t value_to_ignore, uint16_t max_depth, float region_r2,
            Img<const uint16_t> in, Img<uint16_t> out, int tiles_x, int n_tiles) {
  __shared__ uint16_t tile[(kBilTileH + 2 * kMaxBilateralRadius) * (kBilTileW + 2 * kMaxBilateralRadius)];
  __shared__ float spatial[kMaxBilateralRadius * kMaxBilateralRadius + 1];
  const int W = out.width, H = out.height;
  const int tw = kBilTileW + 2 * radius, th = kBilTileH + 2 * radius;
  for (int g2 = threadIdx.x; g2 <= radius_squared; g2 += kThreads) spatial[g2] = (float)(-g2) / denom_xy;
  // the workgroups walk the tiles: the launch can be sized to leave room for concurrent kernels
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
  const int bx = t % tiles_x, by = t / tiles_x;
  __syncthreads();  // (the previous tile's readers are done)
  const int x0 = bx * kBilTileW - radius, y0 = by * kBilTileH - radius;
  for (int i = threadIdx.x; i < tw * th; i += kThreads) {
    const int ty = i / tw, tx = i - ty * tw;
    const int gx = x0 + tx, gy = y0 + ty;
    uint16_t v = 0;
    if (gx >= 0 && gy >= 0 && gx < W && gy < H) v = in(gy, gx);
    tile[i] = v;
  }
  __syncthreads();

  const int lx = threadIdx.x & (kBilTileW - 1), ly = threadIdx.x / kBilTileW;
  const int x = bx * kBilTileW + lx, y = by * kBilTileH + ly;
  if (x >= W || y >= H) continue;
  const unsigned half_w = (unsigned)(W / 2), half_h = (unsigned)(H / 2);
  const unsigned dxc = (unsigned)x - half_w, dyc = (unsigned)y - half_h;
  const float center_distance_squared = (float)(dxc * dxc + dyc * dyc);
  if (center_distance_squared > region_r2) { out(y, x) = value_to_ignore; continue; }
  const uint16_t center_value = tile[(ly + radius) * tw + (lx + radius)];
  if (center_value == value_to_ignore || center_value > max_depth) { out(y, x) = value_to_ignore; continue; }

  const float adapted_sigma_value = (float)center_value * sigma_value_factor;
  const float adapted_denom_value = 2.0f * adapted_sigma_value * adapted_sigma_value;
  const float inv_denom_value = 1.0f / adapted_denom_value;  // one division per pixel, not per tap (arithmetic contract)
  float sum = 0, weight = 0;
  const int min_dy = max(-radius, -y), max_dy = min(radius, H - 1 - y);
  const int min_dx = max(-radius, -x), max_dx = min(radius, W - 1 - x);
  for (int dy = min_dy; dy <= max_dy; ++dy) {
    const uint16_t* trow = &tile[(ly + radius + dy) * tw + (lx + radius)];
    for (int dx = min_dx; dx <= max_dx; ++dx) {
      const int g2 = dx * dx + dy * dy;
      if (g2 > radius_squared) continue;
      const uint16_t sample = trow[dx];
      if (sample == value_to_ignore) continue;
      float vd = (float)((int)center_value - (int)sample);
      vd *= vd;
      const float w = det_expf(spatial[g2] + (-vd) * inv_denom_value);
      sum += w * (float)sample;
      weight += w;
    }
  }
  out(y, x) = (weight == 0) ? value_to_ignore : f2u16(sum / weight + 0.5f);
  }
}

// The same filter with the radius as a compile-time constant: both tap loops unroll completely, the disc test and
// the spatial-term index become constants, the taps are branch-free (an excluded tap contributes +0 to both sums,
// which leaves them bit-identical) and the LDS reads of a row are in flight together.  The generic kernel above
// spends most of its time waiting on one LDS read and three branches per tap.  The scheduler turns the straight-line
// code into a deep software pipeline (260 VGPRs at R = 6, one wavefront per SIMD; 45 us alone).  A variant with a
// rolled row loop (100 VGPRs) runs in 31 us alone but lowers the frame rate by 2 %: the filter runs beside the
// memory-bound surfel kernels, and few fat wavefronts disturb those less than many thin ones.
// det_expf for arguments <= 0 (every tap's exponent is: both terms are non-positive), evaluated unconditionally.  Same
// arithmetic and the same bits as det_expf: the x > 88 branch cannot be taken; the final scaling y * 2^n is exact either
// as a multiplication by the constructed power of two or as v_ldexp_f32 (one instruction instead of add + shift + mul)
// wherever the result is a normal number, and below -86 the result is replaced by 0 anyway.
__device__ __forceinline__ float det_expf_nonpositive(float x) {
  float t = x * 1.44269504088896341f;
  float n = __builtin_rintf(t);
  float r = __builtin_fmaf(n, -0.693359375f, x);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = __builtin_fmaf(p, r2, r);
  y = y + 1.0f;
  const float e = __builtin_amdgcn_ldexpf(y, (int)n);
  return (x < -86.0f) ? 0.0f : e;
}


// The same filter with two taps per instruction.  Alone on the chip the kernel above runs at the VALU issue limit
// (23 instructions per tap, one wavefront per SIMD issuing back to back), so the way to make it cheaper is fewer
// instructions: gfx950 executes v_pk_{mul,add,fma}_f32 -- two independent IEEE single-precision operations per lane --
// at the rate of the scalar forms, and every step of a tap's weight except its final select is such an operation.
//   - Taps are paired so that both have the same spatial term: (-dx, dy) with (+dx, dy), and the centre column's
//     (0, -dy) with (0, +dy).  The weights are formed pairwise, the two sums still add them one at a time in the
//     reference's tap order (row by row, left to right), so every rounding is the one the scalar kernel performs.
//   - The tile is staged as floats (no conversion per tap; the two taps of a pair arrive in one ds_read2_b32, already
//     an aligned register pair).  Cells to ignore and cells outside the image hold -1e30: their squared difference
//     overflows to +inf, the exponent to -inf, and the "x < -86 -> 0" select every tap has anyway turns the weight into
//     0 -- no separate test.  (0 * -1e30 = -0 is added to a sum that is +0 or positive: no change.)
//   - round-to-nearest-integer of t = x * log2(e) as (t + 1.5 * 2^23) - 1.5 * 2^23 (two packed adds instead of
//     v_rndne_f32, exact for |t| < 2^22; a larger |t| has x < -86 and is discarded); the integer sits in the low
//     mantissa bits of the intermediate, so 2^n is one shift-add and the final scaling a packed multiplication
//     (exact: the result is a normal number whenever it is kept).
// 13.5 VALU instructions per tap instead of 23.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
// (N independent pairs advance together, one step of the evaluation at a time: with one wavefront per SIMD nothing else
// hides the latency of a dependent packed operation, and the compiler keeps the order it is given)
template <int N>
__device__ __forceinline__ void det_expf_nonpositive2(const f32x2 (&x)[N], f32x2 (&e)[N]) {
  const f32x2 magic = splat2(12582912.0f);
  f32x2 tm[N], n[N], r[N], p[N], r2[N], y[N];
#pragma unroll
  for (int k = 0; k < N; ++k) tm[k] = x[k] * splat2(1.44269504088896341f);
#pragma unroll
  for (int k = 0; k < N; ++k) tm[k] = tm[k] + magic;
#pragma unroll
  for (int k = 0; k < N; ++k) n[k] = tm[k] - magic;
#pragma unroll
  for (int k = 0; k < N; ++k) r[k] = __builtin_elementwise_fma(n[k], splat2(-0.693359375f), x[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) r[k] = __builtin_elementwise_fma(n[k], splat2(2.12194440e-4f), r[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) p[k] = __builtin_elementwise_fma(splat2(1.9875691500e-4f), r[k], splat2(1.3981999507e-3f));
#pragma unroll
  for (int k = 0; k < N; ++k) p[k] = __builtin_elementwise_fma(p[k], r[k], splat2(8.3334519073e-3f));
#pragma unroll
  for (int k = 0; k < N; ++k) p[k] = __builtin_elementwise_fma(p[k], r[k], splat2(4.1665795894e-2f));
#pragma unroll
  for (int k = 0; k < N; ++k) p[k] = __builtin_elementwise_fma(p[k], r[k], splat2(1.6666665459e-1f));
#pragma unroll
  for (int k = 0; k < N; ++k) p[k] = __builtin_elementwise_fma(p[k], r[k], splat2(5.0000001201e-1f));
#pragma unroll
  for (int k = 0; k < N; ++k) r2[k] = r[k] * r[k];
#pragma unroll
  for (int k = 0; k < N; ++k) y[k] = __builtin_elementwise_fma(p[k], r2[k], r[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) y[k] = y[k] + splat2(1.0f);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const i32x2 scale_bits = (__builtin_bit_cast(i32x2, tm[k]) << 23) + 0x3F800000;
    e[k] = y[k] * __builtin_bit_cast(f32x2, scale_bits);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    e[k].x = (x[k].x < -86.0f) ? 0.0f : e[k].x;
    e[k].y = (x[k].y < -86.0f) ? 0.0f : e[k].y;
  }
}
#ifndef SMX_BIL_PRIO
#define SMX_BIL_PRIO 0
#endif
constexpr float kBilIgnore = -1.0e30f;
constexpr int bil_half_width(int R, int dy) {
  int hw = 0;
  while ((hw + 1) * (hw + 1) + dy * dy <= R * R) ++hw;
  return hw;
}

struct OutlierParams {
  Mat34 T[8];
  Img<const uint16_t> others[8];
};
// everything the outlier test of one pixel needs besides the pixel (also an argument of the fused bilateral kernel)
struct OutlierArgs {
  int required_count; float max_tol, min_tol; float fx, fy, cx, cy; Unproj up; OutlierParams p;
};
// OutlierDepthMapFusionCUDAKernel for ONE pixel (both overloads, cu:168-227 and :337-397): depth d of pixel (x, y) -> d or 0
template <int kOthers>
__device__ __forceinline__ uint16_t outlier_pixel(uint16_t d, int x, int y, int W, int H, const OutlierArgs& a) {
  if (d == 0) return 0;
  const float fd = (float)d;
  Vec3 rp;
  rp.x = fd * (a.up.fx_inv * (float)x + a.up.cx_inv);
  rp.y = fd * (a.up.fy_inv * (float)y + a.up.cy_inv);
  rp.z = fd;
  int ok_count = 0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < kOthers; ++k) {
    const Vec3 o = mul(a.p.T[k], rp);
    bool good = false;
    if (o.z > 0) {
      const float u = a.fx * (o.x / o.z) + a.cx, v = a.fy * (o.y / o.z) + a.cy;
      if (u > -1.0f && v > -1.0f && u < (float)W && v < (float)H) {
        const int px = (int)u, py = (int)v;
        const uint16_t od = a.p.others[k](py, px);
        const float fod = (float)od;
        if (!(od == 0 || fod > a.max_tol * o.z || fod < a.min_tol * o.z)) good = true;
      }
    }
    if (good) ++ok_count;
    else ok = false;
  }
  if (a.required_count < 0) return ok ? d : (uint16_t)0;
  return (ok_count >= a.required_count) ? d : (uint16_t)0;
}

// kOthers > 0: the multi-frame outlier cull of the filtered pixel rides in the same launch (smx_bilateral_outlier_fusion):
// it is a per-pixel test of the pixel's own filtered depth against the RAW depths of the other frames, so it needs nothing
// the filter's other pixels produce -- and inside this kernel it runs on a CU the filter holds to itself anyway, instead of as a
// thin kernel of its own on the SIMDs of the memory-bound surfel kernels (profiles/r17_ab_notes.md: r17p, r17q).
template <int R, int kOthers = 0>
__global__ void __launch_bounds__(kThreads)
k_bilateral_p(float denom_xy, float sigma_value_factor, uint16_t value_to_ignore, uint16_t max_depth, float region_r2,
              Img<const uint16_t> in, Img<uint16_t> out, int tiles_x, int n_tiles, OutlierArgs oa) {
#if SMX_BIL_PRIO
  __builtin_amdgcn_s_setprio(SMX_BIL_PRIO);   // (A/B: the filter's wave priority against the surfel kernels' 1, profiles/r6_ab_notes.md section 17)
#endif
  constexpr int TW = kBilTileW + 2 * R, TH = kBilTileH + 2 * R;
  __shared__ float tile[TH * TW];
  __shared__ float spatial[R * R + 1];
  const int W = out.width, H = out.height;
  for (int g2 = threadIdx.x; g2 <= R * R; g2 += kThreads) spatial[g2] = (float)(-g2) / denom_xy;
  __syncthreads();
  float sp[R * R + 1];  // (constant indices after unrolling: the entries in use live in registers)
#pragma unroll
  for (int g2 = 0; g2 <= R * R; ++g2) sp[g2] = spatial[g2];
  const float fmax_depth = (float)max_depth;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int bx = t % tiles_x, by = t / tiles_x;
    __syncthreads();  // (the previous tile's readers are done)
    const int x0 = bx * kBilTileW - R, y0 = by * kBilTileH - R;
    for (int i = threadIdx.x; i < TW * TH; i += kThreads) {
      const int ty = i / TW, tx = i - ty * TW;
      const int gx = x0 + tx, gy = y0 + ty;
      // cells outside the image read as "ignore": the reference's clamped loop bounds never visit them
      uint16_t v = value_to_ignore;
      if (gx >= 0 && gy >= 0 && gx < W && gy < H) v = in(gy, gx);
      tile[i] = (v == value_to_ignore) ? kBilIgnore : (float)v;
    }
    __syncthreads();
    const int lx = threadIdx.x & (kBilTileW - 1), ly = threadIdx.x / kBilTileW;
    const int x = bx * kBilTileW + lx, y = by * kBilTileH + ly;
    if (x >= W || y >= H) continue;
    const unsigned half_w = (unsigned)(W / 2), half_h = (unsigned)(H / 2);
    const unsigned dxc = (unsigned)x - half_w, dyc = (unsigned)y - half_h;
    const float center_distance_squared = (float)(dxc * dxc + dyc * dyc);
    if (center_distance_squared > region_r2) { out(y, x) = value_to_ignore; continue; }
    const float* tc = &tile[(ly + R) * TW + (lx + R)];
    const float fcenter = tc[0];
    if (fcenter < 0.0f || fcenter > fmax_depth) { out(y, x) = value_to_ignore; continue; }
    const float adapted_sigma_value = fcenter * sigma_value_factor;
    const float adapted_denom_value = 2.0f * adapted_sigma_value * adapted_sigma_value;
    const f32x2 inv_denom_value = splat2(1.0f / adapted_denom_value);
    const f32x2 fcenter2 = splat2(fcenter);
    float sum = 0, weight = 0;
    float wc[R + 1], pc[R + 1];   // centre-column taps (0, +dy), formed together with (0, -dy) and added when row +dy is
#pragma unroll
    for (int dy = -R; dy <= R; ++dy) {
      const int hw = bil_half_width(R, dy);   // (a constant in every unrolled copy)
      const float* row = tc + dy * TW;
      // the row's pairs: slot 0 = the centre column's (0, dy) with (0, -dy) [rows dy <= 0], slot dx = (-dx, dy) with (+dx, dy)
      f32x2 f[R + 1], x[R + 1], w[R + 1], p[R + 1];
#pragma unroll
      for (int dx = 0; dx <= R; ++dx) {
        f[dx] = (dx == 0) ? f32x2{row[0], tc[-dy * TW]} : (dx <= hw) ? f32x2{row[-dx], row[dx]} : splat2(0.0f);
      }
#pragma unroll
      for (int dx = 0; dx <= R; ++dx) {
        f32x2 vd = fcenter2 - f[dx];   // integers below 2^16 (or the -1e30 mark): exact
        vd = vd * vd;
        x[dx] = splat2(sp[(dx <= hw ? dx * dx : 0) + dy * dy]) + (-vd) * inv_denom_value;
      }
      // (unused slots -- beyond the disc, the centre pair of rows dy > 0 -- are dead code after unrolling)
      det_expf_nonpositive2<R + 1>(x, w);
#pragma unroll
      for (int dx = 0; dx <= R; ++dx) p[dx] = w[dx] * f[dx];
      float w0, p0;
      if (dy <= 0) {
        w0 = w[0].x; p0 = p[0].x;
        wc[-dy] = w[0].y; pc[-dy] = p[0].y;
      } else {
        w0 = wc[dy]; p0 = pc[dy];
      }
      // the two sums, in the reference's order: dx = -hw .. hw
#pragma unroll
      for (int dx = R; dx >= 1; --dx) {
        if (dx > hw) continue;
        sum += p[dx].x; weight += w[dx].x;
      }
      sum += p0; weight += w0;
#pragma unroll
      for (int dx = 1; dx <= R; ++dx) {
        if (dx > hw) continue;
        sum += p[dx].y; weight += w[dx].y;
      }
    }
    const uint16_t filtered = (weight == 0) ? value_to_ignore : f2u16(sum / weight + 0.5f);
    out(y, x) = kOthers > 0 ? outlier_pixel<kOthers>(filtered, x, y, W, H, oa) : filtered;   // (fused: value_to_ignore is 0, what the cull calls "no depth")
  }
}

// ---------------------------------------------------------------------------------------------
// Multi-frame outlier cull.  Reference: OutlierDepthMapFusionCUDAKernel (both overloads),
// cuda_depth_processing.cu:168-227 and :337-397.  Matrices and image descriptors travel in the
// kernel-argument segment (SGPR-resident), the 8 scattered u16 probes hit L2 (9 x 600 KB).
template <int kOthers>
__global__ void __launch_bounds__(kThreads)
k_outlier(Img<const uint16_t> in, OutlierArgs a, Img<uint16_t> out) {
  const int x = blockIdx.x * kTileW + (threadIdx.x & (kTileW - 1));
  const int y = blockIdx.y * (kThreads / kTileW) + threadIdx.x / kTileW;
  const int W = out.width, H = out.height;
  if (x >= W || y >= H) return;
  out(y, x) = outlier_pixel<kOthers>(in(y, x), x, y, W, H, a);
}

// ---------------------------------------------------------------------------------------------
// Erosion / border copy.  Reference: ErodeDepthMapCUDAKernel<radius>, cu:514-538;
// CopyWithoutBorderCUDAKernel, cu:589-607.
__global__ void __launch_bounds__(kThreads)
k_erode(int radius, Img<const uint16_t> in, Img<uint16_t> out) {
  const int x = blockIdx.x * kTileW + (threadIdx.x & (kTileW - 1));
  const int y = blockIdx.y * (kThreads / kTileW) + threadIdx.x / kTileW;
  const int W = out.width, H = out.height;
  if (x >= W || y >= H) return;
  if (x < radius || y < radius || x >= W - radius || y >= H - radius) { out(y, x) = 0; return; }
  bool all_valid = true;
  for (int dy = y - radius; dy <= y + radius; ++dy) {
    const uint16_t* row = in.row(dy);
    for (int dx = x - radius; dx <= x + radius; ++dx)
      if (row[dx] == 0) all_valid = false;
  }
  out(y, x) = all_valid ? in(y, x) : (uint16_t)0;
}

// MedianFilterAndDensifyDepthMap, APP/main.cc:206-252 -- in the reference a CPU loop ahead of the upload ("TODO: Do
// this on the GPU", main.cc:928).  3x3 window clipped at the image border, zeros excluded; with fewer than two
// measurements the pixel is copied, with an odd count the median is taken, with an even count the middle element
// closer to the mean (the lower one on a tie of the float differences -- `prev_diff < next_diff` picks the upper).
__global__ void __launch_bounds__(kThreads)
k_median_densify(Img<const uint16_t> in, Img<uint16_t> out) {
  const int x = blockIdx.x * kTileW + (threadIdx.x & (kTileW - 1));
  const int y = blockIdx.y * (kThreads / kTileW) + threadIdx.x / kTileW;
  const int W = out.width, H = out.height;
  if (x >= W || y >= H) return;
  uint16_t v[9];
  int n = 0;
  float sum = 0;  // (integers below 2^24: exact in any order)
  for (int dy = max(0, y - 1); dy <= min(H - 1, y + 1); ++dy) {
    const uint16_t* row = in.row(dy);
    for (int dx = max(0, x - 1); dx <= min(W - 1, x + 1); ++dx) {
      const uint16_t d = row[dx];
      if (d != 0) {
        // insertion into the sorted prefix (registers: the loops are unrolled by the compiler for n <= 9)
        int k = n;
        while (k > 0 && v[k - 1] > d) { v[k] = v[k - 1]; --k; }
        v[k] = d;
        ++n;
        sum += (float)d;
      }
    }
  }
  uint16_t r = in(y, x);
  if (n >= 2) {
    if ((n & 1) == 0) {
      const float average = sum / (float)n;
      const float prev_diff = fabsf((float)v[n / 2 - 1] - average), next_diff = fabsf((float)v[n / 2] - average);
      r = (prev_diff < next_diff) ? v[n / 2 - 1] : v[n / 2];
    } else {
      r = v[n / 2];
    }
  }
  out(y, x) = r;
}

// Image<T>::DownscaleUsingMedianWhileExcluding (VIS/image.h:1003-1053), the depth half of --pyramid_level
// (APP/main.cc:941-962; a CPU loop in the reference).  One output pixel per lane; its source block
// [W x / w, W (x + 1) / w) x [H y / h, H (y + 1) / h) holds at most kMaxDownscaleBlock values.
constexpr int kMaxDownscaleBlock = 81;
__global__ void __launch_bounds__(kThreads)
k_downscale_median(uint16_t value_to_ignore, Img<const uint16_t> in, Img<uint16_t> out) {
  const int x = blockIdx.x * kTileW + (threadIdx.x & (kTileW - 1));
  const int y = blockIdx.y * (kThreads / kTileW) + threadIdx.x / kTileW;
  const unsigned w = out.width, h = out.height, W = in.width, H = in.height;
  if (x >= (int)w || y >= (int)h) return;
  const unsigned sx = (W * (unsigned)x) / w, ex = (W * ((unsigned)x + 1)) / w;
  const unsigned sy = (H * (unsigned)y) / h, ey = (H * ((unsigned)y + 1)) / h;
  uint16_t v[kMaxDownscaleBlock];
  int n = 0;
  float sum = 0;  // (exact: below 2^24)
  for (unsigned oy = sy; oy < ey; ++oy) {
    const uint16_t* row = in.row((int)oy);
    for (unsigned ox = sx; ox < ex; ++ox) {
      const uint16_t d = row[ox];
      if (d != value_to_ignore) {
        int k = n;
        while (k > 0 && v[k - 1] > d) { v[k] = v[k - 1]; --k; }
        v[k] = d;
        ++n;
        sum += (float)d;
      }
    }
  }
  uint16_t r = value_to_ignore;
  if (n > 0) {
    if (n & 1) {
      r = v[n / 2];
    } else {
      const float average = sum / (float)n;
      const uint16_t lo = v[n / 2 - 1], hi = v[n / 2];
      r = (fabsf(average - (float)lo) < fabsf(average - (float)hi)) ? lo : hi;
    }
  }
  out(y, x) = r;
}

// The colour half of --pyramid_level (APP/main.cc:973-981): ImagePyramid(frame, L) = L times
// Image<Vec3u8>::DownscaleToHalfSize (VIS/image_cache.h:203-243, VIS/image.h:929-948), whose per-channel
// a/4 + b/4 + c/4 + d/4 truncates each term (the reference's own TODO notes the missing rounding); the levels are
// nested in registers, one output pixel per lane, so that no intermediate level is stored.
struct Rgb { unsigned r, g, b; };
template <int L>
__device__ __forceinline__ Rgb pyramid_pixel(const Img<const uint8_t>& in, int x, int y) {
  if constexpr (L == 0) {
    const uint8_t* p = in.row(y) + 3 * x;
    return Rgb{p[0], p[1], p[2]};
  } else {
    const Rgb a = pyramid_pixel<L - 1>(in, 2 * x, 2 * y), b = pyramid_pixel<L - 1>(in, 2 * x + 1, 2 * y);
    const Rgb c = pyramid_pixel<L - 1>(in, 2 * x, 2 * y + 1), d = pyramid_pixel<L - 1>(in, 2 * x + 1, 2 * y + 1);
    return Rgb{a.r / 4 + b.r / 4 + c.r / 4 + d.r / 4, a.g / 4 + b.g / 4 + c.g / 4 + d.g / 4,
               a.b / 4 + b.b / 4 + c.b / 4 + d.b / 4};
  }
}
template <int L>
__global__ void __launch_bounds__(kThreads)
k_color_pyramid(Img<const uint8_t> in /* width in pixels */, Img<uint8_t> out) {
  const int x = blockIdx.x * kTileW + (threadIdx.x & (kTileW - 1));
  const int y = blockIdx.y * (kThreads / kTileW) + threadIdx.x / kTileW;
  if (x >= out.width || y >= out.height) return;
  const Rgb v = pyramid_pixel<L>(in, x, y);
  uint8_t* o = out.row(y) + 3 * x;
  o[0] = (uint8_t)v.r; o[1] = (uint8_t)v.g; o[2] = (uint8_t)v.b;
}

__global__ void __launch_bounds__(kThreads)
k_copy_without_border(Img<const uint16_t> in, Img<uint16_t> out) {
  const int x = blockIdx.x * kTileW + (threadIdx.x & (kTileW - 1));
  const int y = blockIdx.y * (kThreads / kTileW) + threadIdx.x / kTileW;
  const int W = out.width, H = out.height;
  if (x >= W || y >= H) return;
  out(y, x) = (x < 1 || y < 1 || x >= W - 1 || y >= H - 1) ? (uint16_t)0 : in(y, x);
}

__device__ __forceinline__ uint16_t rd0(const Img<const uint16_t>& img, int y, int x) {
  // out-of-image reads are defined as 0 (the reference relies on a zeroed border, cu:659-662)
  if (x < 0 || y < 0 || x >= img.width || y >= img.height) return 0;
  return img(y, x);
}

__device__ __forceinline__ Vec3 unproject(int x, int y, float depth, const Unproj& up) {
  Vec3 p;  // APP/cuda_util.cuh:61-69
  p.x = depth * (up.fx_inv * (float)x + up.cx_inv);
  p.y = depth * (up.fy_inv * (float)y + up.cy_inv);
  p.z = depth;
  return p;
}

// ---------------------------------------------------------------------------------------------
// Normals + grazing-angle drop.  Reference: ComputeNormalsAndDropBadPixelsCUDAKernel, cu:642-718.
// One pixel: the centre depth and its four neighbours in, the depth after the drop (0 = dropped) and the normal out.
__device__ __forceinline__ uint16_t normals_pixel(float normal_dot_threshold, float inv_depth_scaling, const Unproj& up,
                                                  int x, int y, uint16_t c, uint16_t left, uint16_t right, uint16_t top,
                                                  uint16_t bottom, float2& normal) {
  if (c == 0 || right == 0 || left == 0 || bottom == 0 || top == 0) {
    normal = make_float2(0, 0);
    return 0;
  }
  const Vec3 lp = unproject(x - 1, y, inv_depth_scaling * (float)left, up);
  const Vec3 tp = unproject(x, y - 1, inv_depth_scaling * (float)top, up);
  const Vec3 rp = unproject(x + 1, y, inv_depth_scaling * (float)right, up);
  const Vec3 bp = unproject(x, y + 1, inv_depth_scaling * (float)bottom, up);
  const Vec3 a = {rp.x - lp.x, rp.y - lp.y, rp.z - lp.z};
  const Vec3 b = {tp.x - bp.x, tp.y - bp.y, tp.z - bp.z};
  Vec3 n = {a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y};
  const float length = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
  if (!(length > 1e-6f)) {
    n.x = 0; n.y = 0; n.z = -1;
  } else {
    const float inv_length = ((up.fy_inv < 0) ? -1.0f : 1.0f) / length;
    n.x *= inv_length; n.y *= inv_length; n.z *= inv_length;
  }
  normal = make_float2(n.x, n.y);
  Vec3 vd = {up.fx_inv * (float)x + up.cx_inv, up.fy_inv * (float)y + up.cy_inv, 1.0f};
  const float inv_dir_length = 1.0f / sqrtf(vd.x * vd.x + vd.y * vd.y + vd.z * vd.z);
  vd.x = inv_dir_length * vd.x; vd.y = inv_dir_length * vd.y; vd.z = inv_dir_length * vd.z;
  const float dot = vd.x * n.x + vd.y * n.y + vd.z * n.z;
  return (dot >= normal_dot_threshold) ? (uint16_t)0 : c;
}

__global__ void __launch_bounds__(kThreads)
k_normals(float normal_dot_threshold, float inv_depth_scaling, Unproj up,
          Img<const uint16_t> in, Img<uint16_t> out, Img<float2> out_normals) {
  const int x = blockIdx.x * kTileW + (threadIdx.x & (kTileW - 1));
  const int y = blockIdx.y * (kThreads / kTileW) + threadIdx.x / kTileW;
  const int W = in.width, H = in.height;
  if (x >= W || y >= H) return;
  float2 n;
  out(y, x) = normals_pixel(normal_dot_threshold, inv_depth_scaling, up, x, y, in(y, x), rd0(in, y, x - 1), rd0(in, y, x + 1),
                            rd0(in, y - 1, x), rd0(in, y + 1, x), n);
  out_normals(y, x) = n;
}

// ---------------------------------------------------------------------------------------------
// Point radii + isolated-pixel removal.  Reference:
// ComputePointRadiiAndRemoveIsolatedPixelsCUDAKernel, cu:765-837.
// One pixel with a non-zero centre: d[3][3] = the 3x3 depths around it (0 outside the image); returns the output depth.
__device__ __forceinline__ uint16_t radii_pixel(float ext2, float clamp_term, float inv_depth_scaling, const Unproj& up,
                                                int x, int y, const uint16_t d[3][3], float& radius_squared_out) {
  const uint16_t c = d[1][1];
  const float depth = inv_depth_scaling * (float)c;
  const Vec3 lp = {depth * (up.fx_inv * (float)x + up.cx_inv), depth * (up.fy_inv * (float)y + up.cy_inv), depth};
  int neighbor_count = 0;
  float radius_squared = 0;
  float min_d2 = __builtin_inff();
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int dx = x - 1 + i, dy = y - 1 + j;
      const float dd = inv_depth_scaling * (float)d[j][i];
      if ((i == 1 && j == 1) || dd <= 0) continue;
      ++neighbor_count;
      const Vec3 op = {dd * (up.fx_inv * (float)dx + up.cx_inv), dd * (up.fy_inv * (float)dy + up.cy_inv), dd};
      const Vec3 v = {op.x - lp.x, op.y - lp.y, op.z - lp.z};
      const float d2 = v.x * v.x + v.y * v.y + v.z * v.z;
      if (d2 > radius_squared) radius_squared = d2;
      if (d2 < min_d2) min_d2 = d2;
    }
  }
  radius_squared *= ext2;
  const float clamp = clamp_term * min_d2;
  if (radius_squared > clamp) radius_squared = clamp;
  radius_squared_out = radius_squared;
  return (neighbor_count < 8) ? (uint16_t)0 : c;
}

__global__ void __launch_bounds__(kThreads)
k_radii(float ext2, float clamp_term, float inv_depth_scaling, Unproj up,
        Img<const uint16_t> in, Img<float> out_radius, Img<uint16_t> out) {
  const int x = blockIdx.x * kTileW + (threadIdx.x & (kTileW - 1));
  const int y = blockIdx.y * (kThreads / kTileW) + threadIdx.x / kTileW;
  const int W = in.width, H = in.height;
  if (x >= W || y >= H) return;
  if (in(y, x) == 0) { out(y, x) = 0; return; }   // (radius left untouched, cu:777-780)
  uint16_t d[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) d[j][i] = rd0(in, y - 1 + j, x - 1 + i);
  float r2;
  out(y, x) = radii_pixel(ext2, clamp_term, inv_depth_scaling, up, x, y, d, r2);
  out_radius(y, x) = r2;
}

// ---------------------------------------------------------------------------------------------
// Erosion (or the border copy) + normals + radii of one frame in ONE launch (the three calls of APP/main.cc:1128-1191 that
// always follow each other): each stage's zeroing feeds the next stage's validity tests, so the intermediate depth
// images of a 32x8 pixel tile and their shrinking halos (erosion radius + 2, 2, 1) live in LDS; only the final depth,
// the normals and the radii are written.  Same per-pixel routines as the separate kernels: identical images.
constexpr int kFuseTW = 32, kFuseTH = 8, kFuseMaxR = 3;
__global__ void __launch_bounds__(kThreads)
k_erode_normals_radii(int radius /* 0 = copy without border */, float normal_dot_threshold, float ext2, float clamp_term,
                      float inv_depth_scaling, Unproj up, Img<const uint16_t> in, Img<uint16_t> out_depth,
                      Img<float2> out_normals, Img<float> out_radius) {
  constexpr int kIW = kFuseTW + 2 * (kFuseMaxR + 2), kIH = kFuseTH + 2 * (kFuseMaxR + 2);
  constexpr int kEW = kFuseTW + 4, kEH = kFuseTH + 4, kNW = kFuseTW + 2, kNH = kFuseTH + 2;
  __shared__ uint16_t ti[kIH * kIW];   // input with halo radius + 2
  __shared__ uint16_t te[kEH * kEW];   // after erosion / border copy, halo 2
  __shared__ uint16_t tn[kNH * kNW];   // after the normals stage, halo 1
  const int W = in.width, H = in.height;
  const int bx = blockIdx.x * kFuseTW, by = blockIdx.y * kFuseTH;
  const int hi = radius + 2, iw = kFuseTW + 2 * hi, ih = kFuseTH + 2 * hi;
  for (int k = threadIdx.x; k < iw * ih; k += kThreads) {
    const int ty = k / iw, tx = k - ty * iw;
    const int gx = bx - hi + tx, gy = by - hi + ty;
    ti[ty * kIW + tx] = (gx >= 0 && gy >= 0 && gx < W && gy < H) ? in(gy, gx) : (uint16_t)0;
  }
  __syncthreads();
  // erosion, cu:514-538 (radius >= 1) / border copy, cu:589-607
  for (int k = threadIdx.x; k < kEW * kEH; k += kThreads) {
    const int ty = k / kEW, tx = k - ty * kEW;
    const int gx = bx - 2 + tx, gy = by - 2 + ty;
    uint16_t v = 0;
    if (gx >= 0 && gy >= 0 && gx < W && gy < H) {
      const uint16_t* c = &ti[(ty + radius) * kIW + (tx + radius)];   // the same pixel in the input tile
      if (radius == 0) {
        v = (gx < 1 || gy < 1 || gx >= W - 1 || gy >= H - 1) ? (uint16_t)0 : c[0];
      } else if (!(gx < radius || gy < radius || gx >= W - radius || gy >= H - radius)) {
        bool all_valid = true;
        for (int dy = -radius; dy <= radius; ++dy)
          for (int dx = -radius; dx <= radius; ++dx)
            if (c[dy * kIW + dx] == 0) all_valid = false;
        v = all_valid ? c[0] : (uint16_t)0;
      }
    }
    te[k] = v;
  }
  __syncthreads();
  // normals, cu:642-718 (pixels outside the image stay 0: the next stage reads them as "no measurement")
  for (int k = threadIdx.x; k < kNW * kNH; k += kThreads) {
    const int ty = k / kNW, tx = k - ty * kNW;
    const int gx = bx - 1 + tx, gy = by - 1 + ty;
    uint16_t v = 0;
    if (gx >= 0 && gy >= 0 && gx < W && gy < H) {
      const uint16_t* c = &te[(ty + 1) * kEW + (tx + 1)];
      float2 n;
      v = normals_pixel(normal_dot_threshold, inv_depth_scaling, up, gx, gy, c[0], c[-1], c[1], c[-kEW], c[kEW], n);
      if (tx >= 1 && ty >= 1 && tx <= kFuseTW && ty <= kFuseTH) out_normals(gy, gx) = n;
    }
    tn[k] = v;
  }
  __syncthreads();
  // radii + isolated-pixel removal, cu:765-837
  const int lx = threadIdx.x & (kFuseTW - 1), ly = threadIdx.x / kFuseTW;
  const int x = bx + lx, y = by + ly;
  if (x >= W || y >= H) return;
  const uint16_t* c = &tn[(ly + 1) * kNW + (lx + 1)];
  if (c[0] == 0) { out_depth(y, x) = 0; return; }   // (radius left untouched, cu:777-780)
  uint16_t d[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) d[j][i] = c[(j - 1) * kNW + (i - 1)];
  float r2;
  out_depth(y, x) = radii_pixel(ext2, clamp_term, inv_depth_scaling, up, x, y, d, r2);
  out_radius(y, x) = r2;
}

inline dim3 grid_rows(int W, int H) { return dim3(div_up(W, kTileW), div_up(H, kThreads / kTileW), 1); }

}  // namespace

extern "C" {

namespace {
int fill_outlier_args(OutlierArgs* oa, int32_t other_count, int32_t required_count, float tolerance, float fx, float fy,
                      float cx, float cy, const smx_buffer_desc* other_depths, const float* others_TR_reference,
                      const smx_buffer_desc* output_depth) {
  SMX_CHECK_ARG(other_depths && others_TR_reference);
  SMX_CHECK_ARG(other_count == 2 || other_count == 4 || other_count == 6 || other_count == 8);  // main.cc:1076-1086
  memset(oa, 0, sizeof(*oa));
  for (int i = 0; i < other_count; ++i) {
    memcpy(oa->p.T[i].m, others_TR_reference + 12 * i, sizeof(float) * 12);
    oa->p.others[i] = as_img<const uint16_t>(&other_depths[i]);
    SMX_CHECK_ARG(other_depths[i].width == output_depth->width && other_depths[i].height == output_depth->height);
  }
  oa->required_count = required_count;
  oa->max_tol = 1 + tolerance; oa->min_tol = 1 - tolerance;                        // cu:255-256
  oa->fx = fx; oa->fy = fy; oa->cx = cx; oa->cy = cy;
  oa->up = make_unproj(fx, fy, cx, cy);
  return SMX_OK;
}

// (other_count = 0: the filter alone)
int launch_bilateral(smx_stream s, float sigma_xy, float sigma_value_factor, uint16_t value_to_ignore, float radius_factor,
                     uint16_t max_depth, float depth_valid_region_radius, const smx_buffer_desc* input_depth,
                     const smx_buffer_desc* output_depth, int other_count, const OutlierArgs& oa) {
  SMX_CHECK_ARG(input_depth && output_depth);
  SMX_CHECK_ARG(input_depth->width == output_depth->width && input_depth->height == output_depth->height);
  const int radius = (int)(radius_factor * sigma_xy + 0.5f);                       // cu:135
  SMX_CHECK_ARG(radius >= 0 && radius <= kMaxBilateralRadius);
  const int tiles_x = div_up(output_depth->width, kBilTileW), n_tiles = tiles_x * div_up(output_depth->height, kBilTileH);
  // At most eight workgroups per CU in the grid (one per tile at 640x480, four tiles per workgroup at 1280x960).  Two
  // of these 256-register workgroups fill a CU's register files, and the filter runs beside the surfel kernels
  // (preprocessing of the next frames overlaps Integrate): with few long-lived workgroups (2 per CU, rounds 1-2) a CU
  // stays closed to everything else for the whole filter; short-lived ones hand it back between tiles
  // (profiles/r09g_bilateral_grid.txt: C2 4557 -> 4755 frames/s).
  const int max_blocks = 8 * device_cu_count();
  const dim3 grid(n_tiles < max_blocks ? n_tiles : max_blocks);
  const float denom_xy = 2.0f * sigma_xy * sigma_xy, region_r2 = depth_valid_region_radius * depth_valid_region_radius;
  const Img<const uint16_t> src = as_img<const uint16_t>(input_depth);
  const Img<uint16_t> dst = as_img<uint16_t>(output_depth);
#define SMX_BILATERAL_N(R, N)                                                                                        \
  hipLaunchKernelGGL((k_bilateral_p<R, N>), grid, dim3(kThreads), 0, (hipStream_t)s, denom_xy, sigma_value_factor,   \
                     value_to_ignore, max_depth, region_r2, src, dst, tiles_x, n_tiles, oa)
#define SMX_BILATERAL(R)                                                                                             \
  case R:                                                                                                            \
    if (other_count == 0) SMX_BILATERAL_N(R, 0);                                                                     \
    else SMX_BILATERAL_N(R, 8);                                                                                      \
    break
  // (the fused form exists for eight other frames, the reference's default; smx_bilateral_outlier_fusion takes the two
  // launches for every other count)
  switch (radius) {
    SMX_BILATERAL(1); SMX_BILATERAL(2); SMX_BILATERAL(3); SMX_BILATERAL(4);
    SMX_BILATERAL(5); SMX_BILATERAL(6); SMX_BILATERAL(7); SMX_BILATERAL(8);
    default:  // radius 0: the generic kernel
      SMX_CHECK_ARG(other_count == 0);
      hipLaunchKernelGGL(k_bilateral, grid, dim3(kThreads), 0, (hipStream_t)s, denom_xy, sigma_value_factor, radius,
                         radius * radius, value_to_ignore, max_depth, region_r2, src, dst, tiles_x, n_tiles);
  }
#undef SMX_BILATERAL
#undef SMX_BILATERAL_N
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}
}  // namespace

int smx_bilateral_filtering_and_depth_cutoff(
    smx_stream s, float sigma_xy, float sigma_value_factor, uint16_t value_to_ignore,
    float radius_factor, uint16_t max_depth, float depth_valid_region_radius,
    const smx_buffer_desc* input_depth, const smx_buffer_desc* output_depth) {
  OutlierArgs oa;
  memset(&oa, 0, sizeof(oa));
  return launch_bilateral(s, sigma_xy, sigma_value_factor, value_to_ignore, radius_factor, max_depth, depth_valid_region_radius,
                          input_depth, output_depth, 0, oa);
}

int smx_outlier_depth_map_fusion(
    smx_stream s, int32_t other_count, int32_t required_count, float tolerance,
    const smx_buffer_desc* input_depth, float fx, float fy, float cx, float cy,
    const smx_buffer_desc* other_depths, const float* others_TR_reference,
    const smx_buffer_desc* output_depth) {
  SMX_CHECK_ARG(input_depth && output_depth);
  OutlierArgs oa;
  { const int rc = fill_outlier_args(&oa, other_count, required_count, tolerance, fx, fy, cx, cy, other_depths, others_TR_reference, output_depth);
    if (rc != SMX_OK) return rc; }
  dim3 grid = grid_rows(output_depth->width, output_depth->height);
  hipStream_t st = (hipStream_t)s;
#define SMX_OUTLIER(N) hipLaunchKernelGGL(k_outlier<N>, grid, dim3(kThreads), 0, st, as_img<const uint16_t>(input_depth), oa, as_img<uint16_t>(output_depth))
  switch (other_count) {
    case 2: SMX_OUTLIER(2); break;
    case 4: SMX_OUTLIER(4); break;
    case 6: SMX_OUTLIER(6); break;
    default: SMX_OUTLIER(8); break;
  }
#undef SMX_OUTLIER
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_bilateral_outlier_fusion(
    smx_stream s, float sigma_xy, float sigma_value_factor, float radius_factor, uint16_t max_depth,
    float depth_valid_region_radius, const smx_buffer_desc* input_depth,
    int32_t other_count, int32_t required_count, float tolerance, float fx, float fy, float cx, float cy,
    const smx_buffer_desc* other_depths, const float* others_TR_reference,
    const smx_buffer_desc* scratch_depth, const smx_buffer_desc* output_depth) {
  SMX_CHECK_ARG(input_depth && output_depth);
  OutlierArgs oa;
  { const int rc = fill_outlier_args(&oa, other_count, required_count, tolerance, fx, fy, cx, cy, other_depths, others_TR_reference, output_depth);
    if (rc != SMX_OK) return rc; }
  const int radius = (int)(radius_factor * sigma_xy + 0.5f);
  if (other_count == 8 && radius >= 1 && radius <= kMaxBilateralRadius)   // one launch
    return launch_bilateral(s, sigma_xy, sigma_value_factor, /*value_to_ignore*/ 0, radius_factor, max_depth, depth_valid_region_radius,
                            input_depth, output_depth, 8, oa);
  // any other shape: the two launches, through the scratch image
  SMX_CHECK_ARG(scratch_depth != nullptr);
  { const int rc = smx_bilateral_filtering_and_depth_cutoff(s, sigma_xy, sigma_value_factor, 0, radius_factor, max_depth,
                                                            depth_valid_region_radius, input_depth, scratch_depth);
    if (rc != SMX_OK) return rc; }
  return smx_outlier_depth_map_fusion(s, other_count, required_count, tolerance, scratch_depth, fx, fy, cx, cy, other_depths,
                                      others_TR_reference, output_depth);
}

int smx_erode_depth_map(smx_stream s, int32_t radius, const smx_buffer_desc* input_depth,
                        const smx_buffer_desc* output_depth) {
  SMX_CHECK_ARG(input_depth && output_depth);
  if (radius < 1 || radius > 3) {                                                  // cu:572-574
    set_error("radius value of %d is not supported.", radius);
    return SMX_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(k_erode, grid_rows(output_depth->width, output_depth->height), dim3(kThreads), 0,
                     (hipStream_t)s, radius, as_img<const uint16_t>(input_depth), as_img<uint16_t>(output_depth));
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_median_filter_and_densify_depth_map(smx_stream s, const smx_buffer_desc* input_depth,
                                            const smx_buffer_desc* output_depth) {
  SMX_CHECK_ARG(input_depth && output_depth && input_depth->address != output_depth->address);
  SMX_CHECK_ARG(input_depth->width == output_depth->width && input_depth->height == output_depth->height);
  hipLaunchKernelGGL(k_median_densify, grid_rows(output_depth->width, output_depth->height), dim3(kThreads), 0,
                     (hipStream_t)s, as_img<const uint16_t>(input_depth), as_img<uint16_t>(output_depth));
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_downscale_using_median_while_excluding(smx_stream s, uint16_t value_to_ignore, const smx_buffer_desc* input,
                                               const smx_buffer_desc* output) {
  SMX_CHECK_ARG(input && output && output->width > 0 && output->height > 0);
  SMX_CHECK_ARG(output->width <= input->width && output->height <= input->height);
  const int bw = (input->width + output->width - 1) / output->width + 1, bh = (input->height + output->height - 1) / output->height + 1;
  if ((bw - 1) * (bh - 1) > kMaxDownscaleBlock) {
    set_error("downscale blocks of up to %d x %d pixels are not supported (at most %d values)", bw - 1, bh - 1, kMaxDownscaleBlock);
    return SMX_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(k_downscale_median, grid_rows(output->width, output->height), dim3(kThreads), 0, (hipStream_t)s,
                     value_to_ignore, as_img<const uint16_t>(input), as_img<uint16_t>(output));
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_color_image_pyramid(smx_stream s, int32_t pyramid_level, const smx_buffer_desc* input,
                            const smx_buffer_desc* output) {
  SMX_CHECK_ARG(input && output && pyramid_level >= 1 && pyramid_level <= 4);
  // DownscaleToHalfSize CHECKs even sizes at every level (VIS/image.h:930-931)
  SMX_CHECK_ARG(input->width % (1 << pyramid_level) == 0 && input->height % (1 << pyramid_level) == 0);
  SMX_CHECK_ARG(output->width == (input->width >> pyramid_level) && output->height == (input->height >> pyramid_level));
  const dim3 grid = grid_rows(output->width, output->height);
  const Img<const uint8_t> in = as_img<const uint8_t>(input);
  const Img<uint8_t> out = as_img<uint8_t>(output);
  hipStream_t st = (hipStream_t)s;
  switch (pyramid_level) {
    case 1: hipLaunchKernelGGL(k_color_pyramid<1>, grid, dim3(kThreads), 0, st, in, out); break;
    case 2: hipLaunchKernelGGL(k_color_pyramid<2>, grid, dim3(kThreads), 0, st, in, out); break;
    case 3: hipLaunchKernelGGL(k_color_pyramid<3>, grid, dim3(kThreads), 0, st, in, out); break;
    default: hipLaunchKernelGGL(k_color_pyramid<4>, grid, dim3(kThreads), 0, st, in, out); break;
  }
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_copy_without_border(smx_stream s, const smx_buffer_desc* input_depth, const smx_buffer_desc* output_depth) {
  SMX_CHECK_ARG(input_depth && output_depth);
  hipLaunchKernelGGL(k_copy_without_border, grid_rows(output_depth->width, output_depth->height), dim3(kThreads), 0,
                     (hipStream_t)s, as_img<const uint16_t>(input_depth), as_img<uint16_t>(output_depth));
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_compute_normals_and_drop_bad_pixels(
    smx_stream s, float observation_angle_threshold_deg, float depth_scaling,
    float fx, float fy, float cx, float cy,
    const smx_buffer_desc* in_depth, const smx_buffer_desc* out_depth, const smx_buffer_desc* out_normals) {
  SMX_CHECK_ARG(in_depth && out_depth && out_normals);
  const float thr = -1 * cosf((float)(M_PI / 180.f * observation_angle_threshold_deg));  // cu:752
  hipLaunchKernelGGL(k_normals, grid_rows(in_depth->width, in_depth->height), dim3(kThreads), 0, (hipStream_t)s,
                     thr, 1.0f / depth_scaling, make_unproj(fx, fy, cx, cy), as_img<const uint16_t>(in_depth),
                     as_img<uint16_t>(out_depth), as_img<float2>(out_normals));
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_compute_point_radii_and_remove_isolated_pixels(
    smx_stream s, float point_radius_extension_factor, float point_radius_clamp_factor,
    float depth_scaling, float fx, float fy, float cx, float cy,
    const smx_buffer_desc* depth_buffer, const smx_buffer_desc* radius_buffer, const smx_buffer_desc* out_depth) {
  SMX_CHECK_ARG(depth_buffer && radius_buffer && out_depth);
  const float ext2 = point_radius_extension_factor * point_radius_extension_factor;
  const float clamp_term = point_radius_clamp_factor * point_radius_clamp_factor * sqrtf(2) * sqrtf(2);  // cu:873
  hipLaunchKernelGGL(k_radii, grid_rows(depth_buffer->width, depth_buffer->height), dim3(kThreads), 0,
                     (hipStream_t)s, ext2, clamp_term, 1.0f / depth_scaling, make_unproj(fx, fy, cx, cy),
                     as_img<const uint16_t>(depth_buffer), as_img<float>(radius_buffer), as_img<uint16_t>(out_depth));
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_erode_normals_radii(smx_stream s, int32_t erosion_radius, float observation_angle_threshold_deg,
                            float point_radius_extension_factor, float point_radius_clamp_factor, float depth_scaling,
                            float fx, float fy, float cx, float cy, const smx_buffer_desc* in_depth,
                            const smx_buffer_desc* out_depth, const smx_buffer_desc* out_normals,
                            const smx_buffer_desc* radius_buffer) {
  return smx_erode_normals_radii_signal(s, erosion_radius, observation_angle_threshold_deg, point_radius_extension_factor,
                                        point_radius_clamp_factor, depth_scaling, fx, fy, cx, cy, in_depth, out_depth, out_normals,
                                        radius_buffer, nullptr);
}

int smx_erode_normals_radii_signal(smx_stream s, int32_t erosion_radius, float observation_angle_threshold_deg,
                                   float point_radius_extension_factor, float point_radius_clamp_factor, float depth_scaling,
                                   float fx, float fy, float cx, float cy, const smx_buffer_desc* in_depth,
                                   const smx_buffer_desc* out_depth, const smx_buffer_desc* out_normals,
                                   const smx_buffer_desc* radius_buffer, smx_event done) {
  SMX_CHECK_ARG(in_depth && out_depth && out_normals && radius_buffer && in_depth->address != out_depth->address);
  SMX_CHECK_ARG(in_depth->width == out_depth->width && in_depth->height == out_depth->height);
  SMX_CHECK_ARG(out_normals->width == in_depth->width && out_normals->height == in_depth->height);
  SMX_CHECK_ARG(radius_buffer->width == in_depth->width && radius_buffer->height == in_depth->height);
  if (erosion_radius < 0 || erosion_radius > kFuseMaxR) {                          // cu:572-574
    set_error("radius value of %d is not supported.", erosion_radius);
    return SMX_ERR_UNSUPPORTED;
  }
  const float thr = -1 * cosf((float)(M_PI / 180.f * observation_angle_threshold_deg));  // cu:752
  const float ext2 = point_radius_extension_factor * point_radius_extension_factor;
  const float clamp_term = point_radius_clamp_factor * point_radius_clamp_factor * sqrtf(2) * sqrtf(2);  // cu:873
  const dim3 grid(div_up(in_depth->width, kFuseTW), div_up(in_depth->height, kFuseTH));
  // (done: the launch's own completion event -- what a record behind it would mark, without a packet of its own)
  hipExtLaunchKernelGGL(k_erode_normals_radii, grid, dim3(kThreads), 0, (hipStream_t)s, nullptr, (hipEvent_t)done, 0, (int)erosion_radius, thr, ext2,
                     clamp_term, 1.0f / depth_scaling, make_unproj(fx, fy, cx, cy), as_img<const uint16_t>(in_depth),
                     as_img<uint16_t>(out_depth), as_img<float2>(out_normals), as_img<float>(radius_buffer));
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

}  // extern "C"
