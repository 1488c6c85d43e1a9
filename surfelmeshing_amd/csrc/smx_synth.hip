// smx_synth.hip -- synthetic RGB-D frame renderer (benchmark input generator, not part of the
// reference's interface).  Renders the analytic room of SURVEY.md 8(d) / surfelmeshing_amd/synth.py
// (closed 6 x 6 x 3 m box with sinusoidal relief) straight into device buffers, so that long
// benchmark streams do not have to be ray-cast on the host.  Counter-based RNG: the frame is a pure
// function of (seed, frame_index, pose), independent of launch geometry.
#include <math.h>

#include "smx_common.hpp"

using namespace smx;

namespace {

__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
__device__ __forceinline__ float u01(uint32_t h) { return ((h >> 8) + 0.5f) * (1.0f / 16777216.0f); }

struct SynthArgs {
  Mat34 G;  // global_T_frame
  float fx, fy, cx, cy;
  uint32_t seed, frame;
  float depth_scaling, noise_sigma, dropout;
};

__global__ void __launch_bounds__(256)
k_synth_room(SynthArgs a, Img<uint16_t> depth, Img<uchar3> color) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int W = depth.width, H = depth.height;
  if (x >= W || y >= H) return;
  const float half[3] = {3.0f, 1.5f, 3.0f};
  const float dc[3] = {((float)x + 0.5f - a.cx) / a.fx, ((float)y + 0.5f - a.cy) / a.fy, 1.0f};
  float d[3], o[3];
  for (int r = 0; r < 3; ++r) {
    d[r] = a.G.m[4 * r] * dc[0] + a.G.m[4 * r + 1] * dc[1] + a.G.m[4 * r + 2] * dc[2];
    o[r] = a.G.m[4 * r + 3];
  }
  float best = __builtin_inff();
  for (int axis = 0; axis < 3; ++axis) {
    const int ia = axis == 0 ? 1 : 0, ib = axis == 2 ? 1 : 2;
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      const float dn = d[axis] * (float)sgn;
      if (!(dn > 1e-9f)) continue;
      float t = (half[axis] - (float)sgn * o[axis]) / dn;
      for (int it = 0; it < 2; ++it) {
        const float ha = o[ia] + t * d[ia], hb = o[ib] + t * d[ib];
        const float relief = 0.02f * sinf(5.0f * ha) * sinf(5.0f * hb);
        t = (half[axis] - relief - 0.03f - (float)sgn * o[axis]) / dn;
      }
      if (t > 0 && t < best) best = t;
    }
  }
  const uint32_t pix = (uint32_t)(y * W + x);
  const uint32_t base = pcg_hash(a.seed ^ pcg_hash(a.frame * 0x9E3779B9u + 0x7F4A7C15u));
  const uint32_t h1 = pcg_hash(base ^ (pix * 2u + 1u)), h2 = pcg_hash(base + 0x68E31DA4u + pix * 2u);
  const float gauss = sqrtf(-2.0f * logf(u01(h1))) * cosf(6.2831853f * u01(h2));
  const uint32_t blk = (uint32_t)((y >> 3) * ((W + 7) >> 3) + (x >> 3));
  bool drop = u01(pcg_hash(base ^ (0xB5297A4Du + blk * 0x1B56C4E9u))) < a.dropout;
  drop = drop || (u01(pcg_hash(h1 ^ h2 ^ 0x2545F491u)) < a.dropout * 0.01f);
  const float zn = best + a.noise_sigma * best * best * gauss;
  const float dv = rintf(a.depth_scaling * zn);
  uint16_t out = 0;
  if (!drop && dv > 0.0f && dv < 65535.0f) out = (uint16_t)(int)dv;
  depth(y, x) = out;
  const float hit[3] = {o[0] + best * d[0], o[1] + best * d[1], o[2] + best * d[2]};
  const int c0 = (int)floorf(hit[0] * 10.0f), c1 = (int)floorf(hit[1] * 10.0f), c2 = (int)floorf(hit[2] * 10.0f);
  const uint32_t hc = (uint32_t)(c0 * 73856093) ^ (uint32_t)(c1 * 19349663) ^ (uint32_t)(c2 * 83492791);
  uchar3 col;
  col.x = (unsigned char)(hc & 255u); col.y = (unsigned char)((hc >> 8) & 255u); col.z = (unsigned char)((hc >> 16) & 255u);
  color(y, x) = col;
}

}  // namespace

extern "C" int smx_synth_render_room(smx_stream s, const smx_buffer_desc* depth_out, const smx_buffer_desc* color_out,
                                     float fx, float fy, float cx, float cy, const float global_T_frame[12],
                                     uint32_t seed, uint32_t frame_index, float depth_scaling, float noise_sigma,
                                     float dropout) {
  SMX_CHECK_ARG(depth_out && color_out && global_T_frame);
  SMX_CHECK_ARG(depth_out->width == color_out->width && depth_out->height == color_out->height);
  SynthArgs a;
  memcpy(a.G.m, global_T_frame, sizeof(float) * 12);
  a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.seed = seed; a.frame = frame_index;
  a.depth_scaling = depth_scaling; a.noise_sigma = noise_sigma; a.dropout = dropout;
  dim3 grid(div_up(depth_out->width, 64), div_up(depth_out->height, 4));
  hipLaunchKernelGGL(k_synth_room, grid, dim3(256), 0, (hipStream_t)s, a, as_img<uint16_t>(depth_out), as_img<uchar3>(color_out));
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}
