/* oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, never part of the product.
 *
 * Runs the REFERENCE'S OWN KERNELS (the two .cu files of puzzlepaint/surfelmeshing, compiled by hipcc from where they
 * lie under /root/reference -- see oracle/ref_build.py; none of the reference's source is copied into this
 * repository) so that the CPU oracle of oracle/ can be pinned against the code it restates:
 *   applications/surfel_meshing/src/surfel_meshing/cuda_depth_processing.cu
 *   applications/surfel_meshing/src/surfel_meshing/cuda_surfel_reconstruction_kernels.cu
 * This file is the host side those kernels need: the reference's host files (cuda_surfel_reconstruction.cc,
 * ..._kernels.cc, libvis CUDABuffer) depend on Eigen / Sophus / Qt, which are not in the image, so the call sequence
 * of CUDASurfelReconstruction::Integrate (cuda_surfel_reconstruction.cc:112-320) and the parameter derivations of
 * ..._kernels.cc:37-511 are restated here around the reference's launcher functions (the Call*Kernel / *CUDA
 * functions of the two .cuh headers), with plain hipMalloc buffers wrapped as CUDABuffer_<T>.
 *
 * Plain C interface with HOST pointers; everything is synchronous.  Built into oracle/_ref/libsmx_ref.so.
 */
#include <libvis/logging.h>

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <limits>
#include <vector>

#include "surfel_meshing/cuda_depth_processing.cuh"
#include "surfel_meshing/cuda_surfel_reconstruction_kernels.cuh"

using namespace vis;

namespace {

constexpr u32 kInvalidIndex = 0xFFFFFFFFu;  // Surfel::kInvalidIndex, surfel.h:63

#define REF_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "ref_harness: %s -> %s\n", #x, hipGetErrorString(e_)); return -2; } } while (0)

template <typename T>
struct Dev {  // tightly pitched device image
  T* p = nullptr; int h = 0, w = 0;
  int alloc(int height, int width) {
    h = height; w = width;
    return hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (size_t)h * w) == hipSuccess ? 0 : -2;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; }
  size_t bytes() const { return sizeof(T) * (size_t)h * w; }
  CUDABuffer_<T> cb() const { return CUDABuffer_<T>(p, h, w, sizeof(T) * (size_t)w); }
  int up(const void* src) { return hipMemcpy(p, src, bytes(), hipMemcpyHostToDevice) == hipSuccess ? 0 : -2; }
  int down(void* dst) const { return hipMemcpy(dst, p, bytes(), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2; }
};

template <typename T>
__global__ void k_fill(T* p, size_t n, T v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
template <typename T>
void fill(Dev<T>& d, T v) {  // CUDABuffer<T>::Clear
  const size_t n = (size_t)d.h * d.w;
  hipLaunchKernelGGL(k_fill<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d.p, n, v);
}

struct RowMajor3x4 {  // what CUDAMatrix3x4's converting constructor reads: matrix(row, col)
  const float* m;
  float operator()(int r, int c) const { return m[4 * r + c]; }
};
CUDAMatrix3x4 mat(const float* m) { return CUDAMatrix3x4(RowMajor3x4{m}); }

int GetBlockCountLocal(int n, int b) { return (n + b - 1) / b; }

}  // namespace

extern "C" {

/* ---- depth preprocessing: one call = upload, the reference's launcher, download -------------------------------- */
int ref_bilateral(int W, int H, const uint16_t* in, float sigma_xy, float sigma_value_factor, uint16_t value_to_ignore,
                  float radius_factor, uint16_t max_depth, float depth_valid_region_radius, uint16_t* out) {
  Dev<u16> a, b;
  if (a.alloc(H, W) || b.alloc(H, W) || a.up(in)) return -2;
  REF_HIP(hipMemset(b.p, 0, b.bytes()));
  CUDABuffer_<u16> o = b.cb();
  BilateralFilteringAndDepthCutoffCUDA(0, sigma_xy, sigma_value_factor, value_to_ignore, radius_factor, max_depth,
                                       depth_valid_region_radius, a.cb(), &o);
  REF_HIP(hipDeviceSynchronize());
  const int rc = b.down(out);
  a.release(); b.release();
  return rc;
}

int ref_outlier_fusion(int W, int H, int other_count /* 2, 4, 6, 8 */, int required_count /* -1: all */, float tolerance,
                       const uint16_t* in, float fx, float fy, float cx, float cy, const uint16_t* const* others,
                       const float* others_TR_reference /* [other_count][12] */, uint16_t* out) {
  Dev<u16> a, b, o[8];
  if (a.alloc(H, W) || b.alloc(H, W) || a.up(in)) return -2;
  REF_HIP(hipMemset(b.p, 0, b.bytes()));
  CUDABuffer_<u16> ocb[8];
  const CUDABuffer_<u16>* optr[8];
  CUDAMatrix3x4 T[8];
  for (int i = 0; i < other_count; ++i) {
    if (o[i].alloc(H, W) || o[i].up(others[i])) return -2;
    ocb[i] = o[i].cb(); optr[i] = &ocb[i];
    T[i] = mat(others_TR_reference + 12 * i);
  }
  CUDABuffer_<u16> dst = b.cb();
#define REF_OUTLIER(N)                                                                                            \
  case N - 1:                                                                                                     \
    if (required_count < 0) OutlierDepthMapFusionCUDA<N, u16>(0, tolerance, a.cb(), fx, fy, cx, cy, optr, T, &dst); \
    else OutlierDepthMapFusionCUDA<N, u16>(0, required_count, tolerance, a.cb(), fx, fy, cx, cy, optr, T, &dst);  \
    break
  switch (other_count) {  // main.cc:1076-1086: count = others + 1
    REF_OUTLIER(3); REF_OUTLIER(5); REF_OUTLIER(7); REF_OUTLIER(9);
    default: return -1;
  }
#undef REF_OUTLIER
  REF_HIP(hipDeviceSynchronize());
  const int rc = b.down(out);
  a.release(); b.release();
  for (int i = 0; i < other_count; ++i) o[i].release();
  return rc;
}

int ref_erode(int W, int H, int radius, const uint16_t* in, uint16_t* out) {
  Dev<u16> a, b;
  if (a.alloc(H, W) || b.alloc(H, W) || a.up(in)) return -2;
  REF_HIP(hipMemset(b.p, 0, b.bytes()));
  CUDABuffer_<u16> o = b.cb();
  if (radius > 0) ErodeDepthMapCUDA<u16>(0, radius, a.cb(), &o);
  else CopyWithoutBorderCUDA<u16>(0, a.cb(), &o);
  REF_HIP(hipDeviceSynchronize());
  const int rc = b.down(out);
  a.release(); b.release();
  return rc;
}

int ref_normals(int W, int H, float observation_angle_threshold_deg, float depth_scaling, float fx, float fy, float cx,
                float cy, const uint16_t* in, uint16_t* out_depth, float* out_normals /* [H][W][2] */) {
  Dev<u16> a, b;
  Dev<float2> n;
  if (a.alloc(H, W) || b.alloc(H, W) || n.alloc(H, W) || a.up(in)) return -2;
  REF_HIP(hipMemset(b.p, 0, b.bytes()));
  REF_HIP(hipMemset(n.p, 0, n.bytes()));
  CUDABuffer_<u16> o = b.cb();
  CUDABuffer_<float2> on = n.cb();
  ComputeNormalsAndDropBadPixelsCUDA(0, observation_angle_threshold_deg, depth_scaling, fx, fy, cx, cy, a.cb(), &o, &on);
  REF_HIP(hipDeviceSynchronize());
  const int rc = b.down(out_depth) | n.down(out_normals);
  a.release(); b.release(); n.release();
  return rc;
}

int ref_radii(int W, int H, float point_radius_extension_factor, float point_radius_clamp_factor, float depth_scaling,
              float fx, float fy, float cx, float cy, const uint16_t* in, float* radius_inout, uint16_t* out_depth) {
  Dev<u16> a, b;
  Dev<float> r;
  if (a.alloc(H, W) || b.alloc(H, W) || r.alloc(H, W) || a.up(in) || r.up(radius_inout)) return -2;
  REF_HIP(hipMemset(b.p, 0, b.bytes()));
  CUDABuffer_<u16> o = b.cb();
  CUDABuffer_<float> orad = r.cb();
  ComputePointRadiiAndRemoveIsolatedPixelsCUDA(0, point_radius_extension_factor, point_radius_clamp_factor, depth_scaling,
                                               fx, fy, cx, cy, a.cb(), &orad, &o);
  REF_HIP(hipDeviceSynchronize());
  const int rc = b.down(out_depth) | r.down(radius_inout);
  a.release(); b.release(); r.release();
  return rc;
}

/* ---- CUDASurfelReconstruction around the reference's kernels ---------------------------------------------------- */
typedef struct {  /* same fields as smx_integrate_params / orc_integrate_params */
  float sensor_noise_factor, max_surfel_confidence, regularizer_weight;
  int32_t regularization_frame_window_size, do_blending, measurement_blending_radius,
      regularization_iterations_per_integration_iteration;
  float radius_factor_for_regularization_neighbors, normal_compatibility_threshold_deg;
  int32_t surfel_integration_active_window_size;
} ref_integrate_params;

struct ref_recon {
  int W, H;
  float fx, fy, cx, cy;
  u32 max_surfels, surfel_count, merge_count;
  Dev<float> surfels;  // [25][max_surfels], cuda_surfel_reconstruction.cc:59
  Dev<u32> supporting, counts, conflicting;
  Dev<float> depth_sums, first_depth, deltas, new_deltas;
  Dev<u8> distance_map, new_distance_map, new_flags;
  Dev<u32> new_indices, num_merges;
  Dev<u16> depth;
  Dev<float2> normals;
  Dev<float> radius;
  Dev<uchar3> color;
  void* scan_temp;
  usize scan_temp_bytes;
  u32 last_new;
  hipEvent_t ev0, ev1;
  float last_ms;   // device time of the last Integrate (clears .. regulariser, without the image upload / download)
};

int ref_recon_create(uint32_t max_surfels, int W, int H, float fx, float fy, float cx, float cy, ref_recon** out) {
  ref_recon* r = new ref_recon();
  r->W = W; r->H = H; r->fx = fx; r->fy = fy; r->cx = cx; r->cy = cy;
  r->max_surfels = max_surfels; r->surfel_count = 0; r->merge_count = 0; r->scan_temp = nullptr; r->scan_temp_bytes = 0;
  r->last_new = 0; r->last_ms = 0;
  if (hipEventCreate(&r->ev0) != hipSuccess || hipEventCreate(&r->ev1) != hipSuccess) return -2;
  int rc = r->surfels.alloc(kSurfelAttributeCount, (int)max_surfels);
  rc |= r->supporting.alloc(H, W) | r->counts.alloc(H, W) | r->conflicting.alloc(H, W) | r->depth_sums.alloc(H, W);
  rc |= r->first_depth.alloc(H, W) | r->deltas.alloc(H, W) | r->new_deltas.alloc(H, W);
  rc |= r->distance_map.alloc(H, W) | r->new_distance_map.alloc(H, W) | r->new_flags.alloc(1, W * H);
  rc |= r->new_indices.alloc(1, W * H) | r->num_merges.alloc(1, 1);
  rc |= r->depth.alloc(H, W) | r->normals.alloc(H, W) | r->radius.alloc(H, W) | r->color.alloc(H, W);
  if (rc) return -2;
  REF_HIP(hipMemset(r->surfels.p, 0, r->surfels.bytes()));
  REF_HIP(hipMemset(r->deltas.p, 0, r->deltas.bytes()));
  REF_HIP(hipMemset(r->new_deltas.p, 0, r->new_deltas.bytes()));
  *out = r;
  return 0;
}

void ref_recon_destroy(ref_recon* r) {
  if (!r) return;
  r->surfels.release(); r->supporting.release(); r->counts.release(); r->conflicting.release(); r->depth_sums.release();
  r->first_depth.release(); r->deltas.release(); r->new_deltas.release(); r->distance_map.release();
  r->new_distance_map.release(); r->new_flags.release(); r->new_indices.release(); r->num_merges.release();
  r->depth.release(); r->normals.release(); r->radius.release(); r->color.release();
  if (r->scan_temp) (void)hipFree(r->scan_temp);
  delete r;
}

/* Integrate, cuda_surfel_reconstruction.cc:112-320.  depth is rewritten by the blending (returned in place).
 * local_T_global is passed in (the reference takes it from Sophus' inverse()) so that both sides of a comparison use
 * the same matrix. */
int ref_recon_integrate(ref_recon* r, uint32_t frame_index, float depth_scaling, uint16_t* depth, const float* normals,
                        const float* radius, const uint8_t* color, const float global_T_local[12],
                        const float local_T_global[12], const ref_integrate_params* p) {
  const int W = r->W, H = r->H;
  if (r->depth.up(depth) || r->normals.up(normals) || r->radius.up(radius) || r->color.up(color)) return -2;
  hipStream_t stream = 0;
  const CUDAMatrix3x4 G = mat(global_T_local), L = mat(local_T_global);
  const float fx = r->fx, fy = r->fy, cx = r->cx, cy = r->cy;
  // Unprojection intrinsics for pixel center convention, kernels.cc:69-74
  const float fx_inv = 1.0f / fx, fy_inv = 1.0f / fy;
  const float cx_inv_pixel_center = -(cx - 0.5f) / fx, cy_inv_pixel_center = -(cy - 0.5f) / fy;
  const float cos_thr = cosf(M_PI / 180.0f * p->normal_compatibility_threshold_deg);  // kernels.cc:261
  const float depth_correction_factor = 1.0f / depth_scaling;

  REF_HIP(hipEventRecord(r->ev0, stream));
  // cc:134-138
  fill(r->supporting, kInvalidIndex);
  fill(r->counts, 0u);
  fill(r->depth_sums, 0.0f);
  fill(r->conflicting, kInvalidIndex);
  fill(r->first_depth, std::numeric_limits<float>::infinity());

  const u32 N = r->surfel_count;
  if (N > 0) {
    const dim3 grid(GetBlockCountLocal((int)N, 1024)), block(1024);
    CallRenderMinDepthCUDAKernel(stream, grid, block, frame_index, p->surfel_integration_active_window_size, fx, fy, cx, cy,
                                 L, N, r->surfels.cb(), r->first_depth.cb());
    CallAssociateSurfelsCUDAKernel(stream, grid, block, frame_index, p->surfel_integration_active_window_size, fx, fy, cx,
                                   cy, L, p->sensor_noise_factor, cos_thr, N, r->surfels.cb(), depth_correction_factor,
                                   r->depth.cb(), r->normals.cb(), r->radius.cb(), r->supporting.cb(), r->counts.cb(),
                                   r->depth_sums.cb(), r->conflicting.cb(), r->first_depth.cb());
    // MergeSurfelsCUDA, kernels.cc:442-511
    fill(r->num_merges, 0u);
    const dim3 mgrid(GetBlockCountLocal((int)N, kMergeBlockWidth)), mblock(kMergeBlockWidth);
    CallMergeSurfelsCUDAKernel(stream, mgrid, mblock, fx, fy, cx, cy, L, p->sensor_noise_factor, cos_thr, N,
                               r->surfels.cb(), depth_correction_factor, r->depth.cb(), r->normals.cb(), r->radius.cb(),
                               r->supporting.cb(), r->counts.cb(), r->depth_sums.cb(), r->conflicting.cb(),
                               r->first_depth.cb(), r->num_merges.cb());
    u32 num_merges = 0;
    REF_HIP(hipMemcpy(&num_merges, r->num_merges.p, 4, hipMemcpyDeviceToHost));
    r->merge_count += num_merges;
  }
  if (p->do_blending) {  // BlendMeasurementsCUDA, kernels.cc:148-205
    fill(r->distance_map, (u8)0);
    fill(r->new_distance_map, (u8)0);
    const dim3 grid(GetBlockCountLocal(W, 32), GetBlockCountLocal(H, 32)), block(32, 32);
    CallBlendMeasurementsCUDAStartKernel(stream, grid, block, 1.0f / depth_correction_factor, r->depth.cb(),
                                         r->supporting.cb(), r->counts.cb(), r->depth_sums.cb(), r->distance_map.cb(),
                                         r->deltas.cb(), r->new_distance_map.cb(), r->new_deltas.cb());
    for (int iteration = 2; iteration < p->measurement_blending_radius; ++iteration)
      CallBlendMeasurementsCUDAIterationKernel(stream, grid, block, iteration,
                                               1.0f / (p->measurement_blending_radius - 1.0f),
                                               1.0f / depth_correction_factor, r->depth.cb(), r->supporting.cb(),
                                               r->distance_map.cb(), r->deltas.cb(), r->new_distance_map.cb(),
                                               r->new_deltas.cb());
  }
  if (N > 0) {
    const dim3 grid(GetBlockCountLocal((int)N, 1024)), block(1024);
    CallIntegrateMeasurementsCUDAKernel(stream, grid, block, frame_index, p->surfel_integration_active_window_size,
                                        p->max_surfel_confidence, p->sensor_noise_factor, cos_thr, 1.0f / depth_scaling,
                                        fx, fy, cx, cy, fx_inv, fy_inv, cx_inv_pixel_center, cy_inv_pixel_center, L, G,
                                        r->depth.cb(), r->normals.cb(), r->radius.cb(), r->color.cb(), r->supporting.cb(),
                                        r->counts.cb(), r->conflicting.cb(), r->first_depth.cb(), N, r->surfels.cb());
    const float rf2 = p->radius_factor_for_regularization_neighbors * p->radius_factor_for_regularization_neighbors;
    CallUpdateNeighborsCUDAKernel(stream, grid, block, frame_index, p->surfel_integration_active_window_size, rf2,
                                  r->supporting.cb(), fx, fy, cx, cy, L, p->sensor_noise_factor, depth_correction_factor,
                                  r->depth.cb(), r->radius.cb(), r->first_depth.cb(), N, r->surfels.cb());
    CallUpdateNeighborsCUDARemoveReplacedNeighborsKernel(stream, grid, block, frame_index, N, r->surfels.cb());
  }
  {  // CreateNewSurfelsCUDA, kernels.cc:37-146
    const dim3 grid(GetBlockCountLocal(W, 32), GetBlockCountLocal(H, 32)), block(32, 32);
    CallCreateNewSurfelsCUDASerializingKernel(stream, grid, block, r->depth.cb(), r->supporting.cb(), r->conflicting.cb(),
                                              r->new_flags.cb());
    if (r->scan_temp_bytes == 0) {
      CallCUBExclusiveSum(r->scan_temp, r->scan_temp_bytes, r->new_flags.p, r->new_indices.p, W * H, stream);
      REF_HIP(hipMalloc(&r->scan_temp, r->scan_temp_bytes));
    }
    CallCUBExclusiveSum(r->scan_temp, r->scan_temp_bytes, r->new_flags.p, r->new_indices.p, W * H, stream);
    u32 new_surfel_count = 0;
    u8 new_surfel_count_2 = 0;
    REF_HIP(hipMemcpy(&new_surfel_count, r->new_indices.p + (W * H - 1), 4, hipMemcpyDeviceToHost));
    REF_HIP(hipMemcpy(&new_surfel_count_2, r->new_flags.p + (W * H - 1), 1, hipMemcpyDeviceToHost));
    if ((uint64_t)r->surfel_count + new_surfel_count + new_surfel_count_2 > r->max_surfels) return -3;  // (unchecked in the reference)
    CallCreateNewSurfelsCUDACreationKernel(
        stream, grid, block, frame_index, 1.0f / depth_scaling, fx_inv, fy_inv, cx_inv_pixel_center, cy_inv_pixel_center, G,
        r->depth.cb(), r->normals.cb(), r->radius.cb(), r->color.cb(), r->supporting.cb(), r->new_flags.cb(),
        r->new_indices.cb(), r->surfel_count, r->surfels.cb(),
        p->radius_factor_for_regularization_neighbors * p->radius_factor_for_regularization_neighbors);
    REF_HIP(hipDeviceSynchronize());
    r->last_new = new_surfel_count + new_surfel_count_2;
    r->surfel_count += r->last_new;  // cc:291
  }
  CUDABuffer_<float> sb = r->surfels.cb();
  if (p->regularization_iterations_per_integration_iteration == 0) {
    RegularizeSurfelsCUDA(stream, /*disable_denoising*/ true, frame_index, p->radius_factor_for_regularization_neighbors,
                          p->regularizer_weight, p->regularization_frame_window_size, r->surfel_count, &sb);
  } else {
    for (int i = 0; i < p->regularization_iterations_per_integration_iteration; ++i)
      RegularizeSurfelsCUDA(stream, /*disable_denoising*/ false, frame_index,
                            p->radius_factor_for_regularization_neighbors, p->regularizer_weight,
                            p->regularization_frame_window_size, r->surfel_count, &sb);
  }
  REF_HIP(hipEventRecord(r->ev1, stream));
  REF_HIP(hipDeviceSynchronize());
  REF_HIP(hipGetLastError());
  REF_HIP(hipEventElapsedTime(&r->last_ms, r->ev0, r->ev1));
  return r->depth.down(depth);
}

int ref_recon_regularize(ref_recon* r, uint32_t frame_index, float regularizer_weight,
                         float radius_factor_for_regularization_neighbors, int regularization_frame_window_size) {
  CUDABuffer_<float> sb = r->surfels.cb();
  RegularizeSurfelsCUDA(0, false, frame_index, radius_factor_for_regularization_neighbors, regularizer_weight,
                        regularization_frame_window_size, r->surfel_count, &sb);
  REF_HIP(hipDeviceSynchronize());
  return 0;
}

float ref_recon_last_integrate_ms(const ref_recon* r) { return r->last_ms; }

void ref_recon_counts(const ref_recon* r, uint32_t* surfels_size, uint32_t* merge_count, uint32_t* last_new) {
  *surfels_size = r->surfel_count; *merge_count = r->merge_count; *last_new = r->last_new;
}

/* rows: [25][count] floats (bit patterns of the u32 rows) */
int ref_recon_download_surfels(const ref_recon* r, float* rows, uint32_t count) {
  for (int k = 0; k < kSurfelAttributeCount; ++k)
    REF_HIP(hipMemcpy(rows + (size_t)k * count, r->surfels.p + (size_t)k * r->max_surfels, 4 * (size_t)count,
                      hipMemcpyDeviceToHost));
  return 0;
}
int ref_recon_upload_surfels(ref_recon* r, const float* rows, uint32_t count, uint32_t merge_count) {
  for (int k = 0; k < kSurfelAttributeCount; ++k)
    REF_HIP(hipMemcpy(r->surfels.p + (size_t)k * r->max_surfels, rows + (size_t)k * count, 4 * (size_t)count,
                      hipMemcpyHostToDevice));
  r->surfel_count = count; r->merge_count = merge_count;
  return 0;
}
/* ExportVertices, cuda_surfel_reconstruction.cc:405-410: positions [3 * count] floats, colours [3 * count] bytes */
int ref_recon_export_vertices(const ref_recon* r, float* positions, uint8_t* colors) {
  const u32 n = r->surfel_count;
  if (n == 0) return 0;
  Dev<float> pos;
  Dev<u8> col;
  if (pos.alloc(1, 3 * (int)n) || col.alloc(1, 3 * (int)n)) return -2;
  CUDABuffer_<float> pb = pos.cb();
  CUDABuffer_<u8> cbuf = col.cb();
  ExportVerticesCUDA(0, n, r->surfels.cb(), &pb, &cbuf);
  REF_HIP(hipDeviceSynchronize());
  const int rc = pos.down(positions) | col.down(colors);
  pos.release(); col.release();
  return rc;
}

/* which: 0 supporting u32, 1 counts u32, 2 depth sums f32, 3 conflicting u32, 4 first depth f32 */
int ref_recon_download_scratch(const ref_recon* r, int which, void* dst) {
  switch (which) {
    case 0: return r->supporting.down(dst);
    case 1: return r->counts.down(dst);
    case 2: return r->depth_sums.down(dst);
    case 3: return r->conflicting.down(dst);
    case 4: return r->first_depth.down(dst);
    default: return -1;
  }
}

}  // extern "C"
