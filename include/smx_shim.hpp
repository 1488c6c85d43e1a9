// smx_shim.hpp -- header-only C++ shim that re-creates the reference's class and function names for the
// surfel-integration path on top of the C-ABI of smx.h, so that reference-style host code (APP/main.cc's
// frame loop, APP/test/test_triangulation.cc's CUDASurfelsCPU protocol) compiles against libsmx.so after
// replacing cudaStream_t by hipStream_t (both are opaque pointers; smx_stream is a void*).
//
//   vis::CUDABuffer<T>, vis::CUDABuffer_<T>           VIS/cuda/cuda_buffer.h:45-129, cuda_buffer.cuh:44-119
//   vis::CUDAMatrix3x4                                 VIS/cuda/cuda_matrix.cuh:67-116 (host part)
//   vis::BilateralFilteringAndDepthCutoffCUDA ...      APP/cuda_depth_processing.cuh:43-122
//   vis::CUDASurfelReconstruction                      APP/cuda_surfel_reconstruction.h:44-176
//   vis::CUDASurfelBuffersCPU, vis::CUDASurfelsCPU     APP/cuda_surfels_cpu.h:40-124
//
// Error behaviour: the reference aborts through LOG(FATAL) / CUDA_CHECKED_CALL
// (VIS/cuda/cuda_util.h:35-49); SMX_SHIM_CHECK prints smx_last_error() and aborts likewise.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "smx.h"

#define SMX_SHIM_CHECK(call)                                                                     \
  do {                                                                                           \
    int smx_rc__ = (call);                                                                       \
    if (smx_rc__ != 0) {                                                                         \
      std::fprintf(stderr, "FATAL %s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, smx_rc__,  \
                   smx_last_error());                                                            \
      std::abort();                                                                              \
    }                                                                                            \
  } while (0)

// float2: the reference's callers spell the normals buffer `CUDABuffer<float2>` with CUDA's global vector type
// (APP/main.cc:804, APP/cuda_surfel_reconstruction.h:64).  Inside a HIP translation unit (or with SMX_SHIM_USE_HIP_TYPES) that
// is HIP's own ::float2; a plain host compiler without the HIP headers gets a stand-in of the same layout at global scope.
#if defined(__HIPCC__) || defined(SMX_SHIM_USE_HIP_TYPES)
#include <hip/hip_vector_types.h>
#elif !defined(SMX_SHIM_NO_FLOAT2)
struct float2 { float x, y; };
#endif

namespace vis {

// Process-level settings the frame loop wants (smx.h: smx_runtime_advice; INTEGRATION.md "Streams, queues, priorities"): call
// at the top of main(), BEFORE the first HIP call of the process -- the HIP runtime reads the variable when it initialises.
// Never overrides a value the user exported.  The library itself does not touch the environment.
inline void SmxSetRecommendedRuntimeDefaults() { setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite*/ 0); }

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef size_t usize;
typedef smx_stream cudaStream_t;  // a hipStream_t

typedef ::float2 float2_;   // (rounds 1-5 spelled the shim's own stand-in this way; kept for their callers)
static_assert(sizeof(::float2) == 8, "float2 must be two packed floats");
// Vec3u8: libvis' Eigen typedef (VIS/eigen.h).  Inside the reference tree define SMX_SHIM_NO_VEC_TYPES and include libvis'
// header first: the shim then takes vis::Vec3u8 as it finds it (any 3-byte type works: only its size is used).
#ifndef SMX_SHIM_NO_VEC_TYPES
struct Vec3u8 { u8 v[3]; };
#endif

// Row-major 3x4 rigid transform (host part of VIS/cuda/cuda_matrix.cuh:67-116).  As in the reference it is
// constructible from ANY matrix type with (row, col) element access -- the call sites hand it an Eigen 3x4 from
// Sophus' SE3f::matrix3x4() (APP/main.cc:1051, 1057) -- so those lines compile unchanged; the storage order of the
// source type does not matter.  The float-pointer constructor (12 row-major floats) is an addition.
struct CUDAMatrix3x4 {
  float m[12];
  CUDAMatrix3x4() {}
  explicit CUDAMatrix3x4(const float* row_major_3x4) { for (int i = 0; i < 12; ++i) m[i] = row_major_3x4[i]; }
  template <typename T, typename = decltype(static_cast<float>(std::declval<const T&>()(0, 0)))>
  explicit CUDAMatrix3x4(const T& matrix) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) m[4 * r + c] = static_cast<float>(matrix(r, c));
  }
};

#ifndef SMX_SHIM_NO_SE3F
// Stand-in for libvis' SE3f (= Sophus::SE3f, VIS/libvis.h) for callers that build without Sophus / Eigen: the members
// the hot path's call sites use (APP/main.cc:1039-1059, 1205-1223) -- matrix3x4(), inverse(), operator*, translation().
// Inside the reference tree define SMX_SHIM_NO_SE3F and keep Sophus: Integrate() below accepts any pose type that
// has matrix3x4().
struct Vec3f_ {
  float v[3];
  float& operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
};
inline Vec3f_ operator*(float s, const Vec3f_& a) { return Vec3f_{{s * a.v[0], s * a.v[1], s * a.v[2]}}; }
struct Matrix3x4f_ {
  float m[12];  // row-major
  float operator()(int r, int c) const { return m[4 * r + c]; }
};
class SE3f {
 public:
  SE3f() : R_{1, 0, 0, 0, 1, 0, 0, 0, 1}, t_{{0, 0, 0}} {}
  explicit SE3f(const float* row_major_3x4) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) R_[3 * r + c] = row_major_3x4[4 * r + c];
      t_.v[r] = row_major_3x4[4 * r + 3];
    }
  }
  Matrix3x4f_ matrix3x4() const {
    Matrix3x4f_ o;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) o.m[4 * r + c] = R_[3 * r + c];
      o.m[4 * r + 3] = t_.v[r];
    }
    return o;
  }
  SE3f inverse() const {  // R^T, -(R^T t)
    SE3f o;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) o.R_[3 * r + c] = R_[3 * c + r];
      o.t_.v[r] = -(R_[0 + r] * t_.v[0] + R_[3 + r] * t_.v[1] + R_[6 + r] * t_.v[2]);
    }
    return o;
  }
  SE3f operator*(const SE3f& b) const {
    SE3f o;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c)
        o.R_[3 * r + c] = R_[3 * r] * b.R_[c] + R_[3 * r + 1] * b.R_[3 + c] + R_[3 * r + 2] * b.R_[6 + c];
      o.t_.v[r] = R_[3 * r] * b.t_.v[0] + R_[3 * r + 1] * b.t_.v[1] + R_[3 * r + 2] * b.t_.v[2] + t_.v[r];
    }
    return o;
  }
  Vec3f_& translation() { return t_; }
  const Vec3f_& translation() const { return t_; }
 private:
  float R_[9];
  Vec3f_ t_;
};
#endif  // SMX_SHIM_NO_SE3F

// Accessors of VIS/camera.h's PinholeCamera4f that the hot path uses.
class PinholeCamera4f {
 public:
  PinholeCamera4f(int width, int height, const float* fx_fy_cx_cy) : width_(width), height_(height) {
    for (int i = 0; i < 4; ++i) p_[i] = fx_fy_cx_cy[i];
  }
  int width() const { return width_; }
  int height() const { return height_; }
  const float* parameters() const { return p_; }  // fx, fy, cx, cy (pixel-corner convention)
 private:
  int width_, height_;
  float p_[4];
};

// Device-side view, same members/layout as the reference's CUDABuffer_<T>.
#if defined(__HIPCC__)
#define SMX_SHIM_HD __host__ __device__
#else
#define SMX_SHIM_HD
#endif
template <typename T>
struct CUDABuffer_ {
  T* address_;
  int height_;
  int width_;
  size_t pitch_;
  SMX_SHIM_HD T* address() const { return address_; }
  SMX_SHIM_HD int width() const { return width_; }
  SMX_SHIM_HD int height() const { return height_; }
  SMX_SHIM_HD size_t pitch() const { return pitch_; }
  const smx_buffer_desc* desc() const { return reinterpret_cast<const smx_buffer_desc*>(this); }
#if defined(__HIPCC__)
  // The element accessors of VIS/cuda/cuda_buffer.cuh:58-96, for a maintainer's OWN kernels that take a CUDABuffer_<T> by
  // value (the reference's visualisation and debug kernels do): element (y, x) at address + y * pitch + x * sizeof(T).
  __device__ __forceinline__ T& operator()(unsigned int y, unsigned int x) {
    return *(reinterpret_cast<T*>(reinterpret_cast<char*>(address_) + y * pitch_) + x);
  }
  __device__ __forceinline__ const T& operator()(unsigned int y, unsigned int x) const {
    return *(reinterpret_cast<const T*>(reinterpret_cast<const char*>(address_) + y * pitch_) + x);
  }
  __device__ __forceinline__ T& operator()(const int2& pixel) { return operator()(pixel.y, pixel.x); }
  __device__ __forceinline__ const T& operator()(const int2& pixel) const { return operator()(pixel.y, pixel.x); }
  __device__ __forceinline__ T& operator()(const uint2& pixel) { return operator()(pixel.y, pixel.x); }
  __device__ __forceinline__ const T& operator()(const uint2& pixel) const { return operator()(pixel.y, pixel.x); }
#endif
};
static_assert(sizeof(CUDABuffer_<float>) == sizeof(smx_buffer_desc), "CUDABuffer_ must alias smx_buffer_desc");

template <typename T>
class CUDABuffer {
 public:
  typedef T Type;
  CUDABuffer(int height, int width) {
    SMX_SHIM_CHECK(smx_buffer_create(height, width, (int32_t)sizeof(T), &handle_));
    smx_buffer_desc d;
    SMX_SHIM_CHECK(smx_buffer_get_desc(handle_, &d));
    data_.address_ = static_cast<T*>(d.address); data_.height_ = d.height; data_.width_ = d.width; data_.pitch_ = d.pitch;
  }
  CUDABuffer(const CUDABuffer<T>&) = delete;
  ~CUDABuffer() { SMX_SHIM_CHECK(smx_buffer_destroy(handle_)); }

  void UploadAsync(cudaStream_t stream, const T* data) { SMX_SHIM_CHECK(smx_buffer_upload(handle_, stream, data, 0)); }
  // (an addition: the copy as a kernel that reads page-locked memory over the bus; false if `data` is not page-locked)
  bool UploadByKernelAsync(cudaStream_t stream, const T* data, smx_event done = nullptr) {
    return smx_buffer_upload_by_kernel(handle_, stream, data, 0, done) == SMX_OK;
  }
  void UploadPitchedAsync(cudaStream_t stream, size_t pitch, const T* data) { SMX_SHIM_CHECK(smx_buffer_upload(handle_, stream, data, pitch)); }
  void UploadPartAsync(size_t start, size_t length, cudaStream_t stream, const T* data) { SMX_SHIM_CHECK(smx_buffer_upload_part(handle_, stream, start, length, data)); }
  void DownloadAsync(cudaStream_t stream, T* data) const { SMX_SHIM_CHECK(smx_buffer_download(handle_, stream, data, 0)); }
  void DownloadPitchedAsync(cudaStream_t stream, size_t pitch, T* data) { SMX_SHIM_CHECK(smx_buffer_download(handle_, stream, data, pitch)); }
  void DownloadPartAsync(size_t start, size_t length, cudaStream_t stream, T* data) const { SMX_SHIM_CHECK(smx_buffer_download_part(handle_, stream, start, length, data)); }
  void DebugUpload(const T* data) { UploadAsync(nullptr, data); SMX_SHIM_CHECK(smx_stream_synchronize(nullptr)); }
  void DebugDownload(T* data) const { DownloadAsync(nullptr, data); SMX_SHIM_CHECK(smx_stream_synchronize(nullptr)); }
  // (VIS/cuda/cuda_buffer.h:65, 82: the synchronous forms with a host pitch)
  void DebugUploadPitched(size_t pitch, const T* data) { UploadPitchedAsync(nullptr, pitch, data); SMX_SHIM_CHECK(smx_stream_synchronize(nullptr)); }
  void DebugDownloadPitched(size_t pitch, T* data) const {
    SMX_SHIM_CHECK(smx_buffer_download(handle_, nullptr, data, pitch));
    SMX_SHIM_CHECK(smx_stream_synchronize(nullptr));
  }
  // The Image<T> overloads (VIS/cuda/cuda_buffer.h:69, 86; cuda_buffer_inl.h: a pitched copy with the image's stride) --
  // what APP/main.cc:1030, 1121, 1147, 1171 call with `&filtered_depth`.  Any image type with data() and stride() (bytes per
  // row) serves: libvis' Image<T> inside the reference tree, a stand-in elsewhere.  The size has to match like there.
  template <typename Img, typename = decltype(std::declval<const Img&>().stride()), typename = decltype(std::declval<const Img&>().data())>
  void UploadAsync(cudaStream_t stream, const Img& data) {
    check_image_size(data);
    SMX_SHIM_CHECK(smx_buffer_upload(handle_, stream, data.data(), (size_t)data.stride()));
  }
  template <typename Img, typename = decltype(std::declval<Img&>().stride()), typename = decltype(std::declval<Img&>().data())>
  void DownloadAsync(cudaStream_t stream, Img* data) const {
    check_image_size(*data);
    SMX_SHIM_CHECK(smx_buffer_download(handle_, stream, data->data(), (size_t)data->stride()));
  }
  void Clear(T value, cudaStream_t stream) { SMX_SHIM_CHECK(smx_buffer_clear(handle_, stream, &value)); }
  void SetTo(const CUDABuffer<T>& other, cudaStream_t stream) { SMX_SHIM_CHECK(smx_buffer_set_to(handle_, other.handle_, stream)); }

  int width() const { return data_.width_; }
  int height() const { return data_.height_; }
  int Size() const { return (int)(data_.pitch_ * data_.height_); }
  const CUDABuffer_<T>& ToCUDA() const { return data_; }
  CUDABuffer_<T>& ToCUDA() { return data_; }

 private:
  template <typename Img>
  void check_image_size(const Img& image) const {   // (the reference: CHECK_EQ on width and height, cuda_buffer_inl.h)
    if ((int)image.width() != data_.width_ || (int)image.height() != data_.height_) {
      std::fprintf(stderr, "FATAL %s:%d: image is %d x %d, the buffer %d x %d\n", __FILE__, __LINE__, (int)image.width(),
                   (int)image.height(), data_.width_, data_.height_);
      std::abort();
    }
  }
  smx_buffer handle_ = nullptr;
  CUDABuffer_<T> data_;
};
template <typename T> using CUDABufferPtr = std::shared_ptr<CUDABuffer<T>>;
template <typename T> using CUDABufferConstPtr = std::shared_ptr<const CUDABuffer<T>>;   // VIS/cuda/cuda_buffer.h:134-135

// ---- APP/cuda_depth_processing.cuh ------------------------------------------------------------------
inline void BilateralFilteringAndDepthCutoffCUDA(cudaStream_t stream, float sigma_xy, float sigma_value_factor,
                                                 u16 value_to_ignore, float radius_factor, u16 max_depth,
                                                 float depth_valid_region_radius, const CUDABuffer_<u16>& input_depth,
                                                 CUDABuffer_<u16>* output_depth) {
  SMX_SHIM_CHECK(smx_bilateral_filtering_and_depth_cutoff(stream, sigma_xy, sigma_value_factor, value_to_ignore,
                                                          radius_factor, max_depth, depth_valid_region_radius,
                                                          input_depth.desc(), output_depth->desc()));
}

// all-must-agree overload (cuda_depth_processing.cu:229-285)
template <int count, typename DepthT>
void OutlierDepthMapFusionCUDA(cudaStream_t stream, float tolerance, const CUDABuffer_<DepthT>& input_depth,
                               float depth_fx, float depth_fy, float depth_cx, float depth_cy,
                               const CUDABuffer_<DepthT>** other_depths, const CUDAMatrix3x4* others_TR_reference,
                               CUDABuffer_<u16>* output_depth) {
  static_assert(sizeof(DepthT) == 2, "u16 depth only, as instantiated by the reference");
  smx_buffer_desc others[count - 1];
  float T[(count - 1) * 12];
  for (int i = 0; i < count - 1; ++i) {
    others[i] = *other_depths[i]->desc();
    for (int k = 0; k < 12; ++k) T[12 * i + k] = others_TR_reference[i].m[k];
  }
  SMX_SHIM_CHECK(smx_outlier_depth_map_fusion(stream, count - 1, -1, tolerance, input_depth.desc(), depth_fx, depth_fy,
                                              depth_cx, depth_cy, others, T, output_depth->desc()));
}
// counting overload (cu:399-455)
template <int count, typename DepthT>
void OutlierDepthMapFusionCUDA(cudaStream_t stream, int required_count, float tolerance,
                               const CUDABuffer_<DepthT>& input_depth, float depth_fx, float depth_fy, float depth_cx,
                               float depth_cy, const CUDABuffer_<DepthT>** other_depths,
                               const CUDAMatrix3x4* others_TR_reference, CUDABuffer_<u16>* output_depth) {
  smx_buffer_desc others[count - 1];
  float T[(count - 1) * 12];
  for (int i = 0; i < count - 1; ++i) {
    others[i] = *other_depths[i]->desc();
    for (int k = 0; k < 12; ++k) T[12 * i + k] = others_TR_reference[i].m[k];
  }
  SMX_SHIM_CHECK(smx_outlier_depth_map_fusion(stream, count - 1, required_count, tolerance, input_depth.desc(), depth_fx,
                                              depth_fy, depth_cx, depth_cy, others, T, output_depth->desc()));
}

// BilateralFilteringAndDepthCutoffCUDA (value_to_ignore 0) + OutlierDepthMapFusionCUDA<count, u16> as the reference's caller
// chains them (APP/main.cc:1015-1115), one launch where the library can fuse them (smx_bilateral_outlier_fusion); required_count
// < 0 selects the all-must-agree overload.  `scratch` is the intermediate image of the two-call form.
template <int count>
void BilateralFilteringAndOutlierFusionCUDA(cudaStream_t stream, float sigma_xy, float sigma_value_factor, float radius_factor,
                                            u16 max_depth, float depth_valid_region_radius, const CUDABuffer_<u16>& input_depth,
                                            int required_count, float tolerance, float depth_fx, float depth_fy, float depth_cx,
                                            float depth_cy, const CUDABuffer_<u16>** other_depths,
                                            const CUDAMatrix3x4* others_TR_reference, CUDABuffer_<u16>* scratch,
                                            CUDABuffer_<u16>* output_depth) {
  smx_buffer_desc others[count - 1];
  float T[(count - 1) * 12];
  for (int i = 0; i < count - 1; ++i) {
    others[i] = *other_depths[i]->desc();
    for (int k = 0; k < 12; ++k) T[12 * i + k] = others_TR_reference[i].m[k];
  }
  SMX_SHIM_CHECK(smx_bilateral_outlier_fusion(stream, sigma_xy, sigma_value_factor, radius_factor, max_depth,
                                              depth_valid_region_radius, input_depth.desc(), count - 1, required_count, tolerance,
                                              depth_fx, depth_fy, depth_cx, depth_cy, others, T, scratch->desc(),
                                              output_depth->desc()));
}

// Image<u16>::DownscaleUsingMedianWhileExcluding (libvis image.h:1003-1053, --pyramid_level's depth image) on device
// buffers; the output buffer's size selects the source blocks.
inline void DownscaleUsingMedianWhileExcludingCUDA(cudaStream_t stream, u16 value_to_ignore,
                                                   const CUDABuffer_<u16>& input, CUDABuffer_<u16>* output) {
  SMX_SHIM_CHECK(smx_downscale_using_median_while_excluding(stream, value_to_ignore, input.desc(), output->desc()));
}

// ImagePyramid(color_frame, pyramid_level) (libvis image_cache.h:203-275 over Image<Vec3u8>::DownscaleToHalfSize,
// image.h:929-948 -- --pyramid_level's colour image, APP/main.cc:973-981) on device buffers.
inline void ColorImagePyramidCUDA(cudaStream_t stream, int pyramid_level, const CUDABuffer_<Vec3u8>& input,
                                  CUDABuffer_<Vec3u8>* output) {
  SMX_SHIM_CHECK(smx_color_image_pyramid(stream, pyramid_level, input.desc(), output->desc()));
}

// MedianFilterAndDensifyDepthMap (APP/main.cc:206-252) is a CPU function in the reference; its TODO (main.cc:928) asks
// for this: the same filter on device buffers, ahead of the bilateral filter.
inline void MedianFilterAndDensifyDepthMapCUDA(cudaStream_t stream, const CUDABuffer_<u16>& input_depth,
                                               CUDABuffer_<u16>* output_depth) {
  SMX_SHIM_CHECK(smx_median_filter_and_densify_depth_map(stream, input_depth.desc(), output_depth->desc()));
}

template <typename DepthT>
void ErodeDepthMapCUDA(cudaStream_t stream, int radius, const CUDABuffer_<DepthT>& input_depth,
                       CUDABuffer_<DepthT>* output_depth) {
  SMX_SHIM_CHECK(smx_erode_depth_map(stream, radius, input_depth.desc(), output_depth->desc()));
}
template <typename DepthT>
void CopyWithoutBorderCUDA(cudaStream_t stream, const CUDABuffer_<DepthT>& input_depth, CUDABuffer_<DepthT>* output_depth) {
  SMX_SHIM_CHECK(smx_copy_without_border(stream, input_depth.desc(), output_depth->desc()));
}
inline void ComputeNormalsAndDropBadPixelsCUDA(cudaStream_t stream, float observation_angle_threshold_deg,
                                               float depth_scaling, float depth_fx, float depth_fy, float depth_cx,
                                               float depth_cy, const CUDABuffer_<u16>& in_depth,
                                               CUDABuffer_<u16>* out_depth, CUDABuffer_<float2_>* out_normals) {
  SMX_SHIM_CHECK(smx_compute_normals_and_drop_bad_pixels(stream, observation_angle_threshold_deg, depth_scaling, depth_fx,
                                                         depth_fy, depth_cx, depth_cy, in_depth.desc(), out_depth->desc(),
                                                         out_normals->desc()));
}
inline void ComputePointRadiiAndRemoveIsolatedPixelsCUDA(cudaStream_t stream, float point_radius_extension_factor,
                                                         float point_radius_clamp_factor, float depth_scaling,
                                                         float depth_fx, float depth_fy, float depth_cx, float depth_cy,
                                                         const CUDABuffer_<u16>& depth_buffer,
                                                         CUDABuffer_<float>* radius_buffer, CUDABuffer_<u16>* out_depth) {
  SMX_SHIM_CHECK(smx_compute_point_radii_and_remove_isolated_pixels(
      stream, point_radius_extension_factor, point_radius_clamp_factor, depth_scaling, depth_fx, depth_fy, depth_cx,
      depth_cy, depth_buffer.desc(), radius_buffer->desc(), out_depth->desc()));
}

// Not in the reference: the last three preprocessing calls of its frame loop (APP/main.cc:1128-1191: erosion or, for
// radius 0, the border copy; normals; radii) as one launch with the intermediate images in LDS -- the same final depth,
// normals and radii.
inline void ErodeNormalsRadiiCUDA(cudaStream_t stream, int erosion_radius, float observation_angle_threshold_deg,
                                  float point_radius_extension_factor, float point_radius_clamp_factor,
                                  float depth_scaling, float depth_fx, float depth_fy, float depth_cx, float depth_cy,
                                  const CUDABuffer_<u16>& in_depth, CUDABuffer_<u16>* out_depth,
                                  CUDABuffer_<float2_>* out_normals, CUDABuffer_<float>* radius_buffer,
                                  smx_event done = nullptr) {   // (done: the launch's completion event, see smx.h)
  SMX_SHIM_CHECK(smx_erode_normals_radii_signal(stream, erosion_radius, observation_angle_threshold_deg,
                                                point_radius_extension_factor, point_radius_clamp_factor, depth_scaling, depth_fx,
                                                depth_fy, depth_cx, depth_cy, in_depth.desc(), out_depth->desc(),
                                                out_normals->desc(), radius_buffer->desc(), done));
}

// ---- APP/cuda_surfels_cpu.h ---------------------------------------------------------------------------
// (SMX_SHIM_PAGELOCKED_SURFEL_BUFFERS -- define it in front of this header -- takes the eight arrays from page-locked
// memory (smx_host_alloc) instead of new[]: the eight row copies of TransferAllToCPU then run asynchronously to the host,
// as cudaMemcpyAsync only does from and to page-locked memory, and at the full PCIe rate.  Off by default: the
// reference allocates with new[] (cuda_surfels_cpu.h:42-51), and page-locking 8 x max_surfel_count x 4 bytes per buffer
// set is a decision for the application.)
struct CUDASurfelBuffersCPU {
  explicit CUDASurfelBuffersCPU(usize max_surfel_count) {
    surfel_x_buffer = alloc<float>(max_surfel_count);
    surfel_y_buffer = alloc<float>(max_surfel_count);
    surfel_z_buffer = alloc<float>(max_surfel_count);
    surfel_radius_squared_buffer = alloc<float>(max_surfel_count);
    surfel_normal_x_buffer = alloc<float>(max_surfel_count);
    surfel_normal_y_buffer = alloc<float>(max_surfel_count);
    surfel_normal_z_buffer = alloc<float>(max_surfel_count);
    surfel_last_update_stamp_buffer = alloc<u32>(max_surfel_count);
  }
  ~CUDASurfelBuffersCPU() {
    release(surfel_x_buffer); release(surfel_y_buffer); release(surfel_z_buffer);
    release(surfel_radius_squared_buffer); release(surfel_normal_x_buffer); release(surfel_normal_y_buffer);
    release(surfel_normal_z_buffer); release(surfel_last_update_stamp_buffer);
  }
  CUDASurfelBuffersCPU(const CUDASurfelBuffersCPU&) = delete;
  CUDASurfelBuffersCPU& operator=(const CUDASurfelBuffersCPU&) = delete;
  u32 frame_index = 0;
  usize surfel_count = 0;
  float* surfel_x_buffer;
  float* surfel_y_buffer;
  float* surfel_z_buffer;
  float* surfel_radius_squared_buffer;
  float* surfel_normal_x_buffer;
  float* surfel_normal_y_buffer;
  float* surfel_normal_z_buffer;
  u32* surfel_last_update_stamp_buffer;
 private:
#ifdef SMX_SHIM_PAGELOCKED_SURFEL_BUFFERS
  template <typename T> static T* alloc(usize n) {
    void* p = nullptr;
    SMX_SHIM_CHECK(smx_host_alloc(&p, (n ? n : 1) * sizeof(T), /*write_combined*/ 0));   // (the mesher reads these arrays)
    return static_cast<T*>(p);
  }
  template <typename T> static void release(T* p) { (void)smx_host_free(p); }
#else
  template <typename T> static T* alloc(usize n) { return new T[n]; }
  template <typename T> static void release(T* p) { delete[] p; }
#endif
};

class CUDASurfelsCPU {
 public:
  explicit CUDASurfelsCPU(usize max_surfel_count)
      : write_buffers_(new CUDASurfelBuffersCPU(max_surfel_count)),
        read_buffers_(new CUDASurfelBuffersCPU(max_surfel_count)) {}
  ~CUDASurfelsCPU() { delete write_buffers_; delete read_buffers_; }
  void LockWriteBuffers() { write_buffers_lock_.lock(); }
  void UnlockWriteBuffers() { debug_wrote_data_ = true; write_buffers_lock_.unlock(); }
  void WaitForLockAndSwapBuffers() {
    std::unique_lock<std::mutex> lock(write_buffers_lock_);
    if (!debug_wrote_data_) {
      std::fprintf(stderr, "FATAL: Trying to swap the CUDASurfelsCPU buffers, but no data was written. "
                           "Possible multi-threading bug!\n");
      std::abort();
    }
    std::swap(write_buffers_, read_buffers_);
    debug_wrote_data_ = false;
  }
  CUDASurfelBuffersCPU* write_buffers() { return write_buffers_; }
  const CUDASurfelBuffersCPU& read_buffers() const { return *read_buffers_; }
 private:
  bool debug_wrote_data_ = false;
  std::mutex write_buffers_lock_;
  CUDASurfelBuffersCPU* write_buffers_;
  CUDASurfelBuffersCPU* read_buffers_;
};

// ---- APP/cuda_surfel_reconstruction.h -----------------------------------------------------------------
class CUDASurfelReconstruction {
 public:
  // The three cudaGraphicsResource_t arguments and the render window of the reference's constructor are
  // viewer plumbing (OpenGL interop); pass nullptr.
  // device_id (an addition): the GPU the object lives on, -1 = the calling thread's current device (as in the reference).
  // (templates, so that APP/main.cc:835-837 compiles as it stands: whatever the caller's cudaGraphicsResource_t and
  // shared_ptr<SurfelMeshingRenderWindow> are, they are accepted and ignored)
  template <typename R1 = void*, typename R2 = void*, typename R3 = void*, typename Window = void*>
  CUDASurfelReconstruction(usize max_surfel_count, const PinholeCamera4f& depth_camera, const R1& = R1(),
                           const R2& = R2(), const R3& = R3(), const Window& = Window(), int device_id = -1) {
    const float* p = depth_camera.parameters();
    SMX_SHIM_CHECK(smx_recon_create((uint32_t)max_surfel_count, depth_camera.width(), depth_camera.height(), p[0], p[1],
                                    p[2], p[3], device_id, &handle_));
  }
  CUDASurfelReconstruction(const CUDASurfelReconstruction&) = delete;
  ~CUDASurfelReconstruction() { SMX_SHIM_CHECK(smx_recon_destroy(handle_)); }

  // Pose: any type with matrix3x4() returning something with (row, col) access -- Sophus::SE3f in the reference
  // (.h:59-77; the reference takes its inverse() on the host, cc:144: done inside libsmx here) or the SE3f above.
  template <typename Pose>
  void Integrate(cudaStream_t stream, u32 frame_index, float depth_scaling, CUDABuffer<u16>* depth_buffer,
                 const CUDABuffer<float2_>& normals_buffer, const CUDABuffer<float>& radius_buffer,
                 const CUDABuffer<Vec3u8>& color_buffer, const Pose& global_T_local, float sensor_noise_factor,
                 float max_surfel_confidence, float regularizer_weight, int regularization_frame_window_size,
                 bool do_blending, int measurement_blending_radius,
                 int regularization_iterations_per_integration_iteration,
                 float radius_factor_for_regularization_neighbors, float normal_compatibility_threshold_deg,
                 int surfel_integration_active_window_size) {
    smx_integrate_params p;
    p.sensor_noise_factor = sensor_noise_factor;
    p.max_surfel_confidence = max_surfel_confidence;
    p.regularizer_weight = regularizer_weight;
    p.regularization_frame_window_size = regularization_frame_window_size;
    p.do_blending = do_blending ? 1 : 0;
    p.measurement_blending_radius = measurement_blending_radius;
    p.regularization_iterations_per_integration_iteration = regularization_iterations_per_integration_iteration;
    p.radius_factor_for_regularization_neighbors = radius_factor_for_regularization_neighbors;
    p.normal_compatibility_threshold_deg = normal_compatibility_threshold_deg;
    p.surfel_integration_active_window_size = surfel_integration_active_window_size;
    last_stream_ = stream;
    const CUDAMatrix3x4 global_T_local_3x4(global_T_local.matrix3x4());
    SMX_SHIM_CHECK(smx_recon_integrate(handle_, stream, frame_index, depth_scaling, depth_buffer->ToCUDA().desc(),
                                       normals_buffer.ToCUDA().desc(), radius_buffer.ToCUDA().desc(),
                                       color_buffer.ToCUDA().desc(), global_T_local_3x4.m, &p));
  }
  void Regularize(cudaStream_t stream, u32 frame_index, float regularizer_weight,
                  float radius_factor_for_regularization_neighbors, int regularization_frame_window_size) {
    SMX_SHIM_CHECK(smx_recon_regularize(handle_, stream, frame_index, regularizer_weight,
                                        radius_factor_for_regularization_neighbors, regularization_frame_window_size));
  }
  // Requires buffers->LockWriteBuffers() to be held (APP/main.cc:1261-1264).
  void TransferAllToCPU(cudaStream_t stream, u32 frame_index, CUDASurfelsCPU* buffers) {
    CUDASurfelBuffersCPU* b = buffers->write_buffers();
    smx_surfel_buffers_cpu pod = {frame_index, 0, b->surfel_x_buffer, b->surfel_y_buffer, b->surfel_z_buffer,
                                  b->surfel_radius_squared_buffer, b->surfel_normal_x_buffer,
                                  b->surfel_normal_y_buffer, b->surfel_normal_z_buffer,
                                  b->surfel_last_update_stamp_buffer};
    SMX_SHIM_CHECK(smx_recon_transfer_all_to_cpu(handle_, stream, frame_index, &pod));
    b->frame_index = pod.frame_index;
    b->surfel_count = pod.surfel_count;
  }
  // Not in the reference: the changed-surfel delta for the mesher, see smx.h.
  void SetDeltaTracking(cudaStream_t stream, bool enabled) {
    SMX_SHIM_CHECK(smx_recon_set_delta_tracking(handle_, stream, enabled ? 1 : 0));
  }
  // Fills `delta` (whose vectors are resized to the capacity first) with the slots changed since the previous call.
  inline void TransferChangedToCPU(cudaStream_t stream, u32 frame_index, struct CUDASurfelDeltaCPU* delta);
  // The loop-closure hook the reference describes but does not ship (README.md:152-176, call site main.cc:1194-1200):
  // surfels created at frame c move by the rigid correction frame_T[c] (3x4 row-major each); see smx.h.
  void DeformByCreationFrame(cudaStream_t stream, const float* frame_T, u32 frame_count, const u8* reactivate,
                             u32 frame_index) {
    SMX_SHIM_CHECK(smx_recon_deform_by_creation_frame(handle_, stream, frame_T, frame_count, reactivate, frame_index, 0));
  }
  // Not in the reference: the per-triangle tests of SurfelMeshing::CheckRemeshing
  // (APP/surfel_meshing.cc:590-650) for `count` triangles (3 surfel indices each) against the device map; flag bits in smx.h.
  void CheckTrianglesForRemeshing(cudaStream_t stream, const u32* triangle_indices, u32 count,
                                  float long_edge_total_factor_squared, u8* flags) {
    SMX_SHIM_CHECK(smx_recon_check_triangles(handle_, stream, triangle_indices, count, long_edge_total_factor_squared,
                                             flags, 0));
  }
  void UpdateVisualizationBuffers(cudaStream_t, u32, u32, u32, int, bool, bool, bool, bool) {}  // viewer only
  void ExportVertices(cudaStream_t stream, CUDABuffer<float>* position_buffer, CUDABuffer<u8>* color_buffer) {
    SMX_SHIM_CHECK(smx_recon_export_vertices(handle_, stream, position_buffer->ToCUDA().desc(), color_buffer->ToCUDA().desc()));
  }
  // APP/cuda_surfel_reconstruction.cc:412-429.  Two of the reference's seven columns are ALWAYS 0 here, because their work
  // has no launch of its own: *surfel_merging (the merges are decided in the association kernel and applied by the
  // integration kernel: its time is inside *data_association and *integration) and *new_surfel_creation (the first
  // workgroups of the neighbour-update launch: inside *neighbor_update).  The seven values still add up to the call; a
  // caller that logs per-column sums like APP/main.cc:1511-1530 should fold columns 2 and 6 into their neighbours.
  void GetTimings(float* data_association, float* surfel_merging, float* measurement_blending, float* integration,
                  float* neighbor_update, float* new_surfel_creation, float* regularization) {
    float t[7];
    SMX_SHIM_CHECK(smx_recon_get_timings(handle_, t));
    *data_association = t[0]; *surfel_merging = t[1]; *measurement_blending = t[2]; *integration = t[3];
    *neighbor_update = t[4]; *new_surfel_creation = t[5]; *regularization = t[6];
  }
  // Not in the reference: the same seven times without the wait for the last call -- those of the newest call that is
  // known to be through (smx.h smx_recon_get_timings_nowait); returns that call's 1-based number, 0 = none yet (zeros).
  // For a frame loop that reads the stage times after every Integrate (APP/main.cc:1511) and must not stall its queue.
  uint64_t GetTimingsNoWait(float* data_association, float* surfel_merging, float* measurement_blending, float* integration,
                       float* neighbor_update, float* new_surfel_creation, float* regularization) {
    float t[7];
    uint64_t call = 0;
    SMX_SHIM_CHECK(smx_recon_get_timings_nowait(handle_, t, &call));
    *data_association = t[0]; *surfel_merging = t[1]; *measurement_blending = t[2]; *integration = t[3];
    *neighbor_update = t[4]; *new_surfel_creation = t[5]; *regularization = t[6];
    return call;
  }
  // Unlike the reference these read the device-side counters (they synchronise the last used stream).
  u32 surfel_count() const { u32 a = 0, b = 0; SMX_SHIM_CHECK(smx_recon_counts(handle_, last_stream_, &a, &b)); return a; }
  u32 surfels_size() const { u32 a = 0, b = 0; SMX_SHIM_CHECK(smx_recon_counts(handle_, last_stream_, &a, &b)); return b; }
  // Not in the reference: frame pipelining inside Integrate (on by default, see smx.h smx_recon_set_overlap).
  void SetFramePipelining(bool enabled) { SMX_SHIM_CHECK(smx_recon_set_overlap(handle_, enabled ? 1 : 0)); }
  smx_recon handle() const { return handle_; }

 private:
  smx_recon handle_ = nullptr;
  cudaStream_t last_stream_ = nullptr;
};

// ---- batched counterpart of CompressedOctree::FindNearestSurfelsWithinRadius (APP/octree.h:470-477) ----
// A uniform-grid index on the GPU, queried for many positions at once.  result_* are [query_count][max_result_count]
// host arrays, result_counts [query_count]; order: ascending (distance^2, index).
class SurfelNeighborIndex {
 public:
  explicit SurfelNeighborIndex(int device_id = -1) { SMX_SHIM_CHECK(smx_nn_create(device_id, &handle_)); }
  SurfelNeighborIndex(const SurfelNeighborIndex&) = delete;
  ~SurfelNeighborIndex() { SMX_SHIM_CHECK(smx_nn_destroy(handle_)); }
  // From three host rows (e.g. CUDASurfelBuffersCPU::surfel_{x,y,z}_buffer).
  void Build(cudaStream_t stream, const float* x, const float* y, const float* z, u32 count, float cell_size) {
    SMX_SHIM_CHECK(smx_nn_build(handle_, stream, x, y, z, count, cell_size, 0));
  }
  // Straight from the device-resident map; merged surfels are left out.  A snapshot: rebuild after Integrate.
  void Build(cudaStream_t stream, const CUDASurfelReconstruction& reconstruction, float cell_size) {
    SMX_SHIM_CHECK(smx_recon_build_neighbor_index(reconstruction.handle(), stream, handle_, cell_size));
  }
  void FindNearestSurfelsWithinRadius(cudaStream_t stream, u32 query_count, const float* x, const float* y,
                                      const float* z, const float* radius_squared, int max_result_count,
                                      const u8* surfel_state, u8 skip_mask, float* result_distances_squared,
                                      u32* result_indices, int* result_counts) {
    SMX_SHIM_CHECK(smx_nn_query_batch(handle_, stream, query_count, x, y, z, radius_squared, max_result_count,
                                      surfel_state, skip_mask, 0, result_indices, result_distances_squared,
                                      result_counts, 0));
  }
  // The candidate lists of SurfelMeshing::TriangulateSurfel (APP/surfel_meshing.cc:417-425) for a batch of surfels:
  // ball = radius_factor_squared * the surfel's radius_squared around its position, both read on the device.
  void FindNeighborCandidates(cudaStream_t stream, const CUDASurfelReconstruction& reconstruction,
                              const u32* surfel_indices, u32 count, float radius_factor_squared, int max_result_count,
                              const u8* surfel_state, u8 skip_mask, float* result_distances_squared,
                              u32* result_indices, int* result_counts) {
    SMX_SHIM_CHECK(smx_recon_neighbor_candidates(reconstruction.handle(), stream, handle_, surfel_indices, count,
                                                 radius_factor_squared, max_result_count, surfel_state, skip_mask, 0,
                                                 result_indices, result_distances_squared, result_counts, 0));
  }
  smx_nn handle() const { return handle_; }

 private:
  smx_nn handle_ = nullptr;
};

// The changed-surfel delta (not in the reference): slot indices, ascending, and the eight attributes TransferAllToCPU
// moves, for those slots only.  ApplyTo patches a CUDASurfelBuffersCPU that holds an earlier full transfer.
struct CUDASurfelDeltaCPU {
  explicit CUDASurfelDeltaCPU(usize capacity) : capacity(capacity) {
    surfel_index.resize(capacity); last_update_stamp.resize(capacity);
    for (auto* v : {&x, &y, &z, &radius_squared, &normal_x, &normal_y, &normal_z}) v->resize(capacity);
  }
  void ApplyTo(CUDASurfelBuffersCPU* b) const {
    for (u32 k = 0; k < count; ++k) {
      const u32 i = surfel_index[k];
      b->surfel_x_buffer[i] = x[k]; b->surfel_y_buffer[i] = y[k]; b->surfel_z_buffer[i] = z[k];
      b->surfel_radius_squared_buffer[i] = radius_squared[k];
      b->surfel_normal_x_buffer[i] = normal_x[k]; b->surfel_normal_y_buffer[i] = normal_y[k];
      b->surfel_normal_z_buffer[i] = normal_z[k];
      b->surfel_last_update_stamp_buffer[i] = last_update_stamp[k];
    }
    b->frame_index = frame_index;
    b->surfel_count = surfel_count;
  }
  usize capacity;
  u32 count = 0, frame_index = 0, surfel_count = 0;
  std::vector<u32> surfel_index, last_update_stamp;
  std::vector<float> x, y, z, radius_squared, normal_x, normal_y, normal_z;
};
inline void CUDASurfelReconstruction::TransferChangedToCPU(cudaStream_t stream, u32 frame_index, CUDASurfelDeltaCPU* d) {
  smx_surfel_delta_cpu pod = {(uint32_t)d->capacity, 0, 0, 0, d->surfel_index.data(), d->x.data(), d->y.data(), d->z.data(),
                              d->radius_squared.data(), d->normal_x.data(), d->normal_y.data(), d->normal_z.data(),
                              d->last_update_stamp.data()};
  SMX_SHIM_CHECK(smx_recon_transfer_changed_to_cpu(handle_, stream, frame_index, &pod));
  d->count = pod.count; d->frame_index = pod.frame_index; d->surfel_count = pod.surfel_count;
}

}  // namespace vis
