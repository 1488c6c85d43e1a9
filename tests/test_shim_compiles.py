"""include/smx_shim.hpp is what a maintainer of the reference compiles against: reference-style host code using every
class and free function of the shim builds with the plain host compiler (no hipcc, no HIP headers) and links
against libsmx.so.  Nothing is executed here (no GPU); the GPU tests run the same calls through the native driver."""
import os
import subprocess

from common import ROOT

SRC = r'''
#include <unordered_map>
#include "smx_shim.hpp"
using namespace vis;

// A pose type shaped like Sophus::SE3f over Eigen (column-major storage, (row, col) access, matrix3x4(), inverse(),
// operator*, translation()): what the reference's call sites hand to CUDAMatrix3x4 and Integrate.  The shim has to take
// it as it is (VIS/cuda/cuda_matrix.cuh:67-116, APP/cuda_surfel_reconstruction.h:59-77).
namespace sophus_like {
struct Mat34 {                       // Eigen::Matrix<float, 3, 4>: column-major
  float d[12];
  float operator()(int r, int c) const { return d[3 * c + r]; }
  float& operator()(int r, int c) { return d[3 * c + r]; }
};
struct Vec3 { float d[3]; };
inline Vec3 operator*(float s, const Vec3& v) { return Vec3{{s * v.d[0], s * v.d[1], s * v.d[2]}}; }
struct SE3 {
  Mat34 M;
  SE3() { for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) M(r, c) = (r == c) ? 1.f : 0.f; }
  Mat34 matrix3x4() const { return M; }
  SE3 inverse() const {
    SE3 o;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) o.M(r, c) = M(c, r);
      o.M(r, 3) = -(M(0, r) * M(0, 3) + M(1, r) * M(1, 3) + M(2, r) * M(2, 3));
    }
    return o;
  }
  SE3 operator*(const SE3& b) const {
    SE3 o;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c)
        o.M(r, c) = M(r, 0) * b.M(0, c) + M(r, 1) * b.M(1, c) + M(r, 2) * b.M(2, c) + (c == 3 ? M(r, 3) : 0.f);
    return o;
  }
  Vec3& translation() { return *reinterpret_cast<Vec3*>(&M.d[9]); }
};
}  // namespace sophus_like

// the reference's pose handling around the outlier cull and Integrate (APP/main.cc:1039-1059, 1205-1223), token for token
// apart from the container, once with the Sophus-shaped type and once with the shim's own SE3f
template <typename Pose>
void poses_like_main_cc(cudaStream_t stream, CUDASurfelReconstruction& reconstruction, CUDABuffer<u16>* depth,
                        CUDABuffer<float2_>& normals_buffer, CUDABuffer<float>& radius_buffer,
                        CUDABuffer<Vec3u8>& color_buffer, Pose frame_T_global, Pose global_T_other, float depth_scaling) {
  Pose input_depth_frame_scaled_frame_T_global = frame_T_global;
  input_depth_frame_scaled_frame_T_global.translation() = depth_scaling * input_depth_frame_scaled_frame_T_global.translation();
  std::vector<Pose> global_TR_others(8);
  std::vector<CUDAMatrix3x4> others_TR_reference(8);
  for (int i = 0; i < 8; ++i) {
    global_TR_others[i] = global_T_other;
    global_TR_others[i].translation() = depth_scaling * global_TR_others[i].translation();
    others_TR_reference[i] = CUDAMatrix3x4((input_depth_frame_scaled_frame_T_global * global_TR_others[i]).inverse().matrix3x4());
  }
  reconstruction.Integrate(stream, 7, depth_scaling, depth, normals_buffer, radius_buffer, color_buffer,
                           frame_T_global.inverse(), 0.05f, 5.f, 10.f, 30, false, 0, 1, 2.f, 40.f, 0x7fffffff);
}
template void poses_like_main_cc<sophus_like::SE3>(cudaStream_t, CUDASurfelReconstruction&, CUDABuffer<u16>*, CUDABuffer<float2_>&,
    CUDABuffer<float>&, CUDABuffer<Vec3u8>&, sophus_like::SE3, sophus_like::SE3, float);
template void poses_like_main_cc<SE3f>(cudaStream_t, CUDASurfelReconstruction&, CUDABuffer<u16>*, CUDABuffer<float2_>&,
    CUDABuffer<float>&, CUDABuffer<Vec3u8>&, SE3f, SE3f, float);

// host-only value checks (run by the test: none of this touches the GPU)
int check_pose_conversions() {
  sophus_like::SE3 a;
  const float rot[9] = {0.f, -1.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f};   // 90 degrees about z
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a.M(r, c) = rot[3 * r + c];
  a.M(0, 3) = 1.f; a.M(1, 3) = 2.f; a.M(2, 3) = 3.f;
  const CUDAMatrix3x4 m(a.matrix3x4());   // column-major source -> row-major rows
  const float want[12] = {0.f, -1.f, 0.f, 1.f, 1.f, 0.f, 0.f, 2.f, 0.f, 0.f, 1.f, 3.f};
  for (int i = 0; i < 12; ++i) if (m.m[i] != want[i]) return 1;
  const SE3f b(want);
  const CUDAMatrix3x4 mb(b.matrix3x4());
  for (int i = 0; i < 12; ++i) if (mb.m[i] != want[i]) return 2;
  const CUDAMatrix3x4 inv_a(a.inverse().matrix3x4()), inv_b(b.inverse().matrix3x4());
  for (int i = 0; i < 12; ++i) if (inv_a.m[i] != inv_b.m[i]) return 3;
  const CUDAMatrix3x4 id((b * b.inverse()).matrix3x4());
  const float ident[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  for (int i = 0; i < 12; ++i) if (id.m[i] != ident[i]) return 4;
  SE3f c = b;
  c.translation() = 5000.f * c.translation();
  if (c.matrix3x4()(0, 3) != 5000.f || c.matrix3x4()(2, 3) != 15000.f || c.matrix3x4()(1, 0) != 1.f) return 5;
  return 0;
}

// the reference caller's sequence (APP/main.cc:1015-1267) in the reference's own names
int frame(cudaStream_t stream, CUDASurfelReconstruction& reconstruction, CUDASurfelsCPU& cuda_surfels_cpu,
          CUDABuffer<u16>& depth_buffer, CUDABuffer<u16>& A, CUDABuffer<u16>& B, CUDABuffer<float2_>& normals_buffer,
          CUDABuffer<float>& radius_buffer, CUDABuffer<Vec3u8>& color_buffer, const PinholeCamera4f& depth_camera,
          const SE3f& global_T_frame, u32 frame_index) {
  const float* p = depth_camera.parameters();
  BilateralFilteringAndDepthCutoffCUDA(stream, 3.f, 0.05f, (u16)0, 2.f, (u16)15000, 333.f, depth_buffer.ToCUDA(), &A.ToCUDA());
  const CUDABuffer_<u16>* other_depths[8] = {&depth_buffer.ToCUDA(), &depth_buffer.ToCUDA(), &depth_buffer.ToCUDA(),
      &depth_buffer.ToCUDA(), &depth_buffer.ToCUDA(), &depth_buffer.ToCUDA(), &depth_buffer.ToCUDA(), &depth_buffer.ToCUDA()};
  CUDAMatrix3x4 others_TR_reference[8];
  OutlierDepthMapFusionCUDA<9, u16>(stream, 0.02f, A.ToCUDA(), p[0], p[1], p[2], p[3], other_depths, others_TR_reference, &B.ToCUDA());
  OutlierDepthMapFusionCUDA<9, u16>(stream, 7, 0.02f, A.ToCUDA(), p[0], p[1], p[2], p[3], other_depths, others_TR_reference, &B.ToCUDA());
  ErodeDepthMapCUDA(stream, 2, B.ToCUDA(), &A.ToCUDA());
  CopyWithoutBorderCUDA(stream, B.ToCUDA(), &A.ToCUDA());
  ComputeNormalsAndDropBadPixelsCUDA(stream, 85.f, 5000.f, p[0], p[1], p[2], p[3], A.ToCUDA(), &B.ToCUDA(), &normals_buffer.ToCUDA());
  ComputePointRadiiAndRemoveIsolatedPixelsCUDA(stream, 1.5f, 1e30f, 5000.f, p[0], p[1], p[2], p[3], B.ToCUDA(),
                                               &radius_buffer.ToCUDA(), &A.ToCUDA());
  MedianFilterAndDensifyDepthMapCUDA(stream, depth_buffer.ToCUDA(), &B.ToCUDA());
  DownscaleUsingMedianWhileExcludingCUDA(stream, (u16)0, depth_buffer.ToCUDA(), &B.ToCUDA());
  ColorImagePyramidCUDA(stream, 1, color_buffer.ToCUDA(), &color_buffer.ToCUDA());
  reconstruction.Integrate(stream, frame_index, 5000.f, &A, normals_buffer, radius_buffer, color_buffer, global_T_frame,
                           0.05f, 5.f, 10.f, 30, true, 1, 1, 2.f, 40.f, 0x7fffffff);
  reconstruction.Regularize(stream, frame_index, 10.f, 2.f, 30);
  cuda_surfels_cpu.LockWriteBuffers();
  reconstruction.TransferAllToCPU(stream, frame_index, &cuda_surfels_cpu);
  cuda_surfels_cpu.UnlockWriteBuffers();
  cuda_surfels_cpu.WaitForLockAndSwapBuffers();
  CUDABuffer<float> position_buffer(1, 3 * reconstruction.surfels_size());
  CUDABuffer<u8> color_out(1, 3 * reconstruction.surfels_size());
  reconstruction.ExportVertices(stream, &position_buffer, &color_out);
  float t[7];
  reconstruction.GetTimings(&t[0], &t[1], &t[2], &t[3], &t[4], &t[5], &t[6]);
  const uint64_t timed_call = reconstruction.GetTimingsNoWait(&t[0], &t[1], &t[2], &t[3], &t[4], &t[5], &t[6]);
  // additions (SURVEY.md 8f)
  reconstruction.SetDeltaTracking(stream, true);
  CUDASurfelDeltaCPU delta(1000);
  reconstruction.TransferChangedToCPU(stream, frame_index, &delta);
  delta.ApplyTo(cuda_surfels_cpu.write_buffers());
  SurfelNeighborIndex index;
  index.Build(stream, reconstruction, 0.02f);
  std::vector<u32> ids(64 * delta.count); std::vector<float> d2(64 * delta.count); std::vector<int> counts(delta.count);
  index.FindNeighborCandidates(stream, reconstruction, delta.surfel_index.data(), delta.count, 4.f, 64, nullptr, 0,
                               d2.data(), ids.data(), counts.data());
  u32 tri[3] = {0, 1, 2}; u8 flags[1];
  reconstruction.CheckTrianglesForRemeshing(stream, tri, 1, 16.f, flags);
  float T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  reconstruction.DeformByCreationFrame(stream, T, 1, nullptr, frame_index);
  reconstruction.SetFramePipelining(true);
  return (int)reconstruction.surfel_count();
}
// ---- APP/main.cc:801-812 and :1029-1030 / :1120-1121, token for token (VERDICT r5 #4, #5): the buffers are declared with
// CUDA's global float2 and libvis' Vec3u8, and the debug path downloads into a libvis Image<u16>.  `Image` below stands in
// for VIS/image.h's class (its header needs Eigen / glog / libpng / Qt): same constructor order (width, height), data(),
// stride() in bytes, width(), height().
template <typename T>
class Image {
 public:
  Image(u32 width, u32 height) : width_(width), height_(height), stride_(((width * sizeof(T) + 63) / 64) * 64), buf_(stride_ * height) {}
  T* data() { return reinterpret_cast<T*>(buf_.data()); }
  const T* data() const { return reinterpret_cast<const T*>(buf_.data()); }
  u32 stride() const { return stride_; }
  u32 width() const { return width_; }
  u32 height() const { return height_; }
 private:
  u32 width_, height_, stride_;
  std::vector<u8> buf_;
};
using std::shared_ptr;
using std::unordered_map;
void buffers_like_main_cc(cudaStream_t stream, int width, int height, bool debug_depth_preprocessing) {
  // Allocate CUDA buffers.
  unordered_map<int, u16*> frame_index_to_depth_buffer_pagelocked;
  unordered_map<int, CUDABufferPtr<u16>> frame_index_to_depth_buffer;
  CUDABuffer<u16> filtered_depth_buffer_A(height, width);
  CUDABuffer<u16> filtered_depth_buffer_B(height, width);

  CUDABuffer<float2> normals_buffer(height, width);
  CUDABuffer<float> radius_buffer(height, width);

  Vec3u8* color_buffer_pagelocked;
  Vec3u8* next_color_buffer_pagelocked;
  shared_ptr<CUDABuffer<Vec3u8>> color_buffer(new CUDABuffer<Vec3u8>(height, width));
  shared_ptr<CUDABuffer<Vec3u8>> next_color_buffer(new CUDABuffer<Vec3u8>(height, width));

  std::vector<u16*> depth_buffers_pagelocked_cache;
  std::vector<CUDABufferPtr<u16>> depth_buffers_cache;

  // DEBUG: Show bilateral filtering result.
  if (debug_depth_preprocessing) {
    Image<u16> filtered_depth(width, height);
    filtered_depth_buffer_A.DownloadAsync(stream, &filtered_depth);
  }
  if (debug_depth_preprocessing) {
    Image<u16> filtered_depth(width, height);
    filtered_depth_buffer_B.DownloadAsync(stream, &filtered_depth);
    filtered_depth_buffer_B.UploadAsync(stream, filtered_depth);     // VIS/cuda/cuda_buffer.h:69
  }
  // the remaining members of VIS/cuda/cuda_buffer.h:61-135
  std::vector<u16> host(64 * height);
  filtered_depth_buffer_A.DebugUploadPitched(64 * sizeof(u16), host.data());
  filtered_depth_buffer_A.DebugDownloadPitched(64 * sizeof(u16), host.data());
  CUDABufferConstPtr<Vec3u8> const_color = color_buffer;
  (void)const_color->ToCUDA(); (void)color_buffer_pagelocked; (void)next_color_buffer_pagelocked;
  // Integrate takes the float2 buffer as the reference declares it (APP/cuda_surfel_reconstruction.h:59-77)
  CUDASurfelReconstruction* reconstruction = nullptr;
  if (debug_depth_preprocessing && width < 0) {
    // APP/main.cc:835-838, token for token: three cudaGraphicsResource_t and the render window (viewer plumbing: ignored)
    struct cudaGraphicsResource; typedef cudaGraphicsResource* cudaGraphicsResource_t;
    struct SurfelMeshingRenderWindow {};
    cudaGraphicsResource_t vertex_buffer_resource = nullptr, neighbor_index_buffer_resource = nullptr, normal_vertex_buffer_resource = nullptr;
    shared_ptr<SurfelMeshingRenderWindow> render_window;
    const float params[4] = {525.f, 525.f, 320.f, 240.f};
    const PinholeCamera4f depth_camera(width, height, params);
    usize max_surfel_count = 1000;
    CUDASurfelReconstruction reconstruction(
        max_surfel_count, depth_camera, vertex_buffer_resource,
        neighbor_index_buffer_resource, normal_vertex_buffer_resource, render_window);
    CUDASurfelsCPU cuda_surfels_cpu_buffers(max_surfel_count);
  }
  if (reconstruction)
    reconstruction->Integrate(stream, 0, 5000.f, &filtered_depth_buffer_A, normals_buffer, radius_buffer, *color_buffer, SE3f(),
                              0.05f, 5.f, 10.f, 30, true, 1, 1, 2.f, 40.f, 0x7fffffff);
}
int main(int argc, char**) { return argc > 1 ? check_pose_conversions() : 0; }
'''


def test_reference_style_host_code_compiles_and_links(tmp_path):
    from surfelmeshing_amd import _lib, build
    build.build(verbose=False)
    src = tmp_path / "caller.cc"
    src.write_text(SRC)
    exe = tmp_path / "caller"
    lib_dir = os.path.dirname(_lib.SO_PATH)
    r = subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-Wno-unused-variable", "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe), "-L", lib_dir, "-l:libsmx.so", "-Wl,-rpath," + lib_dir,
                        "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the pose conversions are pure host code: run them (libsmx.so is only loaded, no entry point is called)
    r = subprocess.run([str(exe), "check"], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    # ... and with the shim's option that takes CUDASurfelBuffersCPU from page-locked memory (smx_host_alloc)
    r = subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-Wno-unused-variable", "-DSMX_SHIM_PAGELOCKED_SURFEL_BUFFERS",
                        "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe) + "_pl", "-L", lib_dir, "-l:libsmx.so",
                        "-Wl,-rpath," + lib_dir, "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # (GetTimingsNoWait is in the GetTimings block of the caller above)
    # ... and inside a HIP translation unit float2 is HIP's own vector type (what a maintainer building main.cc with hipcc gets);
    # SMX_SHIM_NO_VEC_TYPES takes the caller's Vec3u8 (libvis' Eigen typedef inside the reference tree)
    hip_src = tmp_path / "caller_hip.cc"
    # ... with a kernel of the maintainer's own that takes CUDABuffer_<T> by value and uses the reference's accessors
    # (VIS/cuda/cuda_buffer.cuh:58-96)
    own_kernel = (
        "template <typename T> __global__ void k_scale(vis::CUDABuffer_<T> in, vis::CUDABuffer_<float> out, float s) {\n"
        "  const unsigned x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;\n"
        "  if (x < (unsigned)in.width() && y < (unsigned)in.height()) out(y, x) = s * (float)in(make_uint2(x, y));\n"
        "}\n"
        "void launch_own(hipStream_t st, vis::CUDABuffer<vis::u16>& a, vis::CUDABuffer<float>& b) {\n"
        "  hipLaunchKernelGGL(k_scale<vis::u16>, dim3((a.width() + 63) / 64, a.height()), dim3(64), 0, st, a.ToCUDA(), b.ToCUDA(), 0.0002f);\n"
        "}\n")
    hip_src.write_text("#include <hip/hip_runtime.h>\nnamespace vis { struct Vec3u8 { unsigned char v[3]; }; }\n" + SRC + own_kernel)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-Wall", "-Werror", "-Wno-unused-variable",
                        "-Wno-unused-result", "-DSMX_SHIM_NO_VEC_TYPES", "-x", "hip", "-c", "-I", os.path.join(ROOT, "include"), str(hip_src),
                        "-o", str(tmp_path / "caller_hip.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
