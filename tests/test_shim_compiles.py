"""include/smx_shim.hpp is what a maintainer of the reference compiles against: reference-style host code using every
class and free function of the shim builds with the plain host compiler (no hipcc, no HIP headers) and links
against libsmx.so.  Nothing is executed here (no GPU); the GPU tests run the same calls through the native driver."""
import os
import subprocess

from common import ROOT

SRC = r'''
#include "smx_shim.hpp"
using namespace vis;

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
int main() { return 0; }
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
