// smx_common.hpp -- shared host/device helpers of libsmx (gfx950 only).
//
// Arithmetic contract (DESIGN.md "Arithmetic"): the library is compiled with
// -ffp-contract=off, so every a*b+c is an IEEE multiply followed by an IEEE
// add unless __builtin_fmaf is written explicitly; divisions and square roots
// are the correctly rounded ones hipcc emits by default.  Every kernel that
// projects a surfel uses the one project() routine below so that the
// camera-space z it produces is bit-identical across kernels (the reference
// relies on the same property, APP/cuda_surfel_reconstruction_kernels.cu:775,
// 1613, 1885 against :1463).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "smx.h"

namespace smx {

void set_error(const char* fmt, ...);

#define SMX_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      ::smx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return SMX_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

#define SMX_CHECK_ARG(cond)                                                            \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      ::smx::set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__);     \
      return SMX_ERR_INVALID_ARGUMENT;                                                 \
    }                                                                                  \
  } while (0)

#define SMX_LAUNCH_CHECK()                                                             \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      ::smx::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
      return SMX_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

// Objects (smx_recon, smx_nn) remember the device they were created on and make it the calling thread's current
// device for the duration of every entry point, so that one process can drive one object per GPU from one thread
// per GPU -- or from a single thread -- without calling smx_set_device between the calls.  The
// previous device is restored on return.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int device) {
    if (device < 0) return;
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      switched = (err == hipSuccess);
    }
  }
  ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
};
#define SMX_ON_DEVICE(device)                                                          \
  ::smx::DeviceScope smx_device_scope__(device);                                       \
  do {                                                                                 \
    if (smx_device_scope__.err != hipSuccess) {                                        \
      ::smx::set_error("cannot select device %d: %s", (int)(device), hipGetErrorString(smx_device_scope__.err)); \
      return SMX_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

// device_id argument of the create functions: -1 = the calling thread's current device
inline int resolve_device(int32_t device_id, int* out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    set_error("no HIP device available");
    return SMX_ERR_NO_DEVICE;
  }
  int dev = device_id;
  if (dev < 0) SMX_HIP(hipGetDevice(&dev));
  if (dev >= ndev) { set_error("device_id %d out of range (%d devices)", dev, ndev); return SMX_ERR_INVALID_ARGUMENT; }
  *out = dev;
  return SMX_OK;
}

constexpr uint32_t kInvalid = 0xFFFFFFFFu;

// Typed view of smx_buffer_desc for kernels (same layout as CUDABuffer_<T>).
template <typename T>
struct Img {
  T* address;
  int32_t height;
  int32_t width;
  size_t pitch;
  __device__ __forceinline__ T& operator()(int y, int x) const {
    return *reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(address) + (size_t)y * pitch + (size_t)x * sizeof(T));
  }
  __device__ __forceinline__ T* row(int y) const {
    return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(address) + (size_t)y * pitch);
  }
};
template <typename T>
inline Img<T> as_img(const smx_buffer_desc* d) {
  Img<T> i;
  i.address = reinterpret_cast<T*>(d->address);
  i.height = d->height; i.width = d->width; i.pitch = d->pitch;
  return i;
}
template <typename T>
inline Img<T> as_img(const smx_buffer_desc& d) { return as_img<T>(&d); }

struct Mat34 { float m[12]; };  // row-major 3x4, CUDAMatrix3x4 (VIS/cuda/cuda_matrix.cuh:67-116)

struct Vec3 { float x, y, z; };

// CUDAMatrix3x4::operator* -- left-to-right adds
__device__ __forceinline__ Vec3 mul(const Mat34& M, const Vec3& p) {
  Vec3 o;
  o.x = M.m[0] * p.x + M.m[1] * p.y + M.m[2] * p.z + M.m[3];
  o.y = M.m[4] * p.x + M.m[5] * p.y + M.m[6] * p.z + M.m[7];
  o.z = M.m[8] * p.x + M.m[9] * p.y + M.m[10] * p.z + M.m[11];
  return o;
}
__device__ __forceinline__ Vec3 rotate(const Mat34& M, const Vec3& p) {
  Vec3 o;
  o.x = M.m[0] * p.x + M.m[1] * p.y + M.m[2] * p.z;
  o.y = M.m[4] * p.x + M.m[5] * p.y + M.m[6] * p.z;
  o.z = M.m[8] * p.x + M.m[9] * p.y + M.m[10] * p.z;
  return o;
}

// float -> u16 store: truncate toward zero, saturating.
__device__ __forceinline__ uint16_t f2u16(float v) {
  if (!(v > 0.0f)) return 0;
  if (v >= 65535.0f) return 65535;
  return (uint16_t)(int)v;
}

// 2^-32 fixed point used for order-independent (hence deterministic) sums.
__device__ __forceinline__ long long q_from_float(float v) {
  return (long long)((double)v * 4294967296.0);
}
__device__ __forceinline__ float q_to_float(long long s) {
  return (float)((double)s * (1.0 / 4294967296.0));
}

// Regulariser gradient terms: 2^-22 fixed point (0.24 um), clamped to +-16 so that two of them share one 64-bit
// accumulator word and the sum of up to 31 terms cannot leave its 32-bit half (NaN maps to the lower bound).
__device__ __forceinline__ int q22_from_float(float v) {
  // (single precision: the scaling by 2^22 is exact, the same value as the product formed in double precision)
  float d = v * 4194304.0f;
  if (!(d > -67108864.0f)) d = -67108864.0f;
  if (d > 67108864.0f) d = 67108864.0f;
  return (int)d;
}
__device__ __forceinline__ float q22_to_float(long long s) {
  return (float)((double)s * (1.0 / 4194304.0));
}
// word = hi * 2^32 + lo with signed halves: integer sums of such words stay decodable while both half sums fit
// in 32 bits (the borrow of a negative lo is undone by decoding lo first)
__device__ __forceinline__ unsigned long long pack_pair(int hi, int lo) {
  return (unsigned long long)(((long long)hi << 32) + (long long)lo);
}
__device__ __forceinline__ void unpack_pair(long long w, long long& hi, int& lo) {
  lo = (int)(unsigned int)(unsigned long long)w;
  hi = (w - (long long)lo) >> 32;
}

// Deterministic expf (Cody-Waite + degree-6 polynomial with explicit FMAs).
__device__ __forceinline__ float det_expf(float x) {
  if (x < -86.0f) return 0.0f;
  if (x > 88.0f) return __builtin_inff();
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
  int ni = (int)n;
  return y * __uint_as_float((uint32_t)(ni + 127) << 23);
}

inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// Host-side SE3 inverse: R^T, -(R^T t)  (the reference takes it from Sophus,
// APP/cuda_surfel_reconstruction.cc:144).
inline Mat34 se3_inverse(const float* m) {
  Mat34 o;
  for (int i = 0; i < 3; ++i) {
    o.m[4 * i + 0] = m[0 + i]; o.m[4 * i + 1] = m[4 + i]; o.m[4 * i + 2] = m[8 + i];
    o.m[4 * i + 3] = -(o.m[4 * i + 0] * m[3] + o.m[4 * i + 1] * m[7] + o.m[4 * i + 2] * m[11]);
  }
  return o;
}

// Pixel-centre unprojection intrinsics, APP/cuda_surfel_reconstruction_kernels.cc:69-74.
struct Unproj { float fx_inv, fy_inv, cx_inv, cy_inv; };
inline Unproj make_unproj(float fx, float fy, float cx, float cy) {
  Unproj u;
  u.fx_inv = 1.0f / fx; u.fy_inv = 1.0f / fy;
  const float cxp = cx - 0.5f, cyp = cy - 0.5f;
  u.cx_inv = -cxp / fx; u.cy_inv = -cyp / fy;
  return u;
}

}  // namespace smx
