#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Builds oracle/_ref/libsmx_ref.so: the reference's own two CUDA kernel files, compiled by
hipcc for gfx950 FROM WHERE THEY LIE under /root/reference (nothing is copied into this repository), plus
oracle/ref_harness.cpp, the host side they need.  oracle/ref_shim/ holds the four small headers that let hipcc read
the CUDA sources (cuda_runtime.h, math_constants.h, two cub headers mapped to hipCUB).

The product never loads this library.  It exists to pin the CPU oracle against the code it restates
(tests/test_gpu_reference_pin.py).  /root/reference does not exist on the GPU box: the .so is built in the authoring
container and travels with the repository snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored).

Arithmetic: compiled WITHOUT fast-math and with -ffp-contract=off, i.e. the reference's code under the IEEE contract
the oracle and the product use (the reference's own build uses -use_fast_math, which no other compiler reproduces).

    python oracle/ref_build.py            # no-op (returns False) when /root/reference is absent
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SMX_REFERENCE_ROOT", "/root/reference")
APP = os.path.join(REF, "applications", "surfel_meshing", "src")
OUT = os.path.join(HERE, "_ref")
SO = os.path.join(OUT, "libsmx_ref.so")
SOURCES = [os.path.join(APP, "surfel_meshing", "cuda_depth_processing.cu"),
           os.path.join(APP, "surfel_meshing", "cuda_surfel_reconstruction_kernels.cu"),
           os.path.join(HERE, "ref_harness.cpp")]
LOGURU = os.path.join(REF, "libvis", "third_party", "loguru", "loguru.cpp")   # the reference's vendored logger


def available():
    return all(os.path.exists(s) for s in SOURCES + [LOGURU])


def build(force=False, verbose=True):
    if not available():
        return False
    os.makedirs(OUT, exist_ok=True)
    newest = max(os.path.getmtime(s) for s in SOURCES + [__file__])
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= newest:
        return True
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip", "-w",
             "-I", os.path.join(HERE, "ref_shim"), "-I", os.path.join(REF, "libvis", "src"),
             "-I", os.path.join(REF, "libvis", "third_party", "loguru"), "-I", APP]
    objs = []
    for s in SOURCES:
        o = os.path.join(OUT, os.path.splitext(os.path.basename(s))[0] + ".o")
        cmd = [hipcc] + flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(o)
    o = os.path.join(OUT, "loguru.o")
    cmd = ["g++", "-O2", "-std=c++14", "-fPIC", "-w", "-DLOGURU_REPLACE_GLOG=1", "-I", os.path.dirname(LOGURU), "-c", LOGURU, "-o", o]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return True


# ---- the three input-side image functions, CPU code -------------------------------------------------
# MedianFilterAndDensifyDepthMap is a free function of APP/main.cc; the two downscaling functions are members of
# libvis' Image<T>, whose header needs Eigen, glog, libpng and Qt (none of them in this image).  All three only use
# width() / height() / row() / operator()(x, y) / SetSize of Image<T>, so their TEXT -- the line ranges below, read from
# where the reference lies -- is pasted at build time into a generated translation unit around a twenty-line stand-in for
# Image<T> (and for the Eigen byte vector the colour pyramid divides and adds).  The generated file lives in oracle/_ref/
# only (git-ignored): nothing of the reference enters the repository.
MAIN_CC = os.path.join(APP, "surfel_meshing", "main.cc")
IMAGE_H = os.path.join(REF, "libvis", "src", "libvis", "image.h")
IMG_SO = os.path.join(OUT, "libsmx_ref_image.so")
IMG_RANGES = {"median": (MAIN_CC, 207, 252, "void MedianFilterAndDensifyDepthMap("),
              "half": (IMAGE_H, 929, 948, "void DownscaleToHalfSize("),
              "downscale_median": (IMAGE_H, 1003, 1053, "void DownscaleUsingMedianWhileExcluding(")}


def _lines(path, first, last, must_start_with):
    rows = open(path).read().split("\n")[first - 1:last]
    assert rows[0].strip().startswith(must_start_with), "reference moved: %s:%d is %r" % (path, first, rows[0])
    assert rows[-1].strip() == "}", "reference moved: %s:%d is %r" % (path, last, rows[-1])
    return "\n".join(rows)


def image_fns_available():
    return os.path.exists(MAIN_CC) and os.path.exists(IMAGE_H)


def build_image_fns(force=False, verbose=True):
    if not image_fns_available():
        return False
    os.makedirs(OUT, exist_ok=True)
    newest = max(os.path.getmtime(x) for x in (MAIN_CC, IMAGE_H, __file__))
    if not force and os.path.exists(IMG_SO) and os.path.getmtime(IMG_SO) >= newest:
        return True
    part = {k: _lines(*v) for k, v in IMG_RANGES.items()}
    src = """// GENERATED by oracle/ref_build.py -- TEST INFRASTRUCTURE, not committed.  The three function bodies below are the
// reference's own text (%s:207-252, %s:929-948 and :1003-1053) around a stand-in for Image<T>.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <vector>
typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32;
using std::vector;
#define CHECK_EQ(a, b) assert((a) == (b))
// Eigen::Matrix<u8, 3, 1> as far as DownscaleToHalfSize uses it: component-wise, in the scalar type
struct Vec3u8 {
  u8 v[3];
  Vec3u8 operator/(int d) const { Vec3u8 r; for (int k = 0; k < 3; ++k) r.v[k] = (u8)(v[k] / d); return r; }
  Vec3u8 operator+(const Vec3u8& o) const { Vec3u8 r; for (int k = 0; k < 3; ++k) r.v[k] = (u8)(v[k] + o.v[k]); return r; }
};
template <typename T>
class Image {   // a view of caller-owned, densely packed rows
 public:
  Image(u32 w, u32 h, T* data) : w_(w), h_(h), d_(data) {}
  u32 width() const { return w_; }
  u32 height() const { return h_; }
  void SetSize(u32 w, u32 h) { assert(w == w_ && h == h_); (void)w; (void)h; }
  T* row(u32 y) { return d_ + (size_t)y * w_; }
  const T* row(u32 y) const { return d_ + (size_t)y * w_; }
  T& operator()(u32 x, u32 y) { return d_[(size_t)y * w_ + x]; }
  const T& operator()(u32 x, u32 y) const { return d_[(size_t)y * w_ + x]; }
%s
%s
 private:
  u32 w_, h_; T* d_;
};
%s
extern "C" {
void ref_median_filter_and_densify(int w, int h, const u16* in, u16* out) {
  Image<u16> a(w, h, const_cast<u16*>(in)), b(w, h, out);
  MedianFilterAndDensifyDepthMap(a, &b);
}
void ref_downscale_using_median_while_excluding(u16 ignore, int w, int h, const u16* in, int ow, int oh, u16* out) {
  Image<u16> a(w, h, const_cast<u16*>(in)), b(ow, oh, out);
  a.DownscaleUsingMedianWhileExcluding(ignore, ow, oh, &b);
}
void ref_downscale_to_half_size_rgb(int w, int h, const u8* in, u8* out) {
  Image<Vec3u8> a(w, h, reinterpret_cast<Vec3u8*>(const_cast<u8*>(in))), b(w / 2, h / 2, reinterpret_cast<Vec3u8*>(out));
  a.DownscaleToHalfSize(&b);
}
}
""" % (MAIN_CC, IMAGE_H, part["half"], part["downscale_median"], part["median"])
    gen = os.path.join(OUT, "ref_image_fns.cpp")
    open(gen, "w").write(src)
    cmd = ["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-ffp-contract=off", "-w", gen, "-o", IMG_SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.remove(gen)   # (only the library stays, like the kernels' objects: the pasted text is not kept around)
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref/libsmx_ref.so", "built" if ok else "NOT built (no reference sources here)")
    ok = build_image_fns(force="--force" in sys.argv)
    print("oracle/_ref/libsmx_ref_image.so", "built" if ok else "NOT built (no reference sources here)")
