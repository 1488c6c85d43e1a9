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


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref/libsmx_ref.so", "built" if ok else "NOT built (no reference sources here)")
