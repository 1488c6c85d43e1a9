"""Builds surfelmeshing_amd/libsmx.so with hipcc for gfx950 (in-tree, no JIT cache).

    python -m surfelmeshing_amd.build [--force]

-ffp-contract=off is part of the arithmetic contract: kernels must
not fuse a*b+c, so that results are reproducible bit for bit on an IEEE host.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsmx.so")
SOURCES = ["smx_buffer.hip", "smx_depth.hip", "smx_recon.hip", "smx_nn.hip", "smx_synth.hip", "smx_driver.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    extra = os.environ.get("SMX_EXTRA_FLAGS", "").split()   # experiments only (e.g. -DSMX_EXP=1)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers += [os.path.join(ROOT, "include", h) for h in ("smx.h", "smx_shim.hpp", "smx_driver.h")]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + extra + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
