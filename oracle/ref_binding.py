"""TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/libsmx_ref.so -- the reference's own kernels (built by
oracle/ref_build.py from the sources under /root/reference) behind the same numpy-level interface as oracle/binding.py,
so that a test can run the CPU oracle and the reference's real kernels on the same inputs.

Needs a GPU (the reference's kernels are GPU kernels; here they run on the MI355X through hipcc).  Import torch before
this module on a ROCm-PyTorch installation, as for libsmx.so.
"""
import ctypes as C
import os

import numpy as np

from .binding import IntegrateParams

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "_ref", "libsmx_ref.so")
_lib = None


def available():
    return os.path.exists(SO_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(SO_PATH)
    return _lib


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("reference harness: %s failed (%d)" % (what, rc))


def se3_inverse(m):
    """R^T, -(R^T t) in float32 with the oracle's operation order (oracle/smx_oracle_recon.c se3_inverse)."""
    m = np.asarray(m, np.float32).reshape(12)
    o = np.zeros(12, np.float32)
    for i in range(3):
        o[4 * i + 0], o[4 * i + 1], o[4 * i + 2] = m[0 + i], m[4 + i], m[8 + i]
        s = np.float32(o[4 * i + 0] * m[3])
        s = np.float32(s + np.float32(o[4 * i + 1] * m[7]))
        s = np.float32(s + np.float32(o[4 * i + 2] * m[11]))
        o[4 * i + 3] = -s
    return o


# ---- depth preprocessing (same signatures as oracle/binding.py) ---------------------------------------------------
def bilateral_filter_and_cutoff(depth, sigma_xy=3.0, sigma_value_factor=0.05, value_to_ignore=0,
                                radius_factor=2.0, max_depth=15000, depth_valid_region_radius=333.0):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    out = np.zeros_like(depth)
    _check(lib().ref_bilateral(C.c_int(w), C.c_int(h), _p(depth), C.c_float(sigma_xy), C.c_float(sigma_value_factor),
                               C.c_uint16(value_to_ignore), C.c_float(radius_factor), C.c_uint16(max_depth),
                               C.c_float(depth_valid_region_radius), _p(out)), "bilateral")
    return out


def outlier_depth_map_fusion(depth, others, others_TR_reference, fx, fy, cx, cy, tolerance=0.02, required_count=-1):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    others = [_c(o, np.uint16) for o in others]
    T = _c(np.asarray(others_TR_reference, np.float32).reshape(len(others), 12), np.float32)
    ptrs = (C.c_void_p * len(others))(*[o.ctypes.data for o in others])
    out = np.zeros_like(depth)
    _check(lib().ref_outlier_fusion(C.c_int(w), C.c_int(h), C.c_int(len(others)), C.c_int(required_count),
                                    C.c_float(tolerance), _p(depth), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                    C.c_float(cy), ptrs, _p(T), _p(out)), "outlier fusion")
    return out


def erode_depth_map(depth, radius):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    out = np.zeros_like(depth)
    _check(lib().ref_erode(C.c_int(w), C.c_int(h), C.c_int(radius), _p(depth), _p(out)), "erode")
    return out


def compute_normals_and_drop_bad_pixels(depth, fx, fy, cx, cy, observation_angle_threshold_deg=85.0,
                                        depth_scaling=5000.0):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    out = np.zeros_like(depth)
    normals = np.zeros((h, w, 2), np.float32)
    _check(lib().ref_normals(C.c_int(w), C.c_int(h), C.c_float(observation_angle_threshold_deg), C.c_float(depth_scaling),
                             C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), _p(depth), _p(out),
                             _p(normals)), "normals")
    return out, normals


def compute_point_radii_and_remove_isolated_pixels(depth, fx, fy, cx, cy, point_radius_extension_factor=1.5,
                                                   point_radius_clamp_factor=float("inf"), depth_scaling=5000.0,
                                                   radius_init=None):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    out = np.zeros_like(depth)
    radius = np.zeros((h, w), np.float32) if radius_init is None else _c(radius_init, np.float32).copy()
    _check(lib().ref_radii(C.c_int(w), C.c_int(h), C.c_float(point_radius_extension_factor),
                           C.c_float(point_radius_clamp_factor), C.c_float(depth_scaling), C.c_float(fx), C.c_float(fy),
                           C.c_float(cx), C.c_float(cy), _p(depth), _p(radius), _p(out)), "radii")
    return out, radius


# ---- CUDASurfelReconstruction around the reference's kernels ------------------------------------------------------
class Recon:
    ROWS = 25

    def __init__(self, max_surfels, width, height, fx, fy, cx, cy):
        self.width, self.height, self.max_surfels = width, height, max_surfels
        self._r = C.c_void_p()
        _check(lib().ref_recon_create(C.c_uint32(max_surfels), C.c_int(width), C.c_int(height), C.c_float(fx),
                                      C.c_float(fy), C.c_float(cx), C.c_float(cy), C.byref(self._r)), "create")

    def close(self):
        if self._r:
            lib().ref_recon_destroy(self._r)
            self._r = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counts(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().ref_recon_counts(self._r, C.byref(a), C.byref(b), C.byref(c))
        return {"surfels_size": a.value, "merge_count": b.value, "n_new": c.value}

    def integrate(self, frame_index, depth_scaling, depth, normals, radius, color, global_T_local, params=None):
        assert depth.dtype == np.uint16 and depth.flags.c_contiguous and depth.flags.writeable
        params = params or IntegrateParams.defaults()
        normals, radius, color = _c(normals, np.float32), _c(radius, np.float32), _c(color, np.uint8)
        G = _c(np.asarray(global_T_local, np.float32).reshape(12), np.float32)
        L = se3_inverse(G)
        _check(lib().ref_recon_integrate(self._r, C.c_uint32(frame_index), C.c_float(depth_scaling), _p(depth),
                                         _p(normals), _p(radius), _p(color), _p(G), _p(L), C.byref(params)), "integrate")

    def last_integrate_ms(self):
        """Device time of the last integrate() (clears .. regulariser; the reference's two host syncs included)."""
        lib().ref_recon_last_integrate_ms.restype = C.c_float
        return float(lib().ref_recon_last_integrate_ms(self._r))

    def regularize(self, frame_index, regularizer_weight=10.0, radius_factor=2.0, window=30):
        _check(lib().ref_recon_regularize(self._r, C.c_uint32(frame_index), C.c_float(regularizer_weight),
                                          C.c_float(radius_factor), C.c_int(window)), "regularize")

    def surfels(self, count=None):
        n = self.counts()["surfels_size"] if count is None else count
        rows = np.zeros((self.ROWS, n), np.float32)
        if n:
            _check(lib().ref_recon_download_surfels(self._r, _p(rows), C.c_uint32(n)), "download")
        return rows

    def upload_surfels(self, rows, merge_count=0):
        rows = _c(rows, np.float32)
        _check(lib().ref_recon_upload_surfels(self._r, _p(rows), C.c_uint32(rows.shape[1]), C.c_uint32(merge_count)),
               "upload")

    def export_vertices(self):
        n = self.counts()["surfels_size"]
        pos, col = np.zeros(3 * n, np.float32), np.zeros(3 * n, np.uint8)
        if n:
            _check(lib().ref_recon_export_vertices(self._r, _p(pos), _p(col)), "export")
        return pos, col

    def scratch(self):
        out = {}
        for which, (name, dt) in enumerate([("supporting", np.uint32), ("support_counts", np.uint32),
                                            ("depth_sums_f", np.float32), ("conflicting", np.uint32),
                                            ("first_depth", np.float32)]):
            a = np.zeros((self.height, self.width), dt)
            _check(lib().ref_recon_download_scratch(self._r, C.c_int(which), _p(a)), "scratch")
            out[name] = a
        return out
