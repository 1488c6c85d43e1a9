"""ctypes binding of oracle/libsmx_oracle.so (TEST INFRASTRUCTURE ONLY).

The shapes and names mirror oracle/smx_oracle.h; images are dense row-major
numpy arrays (depth u16 [H,W], normals f32 [H,W,2], radius f32 [H,W],
colour u8 [H,W,3]); poses are row-major 3x4 float32.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsmx_oracle.so")

INVALID = 0xFFFFFFFF
ROWS = 25
(ROW_X, ROW_Y, ROW_Z, ROW_SMOOTH_X, ROW_SMOOTH_Y, ROW_SMOOTH_Z, ROW_CONFIDENCE, ROW_RADIUS_SQ,
 ROW_NORMAL_X, ROW_NORMAL_Y, ROW_NORMAL_Z, ROW_GRAD_X, ROW_GRAD_Y, ROW_GRAD_Z,
 ROW_ACCUM_X, ROW_ACCUM_Y, ROW_ACCUM_Z, ROW_CREATION_STAMP, ROW_LAST_UPDATE_STAMP,
 ROW_NEIGHBOR0, ROW_NEIGHBOR1, ROW_NEIGHBOR2, ROW_NEIGHBOR3, ROW_GRAD_COUNT, ROW_COLOR) = range(25)
# rows that are scratch in both implementations and excluded from parity
SCRATCH_ROWS = (11, 12, 13, 14, 15, 16, 23)
SUM_EXACT, SUM_FLOAT_ASCENDING = 0, 1


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


class IntegrateParams(C.Structure):
    _fields_ = [("sensor_noise_factor", C.c_float),
                ("max_surfel_confidence", C.c_float),
                ("regularizer_weight", C.c_float),
                ("regularization_frame_window_size", C.c_int32),
                ("do_blending", C.c_int32),
                ("measurement_blending_radius", C.c_int32),
                ("regularization_iterations_per_integration_iteration", C.c_int32),
                ("radius_factor_for_regularization_neighbors", C.c_float),
                ("normal_compatibility_threshold_deg", C.c_float),
                ("surfel_integration_active_window_size", C.c_int32)]

    @classmethod
    def defaults(cls, **kw):
        # APP/main.cc:323-368
        p = cls(0.05, 5.0, 10.0, 30, 1, 12, 1, 2.0, 40.0, 2147483647)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class _Recon(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("max_surfels", C.c_uint32), ("surfel_count", C.c_uint32), ("merge_count", C.c_uint32),
                ("sum_mode", C.c_int32),
                ("surfels", C.POINTER(C.c_float)), ("grad_acc", C.POINTER(C.c_int64)),
                ("supporting", C.POINTER(C.c_uint32)), ("support_counts", C.POINTER(C.c_uint32)),
                ("depth_sums_f", C.POINTER(C.c_float)), ("depth_sums_q", C.POINTER(C.c_int64)),
                ("conflicting", C.POINTER(C.c_uint32)), ("conflicting_key", C.POINTER(C.c_uint32)),
                ("first_depth", C.POINTER(C.c_float)),
                ("distance_map", C.POINTER(C.c_uint8)), ("new_distance_map", C.POINTER(C.c_uint8)),
                ("deltas", C.POINTER(C.c_float)), ("new_deltas", C.POINTER(C.c_float)),
                ("new_flags", C.POINTER(C.c_uint8)), ("new_indices", C.POINTER(C.c_uint32)),
                ("merge_decision", C.POINTER(C.c_uint8)),
                ("last_n_visible", C.c_uint32), ("last_n_new", C.c_uint32), ("last_n_merged", C.c_uint32),
                ("last_n_recent", C.c_uint32), ("last_n_edges", C.c_uint32),
                ("last_n_integrated", C.c_uint32), ("last_n_replaced", C.c_uint32),
                ("last_n_conflict_hits", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_expf.restype = C.c_float
        L.orc_expf.argtypes = [C.c_float]
        L.orc_recon_create.restype = C.POINTER(_Recon)
        L.orc_recon_create.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]
        L.orc_recon_destroy.argtypes = [C.POINTER(_Recon)]
        L.orc_nn_bruteforce.restype = C.c_int
        _lib = L
    return _lib


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def expf(x):
    x = np.asarray(x, np.float32)
    out = np.empty_like(x)
    f = lib().orc_expf
    flat, o = x.ravel(), out.ravel()
    for i in range(flat.size):
        o[i] = f(float(flat[i]))
    return out


# ---- depth preprocessing -------------------------------------------------------------------
# Row-parallel mode of the per-pixel stages (bilateral, outlier cull, erosion / border copy, normals, radii): every output
# pixel depends on the input images only, so the rows are split into bands, one OS thread per band, each with its own
# thread-local row range (orc_set_row_range) over the same C loops (ctypes releases the GIL during the calls).  Results
# are identical to the single-threaded run.  Used by bench.py's all-host-cores CPU baseline.
_ROW_THREADS = 1
_POOL = None


def set_row_threads(n):
    """Number of row bands / threads of the per-pixel stages (1 = the plain single-threaded loops)."""
    global _ROW_THREADS, _POOL
    n = max(1, int(n))
    if _POOL is not None and n != _ROW_THREADS:
        _POOL.shutdown()
        _POOL = None
    _ROW_THREADS = n


def _rows(height, call):
    """Runs call() over all rows, in bands if row threads are enabled."""
    global _POOL
    n = min(_ROW_THREADS, height)
    if n <= 1:
        call()
        return
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=_ROW_THREADS)
    L = lib()

    def band(lo, hi):
        L.orc_set_row_range(C.c_int(lo), C.c_int(hi))
        try:
            call()
        finally:
            L.orc_set_row_range(C.c_int(0), C.c_int(0x7FFFFFFF))

    edges = [height * k // n for k in range(n + 1)]
    for f in [_POOL.submit(band, edges[k], edges[k + 1]) for k in range(n)]:
        f.result()


def bilateral_filter_and_cutoff(depth, sigma_xy=3.0, sigma_value_factor=0.05, value_to_ignore=0,
                                radius_factor=2.0, max_depth=15000, depth_valid_region_radius=333.0):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    out = np.empty_like(depth)
    _rows(h, lambda: lib().orc_bilateral_filter_and_cutoff(
        C.c_float(sigma_xy), C.c_float(sigma_value_factor), C.c_uint16(value_to_ignore),
        C.c_float(radius_factor), C.c_uint16(max_depth), C.c_float(depth_valid_region_radius),
        C.c_int(w), C.c_int(h), _p(depth), _p(out)))
    return out


def outlier_depth_map_fusion(depth, others, others_TR_reference, fx, fy, cx, cy, tolerance=0.02,
                             required_count=-1):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    others = [_c(o, np.uint16) for o in others]
    T = _c(np.asarray(others_TR_reference, np.float32).reshape(len(others), 12), np.float32)
    ptrs = (C.c_void_p * len(others))(*[o.ctypes.data for o in others])
    out = np.empty_like(depth)
    _rows(h, lambda: lib().orc_outlier_depth_map_fusion(
        C.c_int(len(others)), C.c_int(required_count), C.c_float(tolerance), C.c_int(w), C.c_int(h),
        _p(depth), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), ptrs, _p(T), _p(out)))
    return out


def median_filter_and_densify(depth, iterations=1):
    """MedianFilterAndDensifyDepthMap applied `iterations` times (APP/main.cc:929-939)."""
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    for _ in range(iterations):
        out = np.empty_like(depth)
        lib().orc_median_filter_and_densify(C.c_int(w), C.c_int(h), _p(depth), _p(out))
        depth = out
    return depth


def downscale_using_median_while_excluding(depth, out_width, out_height, value_to_ignore=0):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    out = np.empty((out_height, out_width), np.uint16)
    lib().orc_downscale_using_median_while_excluding(C.c_uint16(value_to_ignore), C.c_int(w), C.c_int(h), _p(depth),
                                                     C.c_int(out_width), C.c_int(out_height), _p(out))
    return out


def color_image_pyramid(color, level):
    color = _c(color, np.uint8)
    h, w, ch = color.shape
    assert ch == 3 and level >= 1 and w % (1 << level) == 0 and h % (1 << level) == 0
    out = np.zeros((h >> level, w >> level, 3), np.uint8)
    lib().orc_color_image_pyramid(C.c_int(w), C.c_int(h), _p(color), C.c_int(level), _p(out))
    return out


def erode_depth_map(depth, radius):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    out = np.empty_like(depth)
    if radius == 0:
        _rows(h, lambda: lib().orc_copy_without_border(C.c_int(w), C.c_int(h), _p(depth), _p(out)))
    else:
        _rows(h, lambda: lib().orc_erode_depth_map(C.c_int(radius), C.c_int(w), C.c_int(h), _p(depth), _p(out)))
    return out


def compute_normals_and_drop_bad_pixels(depth, fx, fy, cx, cy, observation_angle_threshold_deg=85.0,
                                        depth_scaling=5000.0):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    out = np.empty_like(depth)
    normals = np.zeros((h, w, 2), np.float32)
    _rows(h, lambda: lib().orc_compute_normals_and_drop_bad_pixels(
        C.c_float(observation_angle_threshold_deg), C.c_float(depth_scaling),
        C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
        C.c_int(w), C.c_int(h), _p(depth), _p(out), _p(normals)))
    return out, normals


def compute_point_radii_and_remove_isolated_pixels(depth, fx, fy, cx, cy, point_radius_extension_factor=1.5,
                                                   point_radius_clamp_factor=float("inf"),
                                                   depth_scaling=5000.0, radius_init=None):
    depth = _c(depth, np.uint16)
    h, w = depth.shape
    out = np.empty_like(depth)
    # the kernel leaves radius untouched where depth == 0; callers compare only where out > 0
    radius = np.zeros((h, w), np.float32) if radius_init is None else _c(radius_init, np.float32).copy()
    _rows(h, lambda: lib().orc_compute_point_radii_and_remove_isolated_pixels(
        C.c_float(point_radius_extension_factor), C.c_float(point_radius_clamp_factor), C.c_float(depth_scaling),
        C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
        C.c_int(w), C.c_int(h), _p(depth), _p(radius), _p(out)))
    return out, radius


# ---- reconstruction object -----------------------------------------------------------------
class Recon:
    """Mirror of CUDASurfelReconstruction (APP/cuda_surfel_reconstruction.h:44-176) on the CPU."""

    def __init__(self, max_surfels, width, height, fx, fy, cx, cy, sum_mode=SUM_EXACT):
        self._r = lib().orc_recon_create(max_surfels, width, height, fx, fy, cx, cy, sum_mode)
        self.max_surfels, self.width, self.height = max_surfels, width, height

    def close(self):
        if self._r:
            lib().orc_recon_destroy(self._r)
            self._r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def surfels_size(self):
        return int(self._r.contents.surfel_count)

    @property
    def surfel_count(self):
        return int(self._r.contents.surfel_count - self._r.contents.merge_count)

    @property
    def merge_count(self):
        return int(self._r.contents.merge_count)

    def set_counts(self, surfel_count, merge_count):
        self._r.contents.surfel_count = surfel_count
        self._r.contents.merge_count = merge_count

    def stats(self):
        c = self._r.contents
        return {k: int(getattr(c, "last_" + k)) for k in
                ("n_visible", "n_new", "n_merged", "n_recent", "n_edges", "n_integrated", "n_replaced",
                 "n_conflict_hits")}

    def surfels(self):
        """Writable [25, max_surfels] float32 view of the SoA."""
        return np.ctypeslib.as_array(self._r.contents.surfels, shape=(ROWS, self.max_surfels))

    def _img(self, name, dtype):
        a = np.ctypeslib.as_array(getattr(self._r.contents, name), shape=(self.height, self.width))
        assert a.dtype == dtype, (a.dtype, dtype)
        return a

    def scratch(self):
        return {
            "supporting": self._img("supporting", np.uint32),
            "support_counts": self._img("support_counts", np.uint32),
            "depth_sums_q": self._img("depth_sums_q", np.int64),
            "conflicting": self._img("conflicting", np.uint32),
            "first_depth": self._img("first_depth", np.float32),
            "new_flags": self._img("new_flags", np.uint8),
            "new_indices": self._img("new_indices", np.uint32),
        }

    def integrate(self, frame_index, depth_scaling, depth, normals, radius, color, global_T_local, params=None):
        """depth (u16 [H,W]) is mutated in place by blending, as in the reference."""
        assert depth.dtype == np.uint16 and depth.flags.c_contiguous and depth.flags.writeable
        params = params or IntegrateParams.defaults()
        normals = _c(normals, np.float32)
        radius = _c(radius, np.float32)
        color = _c(color, np.uint8)
        T = _c(np.asarray(global_T_local, np.float32).reshape(12), np.float32)
        lib().orc_recon_integrate(self._r, C.c_uint32(frame_index), C.c_float(depth_scaling), _p(depth),
                                  _p(normals), _p(radius), _p(color), _p(T), C.byref(params))

    def regularize(self, frame_index, regularizer_weight=10.0, radius_factor=2.0, window=30):
        lib().orc_recon_regularize(self._r, C.c_uint32(frame_index), C.c_float(regularizer_weight),
                                   C.c_float(radius_factor), C.c_int(window))

    def deform_by_creation_frame(self, frame_T, reactivate=None, frame_index=0):
        T = _c(np.asarray(frame_T, np.float32).reshape(-1, 12), np.float32)
        ra = _c(reactivate, np.uint8) if reactivate is not None else None
        assert ra is None or ra.size == T.shape[0]
        lib().orc_recon_deform_by_creation_frame(self._r, _p(T), C.c_uint32(T.shape[0]),
                                                 _p(ra) if ra is not None else None, C.c_uint32(frame_index))

    def transfer_all(self):
        n = self.surfels_size
        f = [np.empty(n, np.float32) for _ in range(7)]
        s = np.empty(n, np.uint32)
        lib().orc_recon_transfer_all(self._r, *[_p(a) for a in f], _p(s))
        return {"x": f[0], "y": f[1], "z": f[2], "radius_squared": f[3], "normal_x": f[4], "normal_y": f[5],
                "normal_z": f[6], "last_update_stamp": s, "surfel_count": n}

    def export_vertices(self):
        n = self.surfels_size
        pos = np.empty(3 * n, np.float32)
        col = np.empty(3 * n, np.uint8)
        lib().orc_recon_export_vertices(self._r, _p(pos), _p(col))
        return pos, col


# ---- neighbor search -----------------------------------------------------------------------
def set_race_overrides(supporting=None, conflicting=None):
    """See orc_set_race_overrides (smx_oracle.h).  The arrays must stay alive until the overrides are switched off."""
    if supporting is None and conflicting is None:
        lib().orc_set_race_overrides(None, None, C.c_size_t(0))
        return
    assert supporting.dtype == np.uint32 and conflicting.dtype == np.uint32
    assert supporting.flags.c_contiguous and conflicting.flags.c_contiguous
    lib().orc_set_race_overrides(_p(supporting), _p(conflicting), C.c_size_t(supporting.size))


def race_override_stats():
    out = (C.c_uint32 * 4)()
    lib().orc_get_race_override_stats(out)
    return {"applied_supporting": out[0], "rejected_supporting": out[1], "applied_conflicting": out[2],
            "rejected_conflicting": out[3]}


def nn_bruteforce(px, py, pz, q, radius_sq, k, state=None, skip_mask=0):
    px, py, pz = (_c(a, np.float32) for a in (px, py, pz))
    d2 = np.zeros(k, np.float32)
    idx = np.zeros(k, np.uint32)
    st = _p(_c(state, np.uint8)) if state is not None else None
    n = lib().orc_nn_bruteforce(_p(px), _p(py), _p(pz), C.c_uint32(px.size), C.c_float(q[0]), C.c_float(q[1]),
                                C.c_float(q[2]), C.c_float(radius_sq), C.c_int(k), st, C.c_uint8(skip_mask),
                                _p(d2), _p(idx))
    return n, d2, idx


def nn_grid_batch(px, py, pz, cell_size, qx, qy, qz, qr2, k):
    px, py, pz, qx, qy, qz, qr2 = (_c(a, np.float32) for a in (px, py, pz, qx, qy, qz, qr2))
    nq = qx.size
    d2 = np.zeros((nq, k), np.float32)
    idx = np.zeros((nq, k), np.uint32)
    cnt = np.zeros(nq, np.int32)
    lib().orc_nn_grid_batch(_p(px), _p(py), _p(pz), C.c_uint32(px.size), C.c_float(cell_size),
                            _p(qx), _p(qy), _p(qz), _p(qr2), C.c_uint32(nq), C.c_int(k), _p(d2), _p(idx), _p(cnt))
    return cnt, d2, idx


def check_triangles(x, y, z, radius_squared, nx, ny, nz, triangles, long_edge_total_factor_squared):
    """CheckRemeshing's per-triangle tests over host rows; see smx_oracle.h for the flag bits."""
    rows = [_c(a, np.float32) for a in (x, y, z, radius_squared, nx, ny, nz)]
    tri = _c(triangles, np.uint32).reshape(-1, 3)
    flags = np.zeros(tri.shape[0], np.uint8)
    lib().orc_check_triangles(*[_p(a) for a in rows], C.c_uint32(rows[0].size), _p(tri), C.c_uint32(tri.shape[0]),
                              C.c_float(long_edge_total_factor_squared), _p(flags))
    return flags
