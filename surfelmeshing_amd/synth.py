"""Deterministic synthetic RGB-D streams.

Everything is generated on the host with numpy from fixed seeds; nothing is
read from disk.  The scene is a closed 6 x 6 x 3 m room whose walls, floor and
ceiling carry a low-amplitude sinusoidal relief (so normals vary); the camera
moves on a small circle around the room centre while yawing, which keeps it
about 2 m from the walls and gives >= 95 % overlap between frames f-4 .. f+4
(needed by the multi-frame outlier cull).

Conventions follow the reference's TUM reader: depth is u16 = round(5000 z)
with z the camera-space depth in metres (APP/main.cc:279), intrinsics are in
the pixel-CORNER convention (VIS/rgbd_video_io_tum_dataset.h:240-244), poses
are global_T_frame as row-major 3x4 float32.
"""
import numpy as np

ROOM_HALF = np.array([3.0, 1.5, 3.0])  # x, y (down), z half extents in metres
RELIEF_AMPLITUDE = 0.02
RELIEF_FREQ = 5.0  # rad / m


def _relief(a, b):
    return RELIEF_AMPLITUDE * np.sin(RELIEF_FREQ * a) * np.sin(RELIEF_FREQ * b)


class SyntheticStream:
    def __init__(self, width=640, height=480, fx=525.0, fy=525.0, cx=320.0, cy=240.0,
                 seed=0x5EED0001, depth_scaling=5000.0, yaw_deg_per_frame=0.5, step_m_per_frame=0.005,
                 path_radius=1.0, start_yaw_deg=0.0, pitch_deg=0.0, dropout=0.01, noise_sigma=0.001,
                 obstacle_until=-1, obstacle_center=(1.0, 0.1, 1.7), obstacle_radius=0.35):
        self.width, self.height = width, height
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.seed = int(seed)
        self.depth_scaling = float(depth_scaling)
        self.yaw_step = np.deg2rad(yaw_deg_per_frame)
        self.step = float(step_m_per_frame)
        self.path_radius = float(path_radius)
        self.start_yaw = np.deg2rad(start_yaw_deg)
        self.pitch = np.deg2rad(pitch_deg)
        self.dropout = float(dropout)
        self.noise_sigma = float(noise_sigma)
        # a sphere that is present for frames < obstacle_until and then vanishes: its surfels end up in
        # measured free space, which exercises the conflict / replace path of the integration
        self.obstacle_until = int(obstacle_until)
        self.obstacle_center = np.asarray(obstacle_center, np.float64)
        self.obstacle_radius = float(obstacle_radius)
        xs = (np.arange(width, dtype=np.float64) + 0.5 - self.cx) / self.fx
        ys = (np.arange(height, dtype=np.float64) + 0.5 - self.cy) / self.fy
        self._dx, self._dy = np.meshgrid(xs, ys)

    # -- trajectory ------------------------------------------------------------------------
    def pose64(self, f):
        """global_T_frame of frame f as float64 (R [3,3], t [3])."""
        yaw = self.start_yaw + self.yaw_step * f
        # position on a circle, arc length `step` per frame
        ang = self.step * f / max(self.path_radius, 1e-9)
        t = np.array([self.path_radius * np.cos(ang), 0.1 * np.sin(0.01 * f), self.path_radius * np.sin(ang)])
        cy_, sy_ = np.cos(yaw), np.sin(yaw)
        Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
        cp, sp = np.cos(self.pitch), np.sin(self.pitch)
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        return Ry @ Rx, t

    def pose(self, f):
        R, t = self.pose64(f)
        return np.concatenate([R, t[:, None]], axis=1).astype(np.float32)

    # -- rendering -------------------------------------------------------------------------
    def _raycast(self, f):
        R, o = self.pose64(f)
        d = np.stack([self._dx, self._dy, np.ones_like(self._dx)], axis=-1) @ R.T  # world dir, cam z = 1
        best_t = np.full(self._dx.shape, np.inf)
        for axis in range(3):
            a, b = [k for k in range(3) if k != axis]
            for sign in (-1.0, 1.0):
                dn = d[..., axis] * sign
                valid = dn > 1e-9
                plane = ROOM_HALF[axis]
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = (plane - sign * o[axis]) / dn
                    for _ in range(2):  # fixed-point refinement of the relief offset (inward)
                        ha = o[a] + t * d[..., a]
                        hb = o[b] + t * d[..., b]
                        t = (plane - _relief(ha, hb) - 0.03 - sign * o[axis]) / dn
                t = np.where(valid & (t > 0), t, np.inf)
                best_t = np.minimum(best_t, t)
        if f < self.obstacle_until:
            oc = o - self.obstacle_center
            a = (d * d).sum(-1)
            b = 2.0 * (d @ oc)
            c = oc @ oc - self.obstacle_radius ** 2
            disc = b * b - 4 * a * c
            with np.errstate(invalid="ignore"):
                ts = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
            ts = np.where(ts > 0, ts, np.inf)
            best_t = np.minimum(best_t, ts)
        hit = o[None, None, :] + best_t[..., None] * d
        return best_t, hit

    def frame(self, f):
        """Returns (depth u16 [H,W], colour u8 [H,W,3]) of frame f."""
        z, hit = self._raycast(f)
        rng = np.random.Generator(np.random.PCG64([self.seed, int(f)]))
        noise = rng.standard_normal(z.shape)
        # drop-outs: spatially coherent 8x8 holes (like real sensor holes) plus rare single pixels;
        # pure salt-and-pepper at this rate would be amplified by the 9-frame cull and the erosion
        hb, wb = (self.height + 7) // 8, (self.width + 7) // 8
        blocks = rng.random((hb, wb)) < self.dropout
        drop = np.kron(blocks, np.ones((8, 8), bool))[:self.height, :self.width].astype(bool)
        drop |= rng.random(z.shape) < self.dropout * 0.01
        zn = z + self.noise_sigma * z * z * noise
        d = np.rint(self.depth_scaling * zn)
        d = np.where(np.isfinite(d) & (d > 0) & (d < 65535) & ~drop, d, 0).astype(np.uint16)
        cell = np.floor(hit * 10.0).astype(np.int64)  # 10 cm colour cells
        h = (cell[..., 0] * 73856093) ^ (cell[..., 1] * 19349663) ^ (cell[..., 2] * 83492791)
        color = np.stack([(h >> 0) & 255, (h >> 8) & 255, (h >> 16) & 255], axis=-1).astype(np.uint8)
        return d, color

    # -- caller-side argument preparation, APP/main.cc:1039-1059 ----------------------------
    def outlier_frames(self, f, count=8):
        """Frame indices in the reference's order [f-1..f-count/2, f+1..f+count/2]."""
        half = count // 2
        return [f - (i + 1) for i in range(half)] + [f + (i + 1) for i in range(half)]

    def others_TR_reference(self, f, count=8):
        """(ref_scaled_frame_T_global * global_T_other_scaled)^-1 as [count, 3, 4] float32."""
        s = self.depth_scaling
        Rr, tr = self.pose64(f)
        out = []
        for g in self.outlier_frames(f, count):
            Ro, to = self.pose64(g)
            # ref_frame_T_global (scaled) = (Rr^T, -Rr^T tr * s); global_T_other (scaled) = (Ro, to * s)
            R = Rr.T @ Ro
            t = Rr.T @ (to * s) - Rr.T @ (tr * s)
            Ri = R.T
            ti = -Ri @ t
            out.append(np.concatenate([Ri, ti[:, None]], axis=1))
        return np.asarray(out, np.float32)


def config_c1():
    """C1: one 640x480 frame, camera at the origin looking at a tilted plane + sphere (CPU plumbing case)."""
    w, h, fx, fy, cx, cy = 640, 480, 525.0, 525.0, 320.0, 240.0
    xs = (np.arange(w) + 0.5 - cx) / fx
    ys = (np.arange(h) + 0.5 - cy) / fy
    dx, dy = np.meshgrid(xs, ys)
    # plane n.p = d with n = (0.2, 0.1, -1)/|.|, through (0,0,2)
    n = np.array([0.2, 0.1, -1.0])
    n /= np.linalg.norm(n)
    z_plane = (n @ np.array([0, 0, 2.0])) / (n[0] * dx + n[1] * dy + n[2])
    # sphere centre (0.3, -0.1, 1.5) radius 0.3
    c = np.array([0.3, -0.1, 1.5])
    a = dx * dx + dy * dy + 1
    b = -2 * (dx * c[0] + dy * c[1] + c[2])
    cc = c @ c - 0.3 ** 2
    disc = b * b - 4 * a * cc
    z_sphere = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
    z = np.minimum(z_plane, z_sphere)
    depth = np.rint(5000.0 * z).astype(np.uint16)
    return depth, dict(width=w, height=h, fx=fx, fy=fy, cx=cx, cy=cy)


def room_surface_points(n, seed=0x5EED0005):
    """Config C5: about n surfel positions on the six faces of the room (with the same relief), a
    jittered grid of the local spacing; returns ([m, 3] float32, spacing in metres).  Face after face, row-major
    within a face -- the index order carries no more locality than a map built frame by frame would."""
    rng = np.random.default_rng(seed)
    half = ROOM_HALF
    total = sum(2 * 4 * half[a] * half[b] for a, b in ((1, 2), (0, 2), (0, 1)))
    spacing = float(np.sqrt(total / n))
    pts = []
    for axis in range(3):
        a, b = [k for k in range(3) if k != axis]
        for sign in (-1.0, 1.0):
            na, nb = int(2 * half[a] / spacing), int(2 * half[b] / spacing)
            ga, gb = np.meshgrid((np.arange(na) + 0.5) * spacing - half[a], (np.arange(nb) + 0.5) * spacing - half[b])
            p = np.empty((ga.size, 3), np.float32)
            p[:, a] = (ga.ravel() + rng.uniform(-0.3, 0.3, ga.size) * spacing).astype(np.float32)
            p[:, b] = (gb.ravel() + rng.uniform(-0.3, 0.3, ga.size) * spacing).astype(np.float32)
            p[:, axis] = sign * half[axis] + _relief(p[:, a], p[:, b]).astype(np.float32)
            pts.append(p)
    return np.concatenate(pts), spacing
