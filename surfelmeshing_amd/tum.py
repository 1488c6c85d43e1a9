"""Input side of the path: the reference's TUM RGB-D reader, host code.

Mirrors VIS/rgbd_video_io_tum_dataset.h:42-251 (InterpolatePose, ReadTUMRGBDTrajectory,
ReadTUMRGBDDatasetAssociatedAndCalibrated) and the parts of RGBDVideo / ImageFrame the frame loop of APP/main.cc
uses (lazy image loading, per-frame global_T_frame, the shared pinhole camera).  PNG files are decoded by Pillow when
it is importable and by the small decoder below otherwise (8-bit grey / RGB / RGBA and 16-bit grey, non-interlaced --
what the TUM benchmark ships); `write_png` / `write_tum_dataset` exist for tests and tools.
"""
import math
import os
import struct
import zlib

import numpy as np

from .api import PinholeCamera4f

_PNG_MAGIC = b"\x89PNG\r\n\x1a\n"


# ---- PNG ----------------------------------------------------------------------------------------------------------
def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _unfilter(raw, height, stride, bpp):
    """PNG filter reconstruction (ISO/IEC 15948 section 9): rows of 1 filter byte + `stride` bytes."""
    out = np.zeros((height, stride), np.uint8)
    prev = np.zeros(stride, np.uint8)
    pos = 0
    for y in range(height):
        ft = raw[pos]
        line = np.frombuffer(raw, np.uint8, stride, pos + 1)
        pos += 1 + stride
        if ft == 0:
            cur = line.copy()
        elif ft == 1:       # Sub: a running sum per byte lane, modulo 256
            cur = np.cumsum(line.reshape(-1, bpp), axis=0, dtype=np.uint8).reshape(-1)
        elif ft == 2:       # Up
            cur = line + prev
        elif ft in (3, 4):  # Average / Paeth depend on the reconstructed left neighbour: sequential
            cur_l = [0] * stride
            ln, pv = line.tolist(), prev.tolist()
            if ft == 3:
                for i in range(stride):
                    left = cur_l[i - bpp] if i >= bpp else 0
                    cur_l[i] = (ln[i] + ((left + pv[i]) >> 1)) & 255
            else:
                for i in range(stride):
                    left = cur_l[i - bpp] if i >= bpp else 0
                    ul = pv[i - bpp] if i >= bpp else 0
                    cur_l[i] = (ln[i] + _paeth(left, pv[i], ul)) & 255
            cur = np.array(cur_l, np.uint8)
        else:
            raise ValueError("PNG: unknown filter type %d" % ft)
        out[y] = cur
        prev = cur
    return out


def decode_png(data):
    """Decode PNG bytes -> uint8 [H,W] / [H,W,3] or uint16 [H,W].  Alpha is dropped."""
    if data[:8] != _PNG_MAGIC:
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if hdr is None:
        raise ValueError("PNG: no IHDR")
    w, h, depth, ctype, _, _, interlace = hdr
    channels = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if channels is None or depth not in (8, 16) or interlace != 0:
        raise ValueError("PNG: unsupported format (colour type %d, bit depth %d, interlace %d)" % (ctype, depth, interlace))
    bpp = channels * depth // 8
    img = _unfilter(zlib.decompress(b"".join(idat)), h, w * bpp, bpp)
    if depth == 16:
        img = img.reshape(h, w, channels, 2)
        img = (img[..., 0].astype(np.uint16) << 8) | img[..., 1].astype(np.uint16)
    else:
        img = img.reshape(h, w, channels)
    if ctype in (4, 6):
        img = img[..., :-1]
    return np.ascontiguousarray(img[..., 0] if img.shape[-1] == 1 else img)


def encode_png(array, filters=(0,)):
    """uint8 [H,W] / [H,W,3] or uint16 [H,W] -> PNG bytes.  `filters`: the filter type of row y is
    filters[y % len(filters)] (all five types, so that a decoder can be exercised)."""
    a = np.ascontiguousarray(array)
    if a.dtype == np.uint16 and a.ndim == 2:
        ctype, depth, bpp = 0, 16, 2
        rows = a.astype(">u2").view(np.uint8).reshape(a.shape[0], -1)
    elif a.dtype == np.uint8 and a.ndim == 2:
        ctype, depth, bpp, rows = 0, 8, 1, a
    elif a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3:
        ctype, depth, bpp, rows = 2, 8, 3, a.reshape(a.shape[0], -1)
    else:
        raise ValueError("encode_png: uint8 [H,W], uint8 [H,W,3] or uint16 [H,W]")
    h, w = a.shape[:2]
    rows = rows.astype(np.int16)
    zero = np.zeros_like(rows[0])
    out = bytearray()
    for y in range(h):
        ft = filters[y % len(filters)]
        cur, up = rows[y], (rows[y - 1] if y else zero)
        left = np.concatenate([zero[:bpp], cur[:-bpp]])
        ul = np.concatenate([zero[:bpp], up[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - up
        elif ft == 3:
            f = cur - ((left + up) >> 1)
        elif ft == 4:
            p = left + up - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
            f = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
        else:
            raise ValueError("filter type 0..4")
        out.append(ft)
        out += (f & 255).astype(np.uint8).tobytes()

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)

    return (_PNG_MAGIC + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(bytes(out), 6)) + chunk(b"IEND", b""))


def write_png(path, array, filters=(0,)):
    with open(path, "wb") as f:
        f.write(encode_png(array, filters))


def read_png(path, use_pillow=None):
    """use_pillow: None = when importable, True / False = force."""
    if use_pillow is None or use_pillow:
        try:
            from PIL import Image
        except ImportError:
            if use_pillow:
                raise
            Image = None
        if Image is not None:
            with Image.open(path) as im:
                if im.mode in ("I;16", "I;16B", "I;16L", "I"):
                    return np.asarray(im).astype(np.uint16)
                if im.mode in ("RGBA", "P"):
                    im = im.convert("RGB")
                elif im.mode == "LA":
                    im = im.convert("L")
                return np.ascontiguousarray(np.asarray(im))
    with open(path, "rb") as f:
        return decode_png(f.read())


# ---- SE3f (Sophus) as far as the reader needs it ---------------------------------------------------------------------
class SE3f:
    """Unit quaternion (x, y, z, w) + translation, float32 (Sophus::SE3f; its constructor normalises the quaternion)."""

    def __init__(self, quaternion_xyzw=(0, 0, 0, 1), translation=(0, 0, 0)):
        q = np.asarray(quaternion_xyzw, np.float32)
        n = np.float32(math.sqrt(float(np.dot(q, q))))
        self.q = (q / n).astype(np.float32) if n > 0 else np.array([0, 0, 0, 1], np.float32)
        self.t = np.asarray(translation, np.float32).copy()

    def matrix3x4(self):
        """Eigen::Quaternion::toRotationMatrix + translation, row-major 3x4 float32 (global_T_frame)."""
        x, y, z, w = (np.float32(v) for v in self.q)
        tx, ty, tz = x + x, y + y, z + z
        twx, twy, twz = tx * w, ty * w, tz * w
        txx, txy, txz = tx * x, ty * x, tz * x
        tyy, tyz, tzz = ty * y, tz * y, tz * z
        one = np.float32(1)
        return np.array([[one - (tyy + tzz), txy - twz, txz + twy, self.t[0]],
                         [txy + twz, one - (txx + tzz), tyz - twx, self.t[1]],
                         [txz - twy, tyz + twx, one - (txx + tyy), self.t[2]]], np.float32)


def _slerp(qa, qb, t):
    """Eigen::Quaternion::slerp in float32."""
    one = np.float32(1) - np.finfo(np.float32).eps
    d = np.float32(np.dot(qa, qb))
    ad = abs(d)
    t = np.float32(t)
    if ad >= one:
        s0, s1 = np.float32(1) - t, t
    else:
        theta = np.float32(math.acos(float(ad)))
        st = np.float32(math.sin(float(theta)))
        s0 = np.float32(math.sin(float((np.float32(1) - t) * theta))) / st
        s1 = np.float32(math.sin(float(t * theta))) / st
    if d < 0:
        s1 = -s1
    return (s0 * qa + s1 * qb).astype(np.float32)


def InterpolatePose(timestamp, pose_timestamps, poses, max_interpolation_time_extent=float("inf")):
    """VIS/rgbd_video_io_tum_dataset.h:42-86: clamp outside the trajectory, otherwise slerp + linear translation
    between the two enclosing poses; None if either is further away than max_interpolation_time_extent."""
    assert len(pose_timestamps) == len(poses) and len(poses) >= 2
    if timestamp <= pose_timestamps[0]:
        return poses[0]
    if timestamp >= pose_timestamps[-1]:
        return poses[-1]
    i = int(np.searchsorted(np.asarray(pose_timestamps), timestamp, side="left"))   # first index with ts >= timestamp
    # the reference scans linearly for the first interval [i, i+1] that contains the timestamp (:62-63)
    i = max(i - 1, 0)
    while i + 1 < len(pose_timestamps) and not (pose_timestamps[i] <= timestamp <= pose_timestamps[i + 1]):
        i += 1
    if i + 1 >= len(pose_timestamps):
        return None
    if (timestamp - pose_timestamps[i]) > max_interpolation_time_extent or \
            (pose_timestamps[i + 1] - timestamp) > max_interpolation_time_extent:
        return None
    factor = (timestamp - pose_timestamps[i]) / (pose_timestamps[i + 1] - pose_timestamps[i])
    a, b = poses[i], poses[i + 1]
    return SE3f(_slerp(a.q, b.q, factor), a.t + np.float32(factor) * (b.t - a.t))


def ReadTUMRGBDTrajectory(path):
    """VIS/rgbd_video_io_tum_dataset.h:88-128: lines `timestamp tx ty tz qx qy qz qw`, '#' comments, stops at the
    first empty line.  Returns (timestamps, poses) or None."""
    try:
        f = open(path, "r")
    except OSError:
        return None
    ts, poses = [], []
    with f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                break
            if line[0] == "#":
                continue
            parts = line.split()
            if len(parts) < 8:
                return None
            try:
                v = [float(p) for p in parts[1:8]]
                stamp = float(parts[0])
            except ValueError:
                return None
            ts.append(stamp)
            poses.append(SE3f(v[3:7], v[0:3]))
    return ts, poses


# ---- RGBDVideo / ImageFrame ----------------------------------------------------------------------------------------------
class ImageFrame:
    """VIS/image_frame.h as far as main.cc uses it: lazy GetImage(), ClearImageAndDerivedData(), global_T_frame()."""

    def __init__(self, path, timestamp, timestamp_string):
        self.path, self.timestamp, self.timestamp_string = path, timestamp, timestamp_string
        self._pose = SE3f()
        self._image = None

    def SetGlobalTFrame(self, pose):
        self._pose = pose

    def global_T_frame(self):
        return self._pose.matrix3x4()

    def GetImage(self):
        if self._image is None:
            self._image = read_png(self.path)
        return self._image

    def ClearImageAndDerivedData(self):
        self._image = None


class RGBDVideo:
    def __init__(self):
        self.color_frames, self.depth_frames = [], []
        self.color_camera = self.depth_camera = None

    def frame_count(self):
        return len(self.depth_frames)

    def color_frame(self, i):
        return self.color_frames[i]

    def depth_frame(self, i):
        return self.depth_frames[i]


def ReadTUMRGBDDatasetAssociatedAndCalibrated(dataset_folder_path, trajectory_filename=None,
                                              max_interpolation_time_extent=float("inf")):
    """VIS/rgbd_video_io_tum_dataset.h:130-251.  calibration.txt = `fx fy cx cy` (pixel-centre convention; the
    camera gets cx + 0.5, cy + 0.5, :237-241), associated.txt = `rgb_time rgb_file depth_time depth_file` per line
    (associate.py), optional trajectory.  Frames whose pose cannot be interpolated are skipped (:202-214).
    Returns an RGBDVideo or None."""
    try:
        with open(os.path.join(dataset_folder_path, "calibration.txt")) as f:
            cal = f.readline().split()
        fx, fy, cx, cy = (float(v) for v in cal[:4])
    except (OSError, ValueError):
        return None
    ts, poses = [], []
    if trajectory_filename:
        r = ReadTUMRGBDTrajectory(os.path.join(dataset_folder_path, trajectory_filename))
        if r is None:
            return None
        ts, poses = r
    video = RGBDVideo()
    try:
        f = open(os.path.join(dataset_folder_path, "associated.txt"))
    except OSError:
        return None
    width = height = 0
    with f:
        for line in f:
            line = line.rstrip("\n")
            if not line or line[0] == "#":
                continue
            parts = line.split()
            if len(parts) < 4:
                return None
            rgb_time, rgb_file, depth_time, depth_file = parts[:4]
            cpose = dpose = SE3f()
            if poses:
                cpose = InterpolatePose(float(rgb_time), ts, poses, max_interpolation_time_extent)
                if cpose is None:
                    continue
                dpose = InterpolatePose(float(depth_time), ts, poses, max_interpolation_time_extent)
                if dpose is None:
                    continue
            cf = ImageFrame(os.path.join(dataset_folder_path, rgb_file), float(rgb_time), rgb_time)
            cf.SetGlobalTFrame(cpose)
            df = ImageFrame(os.path.join(dataset_folder_path, depth_file), float(depth_time), depth_time)
            df.SetGlobalTFrame(dpose)
            video.color_frames.append(cf)
            video.depth_frames.append(df)
            if width == 0:
                try:
                    img = cf.GetImage()
                except (OSError, ValueError):
                    return None
                height, width = img.shape[:2]
                cf.ClearImageAndDerivedData()
    p = [np.float32(fx), np.float32(fy), np.float32(cx + 0.5), np.float32(cy + 0.5)]
    video.color_camera = PinholeCamera4f(width, height, *p)
    video.depth_camera = PinholeCamera4f(width, height, *p)
    return video


def write_tum_dataset(folder, frames, timestamps, calibration, trajectory=None, trajectory_filename="groundtruth.txt"):
    """Write (depth u16 [H,W], colour u8 [H,W,3]) frames in the layout the reader expects.  calibration =
    (fx, fy, cx, cy) in the file's convention (cx, cy without the half-pixel offset); trajectory = list of
    (timestamp, (tx, ty, tz), (qx, qy, qz, qw))."""
    os.makedirs(os.path.join(folder, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(folder, "depth"), exist_ok=True)
    with open(os.path.join(folder, "calibration.txt"), "w") as f:
        f.write("%r %r %r %r\n" % tuple(float(v) for v in calibration))
    with open(os.path.join(folder, "associated.txt"), "w") as f:
        f.write("# rgb_time rgb_file depth_time depth_file\n")
        for (d, c), t in zip(frames, timestamps):
            name = "%.6f" % t
            write_png(os.path.join(folder, "rgb", name + ".png"), c, filters=(0, 1, 2, 3, 4))
            write_png(os.path.join(folder, "depth", name + ".png"), d, filters=(4, 3, 2, 1, 0))
            f.write("%s rgb/%s.png %s depth/%s.png\n" % (name, name, name, name))
    if trajectory is not None:
        with open(os.path.join(folder, trajectory_filename), "w") as f:
            f.write("# timestamp tx ty tz qx qy qz qw\n")
            for t, tr, q in trajectory:
                f.write("%.6f %r %r %r %r %r %r %r\n" % ((t,) + tuple(float(v) for v in tr) + tuple(float(v) for v in q)))
