"""Input side: the TUM RGB-D reader mirrored from VIS/rgbd_video_io_tum_dataset.h, host code only."""
import os

import numpy as np
import pytest

from common import small_stream
from surfelmeshing_amd import tum


def _quat_from_matrix(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(np.asarray(R, np.float64)).as_quat()      # x, y, z, w


def test_png_roundtrip_all_filters_and_pillow_agreement(tmp_path):
    rng = np.random.default_rng(0)
    images = [rng.integers(0, 65536, (37, 53)).astype(np.uint16), rng.integers(0, 256, (21, 40, 3)).astype(np.uint8),
              rng.integers(0, 256, (9, 17)).astype(np.uint8), np.zeros((4, 4), np.uint16)]
    # smooth content makes the predictors of the Average / Paeth filters non-trivial
    yy, xx = np.mgrid[0:48, 0:64]
    images.append(((yy * 300 + xx * 200) % 65536).astype(np.uint16))
    images.append(np.stack([yy * 3 + xx, yy + 2 * xx, 255 - yy], -1).astype(np.uint8))
    for k, img in enumerate(images):
        for filters in ((0,), (1,), (2,), (3,), (4,), (0, 1, 2, 3, 4)):
            p = str(tmp_path / ("img%d_%d.png" % (k, len(filters) * 10 + filters[0])))
            tum.write_png(p, img, filters)
            got = tum.read_png(p, use_pillow=False)
            assert got.dtype == img.dtype and np.array_equal(got, img), (k, filters)
            try:
                import PIL  # noqa: F401
            except ImportError:
                continue
            assert np.array_equal(tum.read_png(p, use_pillow=True), img), (k, filters)
    with pytest.raises(ValueError):
        tum.decode_png(b"not a png at all")


def test_pillow_written_files_decode(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:60, 0:80]
    rgb = np.stack([yy * 2 + xx, xx * 3, (yy * xx) % 256], -1).astype(np.uint8)
    rgb[::7] = rng.integers(0, 256, rgb[::7].shape)
    d16 = ((yy * 211 + xx * 97) % 65536).astype(np.uint16)
    Image.fromarray(rgb).save(tmp_path / "c.png")                      # zlib / filter choices of another encoder
    Image.fromarray(d16).save(tmp_path / "d.png")
    Image.fromarray(np.dstack([rgb, np.full((60, 80), 200, np.uint8)])).save(tmp_path / "a.png")
    assert np.array_equal(tum.read_png(str(tmp_path / "c.png"), use_pillow=False), rgb)
    assert np.array_equal(tum.read_png(str(tmp_path / "d.png"), use_pillow=False), d16)
    assert np.array_equal(tum.read_png(str(tmp_path / "a.png"), use_pillow=False), rgb)      # alpha dropped
    assert np.array_equal(tum.read_png(str(tmp_path / "a.png")), rgb)


def test_trajectory_and_interpolation(tmp_path):
    from scipy.spatial.transform import Rotation, Slerp
    p = tmp_path / "traj.txt"
    p.write_text("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n"
                 "10.0 0 0 0 0 0 0 2\n"                                  # not normalised: the SE3 constructor does it
                 "11.0 1 2 3 0 0 0.7071067811865476 0.7071067811865476\n"
                 "13.0 1 2 5 0 0 1 0\n"
                 "\n"
                 "99.0 9 9 9 0 0 0 1\n")                                 # after the first empty line: not read (:100)
    ts, poses = tum.ReadTUMRGBDTrajectory(str(p))
    assert ts == [10.0, 11.0, 13.0] and np.allclose(poses[0].q, [0, 0, 0, 1])
    assert tum.ReadTUMRGBDTrajectory(str(tmp_path / "missing.txt")) is None
    (tmp_path / "bad.txt").write_text("1.0 2 3\n")
    assert tum.ReadTUMRGBDTrajectory(str(tmp_path / "bad.txt")) is None
    # clamped outside, exact at the knots
    assert tum.InterpolatePose(5.0, ts, poses) is poses[0] and tum.InterpolatePose(20.0, ts, poses) is poses[-1]
    assert np.allclose(tum.InterpolatePose(11.0, ts, poses).matrix3x4(), poses[1].matrix3x4(), atol=1e-6)
    # inside: slerp + linear translation, against scipy
    rot = Rotation.from_quat(np.array([q.q for q in poses], np.float64))
    slerp = Slerp(ts, rot)
    for t in (10.25, 10.5, 11.5, 12.9):
        got = tum.InterpolatePose(t, ts, poses).matrix3x4()
        i = 0 if t < 11 else 1
        f = (t - ts[i]) / (ts[i + 1] - ts[i])
        want_t = poses[i].t + f * (poses[i + 1].t - poses[i].t)
        assert np.allclose(got[:, :3], slerp([t]).as_matrix()[0], atol=2e-6), t
        assert np.allclose(got[:, 3], want_t, atol=1e-6)
    # max_interpolation_time_extent (:64-67): both neighbours have to be close enough
    assert tum.InterpolatePose(12.0, ts, poses, 0.5) is None and tum.InterpolatePose(12.0, ts, poses, 1.0) is not None
    # the rotation matrix of a unit quaternion is orthonormal; a 90 degree turn about z maps x to y
    R = poses[1].matrix3x4()[:, :3]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and np.allclose(R @ [1, 0, 0], [0, 1, 0], atol=1e-6)


def test_read_dataset_matches_the_stream_it_was_written_from(tmp_path):
    s = small_stream(64, 48)
    n = 6
    frames = [s.frame(f) for f in range(n)]
    stamps = [100.0 + 0.033 * f for f in range(n)]
    traj = []
    for f in range(n):
        T = np.asarray(s.pose(f), np.float64).reshape(3, 4)
        traj.append((stamps[f] + 0.01, T[:, 3], _quat_from_matrix(T[:, :3])))          # poses between the frames
    folder = str(tmp_path / "seq")
    tum.write_tum_dataset(folder, frames, stamps, (s.fx, s.fy, s.cx - 0.5, s.cy - 0.5), traj)
    video = tum.ReadTUMRGBDDatasetAssociatedAndCalibrated(folder, "groundtruth.txt")
    assert video.frame_count() == n and len(video.color_frames) == n
    cam = video.depth_camera
    assert (cam.width(), cam.height()) == (64, 48)
    assert np.allclose(cam.parameters(), [s.fx, s.fy, s.cx, s.cy], rtol=1e-7)          # cx, cy: + 0.5 (:237-241)
    for f in range(n):
        assert np.array_equal(video.depth_frame(f).GetImage(), frames[f][0])
        assert np.array_equal(video.color_frame(f).GetImage(), frames[f][1])
        assert video.depth_frame(f).timestamp_string == "%.6f" % stamps[f]
    video.depth_frame(0).ClearImageAndDerivedData()
    assert np.array_equal(video.depth_frame(0).GetImage(), frames[0][0])
    # frame 0 lies before the first pose: clamped to it; frame 3 lies between the poses of 2 and 3
    assert np.allclose(video.depth_frame(0).global_T_frame(), np.asarray(s.pose(0)).reshape(3, 4), atol=2e-6)
    got = video.depth_frame(3).global_T_frame()
    a, b = np.asarray(s.pose(2)).reshape(3, 4), np.asarray(s.pose(3)).reshape(3, 4)
    f = (stamps[3] - traj[2][0]) / (traj[3][0] - traj[2][0])
    assert np.allclose(got[:, 3], a[:, 3] + f * (b[:, 3] - a[:, 3]), atol=2e-6)
    # without a trajectory every pose is the identity; a tight extent drops frames that are too far from a pose
    v2 = tum.ReadTUMRGBDDatasetAssociatedAndCalibrated(folder, None)
    assert v2.frame_count() == n and np.array_equal(v2.depth_frame(2).global_T_frame(), np.eye(4, dtype=np.float32)[:3])
    v3 = tum.ReadTUMRGBDDatasetAssociatedAndCalibrated(folder, "groundtruth.txt", max_interpolation_time_extent=0.015)
    assert 0 < v3.frame_count() < n
    # missing pieces -> None (the reference returns false)
    assert tum.ReadTUMRGBDDatasetAssociatedAndCalibrated(str(tmp_path / "nope"), None) is None
    assert tum.ReadTUMRGBDDatasetAssociatedAndCalibrated(folder, "no_such_trajectory.txt") is None
    os.remove(os.path.join(folder, "associated.txt"))
    assert tum.ReadTUMRGBDDatasetAssociatedAndCalibrated(folder, None) is None


def test_others_TR_reference_matches_the_stream_generator():
    """pipeline.others_TR_reference (APP/main.cc:1037-1059, from 3x4 poses) against the synthetic stream's own."""
    from surfelmeshing_amd.pipeline import others_TR_reference
    s = small_stream(64, 48)
    f = 9
    others = s.outlier_frames(f)
    got = others_TR_reference(s.pose(f), [s.pose(g) for g in others], s.depth_scaling)
    want = np.asarray(s.others_TR_reference(f)).reshape(len(others), 3, 4)
    # (the generator works from float64 poses, this from their float32 roundings; translations are in depth units)
    assert np.allclose(got[:, :, :3], want[:, :, :3], atol=1e-6) and np.allclose(got[:, :, 3], want[:, :, 3], atol=2e-2)
