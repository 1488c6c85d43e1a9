"""Export side: OBJ / PLY writers mirrored from VIS/mesh.h and VIS/point_cloud.h, host code only."""
import numpy as np

from surfelmeshing_amd import export


def test_write_obj_format(tmp_path):
    pos = np.array([[0, 1.5, -2.25], [1e-5, 123456.789, 1.0 / 3.0], [1e10, -0.0, 2]], np.float32)
    col = np.array([[0, 255, 51], [128, 1, 2], [3, 4, 5]], np.uint8)
    p = str(tmp_path / "m.obj")
    assert export.write_obj(p, pos, col, np.array([[0, 1, 2], [2, 1, 0]]))
    lines = open(p).read().splitlines()
    # ostream << float: %g with 6 significant digits; colours times 1/255 (VIS/point_cloud.h:567-581); 1-based faces
    assert lines[0] == "v 0 1.5 -2.25 0 1 0.2"
    assert lines[1] == "v 1e-05 123457 0.333333 0.501961 0.00392157 0.00784314"
    assert lines[2] == "v 1e+10 -0 2 0.0117647 0.0156863 0.0196078"
    assert lines[3:] == ["f 1 2 3", "f 3 2 1"]
    export.write_obj(p, pos)                                              # positions only (point_cloud.h:557-565)
    assert open(p).read().splitlines() == ["v 0 1.5 -2.25", "v 1e-05 123457 0.333333", "v 1e+10 -0 2"]


def test_write_ply_layout_and_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    pos, nrm = rng.normal(size=(7, 3)).astype(np.float32), rng.normal(size=(7, 3)).astype(np.float32)
    col = rng.integers(0, 256, (7, 3)).astype(np.uint8)
    p = str(tmp_path / "c.ply")
    assert export.write_ply(p, pos, col, nrm)
    raw = open(p, "rb").read()
    head = ("ply\nformat binary_little_endian 1.0\nelement vertex 7\nproperty float x\nproperty float y\n"
            "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\n"
            "property float ny\nproperty float nz\nend_header\n").encode()
    assert raw.startswith(head) and len(raw) == len(head) + 7 * 27       # 12 + 3 + 12 bytes per vertex, packed
    assert raw[len(head):len(head) + 12] == pos[0].astype("<f4").tobytes()
    assert raw[len(head) + 12:len(head) + 15] == col[0].tobytes()
    rec = export.read_ply(p)
    assert np.array_equal(np.stack([rec["x"], rec["y"], rec["z"]], 1), pos)
    assert np.array_equal(np.stack([rec["nx"], rec["ny"], rec["nz"]], 1), nrm)
    assert np.array_equal(np.stack([rec["red"], rec["green"], rec["blue"]], 1), col)
    export.write_ply(p, pos)
    assert export.read_ply(p).dtype.names == ("x", "y", "z")
    export.write_ply(p, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), np.zeros((0, 3), np.float32))
    assert export.read_ply(p).size == 0
