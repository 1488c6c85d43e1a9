"""Export side of the path: the files the reference writes from ExportVertices / the CPU mirror.

Mirrors SaveMeshAsOBJ and SavePointCloudAsPLY (APP/main.cc:128-203) over Mesh::WriteAsOBJ (VIS/mesh.h:106-130) and
PointCloud::WriteAsOBJ / WriteAsPLY (VIS/point_cloud.h:464-540, 557-607): ASCII OBJ with `v x y z r g b` lines (colours
scaled by 1/255, ostream's default %g formatting) and 1-based `f` lines; binary little-endian PLY with float x y z,
uchar red green blue, float nx ny nz.  Host code; the vertex data comes from the GPU through ExportVertices and
TransferAllToCPU.  Merged surfels (NaN positions, cuda_surfel_reconstruction_kernels.cu:2412-2433) are left out and
triangle indices are renumbered accordingly, like SurfelMeshing::ConvertToMesh3fCu8 does for the reference.
"""
import numpy as np

from . import api


def _g(v):
    return "%g" % v


def write_obj(path, positions, colors=None, triangles=None):
    """positions [N,3] float32; colors [N,3] uint8 or None; triangles [T,3] (0-based vertex indices) or None."""
    positions = np.asarray(positions, np.float32).reshape(-1, 3)
    lines = []
    if colors is None:
        for p in positions.tolist():
            lines.append("v %s %s %s\n" % (_g(p[0]), _g(p[1]), _g(p[2])))
    else:
        k = np.float32(1.0) / np.float32(255)              # kNormalizationFactor, point_cloud.h:570-571
        col = (np.asarray(colors, np.uint8).reshape(-1, 3).astype(np.float32) * k).tolist()
        for p, c in zip(positions.tolist(), col):
            lines.append("v %s %s %s %s %s %s\n" % (_g(p[0]), _g(p[1]), _g(p[2]), _g(c[0]), _g(c[1]), _g(c[2])))
    if triangles is not None:
        for t in np.asarray(triangles, np.int64).reshape(-1, 3).tolist():
            lines.append("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))     # mesh.h:116-122
    with open(path, "wb") as f:
        f.write("".join(lines).encode("ascii"))
    return True


def write_ply(path, positions, colors=None, normals=None):
    """Binary little-endian PLY (point_cloud.h:493-531): x y z [red green blue] [nx ny nz]."""
    positions = np.asarray(positions, np.float32).reshape(-1, 3)
    n = positions.shape[0]
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % n
    if colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        header += "property uchar red\nproperty uchar green\nproperty uchar blue\n"
    if normals is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
        header += "property float nx\nproperty float ny\nproperty float nz\n"
    header += "end_header\n"
    rec = np.zeros(n, np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = positions[:, 0], positions[:, 1], positions[:, 2]
    if colors is not None:
        c = np.asarray(colors, np.uint8).reshape(-1, 3)
        rec["red"], rec["green"], rec["blue"] = c[:, 0], c[:, 1], c[:, 2]
    if normals is not None:
        m = np.asarray(normals, np.float32).reshape(-1, 3)
        rec["nx"], rec["ny"], rec["nz"] = m[:, 0], m[:, 1], m[:, 2]
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rec.tobytes())
    return True


def read_ply(path):
    """Reader for the files write_ply produces (tests, tools)."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    n, fields = 0, []
    for line in data[:end].decode("ascii").splitlines():
        p = line.split()
        if p[:2] == ["element", "vertex"]:
            n = int(p[2])
        elif p[0] == "property":
            fields.append((p[2], "<f4" if p[1] == "float" else "u1"))
    return np.frombuffer(data, np.dtype(fields), n, end)


def export_vertices(reconstruction, stream=None):
    """ExportVertices + download (main.cc:141-149): (positions [N,3] with NaN rows for merged surfels, colours [N,3])."""
    n = reconstruction.surfels_size()
    pos, col = api.CUDABuffer(1, max(3 * n, 1), np.float32), api.CUDABuffer(1, max(3 * n, 1), np.uint8)
    reconstruction.ExportVertices(stream, pos, col)
    api.StreamSynchronize(stream)
    p, c = pos.Download().reshape(-1)[:3 * n].reshape(n, 3), col.Download().reshape(-1)[:3 * n].reshape(n, 3)
    pos.close()
    col.close()
    return p, c


def SaveMeshAsOBJ(reconstruction, export_mesh_path, stream=None, triangles=None):
    """main.cc:128-178.  `triangles` [T,3]: surfel (slot) indices as the mesher holds them; triangles that touch a
    merged surfel are dropped, the rest renumbered to the compacted vertex list.  Without triangles the file holds
    the coloured vertices only."""
    p, c = export_vertices(reconstruction, stream)
    live = ~np.isnan(p[:, 0])                                  # main.cc:152-155
    tri = None
    if triangles is not None:
        remap = np.cumsum(live) - 1
        t = np.asarray(triangles, np.int64).reshape(-1, 3)
        t = t[np.all((t >= 0) & (t < live.size), axis=1)]
        t = t[np.all(live[t], axis=1)]
        tri = remap[t]
    return write_obj(export_mesh_path, p[live], c[live], tri)


def SavePointCloudAsPLY(reconstruction, export_point_cloud_path, stream=None, export_colors=False):
    """main.cc:182-203: positions and normals of the live surfels as the mesher's CPU mirror holds them
    (TransferAllToCPU rows); colour is white in the reference (its TODO at :194) unless export_colors."""
    n = reconstruction.surfels_size()
    cpu = api.CUDASurfelsCPU(max(n, 1))
    cpu.LockWriteBuffers()
    reconstruction.TransferAllToCPU(stream, 0, cpu)
    api.StreamSynchronize(stream)
    cpu.UnlockWriteBuffers()
    cpu.WaitForLockAndSwapBuffers()
    b = cpu.read_buffers()
    live = b.surfel_radius_squared_buffer[:n] >= 0
    pos = np.stack([b.surfel_x_buffer[:n], b.surfel_y_buffer[:n], b.surfel_z_buffer[:n]], 1)[live]
    nrm = np.stack([b.surfel_normal_x_buffer[:n], b.surfel_normal_y_buffer[:n], b.surfel_normal_z_buffer[:n]], 1)[live]
    if export_colors:
        col = export_vertices(reconstruction, stream)[1][live]
    else:
        col = np.full((int(live.sum()), 3), 255, np.uint8)
    return write_ply(export_point_cloud_path, pos, col, nrm)
