"""The reference application's reconstruction loop (APP/main.cc:884-1267, without the mesher and the viewer) on a TUM
RGB-D folder: reader -> upload -> preprocessing -> Integrate per frame -> optional OBJ / PLY export.
      python tools/run_tum.py <dataset_folder> [--trajectory groundtruth.txt] [--export_mesh out.obj]
                              [--export_point_cloud out.ply] [--max_surfel_count N] [--pyramid_level L] ...
With --synthetic N it first writes an N-frame synthetic dataset into the folder (the test stream), so that the whole
path can be exercised without data."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset_folder")
    ap.add_argument("--trajectory", default="groundtruth.txt")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--max_surfel_count", type=int, default=5_000_000)
    ap.add_argument("--depth_scaling", type=float, default=5000.0)
    ap.add_argument("--max_depth", type=float, default=3.0)
    ap.add_argument("--outlier_filtering_frame_count", type=int, default=8)
    ap.add_argument("--pyramid_level", type=int, default=0)
    ap.add_argument("--median_filter_and_densify_iterations", type=int, default=0)
    ap.add_argument("--start_frame", type=int, default=0)
    ap.add_argument("--end_frame", type=int, default=2 ** 31)
    ap.add_argument("--export_mesh")
    ap.add_argument("--export_point_cloud")
    args = ap.parse_args()

    import torch  # noqa: F401  (libsmx binds to the HIP runtime torch loaded)
    from surfelmeshing_amd import api, export, tum, _lib
    from surfelmeshing_amd.pipeline import FramePipeline, PreprocessParams, others_TR_reference
    _lib.require_gpu()

    if args.synthetic:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        from common import small_stream
        from scipy.spatial.transform import Rotation
        s = small_stream(320, 240)
        frames = [s.frame(f) for f in range(args.synthetic)]
        stamps = [1000.0 + f / 30.0 for f in range(args.synthetic)]
        traj = []
        for f, t in enumerate(stamps):
            T = np.asarray(s.pose(f), np.float64).reshape(3, 4)
            traj.append((t, T[:, 3], Rotation.from_matrix(T[:, :3]).as_quat()))
        tum.write_tum_dataset(args.dataset_folder, frames, stamps, (s.fx, s.fy, s.cx - 0.5, s.cy - 0.5), traj)
        args.max_depth = 10.0

    video = tum.ReadTUMRGBDDatasetAssociatedAndCalibrated(args.dataset_folder, args.trajectory)
    if video is None:
        sys.exit("Could not read dataset.")
    cam = video.depth_camera
    fx, fy, cx, cy = cam.parameters()
    pre = PreprocessParams(depth_scaling=args.depth_scaling, max_depth=args.max_depth,
                           outlier_filtering_frame_count=args.outlier_filtering_frame_count,
                           depth_valid_region_radius=333.0 * cam.width() / 640.0,
                           pyramid_level=args.pyramid_level,
                           median_filter_and_densify_iterations=args.median_filter_and_densify_iterations)
    pipe = FramePipeline(cam.width(), cam.height(), fx, fy, cx, cy, args.max_surfel_count, pre)
    half = args.outlier_filtering_frame_count // 2
    n = min(video.frame_count(), args.end_frame)
    uploaded = set()
    t0 = time.time()
    done = 0
    for f in range(args.start_frame, n):
        # main.cc:905-968: everything up to f + half + 1 is on the GPU before frame f is processed
        for g in range(f, min(n - 1, f + half + 1) + 1):
            if g not in uploaded:
                pipe.upload(g, video.depth_frame(g).GetImage(), video.color_frame(g).GetImage())
                video.depth_frame(g).ClearImageAndDerivedData()
                video.color_frame(g).ClearImageAndDerivedData()
                uploaded.add(g)
        if f < args.start_frame + half or f >= n - half:      # main.cc:986-995: not enough neighbours
            continue
        others = [f - k for k in range(1, half + 1)] + [f + k for k in range(1, half + 1)]    # main.cc:1039-1059
        G = video.depth_frame(f).global_T_frame()
        T = others_TR_reference(G, [video.depth_frame(g).global_T_frame() for g in others], args.depth_scaling)
        pipe.process(f, others, T, G)
        done += 1
        old = f - half - 1                                                                      # main.cc:1226-1240
        if old in uploaded:
            pipe.release(old)
    api.StreamSynchronize(None)
    dt = time.time() - t0
    rec = pipe.reconstruction
    print("%d frames integrated in %.2f s (%.1f frames/s incl. PNG decoding); %d surfels (%d merged)" % (
        done, dt, done / max(dt, 1e-9), rec.surfels_size(), rec.surfels_size() - rec.surfel_count()))
    if args.export_mesh:
        export.SaveMeshAsOBJ(rec, args.export_mesh)
        print("Wrote %s." % args.export_mesh)
    if args.export_point_cloud:
        export.SavePointCloudAsPLY(rec, args.export_point_cloud, export_colors=True)
        print("Wrote %s." % args.export_point_cloud)


if __name__ == "__main__":
    main()
