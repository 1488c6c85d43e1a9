#!/usr/bin/env python
"""Benchmark of the surfel-integration hot path (BASELINE.json: RGB-D frames/s integrated @640x480 with
5 M live surfels; achieved HBM GB/s).

One "step" = one frame of the reference's per-frame call sequence (APP/main.cc:1015-1223): bilateral filter,
9-frame outlier cull, erosion, normals, radii, CUDASurfelReconstruction::Integrate -- with the raw depth and
colour frames already resident in HBM.  Workload = config C2 of SURVEY.md 8(d): the synthetic room stream at
640x480; the map is first grown to >= 5 M surfels by running the real pipeline over a sweeping trajectory
(untimed), then the timed window re-traverses mapped area (steady state).

    python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU, one independent stream per rank, no data-path
collective -- weak scaling; value = all frames of all ranks / max-over-ranks time.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# ALGORITHMIC bytes per launch of each kernel (DESIGN.md "Kernels and bytes"): every record a kernel has to read
# or write counted once, cache effects and line granularity excluded.  N = slots, V = visible slots, R = slots
# inside the regulariser window, C = slots with a link into the window, Ew = such links, E = all links, P = pixels.
ALG_BYTES = {
    # P records + flag bytes of the segments that are read, 2 flag bytes per slot of the culled ones, list + z-buffer
    "scan_visible": lambda st, P: 18.0 * (st["surfels_size"] - 1024.0 * st.get("n_segments_skipped", 0))
                                  + 2.0 * 1024.0 * st.get("n_segments_skipped", 0) + 24.0 * st["n_visible"],
    "neighbor_scan": lambda st, P: 18.0 * st["surfels_size"] + 1.0 * st["n_edges"] + 4.0 * st["n_recent"],
    # slots served: contributors and recent slots (mostly the same slots): 50 B own records each; 48 B per link into
    # the window (target S + T records, 16 B inbox/accumulator store); 16 B own-term record per recent slot
    "reg_accumulate": lambda st, P: 50.0 * max(st["n_contributors"], st["n_recent"]) + 48.0 * st["n_window_edges"]
                                    + 16.0 * st["n_recent"],
    # P, S, r^2, own-term record, three accumulator channels (32 + 32 + 64 B), S store, inbox re-zeroing
    "reg_step": lambda st, P: 224.0 * st["n_recent"],
    "reg_update": lambda st, P: 36.0 * st["n_recent"],
    "associate": lambda st, P: 90.0 * st["n_visible"],
    "merge_decide": lambda st, P: 60.0 * st["n_visible"],
    "integrate": lambda st, P: 160.0 * st["n_visible"],
    "update_neighbors+create": lambda st, P: 190.0 * st["n_visible"] + 6.0 * P + 122.0 * st["n_new"],
    "blend": lambda st, P: 26.0 * P,
    "clear_assoc": lambda st, P: 26.0 * P,
    "new_flags_scan": lambda st, P: 15.0 * P,
}


def pose64(g, seed_phase=0.0):
    """Growth / re-traversal trajectory: yaw 2 deg/frame, slow Lissajous pitch, small circle around the room
    centre.  Returns global_T_frame (R, t) in float64."""
    yaw = math.radians(2.0 * g) + seed_phase
    pitch = math.radians(62.0) * math.sin(2.0 * math.pi * g / (180.0 * 2.7))
    phi = 0.01 * g + seed_phase
    t = np.array([0.8 * math.cos(phi), 0.25 * math.sin(0.003 * g), 0.8 * math.sin(phi)])
    cy_, sy_ = math.cos(yaw), math.sin(yaw)
    Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
    cp, sp = math.cos(pitch), math.sin(pitch)
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return Ry @ Rx, t


def pose32(g, phase):
    R, t = pose64(g, phase)
    return np.concatenate([R, t[:, None]], axis=1).astype(np.float32)


def others_tr_reference(g, phase, count, scaling):
    """(ref_scaled_frame_T_global * global_T_other_scaled)^-1, APP/main.cc:1039-1059."""
    Rr, tr = pose64(g, phase)
    half = count // 2
    frames = [g - (i + 1) for i in range(half)] + [g + (i + 1) for i in range(half)]
    out = []
    for o in frames:
        Ro, to = pose64(o, phase)
        R = Rr.T @ Ro
        t = Rr.T @ (to * scaling) - Rr.T @ (tr * scaling)
        Ri = R.T
        out.append(np.concatenate([Ri, (-Ri @ t)[:, None]], axis=1))
    return frames, np.asarray(out, np.float32)


class Workload:
    """The synthetic C2 stream driven through the native frame loop (include/smx_driver.h)."""

    def __init__(self, api, width, height, target_surfels, cap_surfels, seed, phase):
        from surfelmeshing_amd.pipeline import NativeFramePipeline, PreprocessParams
        self.api = api
        sc = width / 640.0
        self.w, self.h = width, height
        self.fx = self.fy = 525.0 * sc
        self.cx, self.cy = 320.0 * sc, 240.0 * sc
        self.seed, self.phase = seed, phase
        self.target = target_surfels
        self.pre = PreprocessParams(max_depth=10.0, depth_valid_region_radius=333.0 * sc)
        self.pipe = NativeFramePipeline(width, height, self.fx, self.fy, self.cx, self.cy, cap_surfels, self.pre)
        self.pipe.reconstruction.set_timing_enabled(0)

    def render(self, logical, pose_index):
        """Render logical frame `logical` (noise seed) at trajectory position `pose_index` on the GPU."""
        if logical not in self.pipe.resident:
            self.pipe.render(logical, pose32(pose_index, self.phase), self.seed)

    def plan(self, logical, pose_index):
        """(frame, outlier-cull neighbour frames, their relative poses, camera pose) of one step."""
        frames, T = others_tr_reference(pose_index, self.phase, 8, self.pre.depth_scaling)
        return logical, [logical + (f - pose_index) for f in frames], T, pose32(pose_index, self.phase)

    def steps(self, plans):
        from surfelmeshing_amd.pipeline import DriverStep
        arr = (DriverStep * len(plans))(*[self.pipe.make_step(*p) for p in plans])
        return arr, len(plans)

    def grow(self, log):
        """Untimed: run the real pipeline along the trajectory until the map holds >= target surfels."""
        g = 4
        for f in range(0, 9):
            self.render(f, f)
        n = 0
        t0 = time.time()
        while n < self.target and g < 20000:
            batch = []
            for _ in range(50):
                batch.append(self.plan(g, g))
                g += 1
            # frames g-4 .. g+4 of every step of the batch must be resident while it runs
            for f in range(batch[0][0] - 4, batch[-1][0] + 5):
                self.render(f, f)
            self.pipe.run_array(*self.steps(batch))
            for f in range(batch[0][0] - 4, batch[-1][0] - 3):
                self.pipe.release(f)
            n = self.pipe.reconstruction.surfels_size()
            if log and (g - 4) % 500 == 0:
                print("# grow: frame %d surfels %d (%.1fs)" % (g, n, time.time() - t0), file=sys.stderr, flush=True)
        for f in list(self.pipe.resident):
            self.pipe.release(f)
        self.api.StreamSynchronize(None)
        return g, n


def host_frames_pass(wl, plan, base, count, api, torch):
    """Not `value`: the same frame loop with the inputs arriving over PCIe.  Every step's newest frame (f + 4) is copied
    from page-locked host memory in front of its preprocessing, on the preprocessing stream, beside the previous
    frame's Integrate (smx_driver_run_streamed; the reference caller's staging, APP/main.cc:905-984), into the slot
    the resident run used."""
    pipe = wl.pipe
    warm = min(10, count // 2)
    keep, uploads = [], []
    for j in range(base, base + count):
        f = plan[j][0] + 4
        d, c = pipe.download_frame(f)
        pd, pc = api.PagelockedArray(d.shape, np.uint16), api.PagelockedArray(c.shape, np.uint8)
        pd.array[...] = d
        pc.array[...] = c
        keep += [pd, pc]
        uploads.append((f, pd.array, pc.array))
    steps = [pipe.make_step(*plan[j]) for j in range(base, base + count)]
    pipe.run_streamed(steps[:warm], uploads[:warm])
    torch.cuda.synchronize()
    t = time.perf_counter()
    pipe.run_streamed(steps[warm:], uploads[warm:])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    n = count - warm
    nbytes = wl.w * wl.h * 5
    for k in keep:
        k.close()
    return {"value": n / dt, "unit": "frames/s", "steps": n, "ms_per_step": 1e3 * dt / n,
            "h2d_bytes_per_frame": nbytes, "h2d_GBs": nbytes * n / dt / 1e9,
            "note": "inputs copied from page-locked host memory on the preprocessing stream "
                    "(smx_driver_run_streamed); not the headline value"}


def algorithmic_bytes(st, P):
    """SURVEY.md 8(d) byte model of the REFERENCE's per-frame traffic (for comparison only) and this
    design's own compulsory traffic (DESIGN.md 'Bytes')."""
    N, E = st["surfels_size"], st["n_edges"]
    ref = 140 * N + 8 * E + 460 * st["n_visible"] + 360 * st["n_recent"] + 122 * st["n_new"] + 193 * P
    ours = 32 * N + 8 * E + 460 * st["n_visible"] + 360 * st["n_recent"] + 122 * st["n_new"] + 193 * P
    return ref, ours


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--surfels", type=int, default=5_000_000, help="live surfels to reach before timing")
    ap.add_argument("--cap", type=int, default=0, help="max_surfel_count (default: surfels * 1.1)")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--host-frames", type=int, default=100,
                    help="frames of the extra pass whose inputs arrive from page-locked host memory (0 = skip)")
    ap.add_argument("--no-check", action="store_true", help="skip the full-size GPU-vs-oracle check")
    ap.add_argument("--no-overlap", action="store_true", help="A/B: no frame pipelining inside Integrate")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()

    import torch  # first: libsmx then binds to the HIP runtime torch loaded
    import torch.distributed as dist
    from surfelmeshing_amd import multistream
    rank, local_rank, world = multistream.rank_info()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from surfelmeshing_amd import _lib, api
    _lib.require_gpu()
    _lib.check(_lib.load().smx_set_device(local_rank if world > 1 else 0))
    log = (rank == 0) and not args.quiet

    cap = args.cap or int(args.surfels * 1.1)
    assign = multistream.stream_assignment(rank)   # one independent stream per rank, no data-path collective
    wl = Workload(api, args.width, args.height, args.surfels, cap, assign["seed"], assign["phase"])
    t0 = time.time()
    g_end, n_grown = wl.grow(log)
    if log:
        print("# grown to %d surfels in %d frames, %.1fs" % (n_grown, g_end, time.time() - t0), file=sys.stderr)

    # timed window: re-traverse the start of the trajectory (mapped area) with new frame indices
    K, W = args.steps, args.warmup
    first = g_end + 10
    cal, reps = 10, 20
    total = W + cal + K
    do_host = args.host_frames if (rank == 0 and world == 1) else 0   # PCIe-inclusive pass: rank 0 at N = 1 only
    for j in range(-4, total + 1 + reps + do_host + 4):
        wl.render(first + j, 4 + j)
    plan = [wl.plan(first + j, 4 + j) for j in range(total + 1 + reps + do_host)]
    api.StreamSynchronize(None)

    rec = wl.pipe.reconstruction
    rec.set_stats_enabled(False)   # the distribution counters are single-address atomics: off while timing
    wl.pipe.run_array(*wl.steps(plan[:W]))
    # short calibration pass with HIP events around every kernel: which kernel dominates the frame?
    # (frame pipelining off here and in the per-kernel pass below, so that kernels are timed one at a time)
    rec.set_overlap(False)
    rec.set_timing_enabled(2)
    names = rec.kernel_time_names()
    cal_ms = np.zeros(len(names))
    for j in range(W, W + cal):
        wl.pipe.run_array(*wl.steps(plan[j:j + 1]))
        cal_ms += np.array(rec.kernel_times_ms())
    rec.set_timing_enabled(0)
    rec.set_overlap(not args.no_overlap)
    dominant = names[int(np.argmax(cal_ms))]
    api.StreamSynchronize(None)
    do_cpu = rank == 0 and world == 1 and args.cpu_frames > 0   # CPU baseline: rank 0 at N = 1 only
    state0 = rec.debug_download_surfels() if do_cpu else None
    merge0 = (rec.surfels_size() - rec.surfel_count()) if state0 is not None else 0
    timed_steps = wl.steps(plan[W + cal:W + cal + K])

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # HIP events around the dominant kernel only (2 records per frame on the launch stream) stay on during
    # the timed region; everything else is measured in separate passes.
    rec.profile_begin(dominant, K)
    _lib.check(_lib.load().smx_debug_marker(None, 1))   # delimits the timed region in rocprofv3 kernel traces
    sync_all()
    t_start = time.perf_counter()
    wl.pipe.run_array(*timed_steps)          # K frames: preprocessing + Integrate each, enqueued by the C++ loop
    enqueue_local = time.perf_counter() - t_start   # host side done (returns without synchronising)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t_start
    from surfelmeshing_amd import multistream
    fps, elapsed, _ = multistream.aggregate_throughput(K, elapsed_local, world, dist if world > 1 else None, "cuda")
    if world > 1:
        dist.barrier()
    dom_ms, dom_n = rec.profile_end()
    _lib.check(_lib.load().smx_debug_marker(None, 2))

    # value distributions of one more frame (counters on)
    rec.set_overlap(False)
    rec.set_stats_enabled(True)
    wl.pipe.run_array(*wl.steps(plan[total:total + 1]))
    st = rec.stats()
    rec.set_stats_enabled(False)

    # per-stage and per-kernel device times (separate untimed pass, HIP events on the launch stream)
    rec.set_timing_enabled(3)
    stage_ms = np.zeros(7)
    kernel_ms = np.zeros(len(names))
    for j in range(total + 1, total + 1 + reps):
        wl.pipe.run_array(*wl.steps(plan[j:j + 1]))
        stage_ms += np.array(rec.GetTimings())
        kernel_ms += np.array(rec.kernel_times_ms())
    stage_ms /= reps
    kernel_ms /= reps
    rec.set_timing_enabled(0)

    host_pass = None
    if do_host > 0:
        rec.set_overlap(not args.no_overlap)
        host_pass = host_frames_pass(wl, plan, total + 1 + reps, do_host, api, torch)
        rec.set_overlap(False)

    ref_bytes, own_bytes = algorithmic_bytes(st, args.width * args.height)
    result = {
        "metric": "RGB-D frames/s integrated @640x480, 5M live surfels; achieved HBM GB/s",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "host_enqueue_ms_per_step": 1e3 * enqueue_local / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: synthetic room stream %dx%d, full preprocessing + Integrate per frame, "
                               "%d surfel slots (%d live), steady-state re-traversal" %
                               ("C2" if args.width == 640 else "C3" if args.width == 1280 else "custom", args.width,
                                args.height, st["surfels_size"], st["surfels_size"] - st["merge_count"]),
                   "max_surfel_count": cap, "streams": world, "parallelism": "1 independent stream per GPU"},
        "distributions": st,
        "stage_ms": dict(zip(["data_association", "surfel_merging", "measurement_blending", "integration",
                              "neighbor_update", "new_surfel_creation", "regularization"], [float(x) for x in stage_ms])),
        "reference_model_bytes_per_frame": ref_bytes,
        "design_bytes_per_frame": own_bytes,
        "reference_model_GBs": ref_bytes * (K / elapsed) / 1e9,
    }

    if rank == 0:
        result["roofline"] = roofline_block(st, args.width * args.height, dominant, dom_ms, dom_n,
                                            dict(zip(names, [float(x) for x in kernel_ms])))
        if host_pass is not None:
            result["host_frames"] = host_pass
        if do_cpu:
            result["cpu_baseline"] = cpu_baseline(wl, plan, W + cal, args.cpu_frames, state0, merge0, cap,
                                                  not args.no_check, log)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# kernel-slot name -> kernel name in rocprofv3 output
SLOT_KERNEL = {"reg_accumulate": "k_reg_accumulate", "reg_step": "k_reg_step", "neighbor_scan": "k_neighbor_scan<true, true>",
               "scan_visible": "k_scan_visible", "associate": "k_associate<true>", "merge_decide": "k_merge_decide<true>",
               "integrate": "k_integrate<true>", "update_neighbors+create": "k_update_and_create<true>",
               "blend": "k_blend_fused", "clear_assoc": "k_clear_assoc", "new_flags_scan": "k_new_flags_scan"}


def pmc_traffic(dominant):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/pmc_traffic.json, written by tools/pmc_summary.py; FETCH_SIZE and WRITE_SIZE are in KB and need
    separate passes).  Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
    reports half of the bytes read, so it is doubled; WRITE_SIZE is taken as is.  None if there is no such file."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    try:
        k = json.load(open(path)).get(SLOT_KERNEL.get(dominant, ""), {})
        if "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
            return None, None
        return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0, {"FETCH_SIZE_KB": k["FETCH_SIZE"], "WRITE_SIZE_KB": k["WRITE_SIZE"]}
    except (ValueError, OSError):
        return None, None


def roofline_block(st, P, dominant, dom_ms, dom_n, kernel_ms):
    """Roofline of the dominant kernel: algorithmic bytes per launch (ALG_BYTES) / average launch
    duration measured with HIP events on the launch stream over the timed region."""
    alg = ALG_BYTES[dominant](st, P)
    achieved = alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    per_kernel = {}
    for k, ms in kernel_ms.items():
        if k in ALG_BYTES and ms > 0:
            b = ALG_BYTES[k](st, P)
            # the per-kernel pass brackets every launch with two event records (~6 us of overhead per kernel)
            per_kernel[k] = {"ms_with_event_overhead": ms, "algorithmic_MB": b / 1e6}
    traffic, raw = pmc_traffic(dominant)
    return {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_pmc_raw": raw,
            "algorithmic_bytes_per_launch": alg,
            "avg_launch_ms": dom_ms, "launches_timed": dom_n, "surfel_slots": st["surfels_size"],
            "kernels_untimed_pass": per_kernel}


def cpu_baseline(wl, plan, W, frames, state0, merge0, cap, check, log):
    """The oracle (plain single-threaded C loops) on the first `frames` frames of the timed window, starting
    from the same surfel state; also used as a full-size parity check of the HIP path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    from oracle_pipeline import OraclePipeline
    api = wl.api
    po = OraclePipeline(wl.w, wl.h, wl.fx, wl.fy, wl.cx, wl.cy, cap, wl.pre)
    n0 = state0.shape[1]
    po.recon.surfels()[:, :n0] = state0
    po.recon.set_counts(n0, merge0)
    need = set()
    for j in range(W, W + frames):
        need.add(plan[j][0])
        need.update(plan[j][1])
    host_frames = {f: wl.pipe.download_frame(f) for f in sorted(need)}
    for f, (d, c) in host_frames.items():
        po.upload(f, d, c)
    t0 = time.perf_counter()
    for j in range(W, W + frames):
        po.process(*plan[j])
    dt = time.perf_counter() - t0
    out = {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the timed window from the same %d-surfel state (oracle, -O2, 1 thread)" % (frames, n0)}
    if check:
        from surfelmeshing_amd.pipeline import FramePipeline
        pg = FramePipeline(wl.w, wl.h, wl.fx, wl.fy, wl.cx, wl.cy, cap, wl.pre)
        pg.reconstruction.debug_upload_surfels(state0, merge0)
        for f, (d, c) in host_frames.items():
            pg.upload(f, d, c)
        for j in range(W, W + frames):
            pg.process(*plan[j])
        n = po.recon.surfels_size
        ok = pg.reconstruction.surfels_size() == n
        bad_rows = []
        if ok:
            G = pg.reconstruction.debug_download_surfels(n)
            O = po.recon.surfels()[:, :n]
            for r in range(25):
                if r in orc.SCRATCH_ROWS:
                    continue
                if not np.array_equal(G[r].view(np.uint32), O[r].view(np.uint32)):
                    bad_rows.append(r)
        out["parity_check"] = {"frames": frames, "surfels": int(n), "counts_equal": bool(ok),
                               "rows_not_bit_equal": bad_rows}
    return out


if __name__ == "__main__":
    main()
