#!/usr/bin/env python
"""Benchmark of the surfel-integration hot path (BASELINE.json: RGB-D frames/s integrated @640x480 with
5 M live surfels; achieved HBM GB/s).  One command per BASELINE.json config:

    python bench.py [--config C2] --gpus N --steps K --warmup W     (default; the driver's command)
    python bench.py --config C3      1280x960 stream, 20 M surfel cap        (same code path as C2)
    python bench.py --config C5      50 M surfels, radius-neighbor search for all of them (K = 64)
    python bench.py --config C1      one 640x480 frame through the naive CPU loops (no GPU)

C2 / C3: one "step" = one frame of the reference's per-frame call sequence (APP/main.cc:1015-1223): bilateral filter,
9-frame outlier cull, erosion, normals, radii, CUDASurfelReconstruction::Integrate -- with the raw depth and colour
frames already resident in HBM.  The synthetic room stream of SURVEY.md 8(d); the map is first grown to the target
number of LIVE surfels by running the real pipeline over a sweeping trajectory (untimed), then the timed window
re-traverses mapped area (steady state).
C5: one "step" = every indexed surfel queries its own neighbourhood once (r^2 = its own radius^2, K = 64).

N > 1: launched by torch.distributed.run, one rank per GPU, one independent stream / cloud per rank, no data-path
collective -- weak scaling; value = all units of all ranks / max-over-ranks time.
--dry-run: the rank path (rank_info -> process group -> device selection -> stream assignment -> plan generation ->
barrier -> aggregate) without touching a GPU, over gloo; tests/test_multistream_gloo.py runs it with two ranks.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One hardware queue per busy HIP stream (the runtime's default of 4 lets two of the frame loop's streams share one in about
# one run out of five: 5 670 instead of 6 380 frames/s, profiles/r5_ab_notes.md).  Read when the HIP runtime initialises, so it
# is set before torch is imported; libsmx.so raises the same default when it is loaded.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (vector)

# ALGORITHMIC bytes per launch of each kernel (DESIGN.md "Kernels and bytes"): every record a kernel has to read
# or write counted once, cache effects and line granularity excluded.  N = slots, V = visible slots, R = slots
# inside the regulariser window, C = slots with a link into the window, Ew = such links, E = all links, P = pixels.
ALG_BYTES = {
    # pass A's cull step: box (32 B), previous count, streak byte per segment; one list entry per surviving segment
    "cull_segments": lambda st, P: 37.0 * (st["surfels_size"] / 1024.0) + 4.0 * (st["surfels_size"] / 1024.0 - st.get("n_segments_skipped", 0)),
    # P records + flag bytes of the segments that are read (the culled ones' flag bytes are copied only in the first two
    # calls after a segment drops out of view: not counted), 16 B of chunk descriptors per segment read, list + pairs
    "scan_visible": lambda st, P: (18.0 + 16.0 / 1024.0) * (st["surfels_size"] - 1024.0 * st.get("n_segments_skipped", 0))
                                  + (4.0 + 1.9 * 8.0) * st["n_visible"],
    # T records + flag bytes of the segments that are read (17 B per slot), the hot table + the target-group bitmap (4.5 KB) of
    # the ones that are skipped, one flag byte per link, the recent list and the edge kernel's work list (4 B per entry)
    # (fused -- the default: the segment's workgroup does the edge work itself, the work list stays in LDS -- the launch also
    # moves the edge kernel's bytes, minus the list's 4 + 4 per entry)
    "neighbor_scan": lambda st, P: 17.0 * (st["surfels_size"] - 1024.0 * st.get("n_link_segments_skipped", 0))
                                   + 4608.0 * st.get("n_link_segments_skipped", 0) + 1.0 * st["n_edges"] + 4.0 * st["n_recent"]
                                   + (ALG_BYTES["reg_accumulate"](st, P) - 4.0 * max(st["n_contributors"], st["n_recent"]) if st.get("fused_edges")
                                      else 4.0 * max(st["n_contributors"], st["n_recent"])),
    # entries served: contributors and recent slots (mostly the same slots): the entry (4 B) + T, S, N records (48 B) each;
    # per link into the window the target's S record (16 B) and, for the 29 % of them that leave the segment
    # (tools/far_terms_hist.py), a 16 B record in the target segment's bin; per recent slot its dense record (32 B: in-segment
    # sums | own term)
    "reg_accumulate": lambda st, P: 52.0 * max(st["n_contributors"], st["n_recent"]) + (16.0 + 0.29 * 16.0) * st["n_window_edges"]
                                    + 32.0 * st["n_recent"],
    # list entry, P, S, N, the dense record, S store: 100 B per recent slot; the bins' records
    "reg_step": lambda st, P: 100.0 * st["n_recent"] + 0.29 * 16.0 * st["n_window_edges"],
    "reg_update": lambda st, P: 36.0 * st["n_recent"],
    # association tiles: per pair (~1.9 per visible slot) 8 B + the slot's P and N records (32 B); per pixel the
    # measurement (10 B) and the five images written (24 B); the merge phase's supported-surfel records (32 B per visible slot)
    "assoc_tiles": lambda st, P: 1.9 * 40.0 * st["n_visible"] + 34.0 * P + 32.0 * st["n_visible"],
    "integrate+new_flags": lambda st, P: 160.0 * st["n_visible"] + 15.0 * P,
    "update_neighbors+create": lambda st, P: 190.0 * st["n_visible"] + 6.0 * P + 122.0 * st["n_new"],
    # blend tiles: depth + supporting per region cell (54 x 54 cells per 32 x 32 tile at radius 12) + the blended depth
    "blend": lambda st, P: 6.0 * (54.0 * 54.0 / 1024.0) * P + 2.0 * P,
}

# the driver's preprocessing stages (smx_driver_profile_begin): u16 in + u16 out; + 8 neighbour frames' depths gathered;
# u16 in -> u16 + float2 normals + float radius out
PRE_STAGES = ["bilateral", "outlier_fusion", "erode_normals_radii"]
ALG_BYTES.update({
    "bilateral": lambda st, P: 4.0 * P,
    "outlier_fusion": lambda st, P: 4.0 * P + 2.0 * 8 * P,
    "erode_normals_radii": lambda st, P: 16.0 * P,
})

CONFIGS = {
    "C2": dict(width=640, height=480, surfels=5_000_000),
    "C3": dict(width=1280, height=960, surfels=20_000_000),
}


def source_sha():
    """Hash of the kernel sources: PMC files under profiles/ are only used for the build they were collected on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "surfelmeshing_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


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


def frame_plan(logical, pose_index, phase, depth_scaling):
    """(frame, outlier-cull neighbour frames, their relative poses, camera pose) of one step -- host arithmetic only."""
    frames, T = others_tr_reference(pose_index, phase, 8, depth_scaling)
    return logical, [logical + (f - pose_index) for f in frames], T, pose32(pose_index, phase)


class Workload:
    """The synthetic C2 / C3 stream driven through the native frame loop (include/smx_driver.h)."""

    def __init__(self, api, width, height, target_live, cap_surfels, seed, phase):
        from surfelmeshing_amd.pipeline import NativeFramePipeline, PreprocessParams
        self.api = api
        sc = width / 640.0
        self.w, self.h = width, height
        self.fx = self.fy = 525.0 * sc
        self.cx, self.cy = 320.0 * sc, 240.0 * sc
        self.seed, self.phase = seed, phase
        self.target = target_live
        self.pre = PreprocessParams(max_depth=10.0, depth_valid_region_radius=333.0 * sc)
        self.pipe = NativeFramePipeline(width, height, self.fx, self.fy, self.cx, self.cy, cap_surfels, self.pre)
        # (the stage stamps behind GetTimings stay as the library has them: on)

    def render(self, logical, pose_index):
        """Render logical frame `logical` (noise seed) at trajectory position `pose_index` on the GPU."""
        if logical not in self.pipe.resident:
            self.pipe.render(logical, pose32(pose_index, self.phase), self.seed)

    def plan(self, logical, pose_index):
        return frame_plan(logical, pose_index, self.phase, self.pre.depth_scaling)

    def steps(self, plans):
        from surfelmeshing_amd.pipeline import DriverStep
        arr = (DriverStep * len(plans))(*[self.pipe.make_step(*p) for p in plans])
        return arr, len(plans)

    def grow(self, log, window=None):
        """Run the real pipeline along the trajectory until the map holds >= target LIVE surfels (surfel_count() = slots -
        merged, as BASELINE.json's metric counts them), in batches of 50 frames.
        window = (frames, callback): the EXPLORING regime, timed on the way -- the camera keeps finding new surface,
        thousands of new surfels a frame, every new slot inside the regulariser window.  Every batch runs like the timed
        window of the steady state (frames rendered and step arrays prepared beforehand, no host synchronisation inside);
        its first 10 frames are untimed (the device idles between batches while the host renders and counts), the other 40
        lie between two event records on the caller's stream.  growth_phase = the timed frames of the last batches before
        the target (`frames` of them), growth_curve = every batch.  callback(g, info) is called when the target is reached
        (in-run parity check of the next growth frames)."""
        import torch
        g = 4
        for f in range(0, 9):
            self.render(f, f)
        live = 0
        rec = self.pipe.reconstruction
        rec.set_stats_enabled(False)   # (the distribution counters are single-address atomics: off while timing)
        t0 = time.time()
        curve = []
        B, WARM = 50, 10
        slots = rec.surfels_size()
        while live < self.target and g < 40000:
            batch = []
            for _ in range(B):
                batch.append(self.plan(g, g))
                g += 1
            # frames g-4 .. g+4 of every step of the batch must be resident while it runs
            for f in range(batch[0][0] - 4, batch[-1][0] + 5):
                self.render(f, f)
            warm_steps, timed_steps = self.steps(batch[:WARM]), self.steps(batch[WARM:])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.pipe.run_array(*warm_steps)
            e0.record()
            self.pipe.run_array(*timed_steps)
            e1.record()
            for f in range(batch[0][0] - 4, batch[-1][0] - 3):
                self.pipe.release(f)
            live = rec.surfel_count()      # (synchronises)
            prev_slots, slots = slots, rec.surfels_size()
            ms = e0.elapsed_time(e1)
            curve.append({"frames_done": g - 4, "live": int(live), "slots": int(slots), "timed_frames": B - WARM, "ms": ms,
                          "frames_per_s": (B - WARM) / (ms * 1e-3), "new_slots_per_frame": (slots - prev_slots) / float(B)})
            if log and (g - 4) % 500 == 0:
                print("# grow: frame %d live surfels %d (%.1fs)" % (g, live, time.time() - t0), file=sys.stderr, flush=True)
        self.growth = None
        if window is not None and curve:
            frames, callback = window
            last = curve[-max(1, frames // (B - WARM)):]
            n, ms = sum(c["timed_frames"] for c in last), sum(c["ms"] for c in last)
            rec.set_stats_enabled(True)
            self.render(g + 4, g + 4)
            self.pipe.run_array(*self.steps([self.plan(g, g)]))
            st = rec.stats()
            rec.set_stats_enabled(False)
            g += 1
            self.growth = {"value": n / (ms * 1e-3), "unit": "frames/s", "steps": n, "ms_per_step": ms / n,
                           "new_slots_per_frame": float(np.mean([c["new_slots_per_frame"] for c in last])),
                           "live_surfels_at_start": int(curve[-len(last) - 1]["live"]) if len(curve) > len(last) else 0,
                           "live_surfels_at_end": int(live),
                           "frame_behind_the_window": {k: st[k] for k in ("surfels_size", "n_visible", "n_recent", "n_new", "n_segments_skipped",
                                                                         "n_merged", "n_window_edges")},
                           "regime": "exploring: the sweep that grows the map, the last %d batches of 50 frames before the live-surfel "
                                     "target (40 timed frames each, between two event records on the caller's stream, behind 10 untimed "
                                     "ones); same frame loop, inputs resident, step arrays prepared, no host synchronisation inside" % len(last),
                           "growth_curve": [{k: (round(v, 1) if isinstance(v, float) else v) for k, v in c.items() if k != "timed_frames"}
                                            for c in curve[::max(1, len(curve) // 12)]]}
            if callback is not None:
                callback(g, self.growth)
        for f in list(self.pipe.resident):
            self.pipe.release(f)
        self.api.StreamSynchronize(None)
        return g, live


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
    pipe.upload_counts(reset=True)
    pipe.run_streamed(steps[:warm], uploads[:warm])
    torch.cuda.synchronize()
    t = time.perf_counter()
    pipe.run_streamed(steps[warm:], uploads[warm:])
    t_enq = time.perf_counter() - t           # (the host side of the loop: returns without synchronising)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    n = count - warm
    nbytes = wl.w * wl.h * 5
    for k in keep:
        k.close()
    staged, copy_engine = pipe.upload_counts(reset=True)
    return {"value": n / dt, "unit": "frames/s", "steps": n, "ms_per_step": 1e3 * dt / n,
            "frames_staged_by_copy_kernels": staged, "frames_by_copy_engine": copy_engine,
            "host_enqueue_ms_per_step": 1e3 * t_enq / n,
            "h2d_bytes_per_frame": nbytes, "h2d_GBs": nbytes * n / dt / 1e9,
            "note": "inputs copied from page-locked host memory on the preprocessing stream "
                    "(smx_driver_run_streamed); not the headline value"}


STAGES = ["data_association", "surfel_merging", "measurement_blending", "integration",
          "neighbor_update", "new_surfel_creation", "regularization"]


def handover_probe(wl, n=300):
    """Mean time of an event hand-over between the frame loop's streams on an otherwise idle chip (smx_debug_handover_probe): the
    caller's stream <-> the reconstruction's internal stream <-> the preprocessing queue."""
    import ctypes as C
    from surfelmeshing_amd import _lib
    L = _lib.load()
    internal = C.c_void_p()
    _lib.check(L.smx_recon_debug_internal_stream(wl.pipe.reconstruction._h, C.byref(internal)))
    pre = (C.c_void_p * 2)()
    _lib.check(L.smx_driver_debug_streams(wl.pipe._d, pre))
    caller = wl.pipe._s()
    out = {}
    for name, a, b in (("caller<->internal", caller, internal), ("caller<->preprocessing", caller, C.c_void_p(pre[0])),
                       ("internal<->preprocessing", internal, C.c_void_p(pre[0]))):
        us = C.c_float(0)
        _lib.check(L.smx_debug_handover_probe(a, b, C.c_int32(n), C.byref(us)))
        out[name] = round(float(us.value), 2)
    return out


def stamp_timeline(rec):
    """The pipelined frame as the kernels themselves stamped it (device wall clock, smx_recon_debug_stamp_ring): the last
    calls of the timed region, mean duration of every launch of Integrate in the frame and of the gaps between them -- no
    profiler, no event packets, the run that is timed."""
    ring, khz = rec.debug_stamp_ring()
    recs = {int(r[0]): r.astype(np.int64) for r in ring if r[0] != 0 and r[15] == 0}
    (SEQ, CULL, TILES_END, BLEND_B, BLEND_E, INT_B, INT_E, UPD_B, UPD_E, REG_B, REG_E, SCAN_B, TILES_B, ACC_B, STEP_B) = range(15)
    rows = []
    for q in sorted(recs):
        a, nx = recs[q], recs.get(q + 1)
        if nx is None or min(a[k] for k in (CULL, SCAN_B, TILES_B, BLEND_B, BLEND_E, INT_B, UPD_B, REG_B, STEP_B, REG_E)) == 0:
            continue
        acc_b = a[ACC_B] if a[ACC_B] else a[STEP_B]   # (fused: pass B does the edge work, there is no edge launch)
        us = lambda x, y: (float(y) - float(x)) * 1e3 / khz  # noqa: E731
        rows.append({"cull (+ wait for the previous call's map)": us(a[CULL], a[SCAN_B]), "scan_visible": us(a[SCAN_B], a[TILES_B]),
                     "assoc_tiles": us(a[TILES_B], a[BLEND_B]), "blend": us(a[BLEND_B], a[BLEND_E]),
                     "hand-over to the internal stream (blend end -> integrate begin)": us(a[BLEND_E], a[INT_B]),
                     "integrate+new_flags": us(a[INT_B], a[UPD_B]), "update_neighbors+create": us(a[UPD_B], a[REG_B]),
                     "neighbor_scan": us(a[REG_B], acc_b), "reg_accumulate": us(acc_b, a[STEP_B]), "reg_step": us(a[STEP_B], a[REG_E]),
                     "internal stream: step end -> next integrate begin": us(a[REG_E], nx[INT_B]),
                     "hand-over to the caller's stream (update end -> next pass A begin)": us(a[REG_B], nx[SCAN_B]),
                     "gap integrate end* -> update begin": us(a[INT_E], a[UPD_B]),
                     "gap update end* -> pass B begin": us(a[UPD_E], a[REG_B]),
                     "gap tiles end* -> blend begin": us(a[TILES_END], a[BLEND_B]),
                     "front: cull begin -> blend end": us(a[CULL], a[BLEND_E]),
                     "period (integrate begin -> next integrate begin)": us(a[INT_B], nx[INT_B])})
    if not rows:
        return None
    out = {k: round(float(np.mean([r[k] for r in rows])), 2) for k in rows[0]}
    out["calls_averaged"] = len(rows)
    return out


def timing_passes(wl, plan, base, count, torch):
    """What GetTimings costs the frame loop.  Four variants over 4 x `count` frames behind the timed window, INTERLEAVED in
    chunks of 20 frames (the regime drifts along the trajectory -- visible and recent counts change by tens of percent
    within a few hundred frames -- so every variant samples the whole stretch): stage stamps on (the library default = the
    headline's configuration), stamps off, stamps on + the non-waiting read after every Integrate (APP/main.cc:1511 ported
    with GetTimingsNoWait), and the reference's 14 event records.  No host synchronisation between chunks; a chunk lies
    between two event records on the caller's stream."""
    pipe, rec = wl.pipe, wl.pipe.reconstruction
    variants = (("stamps", 4, 0), ("off", 0, 0), ("stamps_read_every_frame_nowait", 4, 1), ("event_records", 1, 0))
    chunk = 20
    n_chunks = max(len(variants), (4 * count) // chunk)
    arrays = [wl.steps(plan[base + k * chunk:base + (k + 1) * chunk]) for k in range(n_chunks)]
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_chunks + 1)]
    reads = {}
    pipe.timing_sums()
    warm = wl.steps(plan[base - 10:base]) if base >= 10 else None
    if warm is not None:
        pipe.run_array(*warm)
    marks[0].record()
    for k in range(n_chunks):
        name, mode, read = variants[k % len(variants)]
        rec.set_timing_enabled(mode)
        pipe.set_read_timings(read)
        pipe.run_array(*arrays[k])
        marks[k + 1].record()
        if read:
            sums, calls = pipe.timing_sums()
            acc = reads.setdefault(name, [np.zeros(7), 0])
            acc[0] += np.array(sums)
            acc[1] += calls
    torch.cuda.synchronize()
    pipe.set_read_timings(0)
    rec.set_timing_enabled(4)
    out = {}
    for v, (name, mode, read) in enumerate(variants):
        ks = [k for k in range(n_chunks) if k % len(variants) == v]
        ms = sum(marks[k].elapsed_time(marks[k + 1]) for k in ks)
        out[name] = {"value": chunk * len(ks) / (ms * 1e-3), "unit": "frames/s", "steps": chunk * len(ks)}
        if name in reads:
            out[name]["calls_read"] = int(reads[name][1])
            out[name]["mean_stage_ms"] = dict(zip(STAGES, [float(x) / max(reads[name][1], 1) for x in reads[name][0]]))
    ref = out["off"]["value"]
    for name in out:
        out[name]["vs_off"] = out[name]["value"] / ref
    out["note"] = ("interleaved chunks of %d frames behind the timed window (not the headline); the headline runs with the "
                   "library default: stamps on" % chunk)
    return out


def reference_model_bytes(st, P):
    """SURVEY.md 8(d) byte model of the REFERENCE's per-frame traffic (for comparison only: this design does not move
    those bytes; its own compulsory traffic is the sum of ALG_BYTES, `roofline.frame.algorithmic_bytes_per_frame`)."""
    N, E = st["surfels_size"], st["n_edges"]
    return 140 * N + 8 * E + 460 * st["n_visible"] + 360 * st["n_recent"] + 122 * st["n_new"] + 193 * P


def init_ranks(args):
    """rank / world from the launcher's environment; process group (RCCL, or gloo for --dry-run); device selection."""
    from surfelmeshing_amd import multistream
    rank, local_rank, world = multistream.rank_info()
    dist = None
    torch = None
    if world > 1 or not args.dry_run:
        import torch  # first: libsmx then binds to the HIP runtime torch loaded
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run or args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not args.dry_run:
        from surfelmeshing_amd import _lib
        _lib.require_gpu()
        dev = 0
        if world > 1:
            # RCCL wants one rank per device; over gloo the ranks may share one (the streams are independent either way)
            dev = local_rank % _lib.device_count() if args.backend == "gloo" else local_rank
            if args.backend == "gloo":
                torch.cuda.set_device(dev)
        _lib.check(_lib.load().smx_set_device(dev))
    return rank, local_rank, world, dist, torch


def ensure_world(args):
    """--gpus N decides the number of ranks however the script was started.  Under a launcher (WORLD_SIZE set) the two
    must agree; started plainly with --gpus N > 1 the script re-executes itself under torch.distributed.run with N
    ranks on this node (one rank per device, rendezvous on 127.0.0.1) and returns that job's exit code.  None = go on
    in this process."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, env_world), file=sys.stderr)
            return 2
        return None
    if args.gpus <= 1:
        return None
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("# bench.py --gpus %d without a launcher: re-executing under torch.distributed.run" % args.gpus, file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def reduce_device(args):
    return "cpu" if (args.dry_run or args.backend == "gloo") else "cuda"


def finish_ranks(world, dist):
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()



# =====================================================================================================================
# The line the driver reads.  Its record keeps the last 8 000 characters of stdout: the LAST line has to be short (round 5's
# 24 KB line was cut and the driver's record held no headline at all).  Everything measured goes to bench_detail[_<config>].json
# next to this script (and to gpurun_out/ when that directory exists, so that it travels back from the GPU box); the last stdout
# line is compact_line(result): the contract's keys, the two roofline blocks reduced to their figures, the CPU baseline and the
# in-run parity verdicts.  tests/test_bench_line.py holds its length under 6 000 characters on a canned full result.
COMPACT_LIMIT = 6000


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _r(x, digits=5):
    """floats to `digits` significant digits (the detail file keeps the full values)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def _parity_ok(pc):
    """one verdict per in-run parity block: True / False / None (not run)"""
    if not pc:
        return None
    bad = [k for k, v in pc.items() if (v if k == "rows_not_bit_equal" else (k.endswith("_equal") and v is not True))]
    return not bad


def compact_roofline(roof):
    if not roof:
        return None
    out = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                       "avg_launch_ms", "launches_timed"))
    out.setdefault("traffic", None)
    if "avg_launch_ms_by" in roof:
        out["avg_launch_ms_by"] = "launch start/stop events on the launch stream, this run's timed launches"
    if roof.get("trace"):
        out["trace"] = _pick(roof["trace"], ("avg_launch_ms", "launches", "frac", "file"))
    elif roof.get("trace_refused"):
        out["trace_refused"] = str(roof["trace_refused"])[:120]
    raw = roof.get("traffic_pmc_raw") or {}
    if raw.get("traffic_over_algorithmic") is not None:
        out["traffic_over_algorithmic"] = raw["traffic_over_algorithmic"]
    if roof.get("traffic_refused"):
        out["traffic_refused"] = str(roof["traffic_refused"])[:120]
    fr = roof.get("frame")
    if fr:
        out["frame"] = _pick(fr, ("algorithmic_frac", "frac", "algorithmic_bytes_per_frame", "pmc_bytes_per_frame"))
    lk = roof.get("longest_kernel_in_frame")
    if lk:
        out["longest_kernel_in_frame"] = _pick(lk, ("kernel", "in_frame_ms", "bound", "frac_in_frame"))
    return out


def compact_line(d):
    """The short form of a result dictionary (C2 / C3 / C5 alike): see the comment above."""
    out = _pick(d, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "host_enqueue_ms_per_step",
                    "host_enqueue_ms_per_step_per_rank", "host_cores",
                    "higher_is_better", "scaling", "dtype", "data", "per_rank_frames_per_s", "rccl_ranks_seen", "gloo_ranks_seen",
                    "upper_bound"))
    out["vs_baseline"] = d.get("vs_baseline")
    out["config"] = _pick(d.get("config", {}), ("workload", "max_surfel_count", "streams", "parallelism", "backend"))
    out["roofline"] = compact_roofline(d.get("roofline"))
    rv = d.get("roofline_valu")
    if rv:
        out["roofline_valu"] = _pick(rv, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_in_frame",
                                          "avg_launch_ms_alone", "avg_launch_ms_in_frame"))
    cb = d.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "integrate_cores", "kind"))
        c["sample"] = str(cb.get("sample", ""))[:230]
        if cb.get("one_core"):
            c["one_core_value"] = cb["one_core"].get("value")
        if cb.get("c1_single_frame"):
            c["c1_single_frame_ms_all_cores"] = cb["c1_single_frame"].get("ms_all_cores")
        out["cpu_baseline"] = c
        if cb.get("parity_check"):
            out["parity_check"] = dict(cb["parity_check"], ok=_parity_ok(cb["parity_check"]))
    if d.get("parity_check_per_rank"):
        out["parity_check_per_rank"] = d["parity_check_per_rank"]
    if d.get("host_frames"):
        out["host_frames"] = _pick(d["host_frames"], ("value", "unit", "steps", "h2d_GBs"))
        out["host_frames"]["note"] = "PCIe-inclusive pass: inputs arrive from page-locked host memory (not the headline)"
    gp = d.get("growth_phase")
    if gp:
        out["growth_phase"] = _pick(gp, ("value", "unit", "steps", "new_slots_per_frame"))
        if gp.get("parity_check"):
            out["growth_phase"]["parity_ok"] = _parity_ok(gp["parity_check"])
    dist = d.get("distributions") or {}
    out["distributions"] = _pick(dist, ("surfels_size", "merge_count", "n_visible", "n_recent", "n_new", "n_points", "mean_results",
                                        "distance_tests_per_query"))
    tl = d.get("in_frame_timeline_us")
    if tl:
        out["period_us"] = tl.get("period (integrate begin -> next integrate begin)")
    for k in ("index_build", "radius_x2", "general_batch_entry_point"):
        if d.get(k):
            out[k] = _pick(d[k], ("ms", "queries_per_s", "frac"))
    oc = d.get("other_configs")
    if oc:
        out["other_configs"] = {}
        for name, o in oc.items():
            if "error" in o:
                out["other_configs"][name] = {"error": str(o["error"])[:200]}
                continue
            roof = o.get("roofline") or {}
            fr = roof.get("frame") or {}
            cbo = o.get("cpu_baseline") or {}
            e = _pick(o, ("value", "unit", "steps", "warmup", "ms_per_step"))
            e["workload"] = str((o.get("config") or {}).get("workload", ""))[:140]
            e["roofline"] = dict(_pick(roof, ("kernel", "frac", "achieved", "traffic", "avg_launch_ms")),
                                 **({"frame": _pick(fr, ("algorithmic_frac", "frac"))} if fr else {}))
            if o.get("roofline_valu"):
                e["roofline_valu"] = _pick(o["roofline_valu"], ("kernel", "frac", "frac_in_frame"))
            e["parity_ok"] = _parity_ok(o.get("parity_check"))
            e["cpu_baseline"] = _pick(cbo, ("value", "unit", "cores", "kind"))
            out["other_configs"][name] = e
    if d.get("detail"):
        out["detail"] = d["detail"]
    line = json.dumps(_r(out))
    if len(line) > COMPACT_LIMIT:   # (never in the tests; a guard so that the driver's record keeps the headline whatever happens)
        for k in ("other_configs", "distributions", "growth_phase", "host_frames", "roofline_valu"):
            out.pop(k, None)
            line = json.dumps(_r(out))
            if len(line) <= COMPACT_LIMIT:
                break
    return line


def emit(result, args):
    """Full result -> bench_detail[_<config>].json (+ gpurun_out/), compact form -> the last stdout line."""
    name = "bench_detail.json" if args.config == "C2" else "bench_detail_%s.json" % args.config
    paths = [args.detail_out] if getattr(args, "detail_out", "") else [os.path.join(ROOT, name)]
    if not getattr(args, "detail_out", "") and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", name))
    written = []
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(result, f, indent=1)
            written.append(os.path.relpath(p, ROOT))
        except OSError as e:   # (a read-only checkout must not cost the run its line)
            print("# could not write %s: %s" % (p, e), file=sys.stderr)
    result["detail"] = written[0] if written else None
    print(json.dumps(result) if getattr(args, "full_line", False) else compact_line(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["C1", "C2", "C3", "C5"], default="C2", help="BASELINE.json config (SURVEY.md 8d)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--surfels", type=int, default=0, help="LIVE surfels to reach before timing (C2: 5 M, C3: 20 M)")
    ap.add_argument("--cap", type=int, default=0, help="max_surfel_count (default: surfels * 1.25)")
    ap.add_argument("--points", type=int, default=50_000_000, help="C5: surfel positions in the index")
    ap.add_argument("--nn-mode", type=int, default=2, help="C5 A/B: smx_nn_set_query_mode (2 = default)")
    ap.add_argument("--cpu-frames", type=int, default=None, help="frames of the CPU baseline sample (0 = skip; C3 default 4)")
    ap.add_argument("--copy-engine-uploads", action="store_true", help="A/B of the PCIe-inclusive pass: the frames are copied by the copy "
                    "engine in front of their step's preprocessing, in the same queue (rounds 1-4), instead of by kernels on a staging queue")
    ap.add_argument("--host-frames", type=int, default=100,
                    help="frames of the extra pass whose inputs arrive from page-locked host memory (0 = skip)")
    ap.add_argument("--no-check", action="store_true", help="skip the full-size GPU-vs-oracle check")
    ap.add_argument("--growth-frames", type=int, default=200,
                    help="frames of the timed EXPLORING-regime window: the last growth frames before the live-surfel target (0 = skip)")
    ap.add_argument("--timing-frames", type=int, default=100,
                    help="frames of each of the extra passes that price GetTimings (stamps off / read after every frame / the "
                         "reference's 14 event records); 0 = skip")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default C2 run at N = 1: do not run the short C3 (1280x960, 20 M) and C5 (50 M-point search) benches "
                         "whose lines are embedded under other_configs")
    ap.add_argument("--no-overlap", action="store_true", help="A/B: no frame pipelining inside Integrate")
    ap.add_argument("--caller-stream", choices=["null", "plain", "high", "low"], default="null",
                    help="A/B: the stream Integrate is called on (null = the legacy default stream)")
    ap.add_argument("--split-pre", type=int, default=-1, help="A/B: 1 / 0 = two preprocessing queues on / off (default: the library's)")
    ap.add_argument("--stage-timing", choices=["stamps", "off", "events"], default="stamps",
                    help="GetTimings in the timed region: stamps = the library default (the kernels stamp the stage boundaries "
                         "themselves), off, events = the reference's 14 event records (measurement)")
    ap.add_argument("--read-timings", choices=["off", "nowait", "block"], default="off",
                    help="the frame loop reads GetTimings after every Integrate like APP/main.cc:1511: nowait = "
                         "GetTimingsNoWait, block = the reference's waiting call")
    ap.add_argument("--fused-head", action="store_true", help="A/B: bilateral filter and outlier cull in one launch (same images; slower)")
    ap.add_argument("--profile-frames", type=int, default=0, help="A/B: only the first N timed launches of the roofline kernel carry the start / stop events (0 = all of them)")
    ap.add_argument("--profile-kernel", default="", help="A/B: the kernel whose launches carry the start / stop events in the timed region (default: the longest HBM-side kernel)")
    ap.add_argument("--dump-stamps", default="", help="-DSMX_STAMPS builds: the per-workgroup stamps of the last timed frames (tile kernel, blend, edge kernel) and the stage-stamp ring go to this .npz")
    ap.add_argument("--handover", type=int, default=-1, help="A/B: smx_recon_set_handover_mode (1 = device word + gate kernel, 0 = event; default: the library's)")
    ap.add_argument("--pre-cus", type=int, default=0, help="A/B: the preprocessing queues on the first N compute units of the CU mask (N / 8 per XCD); 0 = no partition")
    ap.add_argument("--cu-exclusive", action="store_true", help="with --pre-cus: the internal stream and the caller's stream on the OTHER compute units")
    ap.add_argument("--run-ahead", action="store_true", help="A/B: preprocessing two steps ahead, waits routed off the caller's stream")
    ap.add_argument("--scan-mode", type=int, default=0, help="A/B: smx_recon_set_scan_mode bits (1 = all-slot scans, 2 = multi-launch blend, 4 = no hot-group filter in pass B, 512 = pass B and the edge kernel fused into one launch)")
    ap.add_argument("--ub", default="", help="TIMING-ONLY upper bounds, comma list of: hoist-pre (every timed frame preprocessed "
                    "before the timed region: same results), no-reg (regulariser left out: WRONG map), front-only (pass A + "
                    "association tiles + blend only: WRONG map).  No parity check, no CPU leg; the line carries 'upper_bound'")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process group of --gpus N > 1: nccl (= RCCL, one rank per GPU) or gloo (CPU reductions; ranks may "
                         "share a GPU: rank r uses device r mod device_count -- the C4 path on a 1-GPU box)")
    ap.add_argument("--check-all-ranks", action="store_true", help="N > 1: every rank runs the in-run parity check (oracle) on its own stream")
    ap.add_argument("--dry-run", action="store_true", help="rank path only (no GPU, gloo): see the module docstring")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--full-line", action="store_true", help="tools: print the full result as the last line (not for the driver: it "
                    "keeps 8 000 characters of stdout)")
    ap.add_argument("--detail-out", default="", help="where the full result goes (default: bench_detail[_<config>].json next to "
                    "this script, and gpurun_out/ when it exists); the last stdout line is the compact form")
    args = ap.parse_args()
    if args.config == "C1":
        return run_c1(args)
    rc = ensure_world(args)
    if rc is not None:
        return rc
    if args.config == "C5":
        return run_c5(args)
    return run_integrate(args)


# =====================================================================================================================
# C2 / C3
def run_integrate(args):
    cfg = CONFIGS[args.config]
    width, height = args.width or cfg["width"], args.height or cfg["height"]
    target_live = args.surfels or cfg["surfels"]
    K = args.steps if args.steps is not None else (300 if args.config == "C2" else 100)
    W = args.warmup if args.warmup is not None else (20 if args.config == "C2" else 10)
    cpu_frames = args.cpu_frames if args.cpu_frames is not None else (16 if args.config == "C2" else 4)
    from surfelmeshing_amd import multistream
    rank, local_rank, world, dist, torch = init_ranks(args)
    log = (rank == 0) and not args.quiet
    cap = args.cap or int(target_live * 1.25)
    assign = multistream.stream_assignment(rank)   # one independent stream per rank, no data-path collective

    if args.dry_run:
        # everything the rank does on the host for one run: the plans of the timed window (poses, outlier-cull
        # neighbours and their relative transforms), then the same barrier / reduction protocol as the real run
        t0 = time.perf_counter()
        first = 1000
        plans = [frame_plan(first + j, 4 + j, assign["phase"], 5000.0) for j in range(W + K)]
        assert all(len(p[1]) == 8 and p[2].shape == (8, 3, 4) and p[3].shape == (3, 4) for p in plans)
        if world > 1:
            dist.barrier()
        elapsed_local = max(time.perf_counter() - t0, 1e-9)
        fps, elapsed, units = multistream.aggregate_throughput(K, elapsed_local, world, dist, "cpu")
        each = multistream.per_rank(K / elapsed_local, world, dist, "cpu")
        seen = multistream.ranks_seen(world, dist, "cpu")
        if rank == 0:
            print(json.dumps({"dry_run": True, "config": {"workload": args.config}, "n_gpus": world, "steps": K, "warmup": W,
                              "host_cores": os.cpu_count(),
                              "value": fps, "unit": "frames/s (host-side plan generation only)", "units_all_ranks": units,
                              "per_rank_value": each, "ranks_seen": seen,
                              "seed": assign["seed"], "scaling": "weak"}))
        finish_ranks(world, dist)
        return 0

    from surfelmeshing_amd import _lib, api
    wl = Workload(api, width, height, target_live, cap, assign["seed"], assign["phase"])
    t0 = time.time()
    do_cpu_any = rank == 0 and world == 1 and not args.ub

    def growth_check(g, info):
        """in-run parity check of the exploring regime: the 8 growth frames behind the timed growth window, GPU against
        oracle from the same map state (the main pipeline runs the same frames afterwards)"""
        if not (do_cpu_any and not args.no_check and (args.cpu_frames is None or args.cpu_frames > 0)):
            return
        api.StreamSynchronize(None)
        rec_ = wl.pipe.reconstruction
        state0 = rec_.debug_download_surfels()
        merge0 = rec_.surfels_size() - rec_.surfel_count()
        nchk = 8
        plans = [wl.plan(g + j, g + j) for j in range(nchk)]
        for f in range(g - 4, g + nchk + 5):
            wl.render(f, f)
        api.StreamSynchronize(None)
        r = cpu_baseline(wl, plans, 0, nchk, state0, merge0, cap, True, False, time_one_core=False)
        info["parity_check"] = r.get("parity_check")
        info["cpu_frames_per_s"] = r["value"]

    probe_before = handover_probe(wl)
    g_end, n_live = wl.grow(log, (args.growth_frames, growth_check) if args.growth_frames > 0 else None)
    if log:
        print("# grown to %d live surfels in %d frames, %.1fs" % (n_live, g_end, time.time() - t0), file=sys.stderr)

    # timed window: re-traverse the start of the trajectory (mapped area) with new frame indices.  The first SETTLE frames
    # of the re-traversal are untimed whatever --warmup says: the regulariser window (30 frames) and the integration
    # window still hold the end of the growth phase when it begins, and a short warm-up would time a different regime
    # (round 2: 4144 frames/s at --steps 20 --warmup 5 against 3873 at --steps 300 --warmup 20).
    first = g_end + 10
    ub = set(x for x in args.ub.split(",") if x)
    assert ub <= {"hoist-pre", "no-reg", "front-only", "no-front-wait", "no-upd-wait", "split", "release-after-integrate", "edge-first-chunk", "blend-first-ring"}, "unknown --ub item"
    rec0 = wl.pipe.reconstruction
    names = rec0.kernel_time_names()
    # kernels judged in the frame: the Integrate slots (not the empty slot that measures the time stamps themselves, not the
    # copy-only update that only runs with 0 regulariser iterations) and the driver's three preprocessing stages
    cal_names = [n for n in names if n in ALG_BYTES] + PRE_STAGES
    per, reps = 4, 21
    cal = per * len(cal_names) + 1
    SETTLE = 64
    # Order of the frames: SETTLE untimed frames, the calibration passes (cal frames, counters / events on), the W warm-up
    # frames and directly behind them the K timed ones -- every step array prepared on the host BEFORE the first of
    # them runs, so that the device does not idle (and clock down: the first ~15 frames after a pause of a few
    # milliseconds run 10 % slower) anywhere between the settle frames and the end of the timed region.  The passes that
    # need the host (statistics, per-kernel times, PCIe-inclusive pass, the snapshot for the CPU baseline) follow.
    W_user, W = W, SETTLE
    total = W + cal + W_user + K
    do_host = args.host_frames if (rank == 0 and world == 1 and not ub) else 0   # PCIe-inclusive pass: rank 0 at N = 1 only
    do_cpu = rank == 0 and world == 1 and cpu_frames > 0 and not ub   # CPU baseline: rank 0 at N = 1 only
    chk_frames = max(1, cpu_frames) if (world > 1 and args.check_all_ranks) else 0
    do_timing = args.timing_frames if (rank == 0 and world == 1 and not ub) else 0   # the passes that price GetTimings
    stamps_start = total + 1 + reps                        # stage times by the stamps (reps frames, like the events' pass)
    host_start = stamps_start + reps
    timing_start = host_start + do_host
    cpu_start = timing_start + 4 * do_timing               # its frames: behind everything else
    n_plan = cpu_start + (cpu_frames if do_cpu else 0) + chk_frames
    for j in range(-4, n_plan + 4):
        wl.render(first + j, 4 + j)
    plan = [wl.plan(first + j, 4 + j) for j in range(n_plan)]
    api.StreamSynchronize(None)

    rec = wl.pipe.reconstruction
    if args.caller_stream != "null":
        api.StreamSynchronize(None)
        wl.pipe.stream = api.Stream({"plain": None, "high": 1, "low": -1}[args.caller_stream])
    rec.set_stats_enabled(False)   # the distribution counters are single-address atomics: off while timing
    if args.handover >= 0:
        rec.set_handover_mode(args.handover)
    if args.pre_cus:
        def cu_mask(lo, hi, total=256):
            return [sum(1 << b for b in range(32) if lo <= 32 * w + b < hi) for w in range(total // 32)]
        wl.pipe.set_pre_cu_mask(cu_mask(0, args.pre_cus))
        if args.cu_exclusive:
            api.StreamSynchronize(None)
            rec.set_internal_cu_mask(cu_mask(args.pre_cus, 256))
            wl.pipe.stream = api.Stream(cu_mask=cu_mask(args.pre_cus, 256))
    if args.run_ahead:
        wl.pipe.set_run_ahead(True)
    if args.fused_head:
        wl.pipe.set_fused_head(True)
    timing_mode = {"stamps": 4, "off": 0, "events": 1}[args.stage_timing]
    if args.split_pre >= 0:
        wl.pipe.set_split_preprocessing(args.split_pre == 1)
    if args.scan_mode:
        rec.set_scan_mode(args.scan_mode)
    settle_steps = wl.steps(plan[:W])
    cal_steps = [wl.steps(plan[W + idx * per:W + (idx + 1) * per]) for idx in range(len(cal_names))]
    stats_step = wl.steps(plan[W + cal - 1:W + cal])
    warm_steps = wl.steps(plan[W + cal:W + cal + W_user]) if W_user > 0 else None
    timed_steps = wl.steps(plan[W + cal + W_user:W + cal + W_user + K])
    wl.pipe.run_array(*settle_steps)
    # short calibration pass: which kernel of the frame lasts longest?  Judged IN the frame -- pipelining on, time stamps
    # around one kernel at a time for a few frames each (alone on the chip the candidates lie within 10 % of each other
    # and the choice flipped from run to run; beside the other chains of the frame they do not).  Every kernel of the
    # frame takes part: the Integrate slots on their streams and the three preprocessing stages on the driver's.
    cal_ms = {}
    rec.set_overlap(not args.no_overlap)
    for idx, name in enumerate(cal_names):
        if name in PRE_STAGES:
            wl.pipe.profile_begin(PRE_STAGES.index(name), per)
        else:
            rec.profile_begin(name, per)
        wl.pipe.run_array(*cal_steps[idx])
        api.StreamSynchronize(None)
        ms, n = wl.pipe.profile_end() if name in PRE_STAGES else rec.profile_end()
        cal_ms[name] = ms if n > 0 else 0.0
    rec.set_overlap(False)
    # value distributions of the frame in front of the warm-up frames (counters on for this frame only)
    rec.set_stats_enabled(True)
    wl.pipe.run_array(*stats_step)
    st_before = rec.stats()
    rec.set_stats_enabled(False)
    rec.set_overlap(not args.no_overlap)
    in_timed = [n for n in cal_names if not ("hoist-pre" in ub and n in PRE_STAGES)]
    dominant = max(in_timed, key=lambda n: cal_ms[n])
    dominant_hbm = max((n for n in in_timed if n not in PRE_STAGES), key=lambda n: cal_ms[n])
    # The kernel that is timed with events in the timed region is the frame's longest HBM-side kernel -- the edge kernel
    # (reg_accumulate) in every kernel trace of rounds 3 - 6 -- unless the 4-frame calibration finds another one MORE THAN 15 %
    # longer: pass B and the blend calibrate within a few per cent of it, the choice used to flip from run to run, and WHICH
    # kernel carries the events changes the frame rate (blend, with the event hand-over of rounds 3 - 6: - 10 %, the "slow
    # mode"; pass B: - 3.6 %; the edge kernel: - 0.6 %; profiles/r6_ab_notes.md section 14)
    if "reg_accumulate" in in_timed and cal_ms.get("reg_accumulate", 0.0) > 0 and cal_ms[dominant_hbm] <= 1.15 * cal_ms["reg_accumulate"]:
        dominant_hbm = "reg_accumulate"

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if "hoist-pre" in ub:   # (upper bound: the timed frames find their images preprocessed; same results)
        if warm_steps is not None:
            wl.pipe.prepare_array(*warm_steps)
        wl.pipe.prepare_array(*timed_steps)
    rec.debug_set_skip((1 if "no-reg" in ub else 0) | (2 if "front-only" in ub else 0) | (4 if "no-front-wait" in ub else 0) |
                       (8 if "no-upd-wait" in ub else 0) | (16 if "split" in ub else 0) | (64 if "release-after-integrate" in ub else 0) | (128 if "edge-first-chunk" in ub else 0) | (256 if "blend-first-ring" in ub else 0))
    # Time stamps around the dominant kernel only (2 records per frame on the stream it is launched on) stay on during
    # the timed region; everything else is measured in separate passes.
    if warm_steps is not None:
        wl.pipe.run_array(*warm_steps)       # the W warm-up frames: directly in front of the timed region
    # (the roofline block is the HBM roofline of the longest HBM-side kernel; when a preprocessing stage is the longest kernel
    # of the frame -- the VALU-bound bilateral filter -- it is named in the block with its VALU fractions, `roofline_valu`)
    longest = dominant
    dominant = args.profile_kernel or dominant_hbm
    rec.set_timing_enabled(timing_mode)
    wl.pipe.set_read_timings({"off": 0, "nowait": 1, "block": 2}[args.read_timings])
    rec.profile_begin(dominant, args.profile_frames or K)
    _lib.check(_lib.load().smx_debug_marker(None, 1))   # delimits the timed region in rocprofv3 kernel traces
    sync_all()
    t_start = time.perf_counter()
    wl.pipe.run_array(*timed_steps)          # K frames: preprocessing + Integrate each, enqueued by the C++ loop
    enqueue_local = time.perf_counter() - t_start   # host side done (returns without synchronising)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t_start
    fps, elapsed, _ = multistream.aggregate_throughput(K, elapsed_local, world, dist if world > 1 else None, reduce_device(args))
    each_rank = multistream.per_rank(K / elapsed_local, world, dist if world > 1 else None, reduce_device(args))
    seen = multistream.ranks_seen(world, dist if world > 1 else None, reduce_device(args))
    # (one host core per rank enqueues that rank's frames: 0.06 - 0.08 ms of a 0.16 ms step at C2 -- eight ranks want eight free cores)
    enqueue_each = multistream.per_rank(1e3 * enqueue_local / K, world, dist if world > 1 else None, reduce_device(args))
    if world > 1:
        dist.barrier()
    dom_ms, dom_n = rec.profile_end()
    wl.pipe.set_read_timings(0)
    timeline = stamp_timeline(rec) if timing_mode == 4 else None
    if args.dump_stamps:   # (diagnosis: a -DSMX_STAMPS build only)
        import ctypes as _C
        wg = np.zeros((3, 8192, 16), np.uint64)
        _lib.check(_lib.load().smx_recon_debug_download_stamps(rec._h, wg.ctypes.data_as(_C.c_void_p)))
        ring, khz = rec.debug_stamp_ring()
        np.savez_compressed(args.dump_stamps, wg=wg, ring=ring, khz=khz, fps=fps)
    read_sums, read_calls = wl.pipe.timing_sums()
    rec.set_timing_enabled(4)
    _lib.check(_lib.load().smx_debug_marker(None, 2))
    rec.debug_set_skip(0)
    P = width * height

    if ub:
        # timing-only run: no statistics of a map that may be wrong, no roofline, no CPU leg
        if rank == 0:
            print(json.dumps({
                "metric": "UPPER BOUND (timing only), frames/s", "upper_bound": sorted(ub),
                "note": "hoist-pre: the timed frames were preprocessed before the timed region (results unchanged); no-reg: "
                        "pass B / edges / step left out (map WRONG); front-only: pass A + tiles + blend only (map WRONG); "
                        "no-front-wait / no-upd-wait: the blend -> integrate / update -> pass A hand-over left out (races: map undefined)",
                "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W_user, "ms_per_step": 1e3 * elapsed / K,
                "config": {"workload": "%s %dx%d" % (args.config, width, height)},
                "in_frame_ms_before_the_bound_was_applied": cal_ms,
                "dominant_in_timed_region": {"kernel": dominant, "avg_launch_ms": dom_ms, "launches": dom_n},
                "frame_before_timed_window": {k: st_before[k] for k in ("surfels_size", "n_visible", "n_recent", "n_new")}}))
        finish_ranks(world, dist)
        return 0

    # value distributions of one more frame (counters on)
    rec.set_overlap(False)
    rec.set_stats_enabled(True)
    wl.pipe.run_array(*wl.steps(plan[total:total + 1]))
    st = rec.stats()
    rec.set_stats_enabled(False)
    # (with the counters on the link scan reads every segment; what it skips otherwise: one more frame, counters off)
    wl.pipe.run_array(*wl.steps(plan[total:total + 1]))
    st["n_link_segments_skipped"] = rec.debug_count_skipped_segments()

    # per-stage and per-kernel device times (separate untimed pass, time stamps on the launch stream; the preprocessing
    # stages one at a time, in turn)
    rec.set_timing_enabled(3)
    stage_ms = np.zeros(7)
    kernel_ms = np.zeros(len(names))
    pre_alone = {n: [] for n in PRE_STAGES}
    for j in range(total + 1, total + 1 + reps):
        which = (j - total - 1) % len(PRE_STAGES)
        wl.pipe.profile_begin(which, 1)
        wl.pipe.run_array(*wl.steps(plan[j:j + 1]))
        stage_ms += np.array(rec.GetTimings())
        kernel_ms += np.array(rec.kernel_times_ms())
        pre_alone[PRE_STAGES[which]].append(wl.pipe.profile_end()[0])
    stage_ms /= reps
    kernel_ms /= reps
    # the same stages by the stage stamps (what GetTimings serves by default), the same way: un-pipelined, one frame at a time
    rec.set_timing_enabled(4)
    stage_ms_stamps = np.zeros(7)
    for j in range(stamps_start, stamps_start + reps):
        wl.pipe.run_array(*wl.steps(plan[j:j + 1]))
        stage_ms_stamps += np.array(rec.GetTimings())
    stage_ms_stamps /= reps
    alone_ms = dict(zip(names, [float(x) for x in kernel_ms]))
    alone_ms.update({n: float(np.mean(v)) for n, v in pre_alone.items() if v})
    if alone_ms.get("reg_accumulate", 0.0) <= 0.0:   # (no edge launch: pass B carries its work and its bytes)
        st["fused_edges"] = 1
        st_before["fused_edges"] = 1

    host_pass = None
    if do_host > 0:
        rec.set_overlap(not args.no_overlap)
        wl.pipe.set_staged_uploads(not args.copy_engine_uploads)
        host_pass = host_frames_pass(wl, plan, host_start, do_host, api, torch)
        host_pass["uploads"] = "copy engine, in the preprocessing queue" if args.copy_engine_uploads else "copy kernels on a staging queue"
        rec.set_overlap(False)
    timing_cost = None
    if do_timing > 0:
        rec.set_overlap(not args.no_overlap)
        timing_cost = timing_passes(wl, plan, timing_start, do_timing, torch)
        rec.set_overlap(False)

    ref_bytes = reference_model_bytes(st, P)
    live = st["surfels_size"] - st["merge_count"]
    result = {
        "metric": "RGB-D frames/s integrated @640x480, 5M live surfels; achieved HBM GB/s" if args.config == "C2" else
                  "RGB-D frames/s integrated @%dx%d, %dM surfel cap; achieved HBM GB/s" % (width, height, target_live // 1000000),
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W_user,
        "ms_per_step": 1e3 * elapsed / K, "host_enqueue_ms_per_step": 1e3 * enqueue_local / K,
        "host_enqueue_ms_per_step_per_rank": enqueue_each, "host_cores": os.cpu_count(), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "per_rank_frames_per_s": each_rank, ("rccl_ranks_seen" if args.backend == "nccl" else "gloo_ranks_seen"): seen,
        "config": {"workload": "%s: synthetic room stream %dx%d, raw depth + colour frames resident in HBM, full preprocessing + "
                               "Integrate per frame, %d live surfels (%d slots), steady-state re-traversal" %
                               (args.config, width, height, live, st["surfels_size"]),
                   "max_surfel_count": cap, "streams": world, "parallelism": "1 independent stream per GPU",
                   "backend": args.backend if world > 1 else None},
        "distributions": st,
        "steady_state": {"settle_frames": SETTLE, "note": "untimed frames of the re-traversal in front of the calibration passes, the --warmup frames and the timed window",
                         "frame_before_timed_window": {k: st_before[k] for k in ("surfels_size", "n_visible", "n_recent", "n_new")},
                         "frame_after_timed_window": {k: st[k] for k in ("surfels_size", "n_visible", "n_recent", "n_new")}},
        "stage_ms": dict(zip(STAGES, [float(x) for x in stage_ms_stamps])),
        "stage_ms_by_event_records": dict(zip(STAGES, [float(x) for x in stage_ms])),
        "stage_ms_note": "GetTimings, un-pipelined, one frame at a time, %d frames each: by the kernels' own stage stamps (the "
                         "library default) and by the reference's 14 event records (measurement mode); stages fused into another "
                         "stage's launch report 0 by stamps" % reps,
        "stage_timing_in_timed_region": {"mode": args.stage_timing, "read_every_frame": args.read_timings,
                                         "calls_read": read_calls,
                                         "mean_stage_ms_read": dict(zip(STAGES, [x / max(read_calls, 1) for x in read_sums])) if read_calls else None},
        "in_frame_timeline_us": timeline,
        "handover_probe_us": {"before_the_run": probe_before, "behind_the_timed_window": handover_probe(wl),
                              "note": "event hand-over between two of the frame loop's streams, empty kernels, idle chip"},
        "growth_phase": getattr(wl, "growth", None),
        "reference_model_bytes_per_frame": ref_bytes,
        "reference_model_GBs": ref_bytes * (K / elapsed) / 1e9,
    }

    parity_all = None
    if world > 1 and args.check_all_ranks:
        # C4 readiness: every rank checks its own stream against the oracle (a few frames from its own map)
        api.StreamSynchronize(None)
        state0 = rec.debug_download_surfels()
        merge0 = rec.surfels_size() - rec.surfel_count()
        nchk = max(1, cpu_frames)
        mine = cpu_baseline(wl, plan, cpu_start, nchk, state0, merge0, cap, True, False, time_one_core=False).get("parity_check")
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        parity_all = gathered
    if rank == 0:
        result["roofline_valu"] = bilateral_valu_roofline(wl, api, torch, plan[0][0], cal_ms.get("bilateral"))
        result["roofline"] = roofline_block(st, P, dominant, longest, dom_ms, dom_n, alone_ms, cal_ms,
                                            1e3 * elapsed / K, args.config, result["roofline_valu"],
                                            dict(st_before, n_link_segments_skipped=st.get("n_link_segments_skipped", 0)))
        if parity_all is not None:
            result["parity_check_per_rank"] = parity_all
        if host_pass is not None:
            result["host_frames"] = host_pass
        if timing_cost is not None:
            result["stage_timing_cost"] = timing_cost
        if do_cpu:
            api.StreamSynchronize(None)
            state0 = rec.debug_download_surfels()
            merge0 = rec.surfels_size() - rec.surfel_count()
            result["cpu_baseline"] = cpu_baseline(wl, plan, cpu_start, cpu_frames, state0, merge0, cap,
                                                  not args.no_check, log)
        if (args.config == "C2" and world == 1 and not args.no_other_configs and not args.surfels and not args.width
                and not args.height and not args.scan_mode and do_cpu):
            result["other_configs"] = other_configs(args, log)
        emit(result, args)
    finish_ranks(world, dist)
    return 0


def other_configs(args, log):
    """BASELINE.json's other single-GPU configurations, short runs of this same script in processes of their own behind the
    C2 work (this process keeps its ~2 GB map; the chip has 288 GB): C3 = 1280x960 stream / 20 M surfel cap, C5 = 50 M-point
    radius search.  Their lines -- value, roofline, in-run parity check against the oracle, CPU leg -- are embedded."""
    import subprocess
    runs = {"C3": ["--config", "C3", "--steps", "60", "--warmup", "10", "--cpu-frames", "2", "--host-frames", "0",
                   "--timing-frames", "0", "--growth-frames", "100"],
            "C5": ["--config", "C5", "--steps", "3", "--warmup", "1"]}
    out = {}
    for name, extra in runs.items():
        t0 = time.time()
        # (the child's full result travels through a file in a temporary directory -- the checkout may be read-only -- and is
        # copied next to this script, and to gpurun_out/, afterwards)
        import shutil
        import tempfile
        tmpdir = tempfile.mkdtemp(prefix="smx_bench_")
        detail = os.path.join(tmpdir, "bench_detail_%s.json" % name)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--quiet", "--no-other-configs", "--detail-out", detail] + extra
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
            if r.returncode != 0 or not os.path.exists(detail):
                out[name] = {"error": "rc %d: %s" % (r.returncode, r.stderr[-600:])}
                shutil.rmtree(tmpdir, ignore_errors=True)
                continue
            d = json.load(open(detail))
        except (subprocess.TimeoutExpired, ValueError, OSError) as e:
            out[name] = {"error": repr(e)[:600]}
            shutil.rmtree(tmpdir, ignore_errors=True)
            continue
        for dst in (ROOT, os.path.join(ROOT, "gpurun_out")):
            if os.path.isdir(dst):
                try:
                    shutil.copy(detail, os.path.join(dst, os.path.basename(detail)))
                except OSError:
                    pass
        shutil.rmtree(tmpdir, ignore_errors=True)
        roof = dict(d.get("roofline") or {})
        roof.pop("kernels", None)
        roof.pop("traffic_pmc_raw", None)
        cb = d.get("cpu_baseline") or {}
        out[name] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"],
                     "ms_per_step": d["ms_per_step"], "config": d["config"], "roofline": roof,
                     "roofline_valu": {k: v for k, v in (d.get("roofline_valu") or {}).items() if k != "note"} or None,
                     "parity_check": cb.get("parity_check"),
                     "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")} if cb else None,
                     "growth_phase": d.get("growth_phase"),
                     "distributions": {k: d.get("distributions", {}).get(k) for k in ("surfels_size", "n_visible", "n_recent", "n_new",
                                                                                      "n_points", "mean_results") if k in d.get("distributions", {})},
                     "command": " ".join(["python", "bench.py"] + cmd[2:]), "detail": os.path.basename(detail),
                     "wall_s": time.time() - t0}
        if log:
            print("# other config %s: %.4g %s (%.0fs)" % (name, d["value"], d["unit"], time.time() - t0), file=sys.stderr, flush=True)
    return out


# kernel-slot name -> kernel name in rocprofv3 output
SLOT_KERNEL = {"cull_segments": "k_cull_segments", "reg_accumulate": "k_reg_accumulate", "reg_step": "k_reg_step", "neighbor_scan": "k_neighbor_scan<true, true",
               "scan_visible": "k_scan_visible", "assoc_tiles": "k_assoc_tiles", "blend": "k_blend_tiles",
               "integrate+new_flags": "k_integrate<true>", "update_neighbors+create": "k_update_and_create<true>",}


def pmc_file(config="C2"):
    """The committed rocprofv3 --pmc summary of this same command (profiles/pmc_traffic[_C3|_C5].json, written by
    tools/pmc_summary.py through tools/profile_round.sh, which stamps it with the hash of the kernel sources it was
    collected on and with the counts of the PMC runs themselves).  A file collected on other sources is REFUSED:
    (None, reason)."""
    name = "pmc_traffic.json" if config == "C2" else "pmc_traffic_%s.json" % config
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no profiles/" + name
    try:
        d = json.load(open(path))
    except (ValueError, OSError) as e:
        return None, "unreadable: %s" % e
    meta = d.get("_meta", {})
    if meta.get("source_sha") != source_sha():
        return None, "stale: collected on kernel sources %s, this build is %s" % (meta.get("source_sha"), source_sha())
    return d, None


def trace_file(config, slot, steps):
    """The committed rocprofv3 --kernel-trace summary of this same command (profiles/trace_timed_region[_C3].json, written by
    tools/prof_summary.py through tools/profile_round.sh): average duration of the slot's kernel between the two marker kernels
    of the timed region, from the trace whose frame count is nearest to this run's.  Refused like the PMC file when it was
    collected on other kernel sources."""
    name = "trace_timed_region.json" if config == "C2" else "trace_timed_region_%s.json" % config
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no profiles/" + name
    try:
        d = json.load(open(path))
    except (ValueError, OSError) as e:
        return None, "unreadable: %s" % e
    meta = d.get("_meta", {})
    if meta.get("source_sha") != source_sha():
        return None, "stale: collected on kernel sources %s, this build is %s" % (meta.get("source_sha"), source_sha())
    runs = [(k, v) for k, v in d.items() if k != "_meta" and v.get("frames")]
    if not runs:
        return None, "empty"
    key, run = min(runs, key=lambda kv: abs(math.log(max(kv[1]["frames"], 1) / max(steps, 1))))
    pref = SLOT_KERNEL.get(slot, slot)
    k = max((v for n, v in run["kernels"].items() if n.startswith(pref)), key=lambda v: v["calls"], default=None)
    if k is None:
        return None, "no kernel %s in the trace" % pref
    return {"avg_launch_ms": k["avg_us"] * 1e-3, "launches": k["calls"], "frames_in_trace": run["frames"],
            "command": meta.get(key), "file": "profiles/" + name}, None


PMC_NOTE = ("HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB).  Calibrated on known byte counts in this design's own access patterns "
            "(profiles/r17_counter_calibration.md): every read request to memory is 128 bytes and FETCH_SIZE books it as 64, for "
            "streams and 16-byte gathers alike (a gather moves the whole 128-byte line); WRITE_SIZE is exact (32-byte requests for "
            "sparse 16-byte stores)")


def pmc_bytes(k):
    """HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (KB; separate passes).  Correction per
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of the bytes read, so it is
    doubled; WRITE_SIZE is taken as is."""
    if "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
        return None
    return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0


def roofline_block(st, P, dominant, longest, dom_ms, dom_n, alone_ms, in_frame_ms, ms_per_step, config="C2", valu=None, st_in_frame=None):
    """HBM roofline of the HBM-side kernel that lasts longest IN THE FRAME: algorithmic bytes per launch (ALG_BYTES,
    DESIGN.md) / average launch duration measured with time stamps on its launch stream over the timed region.  Every
    kernel of the frame is judged (the preprocessing stages included): `longest_kernel_in_frame` names the overall longest
    one -- the VALU-bound bilateral filter, whose fractions of the packed-FMA peak, alone and in the frame, ride along
    (`roofline_valu`).  `frame` = all kernels of one frame / the measured frame time, by
    the algorithmic bytes of this run and by the PMC bytes of the committed profile of the same build.  `kernels` = every
    kernel alone (unpipelined pass, the measured cost of an empty pair of time stamps subtracted) and in the frame (the
    short calibration passes in front of the timed region)."""
    alg = ALG_BYTES[dominant](st, P)
    achieved = alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    overhead = alone_ms.get("empty_slot", 0.0)
    per_kernel = {}
    alg_frame = 0.0
    for k, ms in alone_ms.items():
        if k in ALG_BYTES and ms > 0:
            b = ALG_BYTES[k](st, P)
            alg_frame += b
            net = max(ms - overhead, 1e-6)
            # (the in-frame passes ran in FRONT of the timed window, the stand-alone pass behind it: each with the counts
            # of its own end of the window -- at C3 the visible set grows by half across it)
            b_in = ALG_BYTES[k](st_in_frame, P) if st_in_frame else b
            per_kernel[k] = {"ms_with_event_overhead": ms, "alone_ms": net, "algorithmic_MB": b / 1e6,
                             "alone_frac_of_hbm_peak": b / (net * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "in_frame_ms": in_frame_ms.get(k), "in_frame_algorithmic_MB": b_in / 1e6,
                             "in_frame_frac_of_hbm_peak": (b_in / (in_frame_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS) if in_frame_ms.get(k) else None}
            if per_kernel[k]["alone_frac_of_hbm_peak"] > 0.8:
                per_kernel[k]["note"] = ("above what HBM delivers (6.5 TB/s read): the records this kernel asks for were read by the "
                                         "kernel in front of it and part of them is still in the 256 MB Infinity Cache / the L2s")
    pmc, why = pmc_file(config)
    traffic = raw = None
    frame = {"algorithmic_bytes_per_frame": alg_frame, "algorithmic_GBs": alg_frame / (ms_per_step * 1e-3) / 1e9,
             "algorithmic_frac": alg_frame / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "note": "sum of the kernels' algorithmic bytes (ALG_BYTES, preprocessing included) / the measured frame time"}
    if pmc is not None:
        at = pmc.get("_meta", {}).get("surfel_slots")
        if not at or abs(at - st["surfels_size"]) > 0.1 * st["surfels_size"]:
            pmc, why = None, "collected at %s surfel slots, this run has %d" % (at, st["surfels_size"])
    if pmc is not None:
        # (template instances: the kernel of the slot is the entry whose name begins with the slot's kernel name)
        pref = SLOT_KERNEL.get(dominant, "\0")
        k = max((v for n, v in pmc.items() if n != "_meta" and n.startswith(pref)), key=lambda v: v.get("launches", 0), default={})
        traffic = pmc_bytes(k)
        # like for like: the algorithmic bytes of the PMC run's own counts (its window differs from this run's)
        own = pmc.get("_meta", {}).get("distributions_of_the_pmc_run") or {}
        alg_own = ALG_BYTES[dominant](own, P) if all(x in own for x in ("surfels_size", "n_visible", "n_recent", "n_edges")) else None
        raw = {"FETCH_SIZE_KB": k.get("FETCH_SIZE"), "WRITE_SIZE_KB": k.get("WRITE_SIZE"),
               "collected_at_surfel_slots": pmc.get("_meta", {}).get("surfel_slots"),
               "algorithmic_bytes_of_the_pmc_run": alg_own,
               "traffic_over_algorithmic": (traffic / alg_own) if (traffic and alg_own) else None,
               "counter_note": PMC_NOTE}
        total = sum(b for b in (pmc_bytes(v) for n, v in pmc.items() if n != "_meta") if b)
        frame.update({"pmc_bytes_per_frame": total, "GBs": total / (ms_per_step * 1e-3) / 1e9,
                      "frac": total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "pmc_note": "sum over the kernels of one frame of 2 x FETCH_SIZE + WRITE_SIZE / the measured frame time"})
    trace, trace_why = trace_file(config, dominant, dom_n)
    if trace is not None:
        trace["frac"] = alg / (trace["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        trace["over_this_run"] = trace["avg_launch_ms"] / dom_ms if dom_ms > 0 else None
    return {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_pmc_raw": raw, "traffic_refused": why,
            "frame": frame, "algorithmic_bytes_per_launch": alg,
            "avg_launch_ms": dom_ms, "launches_timed": dom_n, "surfel_slots": st["surfels_size"],
            "avg_launch_ms_by": "the launch's own start / stop events (hipExtLaunchKernelGGL) on its launch stream, averaged over the "
                                "timed launches of THIS run; `trace` = the same kernel in the committed rocprofv3 kernel trace of the "
                                "same command",
            "trace": trace, "trace_refused": trace_why,
            "longest_kernel_in_frame": {"kernel": longest, "in_frame_ms": in_frame_ms.get(longest),
                                        "bound": "valu_fp32" if longest in PRE_STAGES else "hbm",
                                        "frac_in_frame": (valu or {}).get("frac_in_frame") if longest == "bilateral" else None,
                                        "frac_alone": (valu or {}).get("frac") if longest == "bilateral" else None,
                                        "note": "bound by VALU issue, not by HBM (4 bytes per pixel): see roofline_valu"
                                                if longest in PRE_STAGES else None},
            "event_overhead_ms": overhead,
            "kernels": per_kernel}


def bilateral_valu_roofline(wl, api, torch, frame, in_frame_ms=None):
    """The longest single dispatch of a frame is the bilateral filter, and it is VALU-bound (113 taps x exp at radius 6):
    its roofline is the FP32 vector peak.  Flops per in-region pixel and tap: range term 5 (sub, mul, mul, add + the
    spatial table value), det_expf 19 (2 mul/rint, 2 fma range reduction, 6 fma polynomial, 2 mul + add, scale),
    accumulation 4 (cvt + fma, add) = 28.  Timed alone (no other chain on the chip), 50 back-to-back launches."""
    import ctypes as C
    from surfelmeshing_amd import _lib
    L = _lib.load()
    p = wl.pre
    depth = api.CUDABuffer(wl.h, wl.w, np.uint16)
    out = api.CUDABuffer(wl.h, wl.w, np.uint16)
    d, _ = wl.pipe.download_frame(frame)
    depth.UploadAsync(None, d)
    radius = int(p.bilateral_filter_radius_factor * p.bilateral_filter_sigma_xy + 0.5)
    taps = sum(1 for dy in range(-radius, radius + 1) for dx in range(-radius, radius + 1) if dx * dx + dy * dy <= radius * radius)
    yy, xx = np.mgrid[0:wl.h, 0:wl.w]
    inside = ((xx - wl.w // 2) ** 2 + (yy - wl.h // 2) ** 2 <= p.depth_valid_region_radius ** 2) & (d > 0) & (d <= p.max_depth_u16())
    flops = 28.0 * taps * float(inside.sum())

    def run(n):
        for _ in range(n):
            api.BilateralFilteringAndDepthCutoffCUDA(None, p.bilateral_filter_sigma_xy, p.bilateral_filter_sigma_depth_factor, 0,
                                                     p.bilateral_filter_radius_factor, p.max_depth_u16(),
                                                     p.depth_valid_region_radius, depth, out)
    run(5)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(50)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 50 * 1e3
    depth.close()
    out.close()
    tf = flops / (ms * 1e-3) / 1e12
    return {"bound": "valu_fp32", "kernel": "k_bilateral_p<%d>" % radius, "achieved": tf, "peak": VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TFLOPS, "flop_per_launch": flops, "taps": taps,
            "avg_launch_ms_alone": ms,
            "avg_launch_ms_in_frame": in_frame_ms,
            "frac_in_frame": (flops / (in_frame_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS) if in_frame_ms else None,
            "note": "timed alone, back to back; inside the frame it shares the chip with two other chains.  28 flop per "
                    "tap is the algorithm's count; the kernel issues 13.5 VALU instructions per tap (two taps per packed "
                    "fp32 instruction wherever the ISA has one), peak = packed FMA rate"}


def cpu_baseline(wl, plan, W, frames, state0, merge0, cap, check, log, time_one_core=True):
    """The oracle (plain C loops) on the `frames` frames behind the snapshot (taken after all GPU passes: the trajectory goes on behind the timed window), starting from the same surfel
    state: once with the per-pixel stages row-parallel on all host cores (the headline CPU number; Integrate itself is
    a sequential scan over the surfels and stays on one core) and once on a single core; the run also serves as a
    full-size parity check of the HIP path.  Plus config C1 (single frame, bilateral + erosion + normals)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    from oracle import binding
    from oracle_pipeline import OraclePipeline
    api = wl.api
    need = set()
    for j in range(W, W + frames):
        need.add(plan[j][0])
        need.update(plan[j][1])
    host_frames = {f: wl.pipe.download_frame(f) for f in sorted(need)}
    n0 = state0.shape[1]
    cores = min(os.cpu_count() or 1, 32)   # threads actually used (one row band each; more bands than this only add dispatch overhead)
    po = None
    timings = {}
    for threads in ((cores, 1) if time_one_core else (cores,)):
        binding.set_row_threads(threads)
        po = OraclePipeline(wl.w, wl.h, wl.fx, wl.fy, wl.cx, wl.cy, cap, wl.pre)
        po.recon.surfels()[:, :n0] = state0
        po.recon.set_counts(n0, merge0)
        for f, (d, c) in host_frames.items():
            po.upload(f, d, c)
        nf = frames if threads == cores else max(2, frames // 4)
        t_pre = t_int = 0.0
        for j in range(W, W + nf):
            t0 = time.perf_counter()
            po.preprocess(*plan[j][:3])
            t1 = time.perf_counter()
            po.integrate(plan[j][0], plan[j][3])
            t_int += time.perf_counter() - t1
            t_pre += t1 - t0
        timings[threads] = (nf, t_pre, t_int)
        if threads == cores:
            po_full = po
    binding.set_row_threads(1)
    nf, t_pre, t_int = timings[cores]
    nf1, t_pre1, t_int1 = timings.get(1, (1, 0.0, 0.0))
    out = {"value": nf / (t_pre + t_int), "unit": "frames/s", "cores": cores, "integrate_cores": 1, "kind": "port",
           "cores_note": "%d threads serve the per-pixel stages only (%.0f %% of the CPU time per frame is Integrate on ONE core: "
                         "a sequential scan over the surfels)" % (cores, 100.0 * t_int / (t_pre + t_int)),
           "sample": "%d frames behind the snapshot (the trajectory continued behind the timed window) from the same %d-surfel state (oracle, gcc -O2): per-pixel stages "
                     "row-parallel on %d threads (%.1f ms/frame), Integrate on 1 thread (%.1f ms/frame)" %
                     (nf, n0, cores, 1e3 * t_pre / nf, 1e3 * t_int / nf),
           }
    if time_one_core:
        out["one_core"] = {"value": nf1 / (t_pre1 + t_int1), "unit": "frames/s", "cores": 1, "frames": nf1,
                           "per_pixel_stages_ms": 1e3 * t_pre1 / nf1, "integrate_ms": 1e3 * t_int1 / nf1}
        out["c1_single_frame"] = c1_timing(wl.w, wl.h)
    if check:
        from surfelmeshing_amd.pipeline import FramePipeline
        po = po_full
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


# =====================================================================================================================
# C1: BASELINE.json configs[0] -- single 640x480 synthetic depth frame, bilateral filter + erosion + normals via the
# naive CPU loops (erosion supplies the zero border the normals stage assumes, cuda_depth_processing.cu:659-662)
def c1_timing(width=640, height=480, reps_all=10, reps_one=3):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    from oracle import binding
    from surfelmeshing_amd.synth import SyntheticStream
    sc = width / 640.0
    s = SyntheticStream(width=width, height=height, fx=525.0 * sc, fy=525.0 * sc, cx=320.0 * sc, cy=240.0 * sc)
    depth, _ = s.frame(0)
    cores = min(os.cpu_count() or 1, 32)
    res = {}
    for threads, reps in ((cores, reps_all), (1, reps_one)):
        binding.set_row_threads(threads)
        t0 = time.perf_counter()
        for _ in range(reps):
            a = orc.bilateral_filter_and_cutoff(depth, max_depth=50000, depth_valid_region_radius=333.0 * sc)
            a = orc.erode_depth_map(a, 2)
            orc.compute_normals_and_drop_bad_pixels(a, s.fx, s.fy, s.cx, s.cy)
        res[threads] = (time.perf_counter() - t0) / reps
    binding.set_row_threads(1)
    return {"workload": "C1: one %dx%d frame, bilateral + erosion(2) + normals (oracle, gcc -O2)" % (width, height),
            "ms_all_cores": 1e3 * res[cores], "cores": cores, "ms_one_core": 1e3 * res[1],
            "ns_per_pixel_one_core": 1e9 * res[1] / (width * height)}


def run_c1(args):
    r = c1_timing()
    print(json.dumps({"metric": "single 640x480 depth frame: bilateral filter + erosion + normals, naive CPU loop (BASELINE.json configs[0])",
                      "value": 1e3 / r["ms_all_cores"], "unit": "frames/s", "n_gpus": 0, "steps": 10, "warmup": 0,
                      "ms_per_step": r["ms_all_cores"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "f32", "data": "synthetic", "config": {"workload": r["workload"]},
                      "cpu_baseline": {"value": 1e3 / r["ms_all_cores"], "unit": "frames/s", "cores": r["cores"], "kind": "port",
                                       "sample": "10 repetitions of the frame", "one_core_ms": r["ms_one_core"],
                                       "ns_per_pixel_one_core": r["ns_per_pixel_one_core"]}}))
    return 0


# =====================================================================================================================
# C5: BASELINE.json configs[4] -- 50 M live surfels, neighbor search dominant
def run_c5(args):
    import ctypes as C
    from surfelmeshing_amd import multistream
    rank, local_rank, world, dist, torch = init_ranks(args)
    if args.dry_run:
        finish_ranks(world, dist)
        return 0
    from surfelmeshing_amd import _lib, api
    from surfelmeshing_amd.synth import room_surface_points
    L = _lib.load()
    log = (rank == 0) and not args.quiet
    steps = args.steps if args.steps is not None else 5
    warm = args.warmup if args.warmup is not None else 1
    Kn = 64
    t0 = time.time()
    pts, spacing = room_surface_points(args.points, seed=0x5EED0005 + rank)
    n = len(pts)
    r = np.float32(1.5 * spacing)                       # surfel radius = 1.5 x local spacing
    if log:
        print("# C5: %d points, spacing %.2f mm, radius %.2f mm (generated in %.1fs)" % (n, spacing * 1e3, r * 1e3, time.time() - t0), file=sys.stderr)

    def dev(host):
        host = np.ascontiguousarray(host)
        b = api.CUDABuffer(1, host.size, host.dtype)
        b.UploadAsync(None, host.reshape(1, -1))
        return b
    bx, by, bz = (dev(pts[:, k]) for k in range(3))
    br2 = dev(np.full(n, r * r, np.float32))
    ptr = lambda b: C.c_void_p(b.ToCUDA().address)  # noqa: E731
    # result rows: n x K x 8 B (25.6 GB at 50 M) -- torch is the device allocator here (plumbing; 3.2 G elements do not
    # fit the int32 width of a CUDABuffer)
    dev_t = torch.device("cuda", local_rank if world > 1 else 0)
    t_idx = torch.empty(n * Kn, dtype=torch.int32, device=dev_t)
    t_d2 = torch.empty(n * Kn, dtype=torch.float32, device=dev_t)
    t_cnt = torch.zeros(n, dtype=torch.int32, device=dev_t)

    class _Raw:   # (same accessors as a CUDABuffer, for ptr())
        def __init__(self, t):
            self.t = t

        def ToCUDA(self):
            return self

        @property
        def address(self):
            return self.t.data_ptr()
    out_idx, out_d2, out_cnt = _Raw(t_idx), _Raw(t_d2), _Raw(t_cnt)
    api.StreamSynchronize(None)
    torch.cuda.synchronize()
    nn = api.SurfelNeighborIndex()
    nn.set_query_mode(args.nn_mode)

    def build():
        _lib.check(L.smx_nn_build(nn._h, None, ptr(bx), ptr(by), ptr(bz), C.c_uint32(n), C.c_float(float(r)), C.c_int32(1)))

    def query_self(factor):
        _lib.check(L.smx_nn_query_self(nn._h, None, ptr(br2), C.c_float(factor), C.c_int32(Kn), C.c_void_p(0), C.c_uint8(0),
                                       ptr(out_idx), ptr(out_d2), ptr(out_cnt)))

    def timed(fn, reps):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    build()
    t_build = timed(build, 3)
    info = nn.stats()
    for _ in range(warm):
        query_self(1.0)
    _lib.check(L.smx_debug_marker(None, 1))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(steps):
        query_self(1.0)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t_start
    qps, elapsed, _ = multistream.aggregate_throughput(float(n) * steps, elapsed_local, world, dist if world > 1 else None, reduce_device(args))
    _lib.check(L.smx_debug_marker(None, 2))
    # counters of one more pass -> algorithmic bytes per query (16 (position + r^2) + 12 x the candidates
    # staged per tile, amortised over the queries of the tile + 8 per result + 4 (count))
    nn.set_stats_enabled(True)
    query_self(1.0)
    st1 = nn.stats()
    nn.set_stats_enabled(False)
    cnt1 = t_cnt.cpu().numpy().copy()
    bytes_per_query = 16.0 + 12.0 * st1["staged_candidates"] / n + 8.0 * st1["results"] / n + 4.0
    ms_step = 1e3 * elapsed / steps
    achieved = bytes_per_query * n / (ms_step * 1e-3) / 1e9
    # secondary numbers: twice the radius (max search-range factor, main.cc:392), and the general batch entry point
    t_2r = timed(lambda: query_self(4.0), 2)
    cnt2 = t_cnt.cpu().numpy().copy()
    batch = min(n, 16_000_000)
    def query_batch():
        for q0 in range(0, n, batch):
            nq = min(batch, n - q0)
            off = lambda b: C.c_void_p(b.ToCUDA().address + 4 * q0)  # noqa: E731
            _lib.check(L.smx_nn_query_batch(nn._h, None, C.c_uint32(nq), off(bx), off(by), off(bz), off(br2), C.c_int32(Kn),
                                            C.c_void_p(0), C.c_uint8(0), C.c_int32(1), ptr(out_idx), ptr(out_d2),
                                            ptr(out_cnt), C.c_int32(1)))
    query_batch()
    t_batch = timed(query_batch, 2)
    build_bytes = 12.0 * n + 8.0 * n + 16.0 * info["n_bricks"]
    # committed PMC profile of this command (profiles/pmc_traffic_C5.json): HBM traffic and the VALU issue-slot utilisation
    c5_traffic = c5_valu = None
    pmc, c5_why = pmc_file("C5")
    if pmc is not None and pmc.get("_meta", {}).get("surfel_slots") != n:
        pmc, c5_why = None, "collected at %s points, this run has %d" % (pmc.get("_meta", {}).get("surfel_slots"), n)
    if pmc is not None:
        k = next((v for name, v in pmc.items() if name.startswith("k_query_lanes")), {})
        c5_traffic = pmc_bytes(k)
        if "SQ_INSTS_VALU" in k:
            # a wave64 VALU instruction issues over 2 cycles on a SIMD-32, four SIMDs per CU: two per CU per cycle
            # (measured: 1.79 with v_add_f32 at 8 wavefronts per SIMD, profiles/r17_counter_calibration.md)
            peak = 256 * 2 * 2.4e9
            rate = k["SQ_INSTS_VALU"] / (ms_step * 1e-3)
            c5_valu = {"bound": "valu-issue", "kernel": "k_query_lanes", "valu_wave_instructions_per_launch": k["SQ_INSTS_VALU"],
                       "salu_wave_instructions_per_launch": k.get("SQ_INSTS_SALU"), "waves_per_launch": k.get("SQ_WAVES"),
                       "valu_wave_instructions_per_query": k["SQ_INSTS_VALU"] / n,
                       "achieved": rate / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s", "frac": rate / peak,
                       "frac_of_measured_ceiling": rate / (256 * 1.79 * 2.4e9),
                       "note": "SQ_INSTS_VALU of the committed PMC pass / this run's step time; peak = 256 CUs x 2 VALU "
                               "wave-instructions per cycle x 2.4 GHz (4 SIMD-32 per CU, 2 cycles per wave64 instruction); the "
                               "measured ceiling is 1.79 per CU per cycle (profiles/r17_counter_calibration.md)"}
    result = {
        "metric": "radius-neighbor queries/s, every one of 50M surfels queries its own neighbourhood, K=64 (BASELINE.json configs[4])",
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C5: %d surfel positions on the room surface (spacing %.2f mm), index cell = search radius = "
                               "1.5 x spacing, self-queries for all points (smx_nn_query_self), K = 64" % (n, spacing * 1e3),
                   "streams": world, "parallelism": "1 independent cloud per GPU"},
        "distributions": {"n_points": n, "n_bricks": info["n_bricks"], "grid_cells": info["dim"], "key_bits": info["key_bits"],
                          "mean_results": float(cnt1.mean()), "max_results": int(cnt1.max()), "tiles": st1["tiles"],
                          "staged_candidates_per_query": st1["staged_candidates"] / n, "distance_tests_per_query": st1["distance_tests"] / n},
        "roofline": {"bound": "hbm", "kernel": "k_query_lanes", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": c5_traffic, "traffic_refused": c5_why,
                     "algorithmic_bytes_per_launch": bytes_per_query * n,
                     "algorithmic_bytes_per_query": bytes_per_query, "avg_launch_ms": ms_step,
                     "note": "one step = k_query_lanes over all tiles (+ k_query_tiles, which returns at once unless a tile "
                             "was marked for it): the step time is the kernel time; the kernel is bound by VALU issue, "
                             "not by bytes -- roofline_valu, DESIGN.md"},
        "roofline_valu": c5_valu,
        "index_build": {"ms": 1e3 * t_build, "algorithmic_bytes": build_bytes, "GBs": build_bytes / t_build / 1e9,
                        "frac": build_bytes / t_build / 1e9 / HBM_PEAK_GBS, "note": "12 N read + 8 N written + 16 B per occupied brick"},
        "radius_x2": {"queries_per_s": n / t_2r, "ms": 1e3 * t_2r, "mean_results": float(cnt2.mean()), "max_results": int(cnt2.max())},
        "general_batch_entry_point": {"queries_per_s": n / t_batch, "ms": 1e3 * t_batch,
                                      "note": "smx_nn_query_batch over the same positions: keys + radix sort + gather of the queries first"},
    }
    if rank == 0:
        if world == 1 and (args.cpu_frames is None or args.cpu_frames > 0):
            result["cpu_baseline"] = c5_cpu_baseline(nn, pts, r, Kn, not args.no_check)
        emit(result, args)
    nn.close()
    finish_ranks(world, dist)
    return 0


def c5_cpu_baseline(nn, pts, r, Kn, check):
    """The oracle's uniform-grid search (plain C, one thread; pinned to brute force by tests/test_nn_oracle.py) over ALL
    points, on a bounded sample of the self-queries; the same sample through the GPU index is compared bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    n = len(pts)
    rng = np.random.default_rng(0xC5)
    sel = np.sort(rng.choice(n, min(n, 100_000), replace=False))
    x, y, z = (np.ascontiguousarray(pts[:, k]) for k in range(3))
    r2 = np.full(len(sel), r * r, np.float32)
    t0 = time.perf_counter()
    ocnt, od2, oidx = orc.nn_grid_batch(x, y, z, 0.05, x[sel], y[sel], z[sel], r2, Kn)
    dt = time.perf_counter() - t0
    out = {"value": len(sel) / dt, "unit": "queries/s", "cores": 1, "kind": "port",
           "sample": "%d sampled self-queries over all %d points (oracle grid, 5 cm cells, gcc -O2, 1 thread; the time "
                     "includes building its grid)" % (len(sel), n)}
    if check:
        cnt, d2, idx = nn.FindNearestSurfelsWithinRadius(pts[sel], r2, Kn)
        m = np.arange(Kn)[None, :] < ocnt[:, None]
        out["parity_check"] = {"queries": int(len(sel)), "counts_equal": bool(np.array_equal(cnt, ocnt)),
                               "indices_equal": bool(np.array_equal(idx[m], oidx[m])),
                               "dist2_bit_equal": bool(np.array_equal(d2[m].view(np.uint32), od2[m].view(np.uint32)))}
    return out


if __name__ == "__main__":
    sys.exit(main())
