"""The N > 1 path on CPU: two processes over gloo, one independent stream each, no data-path collective; the only
exchanged values are the barrier and the MAX / SUM reductions of the timing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import small_pre  # noqa: F401  (path setup)
from surfelmeshing_amd import multistream


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = multistream.stream_assignment(rank)
        # every rank integrates its own stream with the oracle-free host-side bookkeeping only (no GPU here):
        # the "work" is a stand-in whose duration differs per rank
        frames, seconds = 100, 0.5 + 0.25 * rank
        dist.barrier()
        fps, tmax, total = multistream.aggregate_throughput(frames, seconds, world, dist)
        out.put((rank, a["seed"], a["phase"], fps, tmax, total))
    finally:
        dist.destroy_process_group()


def test_two_ranks_independent_streams():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seeds = {r[1] for r in res}
    assert len(seeds) == world                      # distinct streams
    for r in res:
        assert r[5] == 200.0 and abs(r[4] - 0.75) < 1e-9       # SUM of frames, MAX of time
        assert abs(r[3] - 200.0 / 0.75) < 1e-6                 # identical aggregate on every rank


def test_single_rank_passthrough():
    fps, t, n = multistream.aggregate_throughput(50, 0.25, 1)
    assert fps == 200.0 and t == 0.25 and n == 50
    assert multistream.stream_assignment(3)["seed"] == 0x5EED0004


def test_bench_rank_path_dry_run_two_ranks():
    """bench.py's own N > 1 path, as the driver launches it (python -m torch.distributed.run --nproc-per-node N bench.py
    --gpus N ...), without GPUs: --dry-run keeps rank_info -> process group (gloo instead of RCCL) -> stream assignment ->
    per-rank plan generation -> barrier -> MAX / SUM aggregation -> one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    from common import ROOT
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2
    assert d["units_all_ranks"] == 12.0 and d["value"] > 0 and d["scaling"] == "weak"


def test_bench_dry_run_single_rank():
    import json
    import subprocess
    import sys
    from common import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["dry_run"] and d["n_gpus"] == 1 and d["units_all_ranks"] == 4


def test_bench_gpus_flag_launches_the_ranks_itself():
    """plain `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset): the script starts its own two ranks; a launcher
    whose world size disagrees with --gpus is refused."""
    import json
    import subprocess
    import sys
    from common import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["per_rank_value"]) == 2 and d["units_all_ranks"] == 10.0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="1"), cwd=ROOT)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_bench_gpus_8_dry_run_plain_start():
    """The driver's 8-GPU command shape, started plainly (`python bench.py --gpus 8 ...`): eight ranks over gloo, eight distinct
    streams (seeds / start poses), one line from rank 0 with the whole-job aggregate -- the C4 path minus the GPUs (no 8-GPU
    node was available to any round: SCALE_r0x.json are 'skipped' records)."""
    import json
    import subprocess
    import sys
    from common import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and len(d["per_rank_value"]) == 8 and d["units_all_ranks"] == 160.0
    assert d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["value"] > 0
    from surfelmeshing_amd import multistream
    assert len({multistream.stream_assignment(r_)["seed"] for r_ in range(8)}) == 8


@pytest.mark.gpu
def test_bench_line_of_a_real_run_is_compact_and_complete(tmp_path):
    """The driver's command shape on a small map: ONE stdout line that starts with '{', under 6 000 characters (the driver keeps
    8 000 of stdout), with the contract's keys, the roofline and CPU-baseline blocks and a green in-run parity check; the full
    result lands in the detail file."""
    import json
    import subprocess
    import sys
    from common import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--surfels", "300000", "--cpu-frames", "2", "--host-frames", "20", "--growth-frames", "0", "--timing-frames", "0",
                        "--detail-out", detail], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{") and len(lines[0]) < 6000, (len(lines), len(lines[0]))
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0 and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["avg_launch_ms"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["parity_check"]["ok"] is True and d["host_frames"]["value"] > 0
    full = json.load(open(detail))
    assert full["roofline"]["kernels"] and full["host_frames"]["frames_staged_by_copy_kernels"] == 20


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_over_gloo():
    """C4 readiness without the 8-GPU node: bench.py's REAL timed path with two ranks under torch.distributed.run, process
    group gloo (RCCL refuses two ranks on one device; the streams are independent, the only collectives are the barrier and
    the MAX / SUM reductions of the timing).  Both ranks grow their own map on the one GPU, run the timed window
    concurrently, and check their own stream against the oracle; rank 0 prints one line with n_gpus = 2."""
    import json
    import subprocess
    import sys
    from common import ROOT
    # (started plainly: bench.py --gpus 2 launches its own two ranks under torch.distributed.run)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--backend", "gloo", "--surfels", "300000", "--steps", "20", "--warmup", "5",
                        "--cpu-frames", "2", "--host-frames", "0", "--check-all-ranks", "--quiet"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["backend"] == "gloo" and d["config"]["streams"] == 2
    assert d["gloo_ranks_seen"] == 2 and len(d["per_rank_frames_per_s"]) == 2 and min(d["per_rank_frames_per_s"]) > 0
    checks = d["parity_check_per_rank"]
    assert len(checks) == 2
    for c in checks:
        assert c["counts_equal"] and c["rows_not_bit_equal"] == [] and c["frames"] == 2, checks
