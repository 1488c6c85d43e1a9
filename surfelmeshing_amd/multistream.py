"""Multi-GPU use of the surfel-integration path: independent RGB-D streams, one per GPU, no data-path collective
(one stream <-> one GPU <-> one surfel map).  torch.distributed is plumbing only: rendezvous, a
barrier around the timed region and a MAX-reduction of the elapsed time."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def stream_assignment(rank, base_seed=0x5EED0001):
    """Stream parameters of rank `rank` (config C4: seeds 0x5EED0001+g and different start poses)."""
    return {"seed": (base_seed + rank) & 0xFFFFFFFF, "phase": 0.37 * rank, "stream_id": rank}


def aggregate_throughput(local_units, local_seconds, world, dist=None, device="cpu"):
    """Whole-job throughput = units of all ranks / MAX over ranks of the elapsed time."""
    if world <= 1 or dist is None:
        return local_units / local_seconds, local_seconds, local_units
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / float(t.item()), float(t.item()), float(u.item())


def per_rank(value, world, dist=None, device="cpu"):
    """The value of every rank, in rank order (all_gather through the job's process group: RCCL, or gloo)."""
    if world <= 1 or dist is None:
        return [float(value)]
    import torch
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def ranks_seen(world, dist=None, device="cpu"):
    """How many ranks took part in a SUM all-reduce of 1 over the job's process group (= world when every rank of the
    launch reached the collective: bench.py prints it beside n_gpus)."""
    if world <= 1 or dist is None:
        return 1
    import torch
    t = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))
