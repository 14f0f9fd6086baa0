"""Data-parallel harness pieces (one process per GPU, torch.distributed; backend 'nccl' == RCCL on ROCm).

The encoder has no cross-image operation in eval mode (SyncBatchNorm uses running statistics), so ranks share
nothing on the data path: the global batch is sharded by rank and the only exchange step of the pipeline is ONE
all-gather of the per-rank logits per step (north_star; it replaces the reference's pickled
collect_results_{cpu,gpu} of segmentation/mmseg_custom/apis/test_bs.py:597-682).  On MI355X xGMI is a
fully-connected point-to-point fabric, so a single all_gather_into_tensor (each rank pushes its shard to its 7
peers in one hop) is used instead of a chain of sends."""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(global_batch, rk=None, ws=None):
    """Contiguous shard [lo, hi) of a global batch for rank rk (remainder spread over the first ranks)."""
    rk = rank() if rk is None else rk
    ws = world() if ws is None else ws
    if global_batch < 0 or ws <= 0 or not (0 <= rk < ws):
        raise ValueError("bad shard request")
    base, rem = divmod(global_batch, ws)
    lo = rk * base + min(rk, rem)
    return lo, lo + base + (1 if rk < rem else 0)


def allgather_logits(local, global_batch=None):
    """[B_loc, ...] on every rank -> [global batch, ...] (rank-major) with ONE collective.

    `global_batch` = None: every rank holds the same number of images (the bench's weak-scaling step).  With a global batch that
    `shard_range` split unevenly (not divisible by the world size) pass it: every rank pads its shard to ceil(global_batch / world)
    images, so the collective has identical sizes everywhere, and the padding is dropped after the gather."""
    ws = world()
    if ws == 1:
        return local
    local = local.contiguous()
    if global_batch is None:
        out = torch.empty((ws * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    sizes = [shard_range(global_batch, r, ws) for r in range(ws)]
    mine = sizes[rank()][1] - sizes[rank()][0]
    if local.shape[0] != mine:
        raise ValueError(f"rank {rank()} holds {local.shape[0]} images but its shard of a global batch of {global_batch} is {mine}")
    cap = max(hi - lo for lo, hi in sizes)
    if cap == 0:
        return local
    if mine < cap:
        pad = torch.zeros((cap - mine,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((ws * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    if all(hi - lo == cap for lo, hi in sizes):
        return out
    return torch.cat([out[r * cap:r * cap + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def max_over_ranks(seconds, device):
    """Wall time of the slowest rank (the bench contract's max-over-ranks)."""
    if world() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
