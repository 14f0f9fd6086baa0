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


def allgather_logits(local):
    """[B_loc, ...] on every rank -> [world * B_loc, ...] (rank-major) with ONE collective."""
    ws = world()
    if ws == 1:
        return local
    local = local.contiguous()
    out = torch.empty((ws * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out


def max_over_ranks(seconds, device):
    """Wall time of the slowest rank (the bench contract's max-over-ranks)."""
    if world() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
