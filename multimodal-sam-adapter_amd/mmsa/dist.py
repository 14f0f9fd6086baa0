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


class LogitsGather:
    """The step's all-gather, pipelined behind the NEXT step (round 6).

    `allgather_logits` issued between two graph replays orders replay k + 1 behind gather k: the collective's stream waits for the launch stream and
    the launch stream for the collective.  Nothing in step k + 1 needs gather k -- only the logits BUFFER must not be overwritten before its
    contents have left.  submit(local) therefore, on a side stream of its own (`self.stream`):
        wait for the launch stream's work so far (the replay that wrote `local`)  ->  copy `local` into one of TWO staging buffers  ->  record
        `copied`  ->  the launch stream waits for `copied` ONLY (a 13 MB device-to-device copy: the next replay may overwrite `local` then)  ->
        all_gather_into_tensor(out[k % 2], stage[k % 2]) with async_op=True, issued under the side stream  ->  the side stream waits for the
        collective (so the staging / output buffers of step k are free again when step k + 2 reuses them: everything is ordered on the side stream).
    It returns a handle; `handle.result()` blocks the HOST until that step's gathered tensor is complete and returns it (rank-major, padding of
    ragged shards dropped).  With world size 1 the local tensor is handed back.  On CPU tensors (gloo, the tests) the copy is synchronous and the
    collective asynchronous: `result()` waits for the work handle.  `LogitsGather.issued` lists (step, async_op) of every collective (tests)."""

    class Handle:
        def __init__(self, owner, k, work, event, out, sizes, cap):
            self.owner, self.k, self.work, self.event, self.out, self.sizes, self.cap = owner, k, work, event, out, sizes, cap

        def done(self):
            """True once the gathered tensor is complete (never blocks)."""
            if self.event is not None:
                return self.event.query()
            return self.work is None or self.work.is_completed()

        def result(self):
            if self.event is not None:
                self.event.synchronize()
            elif self.work is not None:
                self.work.wait()
            out = self.out
            if self.sizes is not None and not all(hi - lo == self.cap for lo, hi in self.sizes):
                out = torch.cat([out[r * self.cap:r * self.cap + (hi - lo)] for r, (lo, hi) in enumerate(self.sizes)], 0)
            return out

    def __init__(self):
        self.k = 0
        self.stage = [None, None]
        self.out = [None, None]
        self.stream = None
        self.issued = []

    def submit(self, local, global_batch=None):
        ws = world()
        if ws == 1:
            return LogitsGather.Handle(self, self.k, None, None, local, None, 0)
        sizes, cap = None, local.shape[0]
        if global_batch is not None:
            sizes = [shard_range(global_batch, r, ws) for r in range(ws)]
            mine = sizes[rank()][1] - sizes[rank()][0]
            if local.shape[0] != mine:
                raise ValueError(f"rank {rank()} holds {local.shape[0]} images but its shard of a global batch of {global_batch} is {mine}")
            cap = max(hi - lo for lo, hi in sizes)
        slot = self.k & 1
        shp = (cap,) + tuple(local.shape[1:])
        if self.stage[slot] is None or tuple(self.stage[slot].shape) != shp or self.stage[slot].device != local.device:
            self.stage[slot] = torch.zeros(shp, dtype=local.dtype, device=local.device)    # (zeros: the padding rows of a ragged shard)
            self.out[slot] = torch.empty((ws * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        stage, out = self.stage[slot], self.out[slot]
        event = None
        if local.is_cuda:
            main = torch.cuda.current_stream(local.device)
            if self.stream is None or self.stream.device != local.device:
                self.stream = torch.cuda.Stream(device=local.device)
            side = self.stream
            side.wait_stream(main)                       # the replay that wrote `local`
            with torch.cuda.stream(side):
                stage[:local.shape[0]].copy_(local, non_blocking=True)
                copied = torch.cuda.Event()
                copied.record(side)
                work = dist.all_gather_into_tensor(out, stage, async_op=True)
                work.wait()                              # stream-level: the SIDE stream waits for the collective, the host does not
                event = torch.cuda.Event()
                event.record(side)
            main.wait_event(copied)                      # the next replay is ordered behind the COPY, not behind the collective
            work = None
        else:
            stage[:local.shape[0]].copy_(local)
            work = dist.all_gather_into_tensor(out, stage, async_op=True)
        self.issued.append((self.k, True))
        h = LogitsGather.Handle(self, self.k, work, event, out, sizes, cap)
        self.k += 1
        return h


def max_over_ranks(seconds, device):
    """Wall time of the slowest rank (the bench contract's max-over-ranks)."""
    if world() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
