"""Concurrent chains: one batch as N independent sub-batches on N HIP streams.

Why.  Images are independent in the eval-mode forward, and every GEMM of the path is ONE persistent grid whose
workgroups all reach their epilogue at the same moment: the chip alternates between k loops (matrix pipe busy, memory idle) and
epilogues (a store burst at HBM speed, matrix pipe idle) -- DESIGN.md section 4.1.  Two chains that each own half of the CUs
(`ops.GEMM_MAX_GRID`) run those phases against each other.  Measured on one MI355X, ViT-L 1024^2, batch 2 (tools/chains_bench.py,
bench.py --chains 1 / 2 on the same box): one chain of 2 images 37.3-37.5 ms, two chains of 1 image 35.7-36.4 ms (encoder only), 38.8
against 37.3 ms with the head; the results are bit-identical (the kernels' arithmetic does not depend on the batch size or on the
grid).  Forcing a phase offset between the chains (0-18 ms) changes nothing measurable: they drift apart by themselves.

How.  Each chain is its own HIP graph, captured on its own origin stream (a capture that forks side streams may only join them
into its origin stream on ROCm 7.2, so the chains cannot be branches of one graph); the chains share the packed weights and use
private scratch buffers (`backbone.chain(i)`, `head.buf_tag`).  `replay()` launches the N graphs on their streams behind the caller's
stream and joins them into it."""
import torch

from . import ops


class Chains:
    def __init__(self, backbone, head=None, n=2, emit_planes=True):
        if n < 1:
            raise ValueError("mmsa.Chains: n must be >= 1")
        self.backbone, self.head, self.n = backbone, head, n
        self.emit_planes = emit_planes and head is not None
        self.graphs, self.streams, self.feats = [], [], []
        self.logits = None
        self.x = None

    def _step(self, i, xs):
        """One chain's work on its slice: backbone (+ head into its slice of the shared logits tensor)."""
        with self.backbone.chain(i):
            if self.head is not None:
                self.head.buf_tag = f"chain{i}_"
            try:
                feats, _ = self.backbone(xs)
                if self.head is not None:
                    bc = xs.shape[0]
                    if self.logits is None:      # first call: learn the logits' shape from a stand-alone head pass
                        lg = self.head(feats)
                        self.logits = torch.empty(self.x.shape[0], *lg.shape[1:], device=lg.device)
                    self.head(feats, out=self.logits[i * bc:(i + 1) * bc])   # the chain writes its slice of the step's logits itself
            finally:
                if self.head is not None:
                    self.head.buf_tag = ""
        return feats

    @torch.no_grad()
    def capture(self, x):
        """x: the static input buffer [B, 6, H, W] (B a multiple of n) the graphs will read on every replay."""
        if x.shape[0] % self.n:
            raise RuntimeError(f"mmsa.Chains: batch {x.shape[0]} is not a multiple of {self.n} chains")
        dev = x.device
        self.x = x
        bc = x.shape[0] // self.n
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        keep_cap, keep_emit = ops.GEMM_MAX_GRID, getattr(self.backbone, "emit_planes", False)
        self.backbone.emit_planes = self.emit_planes
        with torch.cuda.device(dev):
            self.backbone(x[:bc])                    # packs the weights and creates the workspace
            torch.cuda.synchronize(dev)
            ops.GEMM_MAX_GRID = cus // self.n if self.n > 1 else 0
            self.graphs, self.streams, self.feats = [], [], []
            try:
                # pass 1, eager: every chain runs its OWN sub-batch once -- allocates its scratch buffers (and the shared logits) and, with
                # the backbone's attention guard on, shows every image of the batch to the per-block precision decision BEFORE anything is
                # captured (a block that moves to bf16 hi/lo operands gets new weight planes: a graph captured earlier would point at the old ones)
                self._warm = []
                for i in range(self.n):
                    s = torch.cuda.Stream(device=dev)
                    self._warm.append(s)
                    s.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(s):
                        self._step(i, x[i * bc:(i + 1) * bc])
                    torch.cuda.synchronize(dev)
                # pass 2: one graph per chain.  Creation order per chain: capture stream, graph (its instantiation creates the streams of its
                # parallel branches), replay stream -- the runtime deals its hardware queues to streams in creation order, and with the
                # streams of both chains created first the chains' branches shared queues: 37.8 ms per step instead of 32.6
                # (profiles/r04_chains_streams.txt)
                for i in range(self.n):
                    s = torch.cuda.Stream(device=dev)
                    s.wait_stream(torch.cuda.current_stream(dev))
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        feats = self._step(i, x[i * bc:(i + 1) * bc])
                    torch.cuda.synchronize(dev)
                    self.graphs.append(g)
                    self.streams.append(torch.cuda.Stream(device=dev))
                    self.feats.append(feats)
            finally:
                ops.GEMM_MAX_GRID = keep_cap
                self.backbone.emit_planes = keep_emit
        return self

    def replay(self, join=True):
        """Enqueue one pass over the batch: every chain starts behind the current stream's work.  join=True: the current stream then
        waits for all chains (the outputs are ready for whatever it does next).  join=False: the chains free-run -- consecutive
        replays queue up per chain and nothing orders chain A's pass k against chain B's (measured within 1 % of the joined form);
        call join() before reading the outputs."""
        main = torch.cuda.current_stream(self.x.device)
        start = torch.cuda.Event()
        start.record(main)
        self._done = []
        for g, s in zip(self.graphs, self.streams):
            s.wait_event(start)
            with torch.cuda.stream(s):
                g.replay()
            done = torch.cuda.Event()
            done.record(s)
            self._done.append(done)
        if join:
            self.join()
        return self.logits if self.head is not None else self.feats

    @torch.no_grad()
    def check_guard(self, recapture=True):
        """Read the backbone's attention logit guard (host sync: call it after the replays whose outputs matter).  [] = every block ran
        inside its operand precision's range.  Otherwise the listed blocks have been moved to bf16 hi/lo operands: the outputs of the
        replays since the last check were computed on fp16 attention beyond the threshold, and the graphs are stale -- with
        `recapture` they are captured again here (same static input buffer), so the next replay() is valid."""
        torch.cuda.synchronize(self.x.device)
        moved = self.backbone.check_attention_guard()
        if moved and recapture:
            self.capture(self.x)
        return moved

    def join(self):
        """Make the current stream wait for the chains' last enqueued pass."""
        main = torch.cuda.current_stream(self.x.device)
        for e in getattr(self, "_done", []):
            main.wait_event(e)
