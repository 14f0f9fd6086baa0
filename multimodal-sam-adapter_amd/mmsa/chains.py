"""Concurrent chains: one batch as N independent sub-batches on N HIP streams.

Why.  Images are independent in the eval-mode forward, and every GEMM of the path is ONE persistent grid whose
workgroups all reach their epilogue at the same moment: the chip alternates between k loops (matrix pipe busy, memory idle) and
epilogues (a store burst at HBM speed, matrix pipe idle) -- LAB_NOTES.md section 4.1.  Two chains that each own half of the CUs
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
from .backbone import OperandRangeError


class AttentionRangeError(RuntimeError):
    """A replayed pass ran outside its operand formats' range: fp16 attention on logits beyond backbone.ATTN_F16_MAX_LOGIT (the blocks concerned have been moved
    to fp16 hi/lo pairs), or a value beyond the fp16-based planes' range (backbone.range_fallback: the model is in its wide-range state now, every block is
    named).  The outputs of the passes named in the message are invalid; the graphs have been captured again: replay those passes."""


class Replay:
    """What Chains.replay() returns: the outputs of ONE pass, readable only once the attention logit guard of that pass has been inspected.  A captured
    graph cannot branch, so a replay cannot re-route a block whose logits outgrew single fp16 operands the way an eager forward does; instead every
    pass is followed by an asynchronous copy of the guard words (4 bytes per ViT block) into pinned memory, and `outputs()` waits for THAT copy -- not
    for the device -- and raises AttentionRangeError instead of handing out tensors computed out of range."""

    def __init__(self, owner, seq):
        self._owner, self.seq = owner, seq

    def outputs(self):
        """logits [B, classes, H/4, W/4] (Chains with a head) or the per-chain feature lists; verified."""
        self._owner._verify(self.seq)
        return self._owner.logits if self._owner.head is not None else self._owner.feats

    @property
    def unverified(self):
        """The same tensors without the check -- for owners that inspect the guard themselves after a timing loop (bench.py) or enqueue further
        kernels on them before the pass is verified (SlideRunner); whatever is READ from them must be verified first."""
        return self._owner.logits if self._owner.head is not None else self._owner.feats


class Chains:
    def __init__(self, backbone, head=None, n=2, emit_planes=True, check_every=1):
        if n < 1:
            raise ValueError("mmsa.Chains: n must be >= 1")
        if check_every < 1:
            raise ValueError("mmsa.Chains: check_every must be >= 1")
        self.backbone, self.head, self.n = backbone, head, n
        self.emit_planes = emit_planes and head is not None
        self.graphs, self.streams, self.feats = [], [], []
        self.logits = None
        self.x = None
        # guard bookkeeping: pass numbers are 1, 2, ...; `_clean_upto` = last pass known to have run in range, `_bad` = [(first, last)] invalid passes
        self.check_every = check_every
        self._seq, self._clean_upto, self._bad, self._pending, self._pool = 0, 0, [], [], []

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
        call join() before reading the outputs.
        Returns a `Replay`: `.outputs()` hands the tensors out once that pass's guard words have been inspected (see Replay).  Passes whose
        guard copies have already arrived are inspected HERE first: if one of them ran out of range this call raises AttentionRangeError
        (blocks re-routed, graphs captured again) before enqueuing anything."""
        self._inspect(wait_for=0)
        main = torch.cuda.current_stream(self.x.device)
        start = torch.cuda.Event()
        start.record(main)
        self._done = []
        self._seq += 1
        for g, s in zip(self.graphs, self.streams):
            s.wait_event(start)
            with torch.cuda.stream(s):
                g.replay()
            done = torch.cuda.Event()
            done.record(s)
            self._done.append(done)
        if join:
            self.join()
        return Replay(self, self._seq)

    def join(self):
        """Make the current stream wait for the chains' last enqueued pass -- and, every `check_every`-th pass, enqueue the copy of the guard words
        behind it (one 4 * depth-byte device-to-host copy into pinned memory + an event; nothing waits for it here)."""
        main = torch.cuda.current_stream(self.x.device)
        for e in getattr(self, "_done", []):
            main.wait_event(e)
        words = self.backbone.attention_guard_words()
        if self._done and words is not None and self._seq % self.check_every == 0 and not (self._pending and self._pending[-1][0] == self._seq):
            host = self._pool.pop() if self._pool else torch.empty(words.numel(), dtype=words.dtype).pin_memory()
            host.copy_(words, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(main)
            self._pending.append((self._seq, ev, host))

    @torch.no_grad()
    def _inspect(self, wait_for):
        """Look at the guard copies that have arrived (and WAIT for the first one that covers pass `wait_for`, if > 0).  The words only ever
        grow (the kernels fold maxima into them), so a copy taken after pass j speaks for every pass up to j."""
        moved_at = None
        while self._pending:
            seq, ev, host = self._pending[0]
            if not ev.query():
                if not (wait_for > 0 and self._clean_upto < wait_for):
                    break
                ev.synchronize()
            self._pending.pop(0)
            try:
                moved = self.backbone.check_attention_guard(vals=host.tolist())
            except OperandRangeError:    # a value was clamped on its way into fp16-based planes and nothing is left to re-route (backbone.range_fallback): the passes are invalid
                self._bad.append((self._clean_upto + 1, self._seq))
                self._clean_upto = self._seq
                torch.cuda.synchronize(self.x.device)   # the copies into the pending host buffers have landed before the buffers go back to the pool
                self._pool.append(host)
                for _, _, h in self._pending:
                    self._pool.append(h)
                self._pending = []
                raise
            self._pool.append(host)
            if moved:
                self._bad.append((self._clean_upto + 1, self._seq))    # everything enqueued since the last clean inspection ran on the stale graphs
                self._clean_upto = self._seq
                for _, _, h in self._pending:
                    self._pool.append(h)
                self._pending = []
                moved_at = (seq, moved)
                break
            self._clean_upto = max(self._clean_upto, seq)
        if moved_at is not None:
            first, last = self._bad[-1]
            torch.cuda.synchronize(self.x.device)
            self.capture(self.x)
            raise AttentionRangeError(f"mmsa.Chains: ViT block(s) {moved_at[1]} ran outside their operand formats' range -- attention logits beyond the fp16 threshold, or (all blocks) "
                                      f"a value beyond the fp16-based planes' range (seen after pass {moved_at[0]}): "
                                      f"the outputs of passes {first}..{last} are invalid.  The blocks now run on hi/lo pairs and the graphs have been "
                                      "captured again -- replay those passes.")

    def _verify(self, seq):
        """Raise unless pass `seq` is known to have run inside the attention kernels' operand range."""
        for a, b in self._bad:
            if a <= seq <= b:
                raise AttentionRangeError(f"mmsa.Chains: pass {seq} ran fp16 attention out of range (passes {a}..{b} are invalid): replay it")
        if seq <= self._clean_upto:
            return
        if not any(s >= seq for s, _, _ in self._pending):   # no copy covers it (check_every > 1, join=False without join()): read the words now
            moved = self.check_guard()
            if moved:
                raise AttentionRangeError(f"mmsa.Chains: ViT block(s) {moved} ran fp16 attention out of range in pass {seq}; blocks re-routed, graphs captured again: replay it")
            return
        self._inspect(wait_for=seq)
        self._verify(seq)

    @torch.no_grad()
    def check_guard(self, recapture=True):
        """Read the backbone's attention logit guard NOW (host sync: waits for everything enqueued).  [] = every block ran inside its operand
        precision's range in all passes so far.  Otherwise the listed blocks have been moved to fp16 hi/lo pairs: the outputs of the
        passes since the last clean check were computed on fp16 attention beyond the threshold (their Replay.outputs() will raise), and the
        graphs are stale -- with `recapture` they are captured again here (same static input buffer), so the next replay() is valid."""
        torch.cuda.synchronize(self.x.device)
        for _, _, h in self._pending:
            self._pool.append(h)
        self._pending = []
        try:
            moved = self.backbone.check_attention_guard()
        except OperandRangeError:   # clamp watch, nothing left to re-route: the passes since the last clean check are invalid
            self._bad.append((self._clean_upto + 1, self._seq))
            self._clean_upto = self._seq
            raise
        if moved:
            self._bad.append((self._clean_upto + 1, self._seq))
        self._clean_upto = self._seq
        if moved and recapture:
            self.capture(self.x)
        return moved
