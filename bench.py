#!/usr/bin/env python3
"""Throughput benchmark of the MI355X encoder forward (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one forward of the drop-in backbone `SAMAdapterbimodalMixModNewInTwinConvNEW` (ViT-L SAM encoder +
RGB+LiDAR adapter, BASELINE configs[1]: 1024x1024, batch 2 per GPU) followed by the drop-in `SegformerHead`
(-> logits [B,25,256,256]) on synthetic tensors that are already resident in HBM, weights of that architecture from
the build's seeded "live" generator (tests/weights.py: every parameter and buffer non-trivial -- the same weights the
golden vectors of tests/golden/model_vitl1024.npz were captured with).  N > 1: one process per GPU, the batch is
sharded (weak scaling: 2 images per rank, nothing shared on the data path) and every step ends with the pipeline's ONE
exchange: an RCCL all-gather of the per-rank logits.  The step is the same at every N (at N=1 the gather is
the identity), so per-N values are comparable.  `--no-head` times the encoder forward alone; the default run also
reports it as `encoder_only`.

`--gpus N` with no WORLD_SIZE in the environment: this process starts the N ranks itself (one child per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set, reference launch pattern segmentation/dist_test.sh:7-9) BEFORE it makes any GPU
call and relays rank 0's JSON line.  Under an existing WORLD_SIZE (torch.distributed.run) it is one of the ranks.

Rank 0 prints ONE JSON line; `value` is the whole-job images/s (max-over-ranks time, barrier + synchronize on both
sides of exactly K steps).

What is timed is what is verified (`verified` object, checked OUTSIDE the timed region on every rank):
  * the HIP-graph replay of the step returns, bit for bit, what the eager launch sequence returns;
  * with the golden input of tests/golden/model_vitl1024.npz copied into image 0 of the graph's input buffer, the
    REPLAYED graph's f1..f4 match the reference's golden probes within the 1e-3 gate.

Extra objects:
  roofline     -- the dominant kernel family (the split3 GEMM: gemm_v2_kernel / gemm_split3_kernel): algorithmic FLOPs
                  (2*M*N*K per launch, NOT counting the 3x split products) / HIP-event time of those launches,
                  measured live in a single-stream profiled pass of one step right after the timed region (events on
                  the launch stream), against the dense bf16 MFMA peak (2.5 PFLOP/s).  Round 6: `achieved` / `frac` are over the launches
                  that CAN be matrix-bound; launches below 100 FLOP per compulsory byte (the skinny stage-0 / neck GEMMs) are scored against
                  HBM in `roofline.hbm_gemms`; `roofline.all_gemm_launches` keeps the definition of rounds 1-5 (every launch against MFMA).
  cpu_baseline -- the CPU oracle (oracle/ref_encoder.py, a PyTorch-CPU restatement validated against the reference)
                  timed on this box's physical host cores on ONE 1024x1024 image (rank 0, N=1 only): one small warm-up
                  forward, then the median of 3 runs, plus the ViT-B 512x512 line of SURVEY 8(d) (median of 5).
  hbm_kernels  -- GB/s of the HBM-bound kernels of the step against 8 TB/s, from the committed counter passes
                  (profiles/rNN_hbm_kernels.json, tools/pmc_all.sh).
  replay_ms    -- the same step replayed 20 more times after the timed region, every replay between its own pair of events:
                  median / min / max (SURVEY 8d asks for the median of >= 20 iterations; `value` stays the contract's K-step total).
  attention_guard -- the per-block logit guard words read back after the timed region (backbone.check_attention_guard): the largest
                  |logit| any attention block scored and whether every block stayed inside its operand precision's range.
  worst_case_precision -- the same step with EVERY GEMM and attention operand in the hi/lo pair formats (h8_sites = (), attention_precision = 'b3'):
                  what a checkpoint whose logits leave the fp16 range in every block would run.
  eager_plugin_api -- the call a reference user makes (segmentors/encoder_decoder.py:63-69): `backbone(x)` + head eagerly, no graph, the
                  attention guard in its default "sync" mode; its logits equal the replayed graph's bit for bit.
  config4_frame -- BASELINE configs[3]: one 1080x1920 frame through mmsa.inference.SlideRunner (six 1024^2 windows, two concurrent chains) ->
                  class map, frames/s; the map equals slide_inference + argmax_map bit for bit.
  vith1024     -- the pinnable half of BASELINE configs[4] (SAM ViT-H behind the same adapter), one HIP graph, golden probes checked.
Counter-derived fields (roofline.traffic, roofline.mfma_counters, hbm_kernels) come from committed profiles and carry the digest of the
kernel sources they were measured on; a profile whose digest is not that of the sources in the tree is reported as stale, not attached.

MMSA_BENCH_STUB=1 (tests only): no GPU -- gloo on CPU tensors, the device step replaced by a stand-in of known duration; exercises
rank spawning, sharding, the barrier / max-over-ranks timing, the logits all-gather and the JSON assembly of the N > 1 path.
"""
import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "multimodal-sam-adapter_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

STUB = os.environ.get("MMSA_BENCH_STUB") == "1"   # tests/test_host_cpu.py: the N > 1 control path without a GPU
PEAK_BF16_DENSE_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
HBM_BOUND_FLOP_PER_BYTE = 100.0   # GEMM launches below this arithmetic intensity are scored against HBM (roofline.hbm_gemms); machine balance: 2.5 PFLOP/s / 8 TB/s = 312
FLOPS_PER_IMAGE = {"vitl1024": 4.5207e12, "vitb512": 0.5411e12}   # algorithmic GEMM/conv/bmm FLOPs per image (vith1024: no figure -> null)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU per step (BASELINE configs[1]: 2)")
    ap.add_argument("--config", default="vitl1024", choices=["vitl1024", "vith1024", "vitb512", "tiny256"])
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--set", action="append", default=[], metavar="ATTR=VALUE",
                    help="set an attribute of the backbone before its first forward (A/B runs: --set fold_adapter_ln=False --set 'h8_sites=(\"vit\",)'); the value is a Python literal")
    ap.add_argument("--no-extras", action="store_true", help="skip the eager_plugin_api / config4_frame / vith1024 lines of the default single-GPU run")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed graph-vs-eager / golden-probe checks")
    ap.add_argument("--default-init", action="store_true", help="A/B aid: default-init weights instead of the seeded live generator")
    ap.add_argument("--no-head", action="store_true", help="time the encoder forward only (no decode head, no all-gather)")
    ap.add_argument("--chains", type=int, default=int(os.environ.get("MMSA_CHAINS", "2")),
                    help="the step's batch as this many independent sub-batch chains on concurrent HIP streams (mmsa.Chains); 1 = one chain")
    ap.add_argument("--force-chains", action="store_true", help="skip the untimed probe that falls back to one chain when the chains are slower on this machine")
    return ap.parse_args()


def spawn_ranks(n):
    """Parent of a self-launched multi-GPU run: starts one child per GPU and relays rank 0's stdout.  Makes no GPU call
    (torch.cuda.device_count() does not initialise the device on this image).  All children are polled: the first one that
    exits non-zero takes the others down with it (a rank that dies at start-up would otherwise leave rank 0 in the rendezvous
    until its timeout), and the whole job is bounded by MMSA_BENCH_TIMEOUT seconds (default 1800)."""
    import tempfile
    import torch
    have = torch.cuda.device_count()
    if have < n and not STUB:
        print(f"[bench] --gpus {n} asked for but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    deadline = time.time() + float(os.environ.get("MMSA_BENCH_TIMEOUT", "1800"))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            c = p.poll()
            if c is not None:
                live.remove(p)
                rc = max(rc, abs(c))
        if (rc or time.time() > deadline) and live:
            print(f"[bench] stopping {len(live)} rank(s): " + ("a rank failed" if rc else "timeout"), file=sys.stderr)
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 3
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return rc


def physical_cores():
    """(physical core count, CPU model name) from /proc/cpuinfo; falls back to os.cpu_count()."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    n = len(cores) or (os.cpu_count() or 1)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return max(n, 1), model


def latest_profile(pattern):
    """(newest profiles/rNN_<pattern> in round-number order or None, its json or None, note).  The json is returned only when the profile
    was measured on the kernel sources that are in the tree now (`source_digest` written by tools/pmc_*.py == build.source_digest())."""
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{pattern}")))
    if not fs:
        return None, None, "no profile committed"
    import build as _b   # multimodal-sam-adapter_amd/build.py
    j = json.load(open(fs[-1]))
    name = os.path.basename(fs[-1])
    have, want = j.get("source_digest"), _b.source_digest()
    if have != want:
        return fs[-1], None, f"profiles/{name} is STALE: measured on kernel sources {str(have)[:12]}, the tree holds {want[:12]} -- not attached (tools/pmc_all.sh)"
    return fs[-1], j, f"profiles/{name} (kernel sources {want[:12]}, commit {j.get('commit', 'n/a')})"


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if STUB and os.environ.get("MMSA_BENCH_STUB_FAIL_RANK") == str(rank):
        sys.exit(7)       # tests: a rank that dies at start-up (the parent must stop the others instead of waiting for their rendezvous)
    if STUB:
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *a_, **k_: None   # this process never touches a GPU
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("MMSA_FORCE_DIST") == "1"   # the latter: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if STUB:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm

    import mmsa
    from mmsa.dist import LogitsGather, allgather_logits
    from tests.configs import CONFIGS, HEAD_CONFIGS, make_input, probe_index
    from tests.weights import seeded_state_dict

    cfg = CONFIGS[a.config]
    torch.manual_seed(1234)
    # every rank generates the same 456 M seeded parameters on the host: N ranks x all host threads each oversubscribe the box N-fold
    host_threads = max(1, (os.cpu_count() or 1) // max(world, 1))
    if world > 1:
        torch.set_num_threads(host_threads)
    if STUB:
        a.no_graph = a.no_verify = a.no_roofline = a.no_cpu_baseline = a.no_head = True
        a.chains = 1
    model = None if STUB else mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    if model is not None and not a.default_init:
        model.load_state_dict(seeded_state_dict(model, seed=cfg["seed"]))
    if model is not None:
        import ast
        for kv in a.set:          # explicit switches of an A/B run (tools/ab_env.py): attributes, not environment variables
            k_, _, v_ = kv.partition("=")
            setattr(model, k_, ast.literal_eval(v_))
    x = make_input(cfg if not STUB else CONFIGS["tiny256"], batch=a.batch, seed=1234 + rank).to(dev)

    head = None
    # stub, optional uneven sharding: MMSA_BENCH_STUB_GLOBAL_BATCH = G splits G images over the ranks like mmsa.dist.shard_range (the first
    # G % world ranks hold one more) and gathers through padded shards -- the N = 8 form of the collective with a ragged last step
    stub_global = int(os.environ["MMSA_BENCH_STUB_GLOBAL_BATCH"]) if STUB and os.environ.get("MMSA_BENCH_STUB_GLOBAL_BATCH") else None
    if stub_global is not None:
        from mmsa.dist import shard_range
        lo_, hi_ = shard_range(stub_global, rank, world)
        stub_local = hi_ - lo_
    else:
        stub_local = a.batch
    stub_logits = torch.full((stub_local, 25, 8, 8), float(rank)) if STUB else None
    if not a.no_head:
        hcfg = HEAD_CONFIGS["head_vitl"]
        hkw = dict(hcfg["kwargs"])
        hkw["in_channels"] = [cfg["kwargs"]["embed_dim"]] * 4
        head = mmsa.build_head(dict(type="SegformerHead", **hkw))
        if not a.default_init:
            head.load_state_dict(seeded_state_dict(head, seed=hcfg["seed"]))
        model.emit_planes = True   # (`--set emit_planes=False` for the A/B) the tail also writes its four maps as planes: the head skips its NCHW -> planes pass

    feats = [None]

    def encoder_step():
        feats[0] = model(x)[0]
        return feats[0]

    def local_step():                      # everything that is captured in the HIP graph
        if STUB:                           # stand-in of known duration: rank r's step takes 10 (r + 1) ms
            time.sleep(0.01 * (rank + 1))
            return stub_logits
        fs = encoder_step()
        return head(fs) if head is not None else fs[0]

    def capture(fn):
        """warm-up (packs weights, sizes the workspace), then HIP-graph capture of the whole local step."""
        for _ in range(max(a.warmup, 1)):
            out = fn()
        torch.cuda.synchronize()
        if a.no_graph:
            return fn, out, False
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            g.replay()
            torch.cuda.synchronize()
            return g.replay, out, True
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"[bench] HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            return fn, fn(), False

    def timed(run, drain=lambda: None):
        for _ in range(a.warmup):
            run()
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run()
        drain()                              # the last step's gather completes inside the timed region
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    nch = a.chains if (a.chains > 1 and a.batch % a.chains == 0 and not a.no_graph) else 1
    chains = None
    if nch > 1:
        # the batch as `nch` concurrent chains (mmsa/chains.py): same kernels, same results, the GEMM phases of the chains overlap
        for _ in range(max(a.warmup, 1)):
            local_out = local_step()       # also the eager reference of the verification below
        torch.cuda.synchronize()
        try:
            chains = mmsa.Chains(model, head, n=nch).capture(x)
            replay, graphed = chains.replay, True
            replay()
            torch.cuda.synchronize()
            if head is not None:
                local_out = chains.logits
            graph_feats = None             # per-chain maps: chains.feats
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"[bench] concurrent chains unavailable ({type(e).__name__}: {e}); one chain", file=sys.stderr)
            torch.cuda.synchronize()
            chains, nch = None, 1
    chain_probe = None
    if chains is not None and not a.force_chains:
        # The overlap of the chains depends on how the runtime maps their streams onto hardware queues (LAB_NOTES.md 4.1: another queue
        # count costs 25-30 %).  Untimed probe: a few replays of both forms; the timed region runs the faster one.
        one_replay, one_out, one_graphed = capture(local_step)
        one_feats = feats[0]

        def probe(fn, n=5):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        t_ch, t_one = probe(chains.replay), probe(one_replay)
        if use_dist:     # every rank must take the same decision
            tt = torch.tensor([t_ch, t_one], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_ch, t_one = float(tt[0]), float(tt[1])
        chain_probe = {"chains_ms": round(t_ch, 3), "one_chain_ms": round(t_one, 3)}
        if t_one < t_ch:
            chains, nch = None, 1
            replay, local_out, graphed, graph_feats = one_replay, one_out, one_graphed, one_feats
    elif STUB and use_dist:
        # the chain probe's collective decision with stand-in timings: every rank must end up with the same (max-over-ranks) pair
        tt = torch.tensor([5.0 + rank, 50.0 - rank], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        chain_probe = {"chains_ms": float(tt[0]), "one_chain_ms": float(tt[1])}
        replay, local_out, graphed = capture(local_step)
        graph_feats = None
    elif chains is None:
        replay, local_out, graphed = capture(local_step)
        graph_feats = feats[0]                 # the four output maps the captured graph writes (fixed addresses)
    # ---- attention logit guard: the step has replayed on this input -- every block must have stayed inside its operand precision's range
    # (a block that did not is moved to bf16 hi/lo operands and the graphs are captured again: mmsa.Chains.check_guard / a second capture)
    guard_rerouted = []
    if model is not None:
        torch.cuda.synchronize()
        for _ in range(cfg["kwargs"]["depth"] + 1):
            moved = model.check_attention_guard()
            if not moved:
                break
            guard_rerouted += moved
            if chains is not None:
                chains.capture(x)
                replay = chains.replay
                if head is not None:
                    local_out = chains.logits
            else:
                replay, local_out, graphed = capture(local_step)
                graph_feats = feats[0]
            replay()
            torch.cuda.synchronize()
    gathered = [None]
    # the pipeline's only exchange step, outside the graph (RCCL owns its stream) and PIPELINED behind the next step (mmsa.dist.LogitsGather, round 6): step
    # k + 1's replay is ordered behind the 13 MB copy of step k's logits into a staging buffer, not behind the collective; a step's gathered tensor is
    # collected (host-side) one step later, the last one by `drain` inside the timed region
    gather = LogitsGather() if (head is not None or STUB) and use_dist else None
    pending = [None]
    overlap = {"next_step_enqueued_before_previous_gather_was_collected": 0, "collected": 0}

    def collect():
        if pending[0] is not None:
            gathered[0] = pending[0].result()
            pending[0] = None
            overlap["collected"] += 1

    def run():
        out_ = replay()                     # chains: both sub-batch chains of the step, joined into this stream (a step ends before the next starts)
        if gather is not None:
            if pending[0] is not None:      # step k is enqueued (or, in the stub, has run): only now is step k - 1's gather waited for
                overlap["next_step_enqueued_before_previous_gather_was_collected"] += 1
            collect()
            if "fallback" in overlap:       # the pipelined form failed on this job's first step (see below): the synchronous collective of rounds 1-5
                gathered[0] = allgather_logits(out_ if STUB else local_out, stub_global)
                return
            try:
                pending[0] = gather.submit(out_ if STUB else local_out, stub_global)
            except Exception as e:  # noqa: BLE001 -- the pipelined gather has only ever run on gloo and on one RCCL rank: a first multi-GPU job must not die of it
                if overlap["collected"] or overlap["next_step_enqueued_before_previous_gather_was_collected"]:
                    raise
                overlap["fallback"] = f"{type(e).__name__}: {e}"[:300]
                if rank == 0:
                    print(f"[bench] pipelined all-gather unavailable ({overlap['fallback']}); synchronous all_gather_into_tensor per step", file=sys.stderr)
                gathered[0] = allgather_logits(out_ if STUB else local_out, stub_global)

    dt = timed(run, collect)
    imgs = (stub_global if stub_global is not None else a.batch * world) * a.steps
    value = imgs / dt
    replay_ms = None
    if not STUB:   # median of >= 20 iterations, each between its own events (untimed by the contract's clock)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(20, a.steps))]
        for e0, e1 in evs:
            e0.record()
            run()
            e1.record()
        collect()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        replay_ms = {"n": len(ts), "median": round(statistics.median(ts), 3), "min": round(ts[0], 3), "max": round(ts[-1], 3),
                     "images_per_s_at_median": round(a.batch * world / (statistics.median(ts) * 1e-3), 3),
                     "note": "this rank's step (+ the all-gather at N > 1) replayed after the timed region, one event pair per replay"}
    guard = None
    if model is not None:
        still = model.check_attention_guard(reroute=False)
        ml = [lg for _, lg in model.attention_modes()]
        guard = {"threshold_fp16": model.ATTN_F16_MAX_LOGIT, "max_abs_logit": round(max(ml), 3) if ml else None,
                 "blocks_rerouted_before_timing": sorted(set(guard_rerouted)), "blocks_out_of_range_in_timed_region": still,
                 "note": "max |scale*q.k + rel-pos| per ViT block, folded into one device word per block by the attention kernels on every replay"}
        if still:
            raise SystemExit(f"[bench] attention blocks {still} ran fp16 attention beyond its logit range inside the timed region")
    # operand precision each ViT block's attention ran at in the timed step (backbone.check_attention_guard) -- taken HERE: the extra legs below re-pack the
    # model (worst-case formats) and drop it (ViT-H), which left this field null in the default run of rounds 4-5
    attn_blocks = None
    if model is not None and getattr(model, "_packed", None):
        modes = [m_ for m_, _ in model.attention_modes()]
        attn_blocks = {"f16": modes.count("f16"), "b3": modes.count("b3"),
                       "interactions_on_pairs": list(model._packed.get("inter_pairs", [])), "wide_range_state": bool(model._packed.get("wide", False))}
    if (head is not None or STUB) and use_dist and rank == 0:
        if stub_global is not None:   # ragged shards: rank r's images sit at shard_range(G, r) and carry its rank
            assert gathered[0].shape[0] == stub_global
            for r in range(world):
                lo_, hi_ = shard_range(stub_global, r, world)
                assert all(float(gathered[0][i, 0, 0, 0]) == float(r) for i in range(lo_, hi_)), f"rank {r}'s shard is not where it belongs"
        else:
            assert gathered[0].shape[0] == world * a.batch
            if STUB:    # every rank's shard arrived, in rank order
                assert [float(gathered[0][r * a.batch, 0, 0, 0]) for r in range(world)] == [float(r) for r in range(world)]

    # ---- what was timed is what is verified (untimed; every rank)
    verified = None
    if not a.no_verify:
        verified = {}
        replay()
        torch.cuda.synchronize()
        got_out = local_out.clone() if head is not None else None
        if graphed:
            e_out = local_step()           # eager, ONE chain over the whole batch, full grids
            torch.cuda.synchronize()
            same = True
            if head is not None:
                same = bool(torch.equal(e_out, got_out))
            for k in range(4):             # map by map: f1 alone is 268 MB per image
                gk = graph_feats[k] if chains is None else torch.cat([chains.feats[c][k] for c in range(nch)], 0)
                same = same and bool(torch.equal(feats[0][k], gk))
                del gk
            verified["graph_replay_equals_eager_bitwise"] = same
            if nch > 1:
                verified["chains"] = f"{nch} concurrent chains of {a.batch // nch} image(s) == one eager chain of {a.batch}, bit for bit"
            if not same:
                raise SystemExit("[bench] the HIP-graph replay does not reproduce the eager step bit for bit")
        gfile = os.path.join(ROOT, "tests", "golden", f"model_{a.config}.npz")
        if os.path.exists(gfile) and not a.default_init:
            import numpy as np
            g = np.load(gfile)
            if any(k.endswith("_probe") for k in g.files):
                # image 0 of the step against tests/golden/model_<config>.npz and, where the reference was also run on a second input
                # (model_<config>_b.npz: VERDICT r03 "weak" 4), image 1 against that -- both images of the benchmarked batch pinned on the reference
                gold = [(0, a.config, g)]
                gfile_b = os.path.join(ROOT, "tests", "golden", f"model_{a.config}_b.npz")
                if a.batch >= 2 and os.path.exists(gfile_b) and (a.config + "_b") in CONFIGS:
                    gold.append((1, a.config + "_b", np.load(gfile_b)))
                keep = x[:len(gold)].clone()
                for img, cname, _ in gold:
                    x[img].copy_(make_input(CONFIGS[cname], batch=1)[0].to(dev))   # the golden inputs into the graph's input buffer
                replay()
                torch.cuda.synchronize()
                # the maps THIS replay wrote: a captured graph writes the tensors of its capture (graph_feats); an eager step (--no-graph,
                # or capture unavailable) allocates fresh outputs on every call, which encoder_step leaves in feats[0]
                cur = graph_feats if graphed else feats[0]
                worst = 0.0
                worst_l2 = worst_mx = 0.0
                bc = a.batch // nch
                for img, cname, gg in gold:
                    for i in range(4):
                        f0 = (cur[i][img] if chains is None else chains.feats[img // bc][i][img % bc])
                        pi = probe_index(f0.numel(), 2048, seed=100 + i).to(dev)
                        got = f0.flatten()[pi].double().cpu()
                        ref = torch.from_numpy(gg[f"f{i+1}_probe"]).double()
                        r = float((got - ref).norm() / ref.norm())
                        mx = float((got - ref).abs().max() / ref.abs().max())
                        worst = max(worst, r, mx)
                        worst_l2, worst_mx = max(worst_l2, r), max(worst_mx, mx)
                verified["replayed_graph_vs_reference_golden_probes_max_rel"] = round(worst, 7)
                # its two parts: the relative L2 error over a map's 2048 probes (a mean-type figure, stable) and the largest single deviation relative to the
                # largest reference value (a maximum over 2048 samples of the operand formats' random rounding: moves by +-20 % with any change upstream)
                verified["golden_probes_rel_l2_worst_map"] = round(worst_l2, 7)
                verified["golden_probes_max_abs_over_max_ref_worst_map"] = round(worst_mx, 7)
                verified["golden"] = ("; ".join(f"image {img}: tests/golden/model_{cname}.npz" for img, cname, _ in gold)
                                      + " (f1..f4, 2048 probes each, outputs of the imported reference on these inputs)")
                if not worst <= 1e-3:
                    raise SystemExit(f"[bench] replayed graph misses the golden probes: {worst:.3e} > 1e-3")
                x[:len(gold)].copy_(keep)
                replay()
                torch.cuda.synchronize()

    encoder_only = None
    if head is not None and world == 1:
        if chains is not None:
            ereplay = mmsa.Chains(model, None, n=nch).capture(x).replay
        else:
            ereplay, _, _ = capture(encoder_step)
        edt = timed(ereplay)
        encoder_only = {"value": round(a.batch * a.steps / edt, 3), "unit": "images/s", "ms_per_step": round(edt / a.steps * 1e3, 3)}

    roofline = None
    if not a.no_roofline and rank == 0:
        prof = []
        model.multistream = False   # one stream: an event pair then brackets exactly one kernel, nothing runs beside it
        local_step()
        torch.cuda.synchronize()
        mmsa.ops.GEMM_PROFILE = prof
        local_step()
        torch.cuda.synchronize()
        mmsa.ops.GEMM_PROFILE = None
        model.multistream = True
        flops_all, ms_all = mmsa.ops.collect_gemm_profile(prof)
        # Two families (round 6): a launch whose algorithmic FLOPs per compulsory byte (operands read once, outputs written once) are below
        # HBM_BOUND_FLOP_PER_BYTE can never be matrix-bound -- at 8 TB/s and 2.5 PFLOP/s the machine balance is 312 FLOP/B, the skinny stage-0 / neck GEMMs
        # (M = 131072, N <= 192) sit at 20-70 -- and is scored against HBM, not averaged into the MFMA fraction.
        la = mmsa.ops.collect_gemm_profile.launches
        hb = [(f_, b_, t_) for f_, b_, t_ in la if f_ / max(b_, 1.0) < HBM_BOUND_FLOP_PER_BYTE]
        mf = [(f_, b_, t_) for f_, b_, t_ in la if f_ / max(b_, 1.0) >= HBM_BOUND_FLOP_PER_BYTE]
        flops, ms = sum(f_ for f_, _, _ in mf), sum(t_ for _, _, t_ in mf)
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        ach_all = flops_all / (ms_all * 1e-3) / 1e12 if ms_all > 0 else 0.0
        hb_ms, hb_bytes = sum(t_ for _, _, t_ in hb), sum(b_ for _, b_, _ in hb)
        hbm_gemms = {"bound": "hbm", "launches_per_step": len(hb), "compulsory_gb_per_step": round(hb_bytes / 1e9, 3), "kernel_ms_per_step": round(hb_ms, 3),
                     "achieved": round(hb_bytes / (hb_ms * 1e-3) / 1e9, 1) if hb_ms > 0 else None, "peak": 8000.0, "unit": "GB/s",
                     "frac": round(hb_bytes / (hb_ms * 1e-3) / 1e9 / 8000.0, 4) if hb_ms > 0 else None,
                     "algorithmic_gflop_per_step": round(sum(f_ for f_, _, _ in hb) / 1e9, 1),
                     "rule": f"launches with 2*M*N*K / compulsory bytes < {HBM_BOUND_FLOP_PER_BYTE:.0f} FLOP/B"}
        traffic = None
        _, tj, tnote = latest_profile("gemm_traffic.json")   # written by tools/pmc_traffic.sh (rocprofv3 --pmc passes)
        if tj and a.config == "vitl1024":
            traffic = round(tj["traffic_bytes_per_launch"])
            tnote = ("HBM-side bytes per launch (average over the step's GEMM launches): rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + --pmc WRITE_SIZE, "
                     "separate passes of this workload; " + tnote)
        _, mj, mnote = latest_profile("mfma_util.json")   # written by tools/pmc_mfma.sh (rocprofv3 --pmc MfmaUtil / MOPS passes)
        mfma = {"source": mnote}
        if mj and a.config == "vitl1024":
            mfma = {"gemm_family_mfma_util_pct": round(mj["gemm_family"]["mfma_util_pct"], 1), "gemm_family_mfma_tflops_counted": round(mj["gemm_family"]["mfma_tflops"], 1),
                    "whole_step_mfma_util_pct": round(mj["whole_step"]["mfma_util_pct"], 1), "whole_step_mfma_tflops_counted": round(mj["whole_step"]["mfma_tflops"], 1),
                    "source": "rocprofv3 --pmc MfmaUtil and --pmc SQ_INSTS_VALU_MFMA_MOPS_{BF16,F16,F8,F32} passes of this workload; " + mnote}
        roofline = {"bound": "mfma", "kernel": "split-operand GEMM (gemm_h8c_kernel + gemm_v2_kernel bf16 hi/lo and h8 flavours + gemm_split3_kernel)", "achieved": round(ach, 2), "peak": PEAK_BF16_DENSE_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": traffic, "traffic_note": tnote,
                    "compulsory_bytes_per_launch": round(sum(b_ for _, b_, _ in mf) / max(len(mf), 1)),
                    "launches_per_step": len(mf), "algorithmic_gflop_per_step": round(flops / 1e9, 1),
                    "kernel_ms_per_step": round(ms, 3), "mfma_counters": mfma,
                    "hbm_gemms": hbm_gemms,
                    "all_gemm_launches": {"launches_per_step": len(la), "algorithmic_gflop_per_step": round(flops_all / 1e9, 1), "kernel_ms_per_step": round(ms_all, 3),
                                          "achieved": round(ach_all, 2), "frac": round(ach_all / PEAK_BF16_DENSE_TFLOPS, 4),
                                          "note": "the definition of rounds 1-5: every GEMM launch of the step against the MFMA peak"},
                    "h8_sites": list(model._h8_sites()),
                    "note": "algorithmic 2*M*N*K FLOPs against the dense bf16/fp16 MFMA peak; for fp32-level parity every operand is a hi + lo pair: bf16 hi/lo "
                            "sites issue 3 bf16 MFMAs per algorithmic product, h8 sites 1 fp16 MFMA + the two cross terms on one block-scaled fp8 MFMA at "
                            "twice the rate (2 units)"}

    # ---- the path a reference user calls: EncoderDecoder.extract_feat runs `backbone(img)` eagerly (segmentors/encoder_decoder.py:63-69)
    # and the decode head on its result -- no graph, no chains, the backbone's attention guard in its default "sync" mode (one 4 * depth-byte read-back and a
    # host sync per forward, and a re-run if a block had to be re-routed).  Same weights, same resident input, same step as `value`.
    extras = not a.no_extras and not STUB and world == 1 and rank == 0 and a.config == "vitl1024" and not a.no_graph
    eager_api = None
    if extras:
        assert model.attention_guard == "sync"
        for _ in range(2):
            eo = local_step()
        torch.cuda.synchronize()
        n_e = max(5, a.steps // 2)
        t0 = time.perf_counter()
        for _ in range(n_e):
            eo = local_step()
        torch.cuda.synchronize()
        edt_ = (time.perf_counter() - t0) / n_e
        same_e = None
        if verified is not None and head is not None and graphed:
            replay()
            torch.cuda.synchronize()
            same_e = bool(torch.equal(eo, local_out))
            if not same_e:
                raise SystemExit("[bench] the eager plugin call does not return the replayed step's logits")
        eager_api = {"value": round(a.batch / edt_, 3), "unit": "images/s", "ms_per_step": round(edt_ * 1e3, 3), "steps": n_e,
                     "call": "backbone(x) -> SegformerHead(feats), eager launches on the current stream, attention_guard = 'sync' (read-back + host sync per forward)",
                     "vs_value": round(a.batch / edt_ / value, 4),
                     "verified": {"logits_equal_the_replayed_graphs_bitwise": same_e}}

    worst = None
    if model is not None and world == 1 and not a.no_roofline and not a.set:
        # every operand of every GEMM and attention kernel as a bf16 hi/lo pair: the operand formats a checkpoint with peaky attention in every
        # block falls back to.  Re-packs the weights (the graphs captured above are dead from here on: nothing replays them again).
        del replay, chains
        keep_sites, keep_prec = getattr(model, "h8_sites", None), model.attention_precision
        model.h8_sites, model.attention_precision = (), "b3"
        try:
            model(x[:1])     # packs
            torch.cuda.synchronize()
            if nch > 1:
                wch = mmsa.Chains(model, head, n=nch).capture(x)
                wrep = wch.replay
            else:
                wrep, _, _ = capture(local_step)
            wdt = timed(wrep)
            worst = {"value": round(a.batch * a.steps / wdt, 3), "unit": "images/s", "ms_per_step": round(wdt / a.steps * 1e3, 3),
                     "formats": "every GEMM and attention operand a 16-bit hi/lo PAIR, three MFMAs per product (h8_sites = (), attention_precision = 'b3': fp16 pairs in the ViT blocks, the attention kernels and the TwinConvNeXt chain, bf16 pairs in the interaction / neck / head GEMMs); same step, same graphs / chains form",
                     "vs_value": round(a.batch * a.steps / wdt / value, 4)}
            del wrep
        finally:
            if keep_sites is None:
                del model.h8_sites
            else:
                model.h8_sites = keep_sites
            model.attention_precision = keep_prec

    # ---- BASELINE configs[3]: one 1080 x 1920 RGB+Event frame, sliding-window inference (crop 1024, stride 640: six windows, ED:191-234) through
    # mmsa.inference.SlideRunner -- the windows cut by one kernel, encoder + head as two concurrent chains of three, overlap-average + resize + argmax
    # in one kernel -> class map.  Verified against the plain functions (slide_inference + argmax_map: pinned on the reference's own
    # slide_inference by tests/golden/slide.npz) bit for bit, every pixel covered.
    frame_line = None
    if extras and head is not None:
        import mmsa.inference as inf
        try:
            del replay
        except NameError:
            pass
        chains = None
        torch.cuda.empty_cache()
        gf = torch.Generator().manual_seed(4321)                     # synthetic frame of the MUSES geometry: RGB ~ N(0, 1), aux = sparse 5 % U(0, 1)
        frame = torch.randn(1, 6, 1080, 1920, generator=gf)
        frame[:, 3:] = (torch.rand(1, 3, 1080, 1920, generator=gf) < 0.05).float() * torch.rand(1, 3, 1080, 1920, generator=gf)
        frame = frame.to(dev)
        sr = inf.SlideRunner(model, head, frame, (1024, 1024), (640, 640), chains=2)
        cm, unc = sr.run().outputs()
        torch.cuda.synchronize()
        want = inf.argmax_map(inf.slide_inference(model, head, frame, (1024, 1024), (640, 640), max_batch=3))
        torch.cuda.synchronize()
        ok_f = bool(torch.equal(cm, want)) and int(unc.item()) == 0 and tuple(cm.shape) == (1, 1080, 1920)
        if not ok_f:
            raise SystemExit("[bench] SlideRunner class map differs from slide_inference + argmax_map")
        n_f = 5
        sr.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_f):
            fr = sr.run()
        fr.outputs()                       # waits for the last frame's guard copy (raises if a frame ran out of the fp16 attention range)
        torch.cuda.synchronize()
        fdt = (time.perf_counter() - t0) / n_f
        frame_line = {"value": round(1.0 / fdt, 3), "unit": "frames/s", "ms_per_frame": round(fdt * 1e3, 2), "crops_per_s": round(6.0 / fdt, 2), "frames": n_f,
                      "workload": "BASELINE configs[3]: 1080x1920 frame -> 6 windows 1024^2 (stride 640), ViT-L encoder + SegformerHead + overlap average + x4 resize + argmax -> uint8 class map",
                      "verified": {"class_map_equals_slide_inference_plus_argmax_bitwise": ok_f, "uncovered_pixels": int(unc.item()), "attention_guard": "every frame inspected (FrameResult.outputs)"}}
        del sr, cm, want, frame
        torch.cuda.empty_cache()

    # ---- the pinnable half of BASELINE configs[4]: SAM ViT-H (1280 wide, 32 blocks, head_dim 80 zero-padded to 96) behind the same RGB+LiDAR adapter, batch 2,
    # one HIP graph, golden probes of the imported reference on image 0 (tests/golden/model_vith1024.npz).  The 3-modality / fp8 half has no reference.
    vith_line = None
    if extras and head is not None:
        import numpy as np
        try:
            hc = CONFIGS["vith1024"]
            model = None
            torch.cuda.empty_cache()
            mh = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **hc["kwargs"]))
            mh.load_state_dict(seeded_state_dict(mh, seed=hc["seed"]))
            hkw_h = dict(HEAD_CONFIGS["head_vitl"]["kwargs"])
            hkw_h["in_channels"] = [hc["kwargs"]["embed_dim"]] * 4
            hh = mmsa.build_head(dict(type="SegformerHead", **hkw_h))
            hh.load_state_dict(seeded_state_dict(hh, seed=HEAD_CONFIGS["head_vitl"]["seed"]))
            mh.emit_planes = True
            xh = make_input(hc, batch=a.batch, seed=1234).to(dev)
            xh[0].copy_(make_input(hc, batch=1)[0].to(dev))          # image 0 = the golden input
            hfe = [None]

            def hstep():
                hfe[0] = mh(xh)[0]
                return hh(hfe[0])
            for _ in range(2):
                hstep()
            torch.cuda.synchronize()
            if mh.check_attention_guard():
                hstep()
                torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                hstep()
            gph.replay()
            torch.cuda.synchronize()
            n_h = 3
            t0 = time.perf_counter()
            for _ in range(n_h):
                gph.replay()
            torch.cuda.synchronize()
            hdt = (time.perf_counter() - t0) / n_h
            gh = np.load(os.path.join(ROOT, "tests", "golden", "model_vith1024.npz"))
            worst_h = 0.0
            for i in range(4):
                f0 = hfe[0][i][0]
                pi = probe_index(f0.numel(), 2048, seed=100 + i).to(dev)
                got = f0.flatten()[pi].double().cpu()
                ref = torch.from_numpy(gh[f"f{i+1}_probe"]).double()
                worst_h = max(worst_h, float((got - ref).norm() / ref.norm()), float((got - ref).abs().max() / ref.abs().max()))
            still_h = mh.check_attention_guard(reroute=False)
            if not worst_h <= 1e-3 or still_h:
                raise SystemExit(f"[bench] ViT-H line: golden probes {worst_h:.3e} (gate 1e-3), blocks out of the fp16 attention range {still_h}")
            vith_line = {"value": round(a.batch / hdt, 3), "unit": "images/s", "ms_per_step": round(hdt * 1e3, 3), "replays": n_h,
                         "workload": f"SAM ViT-H encoder (embed 1280, depth 32, head_dim 80 -> 96 zero-padded) + RGB+LiDAR adapter + SegformerHead, 1024x1024, batch {a.batch}, one HIP graph",
                         "verified": {"replayed_graph_vs_reference_golden_probes_max_rel": round(worst_h, 7), "golden": "image 0: tests/golden/model_vith1024.npz",
                                      "blocks_out_of_fp16_attention_range": still_h}}
            del gph, mh, hh, xh
            torch.cuda.empty_cache()
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001
            vith_line = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if not a.no_cpu_baseline and rank == 0 and world == 1 and a.config == "vitl1024":
        from oracle import ref_encoder as R  # CPU baseline leg only
        ncores, cpu_model = physical_cores()
        torch.set_num_threads(ncores)
        tiny = CONFIGS["tiny256"]
        R.OracleEncoder(**tiny["kwargs"])(make_input(tiny, batch=1))      # warm-up: thread pool, oneDNN primitives

        def cpu_line(name, runs, budget):
            c = CONFIGS[name]
            orc = R.OracleEncoder(**c["kwargs"])
            xc = make_input(c, batch=1, seed=1234)
            times, t_all = [], time.perf_counter()
            while len(times) < runs and (not times or time.perf_counter() - t_all + times[-1] < budget):
                tc = time.perf_counter()
                orc(xc)
                times.append(time.perf_counter() - tc)
            return statistics.median(times), times

        tcpu, times = cpu_line("vitl1024", 3, 240.0)
        tb, times_b = cpu_line("vitb512", 5, 40.0)
        cpu = {"value": round(1.0 / tcpu, 4), "unit": "images/s", "cores": ncores, "kind": "port", "cpu_model": cpu_model,
               "sample": f"1 image 1024x1024 ViT-L RGB+LiDAR, fp32 PyTorch-CPU oracle, {ncores} threads (physical cores), tiny warm-up forward then "
                         f"median of {len(times)} run(s): " + ", ".join(f"{t:.1f}" for t in times) + " s",
               "vitb512": {"value": round(1.0 / tb, 4), "unit": "images/s",
                           "sample": f"1 image 512x512 ViT-B (BASELINE configs[0]), same threads, median of {len(times_b)} run(s): "
                                     + ", ".join(f"{t:.2f}" for t in times_b) + " s"}}

    if rank == 0:
        fpi = FLOPS_PER_IMAGE.get(a.config) if not STUB else None
        headline = a.config == "vitl1024" and not STUB
        size = cfg["kwargs"]["img_size"] if not STUB else 0
        arch = {"vitl1024": "ViT-L", "vith1024": "ViT-H", "vitb512": "ViT-B", "tiny256": "tiny fixture model"}[a.config]
        hbm = None
        if headline and not a.no_roofline:
            _, hj, hnote = latest_profile("hbm_kernels.json")
            hbm = {"source": hnote}
            if hj:
                hbm = {"peak_GBps": hj.get("peak_GBps", 8000.0), "source": hnote,
                       "kernels": {k: {"GBps": v["GBps"], "frac": v["frac"]} for k, v in hj["kernels"].items()}}
        out = {
            "metric": f"images/sec encoder fwd @{size}x{size} RGB+LiDAR {arch}" if not STUB else "stub (no GPU): control path of the N > 1 bench",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "split-operand MFMA: fp16 hi + e5m2 cross terms (h8c / h8 planes, ViT / interaction / up-conv GEMMs) fp16 hi/lo pairs x3 (TwinConvNeXt: f3 planes) and bf16 hi/lo x3 (neck, head); attention blocks with the rel-pos terms fused: every contraction on one fp16 MFMA (q, k, v, P rounded to fp16) in the blocks whose measured max |logit| is below 8, fp16 hi/lo pairs (and fp16 hi/lo pair block GEMMs) in the others (config.attention_blocks); fp32 accumulate, fp32 activations", "data": "synthetic",
            "config": {"workload": f"{a.config}: SAM ViT-L encoder + RGB+LiDAR adapter forward, 1024x1024, batch {a.batch} per GPU"
                       if headline else f"{a.config} (NOT the BASELINE headline workload)",
                       "stage": "encoder forward only" if head is None else
                                "encoder forward + SegformerHead logits [B,25,H/4,W/4] + all-gather of logits across ranks",
                       "weights": "default init" if a.default_init else "seeded live generator (tests/weights.py), every parameter / buffer non-trivial",
                       "global_batch": stub_global if stub_global is not None else a.batch * world, "parallelism": f"dp{world}", "hip_graph": bool(graphed),
                       "chains_per_gpu": nch, "chains_probe_ms": chain_probe, "attention_blocks": attn_blocks,
                       "collective": ("one RCCL all_gather_into_tensor of the logits per step, pipelined behind the next step (mmsa.dist.LogitsGather)" if (head is not None and use_dist)
                                      else "none (single rank)" if head is not None else "none (encoder only)"),
                       "collective_overlap": overlap if gather is not None else None, "host_threads_per_rank": host_threads if world > 1 else None},
            "verified": verified,
            "encoder_only": encoder_only,
            "end_to_end_algorithmic_tflops": round(value * fpi / 1e12, 1) if fpi else None,
            "chains_probe_ms": chain_probe, "replay_ms": replay_ms, "attention_guard": guard, "worst_case_precision": worst,
            "eager_plugin_api": eager_api, "config4_frame": frame_line, "vith1024": vith_line,
            "roofline": roofline, "hbm_kernels": hbm, "cpu_baseline": cpu,
        }
        try:   # RCCL prints a banner through C stdio; flush it so that the JSON line is the LAST line of stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
