"""CPU tests of the host side: C-ABI surface, plugin API / state_dict contract, pack-time index and
interpolation helpers, loud failure without a GPU, and the world_size-2 data-parallel harness over gloo."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import mmsa
    hdr = open(os.path.join(ROOT, "include", "mmsa.h")).read()
    declared = set(re.findall(r"\b(mmsa_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(mmsa.lib.raw, name), f"{name} declared in include/mmsa.h but not exported"
    assert declared == set(mmsa.lib.SIGNATURES), "ctypes signature table and header disagree"
    assert mmsa.lib.version() == mmsa.lib.ABI_VERSION == int(re.search(r"#define MMSA_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "mmsa_version.h")).read()).group(1))


def _header_prototypes():
    """(name -> (return kind, [argument kinds])) parsed from include/mmsa.h.  Kinds: P pointer, I int / unsigned, L long / int64, F float, D double, Z size_t."""
    hdr = open(os.path.join(ROOT, "include", "mmsa.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)                      # comments (also the ones between arguments)
    hdr = "\n".join(l for l in hdr.split("\n") if not l.lstrip().startswith("#"))

    def kind(t):
        t = t.strip()
        if "*" in t or t.startswith("mmsa_stream_t"):
            return "P"
        t = re.sub(r"\b(const|unsigned)\b", " ", t).split()
        base = t[0] if t else "int"                                       # `unsigned pattern` -> base "pattern" is the NAME: unsigned int
        return {"int": "I", "long": "L", "int64_t": "L", "float": "F", "double": "D", "size_t": "Z", "char": "I"}.get(base, "I")

    protos = {}
    for ret, name, args in re.findall(r"\b((?:const\s+)?[a-z_0-9]+\s*\*?)\s*(mmsa_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = " ".join(args.split())
        protos[name] = (kind(ret), [] if args in ("", "void") else [kind(a) for a in args.split(",")])
    return protos


def test_header_prototypes_match_ctypes_table():
    """the header, the definitions and the ctypes table were three hand-kept copies compared by NAME only.  The definitions are
    now compiled against the header (csrc/common.h includes it; next test); this one compares every prototype's argument COUNT and KIND with
    mmsa/lib.py SIGNATURES, so that a trailing argument dropped from one of the two fails here and not as stack garbage in a launch."""
    import ctypes
    import mmsa
    protos = _header_prototypes()
    assert set(protos) == set(mmsa.lib.SIGNATURES) and len(protos) >= 45
    ckind = {ctypes.c_void_p: "P", ctypes.c_int: "I", ctypes.c_uint: "I", ctypes.c_long: "L", ctypes.c_float: "F", ctypes.c_double: "D",
             ctypes.c_size_t: "Z", ctypes.c_char_p: "P"}

    def ck(t):
        return ckind.get(t, "P")        # POINTER(...) types

    for name, (ret, args) in protos.items():
        got = [ck(t) for t in mmsa.lib.SIGNATURES[name]]
        want = ["L" if a == "Z" else a for a in args]
        got = ["L" if a == "Z" else a for a in got]                     # size_t and long: one 64-bit integer register either way
        assert got == want, f"{name}: include/mmsa.h declares {''.join(want)} ({len(want)} arguments), mmsa/lib.py binds {''.join(got)} ({len(got)})"
        rt = ck(mmsa.lib._RESTYPES.get(name, ctypes.c_int))
        assert rt == ret, f"{name}: return kind {ret} in the header, {rt} in mmsa/lib.py"
    # the check itself: dropping a trailing argument from either side is seen
    broken = dict(mmsa.lib.SIGNATURES, mmsa_gemm_split3=mmsa.lib.SIGNATURES["mmsa_gemm_split3"][:-1])
    assert [ck(t) for t in broken["mmsa_gemm_split3"]] != protos["mmsa_gemm_split3"][1]


def test_the_library_sources_are_compiled_against_the_header(tmp_path):
    """Every source that defines an entry point includes include/mmsa.h (through csrc/common.h, with the stream spelled hipStream_t), so a
    definition that disagrees with its prototype is a compile error of the library build; shown here on a stand-in definition with g++
    (the rule is the language's: two `extern "C"` declarations of one name with different parameter lists conflict)."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "multimodal-sam-adapter_amd", "csrc")
    common = open(os.path.join(csrc, "common.h")).read()
    assert '#include "../../include/mmsa.h"' in common and "#define MMSA_BUILDING_LIBRARY" in common
    for f in glob.glob(os.path.join(csrc, "*.hip")):
        src = open(f).read()
        if re.search(r'extern "C" (?:int|long|const char\*) mmsa_', src):
            assert re.search(r'#include "(common|gemm_v2_shared)\.h"', src), f"{f} defines entry points without the header"
    inc = os.path.join(ROOT, "include")
    good = tmp_path / "good.cpp"
    good.write_text('#include "mmsa.h"\nextern "C" int mmsa_zero_bytes(void* p, size_t bytes, mmsa_stream_t stream) { (void)p; (void)bytes; (void)stream; return 0; }\n')
    bad = tmp_path / "bad.cpp"
    bad.write_text('#include "mmsa.h"\nextern "C" int mmsa_zero_bytes(void* p, size_t bytes) { (void)p; (void)bytes; return 0; }\n')
    assert subprocess.run(["g++", "-fsyntax-only", "-I", inc, str(good)], capture_output=True).returncode == 0
    r = subprocess.run(["g++", "-fsyntax-only", "-I", inc, str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "conflict" in r.stderr
    # and the header is plain C (a cgo / JNI binding compiles it as such)
    assert subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-pedantic", "-Werror", os.path.join(inc, "mmsa.h")], capture_output=True).returncode == 0


def test_state_dict_contract_with_the_constructor_switches_off(golden_dir):
    """with_cffn / use_extra_extractor / add_vit_feature = False (BK:32-34): the parameter tree of the reference built that way."""
    import mmsa
    from tests.configs import CONFIGS
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **CONFIGS["tiny256_plain"]["kwargs"]))
    lines = [l.rstrip("\n").split(" ", 1) for l in open(os.path.join(golden_dir, "state_dict_keys_tiny_plain.txt"))]
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _ in lines]
    assert all(list(sd[k].shape) == eval(s) for k, s in lines)
    assert not any(".ffn" in k or "extra_extractors" in k for k in sd)
    # ... and of a ViT without rel-pos tables / qkv bias (IE:317,320-327); use_abs_pos=False is refused (the reference's forward fails on it, BK:276)
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **CONFIGS["tiny256_norel"]["kwargs"]))
    lines = [l.rstrip("\n").split(" ", 1) for l in open(os.path.join(golden_dir, "state_dict_keys_tiny_norel.txt"))]
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _ in lines]
    assert all(list(sd[k].shape) == eval(s) for k, s in lines)
    assert not any("rel_pos" in k or k.endswith("qkv.bias") for k in sd)
    with pytest.raises(NotImplementedError):
        mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **dict(CONFIGS["tiny256"]["kwargs"], use_abs_pos=False)))


def test_plugin_registry_and_state_dict_contract(golden_dir):
    import mmsa
    from tests.configs import CONFIGS
    for name in ("SAMAdapterbimodalMixModNewInTwinConvNEW", "SAMAdapterbimodalMixModNewInTwinConvNEWwithcp"):
        m = mmsa.build_backbone(dict(type=name, conv_drop_path_rate=0.3, **CONFIGS["tiny224"]["kwargs"]))
        lines = [l.rstrip("\n").split(" ", 1) for l in open(os.path.join(golden_dir, "state_dict_keys_tiny.txt"))]
        sd = m.state_dict()
        assert list(sd.keys()) == [k for k, _ in lines]
        assert all(list(sd[k].shape) == eval(s) for k, s in lines)
        assert not m.training


def test_vitl_state_dict_contract(golden_dir):
    from mmsa.params import param_spec
    from tests.configs import CONFIGS
    kw = dict(CONFIGS["vitl1024"]["kwargs"])
    cfg = dict(embed_dim=kw["embed_dim"], depth=kw["depth"], num_heads=kw["num_heads"], mlp_ratio=kw["mlp_ratio"],
               patch_size=16, pretrained_size=1024, img_size=1024, window_size=14, global_attn_indexes=kw["global_attn_indexes"],
               conv_inplane=48, n_points=4, deform_num_heads=16, init_values=1e-6, interaction_indexes=kw["interaction_indexes"],
               cffn_ratio=0.25, deform_ratio=0.5, arch="small", use_extra_extractor=True)
    spec = param_spec(cfg)
    lines = [l.rstrip("\n").split(" ", 1) for l in open(os.path.join(golden_dir, "state_dict_keys_vitl.txt"))]
    assert [n for n, _, _ in spec] == [k for k, _ in lines]
    assert all(list(s) == eval(t) for (_, s, _), (_, t) in zip(spec, lines))
    assert sum(int(np.prod(s)) for _, s, k in spec if k == "param") == 455_9 * 10 ** 5 + sum(
        int(np.prod(s)) for _, s, k in spec if k == "param") - 455_9 * 10 ** 5  # tautology guard; real check below
    n_params = sum(int(np.prod(s)) if len(s) else 1 for _, s, k in spec if k == "param")
    assert abs(n_params - 455.9e6) < 0.1e6


def test_no_cpu_fallback_and_unsupported_configs():
    import mmsa
    from tests.configs import CONFIGS
    kw = CONFIGS["tiny224"]["kwargs"]
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **kw))
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 6, 224, 224))
    with pytest.raises(RuntimeError, match="GPU"):
        mmsa.ops.layernorm(torch.zeros(4, 32), torch.ones(32), torch.zeros(32), 1e-6, torch.zeros(4, 32))
    with pytest.raises(NotImplementedError):
        mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **dict(kw, modalities_name=["rgb"], modalities_ch=[3])))
    with pytest.raises(NotImplementedError):
        m.train()
    with pytest.raises(KeyError):
        mmsa.build_backbone(dict(type="NoSuchBackbone"))


def test_rel_pos_tables_bit_exact(golden_dir):
    """get_rel_pos gather incl. the linear-interpolation branch (IE:554-584) -- index table bit-exact."""
    from mmsa.backbone import _rel_table, rel_pos_index
    g = np.load(os.path.join(golden_dir, "bookkeeping.npz"))
    for (q, L) in ((14, 27), (64, 127), (14, 31), (20, 31), (16, 31), (32, 127)):
        tab = torch.from_numpy(g[f"rp_in_{q}_{L}"])
        out = _rel_table(q, tab)
        ref = torch.from_numpy(g[f"rp_out_{q}_{L}"])
        if L == 2 * q - 1:
            assert torch.equal(out, ref)  # pure gather
        else:
            assert torch.allclose(out, ref, rtol=1e-6, atol=2e-6)  # fp32 lerp rounding
        idx = rel_pos_index(q, q)
        assert idx.dtype == torch.int64 and idx.min() == 0 and idx.max() == 2 * q - 2
        assert torch.equal(idx, (torch.arange(q)[:, None] - torch.arange(q)[None, :]) + q - 1)


def test_pos_embed_bicubic_matches_torch():
    from mmsa.backbone import _bicubic_resize
    src = torch.randn(1, 16, 16, 8)
    for (H, W) in ((14, 14), (20, 20), (16, 16), (32, 24)):
        ref = F.interpolate(src.permute(0, 3, 1, 2), size=(H, W), mode="bicubic", align_corners=False).permute(0, 2, 3, 1)[0]
        assert torch.allclose(_bicubic_resize(src[0], H, W), ref, rtol=1e-5, atol=1e-5)


def test_reference_points_match_oracle():
    from mmsa.backbone import _ref_points
    from oracle import ref_encoder as R
    for shapes in ([(14, 14)], [(28, 28), (14, 14), (7, 7)], [(128, 128), (64, 64), (32, 32)]):
        assert torch.equal(_ref_points(shapes), R.get_reference_points(shapes, torch.float32).reshape(-1, 2))


def test_shard_ranges_cover_batch():
    from mmsa.dist import shard_range
    for gb in (0, 1, 7, 16, 17):
        for ws in (1, 2, 3, 8):
            spans = [shard_range(gb, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmsa import dist as D
    lo, hi = D.shard_range(6, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 2, 3).contiguous() * 10 + rank
    allg = D.allgather_logits(local)
    slow = D.max_over_ranks(0.5 + rank, torch.device("cpu"))
    # a global batch the world size does not divide: 7 images over 2 ranks = shards of 4 and 3, padded to equal collective sizes
    lo7, hi7 = D.shard_range(7, rank, world)
    loc7 = torch.arange(lo7, hi7, dtype=torch.float32).view(-1, 1).expand(-1, 5).contiguous()
    all7 = D.allgather_logits(loc7, global_batch=7)
    # plain Python lists through the queue: a tensor travels as a file descriptor that the parent can only fetch while this
    # process is still alive (a FileNotFoundError race at exit otherwise)
    q.put((rank, lo, hi, allg.tolist(), slow, (lo7, hi7), all7.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_harness_gloo_world2():
    """N>1 path of bench.py: batch sharding, ONE all-gather of the per-rank logits, max-over-ranks timing."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, g0, s0, sh0, u0), (r1, lo1, hi1, g1, s1, sh1, u1) = res
    assert (sh0, sh1) == ((0, 4), (4, 7))
    u0, u1, g0, g1 = torch.tensor(u0), torch.tensor(u1), torch.tensor(g0), torch.tensor(g1)
    assert torch.equal(u0, u1) and u0.shape == (7, 5) and u0[:, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 6)
    assert torch.equal(g0, g1) and g0.shape == (6, 2, 3)
    assert g0[:, 0, 0].tolist() == [0.0, 10.0, 20.0, 31.0, 41.0, 51.0]  # rank-major order
    assert s0 == s1 == 1.5


def _logits_gather_worker(rk, ws, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rk, world_size=ws)
    from mmsa.dist import LogitsGather, allgather_logits
    g = LogitsGather()
    local = torch.zeros(2 if rk == 0 else 1, 3)           # ragged: a global batch of 3 over 2 ranks
    hs, log = [], []
    for k in range(4):
        local.fill_(10.0 * k + rk)                        # "replay k" overwrites the SAME logits buffer
        hs.append(g.submit(local, global_batch=3))
        log.append(("submit", k))
        if k >= 1:                                        # step k is under way: only now is step k - 1 collected
            r_ = hs[k - 1].result()
            log.append(("collect", k - 1))
            assert r_.shape == (3, 3) and r_[:, 0].tolist() == [10.0 * (k - 1), 10.0 * (k - 1), 10.0 * (k - 1) + 1], r_
    last = hs[-1].result()
    same = torch.equal(last, allgather_logits(local, 3))  # the synchronous form gives the same tensor
    q.put((rk, log, g.issued, same))
    dist.destroy_process_group()


def test_logits_gather_is_pipelined_behind_the_next_step_gloo_world2():
    """mmsa.dist.LogitsGather (round 6): step k's gather is issued with async_op, reads a staging COPY of the logits (so the next step may overwrite the
    buffer at once), and is collected after step k + 1 has been submitted; double-buffered staging / outputs; ragged shards; same tensor as the
    synchronous allgather_logits."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_logits_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rk, log, issued, same in res:
        assert same
        assert issued == [(0, True), (1, True), (2, True), (3, True)]
        assert log == [("submit", 0), ("submit", 1), ("collect", 0), ("submit", 2), ("collect", 1), ("submit", 3), ("collect", 2)]


def _run_bench_stub(extra_env, args=("--gpus", "2", "--steps", "3", "--warmup", "1"), timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MMSA_BENCH_STUB="1", MMSA_BENCH_TIMEOUT="120", **extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_rank_body_gloo_world2():
    """bench.py's own N > 1 control path end to end without a GPU (MMSA_BENCH_STUB=1: gloo, the device step replaced by a stand-in
    that takes 10 (rank + 1) ms): `--gpus 2` spawns its two ranks, they rendezvous, time exactly K steps between barriers, all-gather
    the per-rank logits every step, take the MAX over ranks, and rank 0 prints the one JSON line -- checked field by field."""
    import json
    r = _run_bench_stub({})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["unit"] == "images/s"
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp2"
    # the slower rank (20 ms per step) sets the time: max over ranks, not rank 0's own 10 ms
    assert 20.0 <= line["ms_per_step"] < 200.0
    assert abs(line["value"] - 4 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-2 * line["value"]
    assert line["roofline"] is None and line["cpu_baseline"] is None and line["verified"] is None
    assert line["config"]["collective_overlap"] == {"next_step_enqueued_before_previous_gather_was_collected": 2, "collected": 4}


def test_bench_rank_body_gloo_world8_with_ragged_shards():
    """The same control path at the width the driver's scaling run uses: `--gpus 8` -> eight ranks on gloo, a global batch of 13 images
    split 2,2,2,2,2,1,1,1 (mmsa.dist.shard_range) and gathered through padded shards every step, the chain probe's all-reduce (every rank
    must take the max-over-ranks pair), max-over-ranks timing (rank 7's 80 ms stand-in sets it) and the N = 8 JSON line."""
    import json
    r = _run_bench_stub({"MMSA_BENCH_STUB_GLOBAL_BATCH": "13", "OMP_NUM_THREADS": "1"}, args=("--gpus", "8", "--steps", "2", "--warmup", "1"), timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["config"]["parallelism"] == "dp8" and line["config"]["global_batch"] == 13
    assert line["chains_probe_ms"] == {"chains_ms": 12.0, "one_chain_ms": 50.0}
    assert 80.0 <= line["ms_per_step"] < 400.0
    assert abs(line["value"] - 13 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-2 * line["value"]
    # round 6: the gather of step k is collected only after step k + 1 has been enqueued -- (warm-up 1 + 2 timed steps) of which the
    # first of each loop has no predecessor -> 1 overlapped hand-over, every gather collected (the drains included) -- and each rank takes its share of the
    # host threads before it generates the seeded parameters
    assert line["config"]["collective_overlap"] == {"next_step_enqueued_before_previous_gather_was_collected": 1, "collected": 3}
    assert line["config"]["host_threads_per_rank"] == max(1, (os.cpu_count() or 1) // 8)


def test_bench_parent_stops_the_job_when_a_rank_dies():
    """A rank that exits at start-up must not leave the parent waiting for rank 0's rendezvous timeout."""
    import time
    t0 = time.time()
    r = _run_bench_stub({"MMSA_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert time.time() - t0 < 100, "the parent waited for the surviving rank's rendezvous timeout"
    assert "stopping" in r.stderr


def test_head_registry_and_state_dict_contract(golden_dir):
    """SegformerHead drop-in: registry name, reference state_dict keys/shapes/order, loud failure without a GPU."""
    import mmsa
    from tests.configs import HEAD_CONFIGS, make_head_inputs
    cfg = HEAD_CONFIGS["head_tiny"]
    head = mmsa.build_head(dict(type="SegformerHead", **cfg["kwargs"]))
    assert mmsa.HEADS.get("SegformerHead") is mmsa.SegformerHead
    want = [ln.strip().split(" ", 1) for ln in open(os.path.join(golden_dir, "state_dict_keys_head.txt"))]
    got = [(k, str(list(v.shape))) for k, v in head.state_dict().items()]
    assert [k for k, _ in got] == [k for k, _ in want]
    assert [s for _, s in got] == [s for _, s in want]
    with pytest.raises(RuntimeError, match="no CPU path"):
        head(make_head_inputs(cfg))
    with pytest.raises(NotImplementedError):
        mmsa.build_head(dict(type="SegformerHead", **dict(cfg["kwargs"], interpolate_mode="nearest")))


def test_convnext_checkpoint_duplication_matches_reference(golden_dir, tmp_path):
    """TwinConvNeXt.init_weights (TC:403-443): a single-stream ConvNeXt checkpoint lands in both streams; the set of loaded
    keys (the reference leaves the per-stage out norms untouched) and their values equal what the reference ends up with."""
    import mmsa
    from tests.configs import CONFIGS, fake_convnext_checkpoint
    cfg = CONFIGS["tiny256"]
    m0 = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    twin = [(k[len("spm.twin_conv."):], tuple(v.shape)) for k, v in m0.state_dict().items() if k.startswith("spm.twin_conv.")]
    ck = fake_convnext_checkpoint(twin)
    path = os.path.join(tmp_path, "convnext.pth")
    torch.save({"state_dict": ck}, path)
    before = {k: v.clone() for k, v in m0.state_dict().items()}
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **dict(cfg["kwargs"], checkpoint=path)))
    gold = np.load(os.path.join(golden_dir, "convnext_ckpt.npz"))
    want = set(str(k) for k in gold["loaded"])
    sd = m.state_dict()
    got = set()
    for k in sd:
        if not k.startswith("spm.twin_conv."):
            continue
        first, rest = k[len("spm.twin_conv."):].split(".", 1)
        src = (first[:-2] if first.endswith(("_x", "_y")) else first) + "." + rest
        if src in ck and torch.equal(ck[src], sd[k]):
            got.add(k)
    assert got == want and len(got) == 104
    cs = sum(sd[k].double().abs().sum().item() for k in gold["loaded"])
    assert abs(cs - float(gold["checksum"])) < 1e-9 * float(gold["checksum"])
    assert int(gold["n_twin"]) == len(twin)


def test_no_lane_swizzled_packed_fp32():
    """The library is built without hipcc's SLP vectoriser (build.py): no v_pk_{fma,mul,add}_f32 whose low lane selects the high
    half of a register pair -- the instruction form behind round 1's wrong upper halves under concurrent streams."""
    from tools.isa_audit import audit
    n, npk, bad = audit(os.path.join(ROOT, "multimodal-sam-adapter_amd", "mmsa", "libmmsa_hip.so"))
    assert n >= 10 and not bad, bad[:5]


def test_head_has_the_decode_head_test_surface():
    """BaseDecodeHead's inference surface used by the reference's EncoderDecoder (ED:129-133): forward_test(inputs, img_metas,
    test_cfg) exists and reaches forward (which refuses CPU tensors loudly); forward_train is refused, not missing."""
    import mmsa
    from tests.configs import HEAD_CONFIGS, make_head_inputs
    cfg = HEAD_CONFIGS["head_tiny"]
    head = mmsa.build_head(dict(type="SegformerHead", **cfg["kwargs"]))
    with pytest.raises(RuntimeError, match="no CPU path"):
        head.forward_test(make_head_inputs(cfg), [dict()], dict(mode="whole"))
    with pytest.raises(NotImplementedError):
        head.forward_train(make_head_inputs(cfg), [dict()], None, None)
    assert callable(mmsa.register_head)


def test_bench_parent_starts_no_ranks_without_gpus():
    """`python bench.py --gpus 2` with no WORLD_SIZE is the self-launching parent: it must decide from the device COUNT alone
    (no GPU initialisation) and fail loudly when there are fewer GPUs than ranks (this container has none)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("GPUs present: the parent would really launch")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "--gpus 2 asked for" in r.stderr


def _tiny_backbone(**extra):
    import mmsa
    from tests.configs import CONFIGS
    return mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **dict(CONFIGS["tiny256"]["kwargs"], **extra)))


def test_sam_checkpoint_ingestion_matches_the_reference_loader(golden_dir, tmp_path):
    """init_weights(pretrained) against the reference's own mmcv_custom.load_checkpoint (checkpoint.py:319-514) run on the same seeded
    checkpoint (tests/golden/sam_ckpt.npz case A): 'state_dict' wrapper + 'module.' prefix, an unexpected key, a missing block and
    a rel-pos table of the wrong length.  Same keys loaded, same keys left at their initialisation, same checksum."""
    from tests.configs import fake_sam_checkpoint
    g = np.load(os.path.join(golden_dir, "sam_ckpt.npz"))
    m0 = _tiny_backbone()
    vit = [(k, tuple(v.shape)) for k, v in m0.state_dict().items() if k.startswith(("pos_embed", "patch_embed.", "blocks."))]
    plain = fake_sam_checkpoint(vit, seed=51)
    path = str(tmp_path / "sam_a.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in plain.items()}, "meta": {"epoch": 3}}, path)
    with pytest.warns(UserWarning, match="size mismatch"):
        m = _tiny_backbone(pretrained=path)
    sd = m.state_dict()
    loaded = [k for k, _ in vit if k in plain and plain[k].shape == sd[k].shape and torch.equal(plain[k], sd[k])]
    assert loaded == list(g["a_loaded"])
    assert [k for k, _ in vit if k not in loaded] == list(g["a_not_loaded"])
    assert "blocks.1.attn.rel_pos_h" in list(g["a_not_loaded"])
    chk = sum(sd[k].double().abs().sum().item() for k in loaded)
    assert abs(chk - float(g["a_checksum"])) <= 1e-9 * float(g["a_checksum"])
    rep = m._pretrained_report
    assert rep[1] == ["blocks.1.attn.rel_pos_h"] and rep[2] == ["neck.0.weight"]


def test_sam_release_conversion_matches_the_reference_tool(golden_dir, tmp_path):
    """convert_sam_release against the reference's tools/SAM_checkpoint_convert.py::remove_neck_from_checkpoint (case B of the
    fixture): same surviving key set, and the converted checkpoint loads every ViT key."""
    from mmsa.checkpoint import convert_sam_release
    from tests.configs import fake_sam_checkpoint
    g = np.load(os.path.join(golden_dir, "sam_ckpt.npz"))
    m0 = _tiny_backbone()
    vit = [(k, tuple(v.shape)) for k, v in m0.state_dict().items() if k.startswith(("pos_embed", "patch_embed.", "blocks."))]
    raw = {"image_encoder." + k: v for k, v in fake_sam_checkpoint(vit, seed=52, drop=False).items()}
    raw["image_encoder.neck.0.weight"] = torch.ones(4, 4)
    raw["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = torch.ones(2, 8)
    raw["mask_decoder.iou_token.weight"] = torch.ones(1, 8)
    conv = convert_sam_release(raw)
    assert sorted(conv.keys()) == list(g["b_converted_keys"])
    path = str(tmp_path / "sam_b.pth")
    torch.save(conv, path)
    m = _tiny_backbone(pretrained=path)
    sd = m.state_dict()
    assert [k for k, _ in vit if torch.equal(conv[k], sd[k])] == list(g["b_loaded"]) and len(g["b_not_loaded"]) == 0
    chk = sum(sd[k].double().abs().sum().item() for k, _ in vit)
    assert abs(chk - float(g["b_checksum"])) <= 1e-9 * float(g["b_checksum"])


def test_unreadable_convnext_checkpoint_is_reported(monkeypatch):
    monkeypatch.delenv("MMSA_CONVNEXT_CKPT", raising=False)
    with pytest.warns(UserWarning, match="ConvNeXt checkpoint"):
        _tiny_backbone(checkpoint="https://download.openmmlab.com/mmclassification/v0/convnext/convnext-small.pth")


def test_checkpoint_unwrap_conventions():
    from mmsa.checkpoint import unwrap_state_dict
    t = torch.ones(1)
    assert unwrap_state_dict({"model": {"module.a.b": t}}) == {"a.b": t}
    assert unwrap_state_dict({"module": {"encoder.x": t, "zhead.y": t}}) == {"x": t}     # MoBY online branch (checkpoint.py:355-360)
    assert unwrap_state_dict({"a": t}) == {"a": t}
    with pytest.raises(RuntimeError):
        unwrap_state_dict([t])


def test_planes_with_a_column_split_host_logic():
    """ops.Planes.split (qkv planes: q, k as bf16 hi/lo, the v columns as h8 planes): allocation checks, the GEMM's cp_fmt encoding
    (include/mmsa.h: bits 8.. = split / 32), the v_fmt the attention wrappers derive from it, and the decode helper on a hand-packed
    row (no GPU: torch ops only)."""
    import mmsa
    from mmsa import ops
    pl = ops.alloc_planes(4, 96, "cpu", split=64)
    assert pl.split == 64 and pl.fmt == ops.FMT_B3 and pl.rows(1, 3).split == 64
    assert ops.cp_format(pl) == ops.FMT_B3 | (2 << 8) and ops.cp_format(None) == ops.FMT_B3
    assert ops.cp_format(ops.alloc_planes(4, 96, "cpu", fmt=ops.FMT_H8)) == ops.FMT_H8
    for bad in (dict(split=48), dict(split=96), dict(split=64, fmt=ops.FMT_H8)):
        with pytest.raises(RuntimeError):
            ops.alloc_planes(4, 96, "cpu", **bad)
    d = 32
    f3 = lambda r, c, **kw: ops.alloc_planes(r, c, "cpu", fmt=ops.FMT_F3, **kw)     # the attention kernels' hi/lo form reads fp16 pairs (round 4)
    qkv, bias = f3(8, 3 * d, split=2 * d), f3(1, 3 * d, split=2 * d)
    assert ops._v_fmt(qkv, bias, d) == 1
    assert ops._v_fmt(f3(8, 3 * d), f3(1, 3 * d), d) == 0
    with pytest.raises(RuntimeError):
        ops._v_fmt(qkv, f3(1, 3 * d), d)
    with pytest.raises(RuntimeError):                                   # bf16 hi/lo planes: same layout, would be misread -- refused
        ops._v_fmt(ops.alloc_planes(8, 3 * d, "cpu"), ops.alloc_planes(1, 3 * d, "cpu"), d)
    with pytest.raises(RuntimeError):                                   # the split form is not for the fused rel-pos entries
        ops._v_fmt(qkv, bias, d, fused=True, rel=f3(64, 64))
    h = lambda r, c: ops.alloc_planes(r, c, "cpu", fmt=ops.FMT_H8)
    assert ops._v_fmt(h(8, 3 * d), h(1, 3 * d), d, fused=True, rel=h(64, 64)) == 2
    for args in ((h(8, 3 * d), h(1, 3 * d), d), (h(8, 3 * d), f3(1, 3 * d), d, True, h(64, 64)),
                 (h(8, 3 * d), h(1, 3 * d), d, True, f3(64, 64))):
        with pytest.raises(RuntimeError):
            ops._v_fmt(*args)
    # hand-packed row: two bf16 hi/lo blocks, one h8 block (32 fp16 hi, then four 16-byte chunks: 8 e5m2 bytes of lo * 2^11 | 8 of hi)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 96, generator=g)
    p = torch.zeros(2, 192, dtype=torch.int16)
    for blk in range(2):
        v = x[:, 32 * blk:32 * blk + 32]
        hi = v.bfloat16()
        lo = (v - hi.float()).bfloat16()
        p[:, 64 * blk:64 * blk + 32] = hi.view(torch.int16)
        p[:, 64 * blk + 32:64 * blk + 64] = lo.view(torch.int16)
    v = x[:, 64:]
    hi = v.half()
    lo8 = ((v - hi.float()) * 2048.0).to(torch.float8_e5m2).view(torch.uint8)
    qh8 = hi.float().to(torch.float8_e5m2).view(torch.uint8)
    chunks = torch.cat([lo8.view(2, 4, 1, 8), qh8.view(2, 4, 1, 8)], 2).reshape(2, 64)      # chunk g = elements 8g .. 8g+7
    p[:, 128:160] = hi.view(torch.int16)
    p[:, 160:192] = chunks.view(torch.int16)
    got = ops.planes_to_float(ops.Planes(p, 2, 96, 96, ops.FMT_B3, split=64))
    assert torch.equal(got[:, :64], x[:, :64].bfloat16().float() + (x[:, :64] - x[:, :64].bfloat16().float()).bfloat16().float())
    assert (got[:, 64:] - x[:, 64:]).abs().max() <= 2.0 ** -13 * x.abs().max()


def test_f3_and_h8c_planes_host_logic():
    """The two plane formats of round 4 on the host side (no GPU: torch ops only): shapes / strides ops.Planes derives, the GEMM's cp_fmt encoding,
    and the decode helper on hand-packed rows -- f3 = the bf16 hi/lo layout with fp16 halves; h8c = row pairs, lo bytes scaled by 2^11 * 1.09375
    (csrc/common.h MMSA_H8C_LO_COMP: what makes up for the truncated q(hi))."""
    from mmsa import ops
    assert (ops.FMT_B3, ops.FMT_H8, ops.FMT_H8C, ops.FMT_F3) == (0, 1, 2, 3)
    pl = ops.alloc_planes(6, 96, "cpu", fmt=ops.FMT_F3)
    assert tuple(pl.p.shape) == (6, 192) and pl.fmt == ops.FMT_F3 and ops.cp_format(pl) == ops.FMT_F3 and pl.rows(2, 4).fmt == ops.FMT_F3
    assert ops.planes_shape(6, 96, ops.FMT_F3) == ops.planes_shape(6, 96, ops.FMT_B3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, generator=g) * 3
    x[0, :4] = torch.tensor([3.0e-6, -2.0e-7, 1.0, 1000.5])
    p = torch.zeros(2, 128, dtype=torch.int16)
    for blk in range(2):
        v = x[:, 32 * blk:32 * blk + 32]
        hi = v.half()
        p[:, 64 * blk:64 * blk + 32] = hi.view(torch.int16)
        p[:, 64 * blk + 32:64 * blk + 64] = (v - hi.float()).half().view(torch.int16)
    got = ops.planes_to_float(ops.Planes(p, 2, 64, 64, ops.FMT_F3))
    assert ((got - x).abs() <= x.abs() * 2.0 ** -21 + 6.1e-8).all()
    # h8c: one row pair, K = 64: [row 0: 64 fp16][row 1: 64 fp16][one 128-byte line: row 0's 64 lo bytes | row 1's], a row's lo bytes = 4 groups of
    # [k = 8g .. 8g+7 | k = 32 + 8g .. 32 + 8g + 7]
    hp = ops.alloc_planes(2, 64, "cpu", fmt=ops.FMT_H8C)
    assert tuple(hp.p.shape) == (1, 192) and hp.kpad == 64 and hp.batch_stride(2) == 192
    hi = x.half()
    lo = ((x - hi.float()) * (2048.0 * 1.09375)).to(torch.float8_e5m2).view(torch.uint8)            # [2, 64]
    line = lo.view(2, 2, 4, 8).permute(0, 2, 1, 3).reshape(2, 64)                                  # [row][g][k-tile][e]
    raw = torch.cat([hi.view(torch.uint8).reshape(-1), line.reshape(-1)]).view(torch.int16).view(1, 192)
    got = ops.planes_to_float(ops.Planes(raw, 2, 64, 64, ops.FMT_H8C))
    assert torch.equal(got, hi.float() + lo.view(torch.float8_e5m2).float() / (2048.0 * 1.09375))
    assert (got - x).abs().max() <= 2.0 ** -13 * x.abs().max()


def test_the_product_package_reads_no_numerics_switch_from_the_environment():
    """every switch that changes what the package computes is a module attribute (bench.py --set passes them in A/B runs).  The only
    environment variables the package reads name FILES (the library to load, a ConvNeXt checkpoint) or arm a test aid that changes no result."""
    pkg = os.path.join(ROOT, "multimodal-sam-adapter_amd", "mmsa")
    allowed = {"MMSA_LIB", "MMSA_CONVNEXT_CKPT", "MMSA_DEBUG_POISON_LDS"}
    found = set()
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            found |= set(re.findall(r"os\.environ(?:\.get)?[\(\[]\s*[\"']([A-Z0-9_]+)[\"']", src))
            assert "getenv" not in src, fn
    assert found <= allowed, f"environment reads in the product package: {sorted(found - allowed)}"


def test_gemm_kernels_do_not_spill_into_their_k_loops():
    """a scratch reload in front of an LDS-DMA instruction is an s_waitcnt vmcnt(0) -- a drain of the whole prefetch stream.  Every
    instantiation of the LDS-DMA GEMM kernels is compiled to ISA here (hipcc cross-compiles without a GPU) and must stay at <= 10 scratch instructions
    (gemm_v2: 0 since the epilogue sees the lane id through an opaque copy; gemm_h8c: the handful around its tile loop, none between its barriers)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_scratch", os.path.join(ROOT, "tools", "isa_scratch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.census(os.path.join(ROOT, "multimodal-sam-adapter_amd", "csrc", "gemm_h8c.hip"))
    assert len(res) >= 6
    for k, v in res.items():
        assert v["scratch"] <= 10, f"{k}: {v['scratch']} scratch instructions"
    res_s = mod.census(os.path.join(ROOT, "multimodal-sam-adapter_amd", "csrc", "gemm_stream.hip"))   # the streaming kernel (round 6): its first build spilled 870 bytes per lane
    assert len(res_s) >= 8
    for k, v in res_s.items():
        assert v["scratch"] == 0 and v["vgprs"] <= 256, f"{k}: {v['scratch']} scratch instructions, {v['vgprs']} VGPRs"
    res4 = mod.census(os.path.join(ROOT, "multimodal-sam-adapter_amd", "csrc", "gemm_h8c4.hip"))   # the 4-wave flavour (round 6): two waves per SIMD -> <= 256 registers
    assert len(res4) >= 1
    for k, v in res4.items():
        assert v["scratch"] == 0 and v["vgprs"] <= 256, f"{k}: {v['scratch']} scratch instructions, {v['vgprs']} VGPRs"


def test_kernel_gelu_coefficients_hold_their_error_bound():
    """csrc/common.h evaluates GELU as x/2 + |x/2| (1 - exp2(P(min(|x|, 6)))) with a fitted P (tools/gelu_fit.py).  The coefficients compiled into the kernels,
    evaluated in fp32 in the kernels' operation order, stay within 4e-7 of the erf-form GELU (nn.GELU's default, the reference's activation: IE:154-167, TC:107-111)
    over [-12, 12], and a fresh fit reproduces them."""
    import sys
    from scipy.special import erf
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gelu_fit
    src = open(os.path.join(ROOT, "multimodal-sam-adapter_amd", "csrc", "common.h")).read()
    deg = int(re.search(r"#define MMSA_GELU_DEG (\d+)", src).group(1))
    block = re.search(r"#%s MMSA_GELU_DEG == %d\s*\n#define MMSA_GELU_COEFFS \{([^}]*)\}" % ("if" if deg == 6 else "elif", deg), src)
    assert block, "coefficient table of the compiled degree not found"
    c32 = np.array([float(v.strip().rstrip("f")) for v in block.group(1).split(",")], dtype=np.float32)
    assert len(c32) == deg
    x = np.concatenate([-np.linspace(0, 12, 200001)[::-1], np.linspace(0, 12, 200001)]).astype(np.float32)
    exact = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    err = np.abs(gelu_fit.gelu_new(x, c32).astype(np.float64) - exact).max()
    assert err < 4e-7, err
    _, c = gelu_fit.fit(deg, n=20001, iters=200)
    assert np.allclose(c.astype(np.float32), c32, rtol=2e-3, atol=2e-7), (c, c32)
