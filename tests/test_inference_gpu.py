"""GPU parity of the segmentor glue (mmsa.inference: encode_decode, slide_inference, argmax map) against the CPU oracle
(oracle/ref_segmentor.py, pinned to the reference's own EncoderDecoder.slide_inference)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_encoder as R
from oracle import ref_head as RH
from oracle import ref_segmentor as RS
from tests.configs import CONFIGS, HEAD_CONFIGS, make_input
from tests.util import assert_close
from tests.weights import seeded_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def models():
    import mmsa
    cfg, hcfg = CONFIGS["tiny256"], HEAD_CONFIGS["head_tiny"]
    orc = R.OracleEncoder(**cfg["kwargs"])
    sd = seeded_state_dict(orc, seed=cfg["seed"])
    orc.load_state_dict(sd)
    horc = RH.OracleSegformerHead(**hcfg["kwargs"])
    hsd = seeded_state_dict(horc, seed=hcfg["seed"])
    horc.load_state_dict(hsd)
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    m.load_state_dict(sd)
    h = mmsa.build_head(dict(type="SegformerHead", **hcfg["kwargs"]))
    h.load_state_dict(hsd)
    return cfg, orc, horc, m, h


def test_bilinear_accum_div_argmax_ops():
    import mmsa.inference as inf
    from mmsa import lib, ops
    g = torch.Generator().manual_seed(5)
    lg = torch.randn(2, 7, 13, 9, generator=g)
    ref = F.interpolate(lg, size=(40, 33), mode="bilinear", align_corners=False)
    canvas = torch.zeros(2, 7, 50, 60, device=DEV)
    count = torch.zeros(2, 50, 60, device=DEV)
    for _ in range(2):
        inf._resize_into(lg.to(DEV), canvas, 6, 11, 40, 33, count=count, accumulate=True)
    want = torch.zeros(2, 7, 50, 60)
    want[:, :, 6:46, 11:44] = 2 * ref
    assert_close(canvas, want, tol=1e-6, what="bilinear accumulate")
    assert torch.equal(count.cpu()[:, 6:46, 11:44], torch.full((2, 40, 33), 2.0)) and float(count.sum()) == 2 * 2 * 40 * 33
    inf._resize_into(lg.to(DEV), canvas, 6, 11, 40, 33)          # overwrite mode
    assert_close(canvas[:, :, 6:46, 11:44], ref, tol=1e-6, what="bilinear write")
    x = torch.randn(2, 7, 50, 60, generator=g)
    x[0, 3, 0, 0] = x[0, 5, 0, 0] = 9.0                          # a tie: first maximum wins
    am = inf.argmax_map(x.to(DEV))
    assert am.dtype == torch.uint8 and torch.equal(am.cpu().long(), x.argmax(1))


def test_encode_decode_and_slide_inference(models):
    import mmsa.inference as inf
    cfg, orc, horc, m, h = models
    x = make_input(cfg, batch=1)
    out = inf.encode_decode(m, h, x.to(DEV))
    ref = RS.encode_decode(orc, horc, x)
    assert_close(out, ref, what="encode_decode")
    # a 320 x 400 frame, 256 x 256 windows, stride 170: 2 x 2 windows with overlaps and a shifted last column / row
    g = torch.Generator().manual_seed(3)
    frame = torch.randn(1, 6, 320, 400, generator=g)
    frame[:, 3:] = (torch.rand(1, 3, 320, 400, generator=g) < 0.05).float() * torch.rand(1, 3, 320, 400, generator=g)
    got = inf.slide_inference(m, h, frame.to(DEV), (256, 256), (170, 170), max_batch=3)
    want = RS.slide_inference(lambda c: RS.encode_decode(orc, horc, c), frame, (256, 256), (170, 170), 7)
    assert_close(got, want, what="slide_inference logits")
    agree = (inf.argmax_map(got).cpu().long() == want.argmax(1)).float().mean().item()
    assert agree > 0.999, f"class maps agree on {agree:.5f} of the pixels"
    with pytest.raises(RuntimeError):
        inf.slide_inference(m, h, frame[:, :, :200].to(DEV), (256, 256), (170, 170))


def test_muses_frame_full_size_batching_invariance():
    """BASELINE.json configs[3] at full size: one [1, 6, 1080, 1920] frame -> six 1024 x 1024 windows
    (y in {0, 56}, x in {0, 640, 896}; ED:198-214 with crop 1024, stride 640) through the ViT-L encoder + head.  The oracle
    needs minutes per window, so the full-size check is a property: the window grid is the reference's, and the six windows
    batched through one encoder call give the logits of six single-window calls (images are independent in eval)."""
    import mmsa
    import mmsa.inference as inf
    assert inf.crop_boxes(1080, 1920, (1024, 1024), (640, 640)) == [(y, x, y + 1024, x + 1024) for y in (0, 56) for x in (0, 640, 896)]
    cfg, hcfg = CONFIGS["vitl1024"], HEAD_CONFIGS["head_vitl"]
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    m.load_state_dict(seeded_state_dict(m, seed=cfg["seed"]))
    h = mmsa.build_head(dict(type="SegformerHead", **hcfg["kwargs"]))
    h.load_state_dict(seeded_state_dict(h, seed=hcfg["seed"]))
    h = h.to(DEV)
    g = torch.Generator().manual_seed(21)
    frame = torch.randn(1, 6, 1080, 1920, generator=g)
    frame[:, 3:] = (torch.rand(1, 3, 1080, 1920, generator=g) < 0.05).float() * torch.rand(1, 3, 1080, 1920, generator=g)
    frame = frame.to(DEV)
    batched = inf.slide_inference(m, h, frame, (1024, 1024), (640, 640), max_batch=6)
    single = inf.slide_inference(m, h, frame, (1024, 1024), (640, 640), max_batch=1)
    assert batched.shape == (1, 25, 1080, 1920) and bool(torch.isfinite(batched).all())
    assert torch.equal(batched, single), "six windows batched vs one by one: the path is batch invariant bit for bit"
    # the top-left window alone covers rows < 56 and columns < 640: no averaging there
    alone = inf.encode_decode(m, h, frame[:, :, :1024, :1024].contiguous())
    assert_close(batched[:, :, :56, :640], alone[:, :, :56, :640], tol=1e-6, what="singly covered region equals that window's logits")


def test_fused_class_maps_equal_the_unfused_path(models):
    """slide_class_map / whole_class_map (crops by one kernel, no [B, classes, H, W] canvas, resize + overlap sum + count division +
    argmax in one pass) return the class map of slide_inference / encode_decode + argmax_map bit for bit, incl. a 2-image batch
    and windows that overlap 1, 2 and 4 times."""
    import mmsa.inference as inf
    cfg, orc, horc, m, h = models
    g = torch.Generator().manual_seed(9)
    frame = torch.randn(2, 6, 320, 400, generator=g)
    frame[:, 3:] = (torch.rand(2, 3, 320, 400, generator=g) < 0.05).float() * torch.rand(2, 3, 320, 400, generator=g)
    frame = frame.to(DEV)
    for stride, mb in (((170, 170), 3), ((256, 256), 8), ((64, 144), 5)):
        want = inf.argmax_map(inf.slide_inference(m, h, frame, (256, 256), stride, max_batch=mb))
        if len(inf.crop_boxes(320, 400, (256, 256), stride)) * 2 > 64:
            continue
        got, unc = inf.slide_class_map(m, h, frame, (256, 256), stride, max_batch=mb)
        assert int(unc.item()) == 0
        assert got.dtype == torch.uint8 and torch.equal(got, want), f"stride {stride}"
    x = make_input(cfg, batch=2, seed=4).to(DEV)
    assert torch.equal(inf.whole_class_map(m, h, x), inf.argmax_map(inf.encode_decode(m, h, x)))
    # a window list that leaves pixels uncovered is reported (ED:220), not silently mapped to class 0
    from mmsa import lib, ops
    import ctypes
    lg = torch.randn(1, 7, 16, 16, device=DEV)
    out = torch.zeros(1, 80, 80, dtype=torch.uint8, device=DEV)
    unc = torch.zeros(1, dtype=torch.int32, device=DEV)
    lib.call("mmsa_slide_argmax", lg.data_ptr(), 1, 7, 16, 16, (ctypes.c_int * 3)(0, 0, 0), out.data_ptr(), 1, 80, 80, 64, 64, unc.data_ptr(), ops._stream())
    assert int(unc.item()) == 80 * 80 - 64 * 64
    with pytest.raises(RuntimeError, match="outside"):
        lib.call("mmsa_slide_argmax", lg.data_ptr(), 1, 7, 16, 16, (ctypes.c_int * 3)(0, 40, 0), out.data_ptr(), 1, 80, 80, 64, 64, unc.data_ptr(), ops._stream())


def test_class_map_paths_round_identically_on_exact_ties():
    """The canvas path (bilinear_accum + div_count + argmax) and the one-pass kernel (slide_argmax) evaluate the same interpolation
    formula; they must also ROUND it the same way (no fused multiply-add in one and not the other), or an exact tie between two
    classes in one path is no tie in the other.  Class 5 is a copy of class 2: every pixel is a tie, the first class must win in
    both paths, over two overlapping windows."""
    import ctypes
    import mmsa.inference as inf
    from mmsa import lib, ops
    g = torch.Generator().manual_seed(11)
    lg = torch.randn(2, 7, 16, 16, generator=g) * 0.1
    lg[:, 2] += 3.0
    lg[:, 5] = lg[:, 2]
    lg = lg.to(DEV)
    H, W, hc, wc = 64, 88, 64, 64
    wins = [(0, 0, 0), (0, 0, 24)]                       # (image, y0, x0): columns 24..63 are covered twice
    canvas = torch.zeros(1, 7, H, W, device=DEV)
    count = torch.zeros(1, H, W, device=DEV)
    for k, (_, y0, x0) in enumerate(wins):
        inf._resize_into(lg[k:k + 1], canvas, y0, x0, hc, wc, count=count, accumulate=True)
    lib.call("mmsa_div_count_nchw", canvas.data_ptr(), count.data_ptr(), 1, 7, H * W, ops._stream())
    want = inf.argmax_map(canvas)
    out = torch.zeros(1, H, W, dtype=torch.uint8, device=DEV)
    unc = torch.zeros(1, dtype=torch.int32, device=DEV)
    tab = (ctypes.c_int * 6)(*[v for w in wins for v in w])
    lib.call("mmsa_slide_argmax", lg.data_ptr(), 2, 7, 16, 16, tab, out.data_ptr(), 1, H, W, hc, wc, unc.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    assert int(unc.item()) == 0
    assert torch.equal(canvas[:, 2], canvas[:, 5]) and bool((want == 2).all()), "the canvas path must see exact ties and pick the first class"
    assert torch.equal(out, want), "the one-pass class map rounds the interpolation differently from the canvas path"


def test_crop_batch_kernel():
    import ctypes
    import mmsa.inference as inf
    g = torch.Generator().manual_seed(2)
    img = torch.randn(3, 6, 70, 90, generator=g).to(DEV)
    chunk = [(2, (3, 5, 35, 69)), (0, (38, 26, 70, 90)), (1, (0, 0, 32, 64))]
    got = inf._crops(img, chunk, (32, 64))
    want = torch.stack([img[b, :, y1:y2, x1:x2] for b, (y1, x1, y2, x2) in chunk], 0)
    assert torch.equal(got, want)


@pytest.mark.parametrize("nch", [2, 3])
def test_concurrent_chains_equal_one_chain_bit_for_bit(models, nch):
    """mmsa.Chains: the batch as `nch` sub-batches on concurrent streams (own HIP graphs, private scratch buffers, shared packed
    weights, GEMM grids capped to a share of the CUs) returns, per chain, the bits of a plain forward + head of that sub-batch, replay
    after replay and for new input contents (bench.py checks the whole ViT-L batch of 2 against one chain of 2 the same way)."""
    import mmsa
    cfg, _, _, m, h = models
    x = make_input(cfg, batch=2 * nch, seed=31).to(DEV)
    ch = mmsa.Chains(m, h, n=nch).capture(x)
    for rep in range(3):
        if rep == 2:
            x.copy_(make_input(cfg, batch=2 * nch, seed=32).to(DEV))     # the graphs read the caller's buffer
        logits = ch.replay().outputs()       # verified: waits for this pass's guard copy, not for the device
        torch.cuda.synchronize()
        for c in range(nch):      # chain c == a plain forward + head of its own slice
            fs, _ = m(x[2 * c:2 * c + 2])
            ref = h(fs)
            torch.cuda.synchronize()
            assert torch.equal(logits[2 * c:2 * c + 2], ref)
            for k in range(4):
                assert torch.equal(ch.feats[c][k], fs[k])
    assert mmsa.ops.GEMM_MAX_GRID == 0 and getattr(h, "buf_tag", "") == ""      # nothing left switched on
    enc = mmsa.Chains(m, None, n=nch).capture(x)                               # encoder-only chains
    feats = enc.replay().outputs()
    torch.cuda.synchronize()
    for c in range(nch):
        fs, _ = m(x[2 * c:2 * c + 2])
        for k in range(4):
            assert torch.equal(feats[c][k], fs[k])
    with pytest.raises(RuntimeError):
        mmsa.Chains(m, None, n=nch).capture(x[:nch + 1])


def test_chains_refuse_unchecked_outputs(models):
    """a captured graph cannot re-route a block whose attention logits outgrow single fp16 operands, so Chains.replay() returns a
    handle and `outputs()` hands tensors out only after THAT pass's guard words have been inspected (an asynchronous 4 * depth-byte copy into pinned
    memory behind every pass).  A guard word beyond the threshold -- written here the way the attention kernels' atomic max would -- makes outputs()
    raise, moves the block to fp16 hi/lo pairs, captures the graphs again; the next pass is valid, the bad one stays refused.  Same for SlideRunner."""
    import mmsa
    import mmsa.inference as inf
    from mmsa.chains import AttentionRangeError
    cfg, _, _, m0, h = models
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))      # own instance: this test changes block modes
    m.load_state_dict(m0.state_dict())
    x = make_input(cfg, batch=2, seed=41).to(DEV)
    ch = mmsa.Chains(m, h, n=2).capture(x)
    r1 = ch.replay()
    good = r1.outputs().clone()
    assert all(mode == "f16" for mode, _ in m.attention_modes())
    m.attention_guard_words()[1] = 3.0 * m.ATTN_F16_MAX_LOGIT      # "block 1 scored a logit of 24 in the next pass"
    r2 = ch.replay()
    with pytest.raises(AttentionRangeError, match=r"block\(s\) \[1\]"):
        r2.outputs()
    assert m.attention_modes()[1][0] == "b3" and m.attention_modes()[0][0] == "f16"
    with pytest.raises(AttentionRangeError):
        r2.outputs()                       # the invalid pass stays refused
    r3 = ch.replay()                       # graphs were captured again on the re-routed block
    out3 = r3.outputs()
    fs, _ = m(x[:1])
    torch.cuda.synchronize()
    assert torch.equal(out3[:1], h(fs)) and out3.shape == good.shape       # block 1 now runs on pairs: the eager path's bits
    assert r1.outputs() is ch.logits       # a pass verified before stays verified
    # the same without reading outputs(): the NEXT replay() notices the arrived copy and raises before enqueuing anything
    m.attention_guard_words()[3] = 2.0 * m.ATTN_F16_MAX_LOGIT
    ch.replay()
    torch.cuda.synchronize()
    with pytest.raises(AttentionRangeError, match=r"block\(s\) \[3\]"):
        ch.replay()
    assert ch.replay().outputs() is ch.logits
    # check_every > 1: a pass without a copy of its own is verified by a synchronous read
    ch5 = mmsa.Chains(m, h, n=2, check_every=5).capture(x)
    assert ch5.replay().outputs() is ch5.logits and not ch5._pending
    # SlideRunner: FrameResult.outputs() goes through the same check
    frame = torch.randn(1, 6, 300, 420, generator=torch.Generator().manual_seed(5)).to(DEV)
    sr = inf.SlideRunner(m, h, frame, (256, 256), (160, 160), chains=2)
    cm, unc = sr.run().outputs()
    m.attention_guard_words()[0] = 2.0 * m.ATTN_F16_MAX_LOGIT
    fr = sr.run()
    with pytest.raises(AttentionRangeError):
        fr.outputs()
    cm2, _ = sr.run().outputs()
    assert cm2.shape == cm.shape


def test_slide_runner_equals_slide_inference(models):
    """mmsa.inference.SlideRunner (windows as concurrent chains, HIP graphs) == slide_inference + argmax_map, bit for bit, run after run."""
    import mmsa.inference as inf
    cfg, _, _, m, h = models
    g = torch.Generator().manual_seed(77)
    frame = torch.randn(1, 6, 300, 420, generator=g).to(DEV)        # 256-pixel windows, stride 160: 1 x 3 = ... windows (even count needed for 2 chains)
    boxes = inf.crop_boxes(300, 420, (256, 256), (160, 160))
    sr = inf.SlideRunner(m, h, frame, (256, 256), (160, 160), chains=2)
    for rep in range(3):
        if rep == 2:
            frame.copy_(torch.randn(1, 6, 300, 420, generator=g).to(DEV))
        cm, unc = sr.run().outputs()
        torch.cuda.synchronize()
        # the plain function on the chains' own sub-batch size: at this toy size the row count decides which GEMM kernel runs (not the
        # same bits for another batch size -- tests/test_backbone_gpu.py; at ViT-L sizes there is one kernel per shape)
        want = inf.argmax_map(inf.slide_inference(m, h, frame, (256, 256), (160, 160), max_batch=len(boxes) // 2))
        assert int(unc.item()) == 0 and torch.equal(cm, want)


def test_whole_dim_modes(models):
    """`test_cfg.mode` 'whole_dim' (DELIVER configs) and 'whole_dim_cut' (FMB configs) on device against the oracle restatement of
    ED:329-413 (pinned to the reference's own methods: tests/golden/whole_dim.npz), plus the mode dispatch of ED:417-447."""
    import mmsa.inference as inf
    cfg, orc, horc, m, h = models
    x = make_input(cfg, batch=2, seed=17)
    ed = lambda im: RS.encode_decode(orc, horc, im)
    xd = x.to(DEV)
    same = inf.whole_inference_dim(m, h, xd, (256, 256))
    assert_close(same, RS.whole_inference_dim(ed, x, (256, 256)), what="whole_dim at the input size")
    assert torch.equal(same, inf.encode_decode(m, h, xd))
    assert_close(inf.whole_inference_dim(m, h, xd, (192, 240)), RS.whole_inference_dim(ed, x, (192, 240)), what="whole_dim to another size")
    got = inf.whole_inference_dim_cut(m, h, xd, (192, 256), (256, 192), rescale=False)     # the FMB form: crop only
    assert got.shape == (2, 7, 192, 256) and got.is_contiguous()
    assert_close(got, RS.whole_inference_dim_cut(ed, x, (192, 256), (256, 192), rescale=False), what="whole_dim_cut, no rescale")
    assert_close(inf.whole_inference_dim_cut(m, h, xd, (200, 300), (260, 150), rescale=True),
                 RS.whole_inference_dim_cut(ed, x, (200, 300), (260, 150), rescale=True), what="whole_dim_cut, rescale")
    with pytest.raises(RuntimeError, match="no defined result"):
        inf.whole_inference_dim(m, h, xd, (256, 256), rescale=False)
    # dispatch
    assert torch.equal(inf.inference(m, h, xd, dict(mode="whole_dim", dim=(256, 256))), same)
    assert torch.equal(inf.inference(m, h, xd, dict(mode="whole_dim_cut", dim=(192, 256), cut_dim=(256, 192)), rescale=False), got)
    assert torch.equal(inf.inference(m, h, xd, dict(mode="whole")), same)
    with pytest.raises(RuntimeError, match="not one of"):
        inf.inference(m, h, xd, dict(mode="slide_mod_sel"))
