"""Inference glue of the reference's EncoderDecoder on device (segmentation/mmseg_custom/models/segmentors/encoder_decoder.py):
`encode_decode` (ED:85-95), `slide_inference` (ED:191-234), `whole_inference`, `whole_inference_dim` (ED:329-362),
`whole_inference_dim_cut` (ED:364-413), the mode dispatch of `inference` (ED:417-447) and the class map of `simple_test` (ED:449,477).

The crops of a sliding-window frame are batched through ONE backbone + head call (the reference runs them one by one,
ED:205-214), and the resize / pad / accumulate / count of every crop is one kernel launch on the logits canvas."""
import torch

from . import lib
from . import ops


def _on_device(fn):
    """Run an entry point with its first tensor argument's device current (launches go to that device's current stream)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kw):
        t = next((a for a in args if isinstance(a, torch.Tensor)), None)
        if t is None or not t.is_cuda:
            return fn(*args, **kw)     # _check raises the "no CPU path" error
        with torch.cuda.device(t.device):
            return fn(*args, **kw)
    return wrapped


def _resize_into(logits, canvas, y0, x0, hc, wc, count=None, accumulate=False):
    b, c, hs, ws = logits.shape
    lib.call("mmsa_bilinear_accum_nchw", logits.data_ptr(), c * hs * ws, b, c, hs, ws, canvas.data_ptr(), canvas.shape[2], canvas.shape[3],
             y0, x0, hc, wc, count.data_ptr() if count is not None else None, 1 if accumulate else 0, ops._stream())


def _pair(backbone, head):
    """With this package's head behind this package's backbone, the backbone's tail also emits its maps as planes for the head."""
    from .head import SegformerHead
    if isinstance(head, SegformerHead) and hasattr(backbone, "_vit"):
        backbone.emit_planes = True


def _check(img):
    if not img.is_cuda or img.dtype != torch.float32 or img.dim() != 4:
        raise RuntimeError("mmsa.inference: img must be a float32 [B, C, H, W] GPU tensor (there is no CPU path)")


@_on_device
@torch.no_grad()
def encode_decode(backbone, head, img):
    """ED:85-95: logits of the head resized (bilinear, align_corners=False) to the input size -> [B, classes, H, W]."""
    _check(img)
    _pair(backbone, head)
    feats, _ = backbone(img)
    lg = head(feats)
    out = torch.empty(img.shape[0], lg.shape[1], img.shape[2], img.shape[3], device=img.device)
    _resize_into(lg, out, 0, 0, img.shape[2], img.shape[3])
    return out


def crop_boxes(h_img, w_img, crop_size, stride):
    """The window grid of ED:198-212 (windows at the right / bottom border are shifted inwards)."""
    h_crop, w_crop = crop_size
    h_stride, w_stride = stride
    h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
    w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
    boxes = []
    for h_idx in range(h_grids):
        for w_idx in range(w_grids):
            y1, x1 = h_idx * h_stride, w_idx * w_stride
            y2, x2 = min(y1 + h_crop, h_img), min(x1 + w_crop, w_img)
            y1, x1 = max(y2 - h_crop, 0), max(x2 - w_crop, 0)
            boxes.append((y1, x1, y2, x2))
    return boxes


@_on_device
@torch.no_grad()
def slide_inference(backbone, head, img, crop_size, stride, max_batch=8):
    """ED:191-234 without the optional rescale: averaged logits [B, classes, H, W] of overlapping windows.  All windows have
    the crop size here (the backbone needs H = W = img_size), i.e. the image must be at least as large as the crop."""
    _check(img)
    B, _, H, W = img.shape
    if H < crop_size[0] or W < crop_size[1]:
        raise RuntimeError("mmsa.slide_inference: the image must be at least as large as the crop")
    boxes = crop_boxes(H, W, crop_size, stride)
    _pair(backbone, head)
    preds = count = None
    jobs = [(b, box) for box in boxes for b in range(B)]
    for s in range(0, len(jobs), max_batch):
        chunk = jobs[s:s + max_batch]
        crops = _crops(img, chunk, crop_size)
        feats, _ = backbone(crops)
        lg = head(feats)                                   # [n, classes, hc/4, wc/4]
        if preds is None:
            preds = torch.zeros(B, lg.shape[1], H, W, device=img.device)
            count = torch.zeros(B, H, W, device=img.device)
        for k, (b, (y1, x1, y2, x2)) in enumerate(chunk):   # preds[b] += pad(resize(logits_k)); count[b, window] += 1
            _resize_into(lg[k:k + 1], preds[b:b + 1], y1, x1, y2 - y1, x2 - x1, count=count[b:b + 1], accumulate=True)
    if bool((count == 0).any()):
        raise RuntimeError("mmsa.slide_inference: windows do not cover the image")   # ED:220
    lib.call("mmsa_div_count_nchw", preds.data_ptr(), count.data_ptr(), B, preds.shape[1], H * W, ops._stream())
    return preds


def _crops(img, chunk, crop_size, out=None):
    """ED:205-212 for a batch of windows: one HIP launch (mmsa_crop_batch_nchw), no ATen slicing / stacking."""
    import ctypes
    n = len(chunk)
    if out is None:
        out = torch.empty(n, img.shape[1], crop_size[0], crop_size[1], device=img.device)
    tab = (ctypes.c_int * (3 * n))(*[v for b, (y1, x1, _, _) in chunk for v in (b, y1, x1)])
    lib.call("mmsa_crop_batch_nchw", img.data_ptr(), img.shape[0], img.shape[1], img.shape[2], img.shape[3], tab, n, out.data_ptr(),
             crop_size[0], crop_size[1], ops._stream())
    return out


MAX_OVERLAP = 8   # windows per pixel mmsa_slide_argmax keeps in registers (csrc/segment.hip)


def _check_overlap(boxes, what):
    """The window grid is a product of row intervals and column intervals, so the largest per-pixel overlap is the product of the
    largest per-row and per-column overlaps: checked on the host BEFORE launching, because the one-pass kernel classifies only pixels
    covered by 1..MAX_OVERLAP windows (others get class 255 and are counted in its `uncovered` word)."""
    def deepest(iv):
        ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv], key=lambda t: (t[0], t[1]))   # half-open: close before open
        cur = best = 0
        for _, d in ev:
            cur += d
            best = max(best, cur)
        return best
    rows = sorted({(y1, y2) for y1, _, y2, _ in boxes})
    cols = sorted({(x1, x2) for _, x1, _, x2 in boxes})
    ov = deepest(rows) * deepest(cols)
    if ov > MAX_OVERLAP:
        raise RuntimeError(f"mmsa.{what}: the window grid covers some pixels {ov} times, the one-pass class-map kernel handles up to {MAX_OVERLAP} "
                           "(use slide_inference + argmax_map, or a larger stride)")


@_on_device
@torch.no_grad()
def slide_class_map(backbone, head, img, crop_size, stride, max_batch=8):
    """`simple_test` of a sliding-window frame (ED:191-234 + ED:449,477) -> uint8 class map [B, H, W], without the
    [B, classes, H, W] logits canvas: every window's logits stay at head resolution and ONE kernel (mmsa_slide_argmax) resizes,
    sums the overlapping windows in window order, divides by the count and takes the argmax -- the same additions in the same order
    as slide_inference + argmax_map, so the same class map bit for bit.  All windows of the frame go through the encoder in
    batches of `max_batch`; with static shapes the whole function is HIP-graph capturable (no host sync inside)."""
    import ctypes
    _check(img)
    img = img.contiguous()
    B, _, H, W = img.shape
    if H < crop_size[0] or W < crop_size[1]:
        raise RuntimeError("mmsa.slide_class_map: the image must be at least as large as the crop")
    boxes = crop_boxes(H, W, crop_size, stride)
    _check_overlap(boxes, "slide_class_map")
    _pair(backbone, head)
    jobs = [(b, box) for box in boxes for b in range(B)]      # the accumulation order of slide_inference
    if len(jobs) > 64:
        raise RuntimeError("mmsa.slide_class_map: at most 64 windows per call")
    lgs = []
    for s in range(0, len(jobs), max_batch):
        chunk = jobs[s:s + max_batch]
        feats, _ = backbone(_crops(img, chunk, crop_size))
        lgs.append(head(feats))
    lg = lgs[0] if len(lgs) == 1 else torch.cat(lgs, 0)
    n = len(jobs)
    tab = (ctypes.c_int * (3 * n))(*[v for b, (y1, x1, _, _) in jobs for v in (b, y1, x1)])
    out = torch.empty(B, H, W, dtype=torch.uint8, device=img.device)
    unc = torch.zeros(1, dtype=torch.int32, device=img.device)
    lib.call("mmsa_slide_argmax", lg.data_ptr(), n, lg.shape[1], lg.shape[2], lg.shape[3], tab, out.data_ptr(), B, H, W,
             crop_size[0], crop_size[1], unc.data_ptr(), ops._stream())
    return out, unc          # unc[0] != 0 <=> some pixel is not covered (ED:220); checked by the caller outside a capture


@_on_device
@torch.no_grad()
def whole_class_map(backbone, head, img):
    """Whole-image `simple_test`: resize x4 (bilinear, align_corners=False) + argmax fused (ED:90-94,449,477) -> uint8 [B, H, W]."""
    import ctypes
    _check(img)
    _pair(backbone, head)
    feats, _ = backbone(img)
    lg = head(feats)
    B, _, H, W = img.shape
    tab = (ctypes.c_int * (3 * B))(*[v for b in range(B) for v in (b, 0, 0)])
    out = torch.empty(B, H, W, dtype=torch.uint8, device=img.device)
    unc = torch.zeros(1, dtype=torch.int32, device=img.device)
    lib.call("mmsa_slide_argmax", lg.data_ptr(), B, lg.shape[1], lg.shape[2], lg.shape[3], tab, out.data_ptr(), B, H, W, H, W,
             unc.data_ptr(), ops._stream())
    return out


@_on_device
@torch.no_grad()
def whole_inference(backbone, head, img):
    """ED: whole-image mode = encode_decode on the full input."""
    return encode_decode(backbone, head, img)


@_on_device
@torch.no_grad()
def whole_inference_dim(backbone, head, img, dim, rescale=True):
    """ED:329-362 -- `test_cfg.mode = 'whole_dim'`, the test mode of every DELIVER config (dim = (1024, 1024)): the logits at input size
    (encode_decode), resized once more (bilinear, align_corners=False) to `dim`.  The reference's method has no result for
    `rescale=False` (it returns None and `inference` fails on it, ED:334-346,448): that call is refused here."""
    if not rescale:
        raise RuntimeError("mmsa.whole_inference_dim: rescale=False has no defined result in the reference (encoder_decoder.py:334-346 "
                           "returns None); use whole_inference or whole_inference_dim_cut")
    y = encode_decode(backbone, head, img)
    if tuple(dim) == tuple(y.shape[2:]):
        return y                      # the second resize is the identity (align_corners=False, same size)
    out = torch.empty(y.shape[0], y.shape[1], dim[0], dim[1], device=y.device)
    _resize_into(y, out, 0, 0, dim[0], dim[1])
    return out


@_on_device
@torch.no_grad()
def whole_inference_dim_cut(backbone, head, img, dim, cut_dim, rescale=True):
    """ED:364-413 -- `test_cfg.mode = 'whole_dim_cut'`, the test mode of every FMB config (rescale=False, dim=(600,800), cut_dim=(800,600)):
    the logits at input size, resized to `dim` when `rescale`, cropped to [:, :, :cut_dim[1], :cut_dim[0]] (a contiguous copy)."""
    y = encode_decode(backbone, head, img)
    if rescale and tuple(dim) != tuple(y.shape[2:]):
        out = torch.empty(y.shape[0], y.shape[1], dim[0], dim[1], device=y.device)
        _resize_into(y, out, 0, 0, dim[0], dim[1])
        y = out
    return y[:, :, :cut_dim[1], :cut_dim[0]].contiguous()


@_on_device
@torch.no_grad()
def inference(backbone, head, img, test_cfg, rescale=True):
    """ED:417-447 dispatch on `test_cfg['mode']` -- 'slide', 'whole', 'whole_dim', 'whole_dim_cut' ('slide_mod_sel' runs the segmentor's
    modality-selection variant, ED:236-308, which needs a backbone with a selection head: not this backbone) -- returning the logits the
    reference softmaxes (ED:448-470; flips are the caller's, as in the reference's test pipeline)."""
    mode = test_cfg["mode"]
    if mode == "slide":
        return slide_inference(backbone, head, img, tuple(test_cfg["crop_size"]), tuple(test_cfg["stride"]))
    if mode == "whole":
        return whole_inference(backbone, head, img)
    if mode == "whole_dim":
        return whole_inference_dim(backbone, head, img, tuple(test_cfg["dim"]), rescale)
    if mode == "whole_dim_cut":
        return whole_inference_dim_cut(backbone, head, img, tuple(test_cfg["dim"]), tuple(test_cfg["cut_dim"]), rescale)
    raise RuntimeError(f"mmsa.inference: test_cfg.mode '{mode}' is not one of slide / whole / whole_dim / whole_dim_cut")


@_on_device
@torch.no_grad()
def argmax_map(seg_logit):
    """ED:449,477: softmax is monotonic, the prediction is the per-pixel argmax over the class axis -> uint8 [B, H, W]."""
    _check(seg_logit)
    seg_logit = seg_logit.contiguous()
    B, C, H, W = seg_logit.shape
    out = torch.empty(B, H, W, dtype=torch.uint8, device=seg_logit.device)
    lib.call("mmsa_argmax_nchw", seg_logit.data_ptr(), out.data_ptr(), B, C, H * W, ops._stream())
    return out


class FrameResult:
    """What SlideRunner.run() returns: the class map of ONE frame, readable once the attention logit guard of its pass has been inspected
    (mmsa.chains.Replay; a pass that ran fp16 attention out of range raises mmsa.chains.AttentionRangeError instead)."""

    def __init__(self, runner, replay):
        self._runner, self._replay = runner, replay

    def outputs(self):
        """(class map uint8 [B, H, W], uncovered-pixel flag) -- verified.  Both are the runner's static buffers: read them before the next run()."""
        self._replay._owner._verify(self._replay.seq)
        return self._runner.out, self._runner.unc

    @property
    def unverified(self):
        return self._runner.out, self._runner.unc


class SlideRunner:
    """Throughput form of slide_class_map for a fixed frame geometry: the frame's windows are cut by one kernel, go through the
    encoder + head as `chains` concurrent sub-batches (mmsa.Chains: one HIP graph per chain, shared packed weights) and one kernel
    (mmsa_slide_argmax) turns the head-resolution logits into the class map.  Same class map as slide_inference + argmax_map, bit
    for bit.  `frame` is the static [B, 6, H, W] buffer the runner reads on every run()."""

    def __init__(self, backbone, head, frame, crop_size, stride, chains=2, check_every=1):
        import ctypes
        from .chains import Chains
        _check(frame)
        _pair(backbone, head)
        self.frame = frame.contiguous()
        B, _, H, W = self.frame.shape
        self.crop_size = tuple(crop_size)
        boxes = crop_boxes(H, W, crop_size, stride)
        _check_overlap(boxes, "SlideRunner")
        self.jobs = [(b, box) for box in boxes for b in range(B)]      # the accumulation order of slide_inference
        n = len(self.jobs)
        if n > 64:
            raise RuntimeError("mmsa.SlideRunner: at most 64 windows per frame batch")
        if n % chains:
            chains = 1
        with torch.cuda.device(self.frame.device):
            self.crops = _crops(self.frame, self.jobs, self.crop_size)          # also the static input buffer of the chains
            self.chains = Chains(backbone, head, n=chains, check_every=check_every).capture(self.crops)
            self.tab = (ctypes.c_int * (3 * n))(*[v for b, (y1, x1, _, _) in self.jobs for v in (b, y1, x1)])
            self.out = torch.empty(B, H, W, dtype=torch.uint8, device=self.frame.device)
            self.unc = torch.zeros(1, dtype=torch.int32, device=self.frame.device)

    @torch.no_grad()
    def run(self):
        """Enqueue one frame on the current stream (asynchronous) -> FrameResult; `.outputs()` = (class map uint8 [B, H, W], uncovered-pixel flag) once the
        attention logit guard of this pass has been inspected.  The inspection costs one 4 * depth-byte copy per `check_every` frames and an event wait,
        no device sync; a frame that scored logits beyond the fp16 range raises mmsa.chains.AttentionRangeError from outputs() -- or from the next run(),
        whichever comes first -- after the blocks concerned have been moved to fp16 hi/lo pairs and the graphs captured again: run that frame again."""
        with torch.cuda.device(self.frame.device):
            _crops(self.frame, self.jobs, self.crop_size, out=self.crops)
            rp = self.chains.replay()
            lg = rp.unverified          # the argmax kernel below is enqueued behind the pass; nothing is read on the host before outputs() verifies it
            B, H, W = self.out.shape
            self.unc.zero_()
            lib.call("mmsa_slide_argmax", lg.data_ptr(), len(self.jobs), lg.shape[1], lg.shape[2], lg.shape[3], self.tab, self.out.data_ptr(),
                     B, H, W, self.crop_size[0], self.crop_size[1], self.unc.data_ptr(), ops._stream())
        return FrameResult(self, rp)

    def check_guard(self):
        """Chains.check_guard for the runner's chains (host sync): [] = the frames since the last check ran inside the attention kernels' operand range;
        otherwise the listed ViT blocks were moved to fp16 hi/lo pairs, the graphs were captured again and run() must be repeated for
        those frames (their FrameResult.outputs() raise)."""
        return self.chains.check_guard()
