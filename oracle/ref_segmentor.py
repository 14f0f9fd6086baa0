"""CPU restatement of the reference segmentor's inference glue -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows segmentation/mmseg_custom/models/segmentors/encoder_decoder.py: `encode_decode` (:85-95), `slide_inference` (:191-234,
without the optional rescale to `ori_shape`), `whole_inference_dim` (:329-362), `whole_inference_dim_cut` (:364-413) and the argmax of
`simple_test` (:449,477).  Pinned by tests/golden/slide.npz, which
tools/oracle/make_golden.py produces by calling the reference's own, unmodified `EncoderDecoder.slide_inference` with a stand-in
`self` whose `encode_decode` is a fixed seeded function (the window grid / pad / count / average logic is what is pinned), and by
tests/golden/whole_dim.npz (the reference's own `whole_inference_dim` / `whole_inference_dim_cut`, same harness)."""
import torch
import torch.nn.functional as F


def encode_decode(backbone, head, img, align_corners=False):
    feats, _ = backbone(img)
    return F.interpolate(head(feats), size=img.shape[2:], mode="bilinear", align_corners=align_corners)


def slide_inference(encode_decode_fn, img, crop_size, stride, num_classes):
    h_stride, w_stride = stride
    h_crop, w_crop = crop_size
    B, _, h_img, w_img = img.shape
    h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
    w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
    preds = img.new_zeros((B, num_classes, h_img, w_img))
    count = img.new_zeros((B, 1, h_img, w_img))
    for h_idx in range(h_grids):
        for w_idx in range(w_grids):
            y1, x1 = h_idx * h_stride, w_idx * w_stride
            y2, x2 = min(y1 + h_crop, h_img), min(x1 + w_crop, w_img)
            y1, x1 = max(y2 - h_crop, 0), max(x2 - w_crop, 0)
            logit = encode_decode_fn(img[:, :, y1:y2, x1:x2])
            preds += F.pad(logit, (x1, w_img - x2, y1, h_img - y2))
            count[:, :, y1:y2, x1:x2] += 1
    assert (count == 0).sum() == 0
    return preds / count


def whole_inference_dim(encode_decode_fn, img, dim, rescale=True, align_corners=False):
    """ED:329-362 behind this backbone (its second return value makes `encode_decode_test` return a tuple, ED:96-107): the logits at
    input size, resized to `dim` when `rescale`.  Without `rescale` the reference's method returns None (it falls off its end,
    ED:334-346): there is no result to restate, and the caller (`inference`, ED:448) fails on it."""
    if not rescale:
        return None
    return F.interpolate(encode_decode_fn(img), size=dim, mode="bilinear", align_corners=align_corners)


def whole_inference_dim_cut(encode_decode_fn, img, dim, cut_dim, rescale=True, align_corners=False):
    """ED:364-413: as above, then the crop [:, :, :cut_dim[1], :cut_dim[0]]; without `rescale` the logits at input size are cropped."""
    y = encode_decode_fn(img)
    if rescale:
        y = F.interpolate(y, size=dim, mode="bilinear", align_corners=align_corners)
    return y[:, :, :cut_dim[1], :cut_dim[0]]
