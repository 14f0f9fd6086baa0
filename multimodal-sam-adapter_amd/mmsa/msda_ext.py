"""Stand-in for the reference's pybind11 extension module `MultiScaleDeformableAttention` (segmentation/ops/src/vision.cpp:13-16)
and for its autograd wrapper `MSDeformAttnFunction` (segmentation/ops/functions/ms_deform_attn_func.py:19-50), over the C ABI of
libmmsa_hip.so.  Same function names, argument order and return structure, so

    import mmsa.msda_ext as MSDA          # instead of: import MultiScaleDeformableAttention as MSDA

is the whole reference-side change.  float32 / float16 / float64 like the reference's dispatch."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    with torch.cuda.device(value.device):
        return ops.msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    with torch.cuda.device(value.device):
        return ops.msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output.contiguous(), im2col_step)


class MSDeformAttnFunction(Function):
    """ms_deform_attn_func.py:19-50: forward saves its inputs, backward returns (grad_value, None, None, grad_sampling_loc,
    grad_attn_weight, None)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        attention_weights = attention_weights.type_as(value)        # ms_deform_attn_func.py:26-27
        sampling_locations = sampling_locations.type_as(value)
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                                        ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value, shapes, lsi, loc, aw, grad_output, ctx.im2col_step)
        return gv, None, None, gl, ga, None
