"""`roi_pool` / `roi_pool_grad`: the Python face of the RoiPool / RoiPoolGrad ops
(lib/roi_pooling_layer/roi_pooling_op.py:4-7, op registration roi_pooling_op.cc:30-49,
gradient wiring roi_pooling_op_grad.py:7-43).

    top, argmax = roi_pool(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale)
    bottom_diff = roi_pool_grad(bottom_data, bottom_rois, argmax, grad, ph, pw, scale)

bottom_data NHWC f32, bottom_rois (R,5) [batch_idx,x1,y1,x2,y2]; numpy in -> numpy out,
torch device tensors in -> device tensors out (autograd-aware through RoiPoolFunction:
gradient w.r.t. the data only, `None` for the rois like roi_pooling_op_grad.py:43)."""
import torch

from .. import ops


class RoiPoolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale):
        data = bottom_data.contiguous()
        rois = bottom_rois.contiguous()
        top, argmax = ops.roi_pool_forward(data, rois, pooled_height, pooled_width, spatial_scale)
        ctx.save_for_backward(rois, argmax)
        ctx.meta = (tuple(data.shape), pooled_height, pooled_width, spatial_scale)
        ctx.mark_non_differentiable(argmax)
        return top, argmax

    @staticmethod
    def backward(ctx, grad_top, _grad_argmax):
        rois, argmax = ctx.saved_tensors
        shape, ph, pw, scale = ctx.meta
        grad = ops.roi_pool_backward(grad_top.contiguous(), rois, argmax, shape, ph, pw, scale)
        return grad, None, None, None, None


def _check(bottom_data, bottom_rois):
    # OP_REQUIRES at roi_pooling_op.cc:83-88
    if bottom_data.ndim != 4:
        raise ValueError("data must be 4-dimensional")
    if bottom_rois.ndim != 2:
        raise ValueError("rois must be 2-dimensional")


def roi_pool(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale, name=None):
    _check(bottom_data, bottom_rois)
    if isinstance(bottom_data, torch.Tensor):
        return RoiPoolFunction.apply(bottom_data, bottom_rois, int(pooled_height), int(pooled_width),
                                     float(spatial_scale))
    data, rois = ops._dev(bottom_data), ops._dev(bottom_rois)
    top, argmax = ops.roi_pool_forward(data, rois, int(pooled_height), int(pooled_width), float(spatial_scale))
    return top.cpu().numpy(), argmax.cpu().numpy()


def roi_pool_grad(bottom_data, bottom_rois, argmax, grad, pooled_height, pooled_width, spatial_scale, name=None):
    _check(bottom_data, bottom_rois)
    as_numpy = not isinstance(bottom_data, torch.Tensor)
    rois = ops._dev(bottom_rois)
    am = ops._dev(argmax, dtype=torch.int32)
    g = ops._dev(grad)
    out = ops.roi_pool_backward(g, rois, am, tuple(bottom_data.shape), int(pooled_height), int(pooled_width),
                                float(spatial_scale))
    return out.cpu().numpy() if as_numpy else out
