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


class RoiPoolViewsFunction(torch.autograd.Function):
    """The RoiPool layers of one step (`pool_5`, `pool_5_2` [, `pool_5_3`]: network.py:199-213 called once per view) as the library's
    PAIR: ONE launch forward (mv3d_roi_pool_forward_views_pair: every view, the argmax plane kept as private one-byte codes) and one
    call backward (mv3d_roi_pool_backward_views_pair without a workspace: ONE launch, every view's map tiles in LDS, the ordered
    sums of roi_pooling_op.cc:319-452).  apply(ph, pw, scale, data_0, rois_0, data_1, rois_1, ...) -> (top_0, top_1, ...);
    gradients for the data tensors only (roi_pooling_op_grad.py:43).  A view whose output gets no gradient (unused in the loss)
    contributes zeros, like an unconnected tf.gradients branch."""

    @staticmethod
    def forward(ctx, pooled_height, pooled_width, spatial_scale, *tensors):
        datas = [t.contiguous() for t in tensors[0::2]]
        rois = [t.contiguous() for t in tensors[1::2]]
        res = ops.roi_pool_forward_views_pair([(d, r, spatial_scale) for d, r in zip(datas, rois)], pooled_height, pooled_width)
        ctx.save_for_backward(*rois, *[am for _, am in res])
        ctx.meta = ([tuple(d.shape) for d in datas], pooled_height, pooled_width, spatial_scale)
        return tuple(top for top, _ in res)

    @staticmethod
    def backward(ctx, *grads):
        shapes, ph, pw, scale = ctx.meta
        n = len(shapes)
        saved = ctx.saved_tensors
        rois, argmax = saved[:n], saved[n:]
        views = []
        for k in range(n):
            g = grads[k]
            if g is None:
                g = torch.zeros((rois[k].shape[0], ph, pw, shapes[k][3]), dtype=torch.float32, device=rois[k].device)
            views.append((g.contiguous(), rois[k], argmax[k], shapes[k], scale))
        outs = ops.roi_pool_backward_views_pair(views, ph, pw, workspace=False)   # (one launch, no scratch memory)
        ret = [None, None, None]
        for k in range(n):
            ret += [outs[k], None]
        return tuple(ret)


def roi_pool_views(views, pooled_height, pooled_width, spatial_scale, top_dtype=None):
    """views: [(bottom_data NHWC device tensor, bottom_rois (R,5) device tensor), ...] -> [top, ...] (autograd-aware).
    top_dtype (inference only, torch.float16 / bfloat16, maps of 256 / 512 / 1024 channels): the pooled maps in that type -- what the
    16-bit head would cast them to anyway -- written by the pooling launch itself."""
    flat = []
    for d, r in views:
        _check(d, r)
        flat += [d, r]
    if not (torch.is_grad_enabled() and any(d.requires_grad for d, _ in views)):
        # inference: no gradient will be asked for, so the argmax planes (as many bytes again as the pooled output) are not written
        half = top_dtype if top_dtype in (torch.float16, torch.bfloat16) and all(d.shape[3] in (256, 512, 1024) for d, _ in views) else None
        res = ops.roi_pool_forward_views([(d.contiguous(), r.contiguous(), float(spatial_scale)) for d, r in views], int(pooled_height),
                                         int(pooled_width), want_argmax=False, top_dtype=half)
        return [top for top, _ in res]
    return list(RoiPoolViewsFunction.apply(int(pooled_height), int(pooled_width), float(spatial_scale), *flat))


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
