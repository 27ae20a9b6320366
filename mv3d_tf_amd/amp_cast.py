"""Reduced-precision copies of the dense head's fp32 master parameters for ONE training step, made by one multi-tensor launch.

Under `torch.autocast` every `F.linear` of the fusion head (fc6_x / fc7_x / cls_score / bbox_pred of lib/networks/MV3D_train.py:
108-136) and of the 1x1 RPN heads (:88-97) casts its weight and its bias on its own, and autograd casts every gradient back on its
own: ~20 + ~20 short launches per step with a host-bound gap behind each (profiles/r04_train_tail_bf16.txt: 10.8 us x 28).  Here
the copies of all of them come from one `torch._foreach_copy_` (a multi-tensor kernel), and the gradients go back to fp32 through
one more.  Plumbing around torch's GEMMs -- no kernel of this library is involved."""
import torch


class CastMany(torch.autograd.Function):
    """apply(dtype, *fp32 parameters) -> the same tensors in `dtype`; backward: the gradients in fp32 (one launch each way)"""

    @staticmethod
    def forward(ctx, dtype, *params):
        outs = [torch.empty_like(p, dtype=dtype) for p in params]
        torch._foreach_copy_(outs, [p.detach() for p in params])
        ctx.src = [(p.dtype, p.shape) for p in params]
        ctx.set_materialize_grads(False)          # a copy nobody used keeps grad None (no zero tensor per parameter)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        have = [k for k, g in enumerate(grads) if g is not None]
        res = [None] * len(grads)
        if have:
            dst = [torch.empty(ctx.src[k][1], dtype=ctx.src[k][0], device=grads[k].device) for k in have]
            torch._foreach_copy_(dst, [grads[k].contiguous() for k in have])
            for k, t in zip(have, dst):
                res[k] = t
        return (None,) + tuple(res)


def cast_params(params, names, dtype):
    """{name: [w, b]} fp32 -> {name: (w, b) in dtype} for `names`, differentiable w.r.t. the fp32 tensors"""
    flat = []
    for n in names:
        flat += list(params[n])
    outs = CastMany.apply(dtype, *flat)
    return {n: (outs[2 * k], outs[2 * k + 1]) for k, n in enumerate(names)}
