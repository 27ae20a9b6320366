"""The fusion head of the training graph (lib/networks/MV3D_train.py:159-182: per view pool_5* -> fc6 -> dropout -> fc7 -> dropout,
concat, cls_score / bbox_pred) as ONE autograd function.

Run op by op under autocast the head is ~45 forward and ~60 backward launches per step, each behind a Python / autograd dispatch the
device outruns (profiles/r05_train_tail_bf16.txt: "torch -> torch" gaps 104 per step, 2.6 ms of idle).  Here the same arithmetic is
issued back to back with no graph nodes in between:
  * the views' pooled maps go to ONE (views, rows, 25088) matrix in the GEMM type ((c, h, w) order: the reference's flattening of
    the NCHW blob, network.py:373-377), the master weights of the step to stacked (views, out, in) buffers by one multi-tensor copy;
  * fc6 / fc7 of all views are ONE strided-batched GEMM each (bias through the GEMM's beta operand), ReLU in place, dropout by
    torch.native_dropout (output + mask in one kernel; ReLU and dropout commute, the mask scale being positive);
  * cls_score and bbox_pred are ONE GEMM on the concatenated towers (weights stacked to (2 + 48, views * 2048));
  * backward: the same GEMMs transposed, one masked elementwise per layer; the weight-gradient GEMMs write f32 directly where the
    installed torch offers `out_dtype` (else their results are cast).
The GEMMs themselves are the vendor library's (a dense contraction: hipBLASLt behind torch.baddbmm / addmm); the results equal the
op-by-op graph up to GEMM batching (tests/test_train_entry.py::test_fused_head_equals_the_op_by_op_head)."""
import torch

_F32_OUT = [None]         # does this torch take bmm / mm(..., out_dtype=torch.float32) on the device?  (probed once)


def _mm_f32(a, b, batched):
    """a @ b with 16-bit operands and an f32 RESULT straight from the GEMM (aten::bmm.dtype / mm.dtype: the weight gradients then need
    neither a 16-bit intermediate nor a cast launch); falls back to the product in the operands' type, cast afterwards."""
    if a.dtype != torch.float32 and a.is_cuda and _F32_OUT[0] is not False:
        try:
            out = (torch.bmm if batched else torch.mm)(a, b, out_dtype=torch.float32)
            _F32_OUT[0] = True
            return out
        except (TypeError, RuntimeError, NotImplementedError):
            _F32_OUT[0] = False
    return (torch.bmm if batched else torch.mm)(a, b).float()


class FusedHead(torch.autograd.Function):
    """apply(keep_prob, gemm_dtype, n_views, *tensors) with tensors = pools (n_views x (R,7,7,C) f32), then per view w6, b6, w7, b7
    (fp32 masters, reference layout (out, in) with in = (c, h, w)), then w_cls, b_cls, w_box, b_box.
    Returns (cls_score f32 (R,2), bbox_pred f32 (R,48), fc7 towers (n_views, R, 2048) in gemm_dtype)."""

    @staticmethod
    def forward(ctx, keep_prob, dt, V, held, *ts):
        pools, wts = ts[:V], ts[V:]
        R = pools[0].shape[0]
        dev = pools[0].device
        K = pools[0][0].numel()
        w6 = [wts[4 * v] for v in range(V)]; b6 = [wts[4 * v + 1] for v in range(V)]
        w7 = [wts[4 * v + 2] for v in range(V)]; b7 = [wts[4 * v + 3] for v in range(V)]
        wc, bc, wb, bb = wts[4 * V:4 * V + 4]
        N6, N7 = w6[0].shape[0], w7[0].shape[0]
        nc, nb = wc.shape[0], wb.shape[0]
        # ---- the step's copies of the master weights in the GEMM type, stacked per layer.  `held` = (buffers, current): the network keeps
        # the stacked buffers from step to step and the optimizer's launch writes the updated weights into them (optim.Adam.register_lowp);
        # when they are current nothing is cast here, else one multi-tensor launch fills them
        if held is not None:
            (W6, B6, W7, B7, WH, BH), current = held
        else:
            W6 = torch.empty((V, N6, K), dtype=dt, device=dev); B6 = torch.empty((V, 1, N6), dtype=dt, device=dev)
            W7 = torch.empty((V, N7, N6), dtype=dt, device=dev); B7 = torch.empty((V, 1, N7), dtype=dt, device=dev)
            WH = torch.empty((nc + nb, V * N7), dtype=dt, device=dev); BH = torch.empty((nc + nb,), dtype=dt, device=dev)
            current = False
        if not current:
            dst = [W6[v] for v in range(V)] + [B6[v, 0] for v in range(V)] + [W7[v] for v in range(V)] + [B7[v, 0] for v in range(V)] + [WH[:nc], WH[nc:], BH[:nc], BH[nc:]]
            src = [t.detach() for t in (w6 + b6 + w7 + b7 + [wc, wb, bc, bb])]
            torch._foreach_copy_(dst, src)
        # ---- the pooled maps, (c, h, w)-flattened, in the GEMM type
        X = torch.empty((V, R, K), dtype=dt, device=dev)
        for v in range(V):
            p = pools[v]
            X[v].view(R, p.shape[3], p.shape[1], p.shape[2]).copy_(p.detach().permute(0, 3, 1, 2))
        train = keep_prob < 1.0
        H6 = torch.baddbmm(B6, X, W6.transpose(1, 2))                  # (V, R, N6), bias through beta
        H6.relu_()
        D6, M6 = torch.native_dropout(H6, 1.0 - keep_prob, train) if train else (H6, None)
        H7 = torch.baddbmm(B7, D6, W7.transpose(1, 2))
        H7.relu_()
        D7, M7 = torch.native_dropout(H7, 1.0 - keep_prob, train) if train else (H7, None)
        F_ = D7.permute(1, 0, 2).reshape(R, V * N7)                    # concat of the towers (MV3D_train.py:175)
        out = torch.addmm(BH, F_, WH.t()).float()
        ctx.save_for_backward(X, W6, W7, WH, H6, H7, D6, M6, M7, F_)
        ctx.meta = (keep_prob, dt, V, R, K, N6, N7, nc, nb, [tuple(p.shape) for p in pools], [t.requires_grad for t in ts])
        ctx.mark_non_differentiable(D7)
        return out[:, :nc].contiguous(), out[:, nc:].contiguous(), D7

    @staticmethod
    def backward(ctx, g_cls, g_box, _g_tower):
        X, W6, W7, WH, H6, H7, D6, M6, M7, F_ = ctx.saved_tensors
        keep, dt, V, R, K, N6, N7, nc, nb, pshapes, needs = ctx.meta
        dev = X.device
        gout = torch.empty((R, nc + nb), dtype=dt, device=dev)
        if g_cls is not None:
            gout[:, :nc].copy_(g_cls)
        else:
            gout[:, :nc].zero_()
        if g_box is not None:
            gout[:, nc:].copy_(g_box)
        else:
            gout[:, nc:].zero_()
        gWH = _mm_f32(gout.t(), F_, False)                             # (nc + nb, V * N7), f32
        gBH = gout.float().sum(0)
        gF = gout.mm(WH).view(R, V, N7).permute(1, 0, 2).contiguous()  # -> (V, R, N7)
        scale = 1.0 / keep

        def through(g, H, M):                                          # dropout (mask, scale) and ReLU (H > 0: H is the ReLU output) backward
            if M is not None:
                g = torch.ops.aten.native_dropout_backward(g, M, scale)
            return torch.ops.aten.threshold_backward(g, H, 0.0)
        gH7 = through(gF, H7, M7)
        gW7 = _mm_f32(gH7.transpose(1, 2), D6, True)                   # (V, N7, N6), f32
        gB7 = gH7.float().sum(1)
        gH6 = through(torch.bmm(gH7, W7), H6, M6)
        gW6 = _mm_f32(gH6.transpose(1, 2), X, True)                    # (V, N6, K), f32: the masters' gradients as the GEMM writes them
        gB6 = gH6.float().sum(1)
        gpools = [None] * V
        if any(needs[:V]):
            gX = torch.bmm(gH6, W6)                                    # (V, R, K)
            for v in range(V):
                if needs[v]:
                    R_, h, w, c = pshapes[v]
                    gp = torch.empty(pshapes[v], dtype=torch.float32, device=dev)
                    gp.permute(0, 3, 1, 2).copy_(gX[v].view(R_, c, h, w))
                    gpools[v] = gp
        # ---- the weight gradients are f32 already (views of the batched results)
        dsts = [gW6[v] for v in range(V)] + [gW7[v] for v in range(V)] + [gWH[:nc], gWH[nc:]]
        grads = []
        for v in range(V):
            grads += [dsts[v], gB6[v], dsts[V + v], gB7[v]]
        grads += [dsts[2 * V], gBH[:nc].contiguous(), dsts[2 * V + 1], gBH[nc:].contiguous()]
        return (None, None, None, None) + tuple(gpools) + tuple(grads)


def head_buffers(params, names6, names7, dt, device):
    """the stacked 16-bit weight buffers of the fused head for a network to keep, and which piece of them holds which master parameter:
    ((W6, B6, W7, B7, WH, BH), [(param, view of its copy), ...])"""
    V = len(names6)
    N6, K = params[names6[0]][0].shape
    N7 = params[names7[0]][0].shape[0]
    nc, nb = params["cls_score"][0].shape[0], params["bbox_pred"][0].shape[0]
    W6 = torch.empty((V, N6, K), dtype=dt, device=device); B6 = torch.empty((V, 1, N6), dtype=dt, device=device)
    W7 = torch.empty((V, N7, N6), dtype=dt, device=device); B7 = torch.empty((V, 1, N7), dtype=dt, device=device)
    WH = torch.empty((nc + nb, V * N7), dtype=dt, device=device); BH = torch.empty((nc + nb,), dtype=dt, device=device)
    pieces = []
    for v, (n6, n7) in enumerate(zip(names6, names7)):
        pieces += [(params[n6][0], W6[v]), (params[n6][1], B6[v, 0]), (params[n7][0], W7[v]), (params[n7][1], B7[v, 0])]
    pieces += [(params["cls_score"][0], WH[:nc]), (params["bbox_pred"][0], WH[nc:]), (params["cls_score"][1], BH[:nc]), (params["bbox_pred"][1], BH[nc:])]
    return (W6, B6, W7, B7, WH, BH), pieces


def fused_head(pools, params, names6, names7, keep_prob, gemm_dtype, held=None):
    """pools: the views' pooled maps; params: {layer: [w, b]} fp32 masters; names6 / names7: the views' fc6 / fc7 layer names; held:
    None, or (head_buffers(...)[0], are the copies current?).  Returns (cls_score, bbox_pred, [fc7 tower per view])."""
    V = len(pools)
    ts = list(pools)
    for n6, n7 in zip(names6, names7):
        ts += [params[n6][0], params[n6][1], params[n7][0], params[n7][1]]
    ts += [params["cls_score"][0], params["cls_score"][1], params["bbox_pred"][0], params["bbox_pred"][1]]
    cls, box, towers = FusedHead.apply(float(keep_prob), gemm_dtype, V, held, *ts)
    return cls, box, [towers[v] for v in range(V)]


class RpnHeads(torch.autograd.Function):
    """The two 1 x 1 RPN heads (rpn_cls_score 512 -> 8, rpn_bbox_pred 512 -> 24; lib/networks/MV3D_train.py:88-97) as ONE GEMM on the
    rpn_conv/3x3 map with the filters stacked to (32, 512): apply(gemm_dtype, rpn (B,H,W,512) f32, w_cls, b_cls, w_box, b_box) ->
    (score (B,H,W,8) f32, pred (B,H,W,24) f32).  Six launches forward, seven backward instead of ~9 / ~12 under autocast."""

    @staticmethod
    def forward(ctx, dt, rpn, wc, bc, wb, bb):
        B, H, W, Cin = rpn.shape
        nc, nb = wc.shape[0], wb.shape[0]
        dev = rpn.device
        Wt = torch.empty((nc + nb, Cin), dtype=dt, device=dev)
        Bv = torch.empty((nc + nb,), dtype=dt, device=dev)
        torch._foreach_copy_([Wt[:nc], Wt[nc:], Bv[:nc], Bv[nc:]], [wc.detach().reshape(nc, Cin), wb.detach().reshape(nb, Cin), bc.detach(), bb.detach()])
        X = rpn.detach().reshape(-1, Cin).to(dt)
        out = torch.addmm(Bv, X, Wt.t()).float()
        ctx.save_for_backward(X, Wt)
        ctx.meta = (dt, (B, H, W, Cin), nc, nb, tuple(wc.shape), tuple(wb.shape))
        return out[:, :nc].reshape(B, H, W, nc).contiguous(), out[:, nc:].reshape(B, H, W, nb).contiguous()

    @staticmethod
    def backward(ctx, g_score, g_pred):
        X, Wt = ctx.saved_tensors
        dt, (B, H, W, Cin), nc, nb, wcs, wbs = ctx.meta
        dev = X.device
        g = torch.empty((X.shape[0], nc + nb), dtype=dt, device=dev)
        if g_score is not None:
            g[:, :nc].copy_(g_score.reshape(-1, nc))
        else:
            g[:, :nc].zero_()
        if g_pred is not None:
            g[:, nc:].copy_(g_pred.reshape(-1, nb))
        else:
            g[:, nc:].zero_()
        gW = _mm_f32(g.t(), X, False)
        gB = g.float().sum(0)
        gX = g.mm(Wt).float().view(B, H, W, Cin) if ctx.needs_input_grad[1] else None
        return None, gX, gW[:nc].reshape(wcs), gB[:nc].contiguous(), gW[nc:].reshape(wbs), gB[nc:].contiguous()
