"""Training entry points with the reference's names and arguments (lib/fast_rcnn/train_mv.py): `train_net` (:373),
`SolverWrapper` with `snapshot` (:49-65) and `train_model` (:87-219), `get_training_roidb` (:315), `get_data_layer` (:333),
`filter_roidb` (:345); plus the loss side: `modified_smooth_l1` (:74-90) and the four losses of `train_model` (:92-130) as
autograd functions over the fused device kernels (csrc/losses.hip), the snapshot file name and the `.npy` weight-dict format
that `network.load` reads (network.py:45-64).

The loop is the reference's: Adam(lr = 1e-5) on cross_entropy + loss_box + rpn_cross_entropy + rpn_loss_box, one frame per
iteration from `RoIDataLayer`, the `iter: ... / speed: ...` lines every cfg.TRAIN.DISPLAY iterations, a snapshot every
cfg.TRAIN.SNAPSHOT_ITERS and at the end.  `sess` / `saver` are opaque here (no TensorFlow): pass None.

Data parallel (BASELINE configs[3]): when torch.distributed is initialised (one process per GPU, backend
"nccl" = RCCL), every rank trains on its shard of the roidb (frame r, r + W, ...) and the gradients are averaged by
`sharding.GradBucketer` (25 MB buckets, last layer first, overlapping backward) before the optimiser step; rank 0 writes
the snapshots and prints.  `frames_per_step` > 1 = that many frames per rank and step, run as ONE batch through the graph (per-GPU batch)."""
import os
import time

import numpy as np
import torch

from .. import ops
from .config import cfg


class _RpnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_score, labels, bbox_pred, bbox_targets, sigma):
        losses, d_cls, d_pred = ops.rpn_loss(cls_score.contiguous(), labels.contiguous(), bbox_pred.contiguous(),
                                             bbox_targets.contiguous(), sigma)
        ctx.save_for_backward(d_cls, d_pred)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_ce, g_box):
        d_cls, d_pred = ctx.saved_tensors
        return d_cls * g_ce, None, d_pred * g_box, None, None


class _RcnnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_score, labels, bbox_pred, bbox_targets, sigma):
        losses, d_cls, d_pred = ops.rcnn_loss(cls_score.contiguous(), labels.contiguous(), bbox_pred.contiguous(),
                                              bbox_targets.contiguous(), sigma)
        ctx.save_for_backward(d_cls, d_pred)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_ce, g_box):
        d_cls, d_pred = ctx.saved_tensors
        return d_cls * g_ce, None, d_pred * g_box, None, None


def modified_smooth_l1(sigma, bbox_pred, bbox_targets):
    """Element-wise form of train_mv.py:74-90 (torch expression of the same formula; the fused kernels are used by
    the losses below)."""
    sigma2 = sigma * sigma
    diffs = bbox_pred - bbox_targets
    sign = (diffs.abs() < 1.0 / sigma2).to(diffs.dtype)
    return (diffs * diffs) * (0.5 * sigma2) * sign + (diffs.abs() - 0.5 / sigma2) * (sign - 1.0).abs()


def rpn_losses(rpn_cls_score_reshape, rpn_data, rpn_bbox_pred, sigma=3.0):
    """train_mv.py:92-113: (rpn_cross_entropy, rpn_loss_box) from the `rpn_cls_score_reshape` output (..., 2), the
    `rpn_data` tuple (labels, bbox_targets, ...) of anchor_target_layer and `rpn_bbox_pred`."""
    cls = rpn_cls_score_reshape.reshape(-1, 2)
    labels = torch.as_tensor(rpn_data[0], dtype=torch.float32, device=cls.device).reshape(-1)
    targets = torch.as_tensor(rpn_data[1], dtype=torch.float32, device=cls.device).reshape(-1, 6)
    return _RpnLoss.apply(cls, labels, rpn_bbox_pred.reshape(-1, 6), targets, float(sigma))


def rcnn_losses(cls_score, roi_data_3d, bbox_pred, sigma=3.0):
    """train_mv.py:115-127: (cross_entropy, loss_box) from `cls_score`, the `roi_data_3d` tuple (.., labels, targets, ..)
    of proposal_target_layer_3d and `bbox_pred`."""
    labels = torch.as_tensor(roi_data_3d[2], device=cls_score.device).reshape(-1).to(torch.int32)
    targets = torch.as_tensor(roi_data_3d[3], dtype=torch.float32, device=cls_score.device)
    return _RcnnLoss.apply(cls_score, labels, bbox_pred, targets, float(sigma))


def total_loss(net_layers, sigma=3.0):
    """train_mv.py:130: cross_entropy + loss_box + rpn_cross_entropy + rpn_loss_box from the train graph's layers.  For a batch
    of B frames every loss is the mean over the frames of the reference's per-frame loss (what B single-frame steps with
    averaged gradients give, and what the data-parallel all-reduce averages across ranks)."""
    L = net_layers
    rpn_data = L['rpn_data']
    if rpn_data[0].dim() == 1:                                  # one frame: the reference's shapes
        rpn_ce, rpn_box = rpn_losses(L['rpn_cls_score_reshape'], rpn_data, L['rpn_bbox_pred'], sigma)
        ce, box = rcnn_losses(L['cls_score'], L['roi_data_3d'], L['bbox_pred'], sigma)
        return ce + box + rpn_ce + rpn_box, (ce, box, rpn_ce, rpn_box)
    B = rpn_data[0].shape[0]
    rows = L['roi_rows']
    vals = []
    o = 0
    for b in range(B):
        r_ce, r_box = rpn_losses(L['rpn_cls_score_reshape'][b], (rpn_data[0][b], rpn_data[1][b]), L['rpn_bbox_pred'][b], sigma)
        sl = slice(o, o + rows[b])
        o += rows[b]
        data = L['roi_data_3d']
        ce, box = rcnn_losses(L['cls_score'][sl], (None, None, data[2][sl], data[3][sl]), L['bbox_pred'][sl], sigma)
        vals += [ce, box, r_ce, r_box]
    # the frames' means with ONE stack + ONE reduction (4 B scalars added and divided one by one were 4 B + 4 B launches each way)
    parts = torch.stack(vals).view(B, 4).mean(0)
    return parts.sum(), tuple(parts.unbind(0))


def stack_blobs(frames):
    """blobs of several frames (RoIDataLayer.forward() each) -> the feed of one batched pass: images / BEV maps stacked on
    axis 0 (same size per batch, as KITTI crops are), im_info (B,3), calib (B,4,12), ground truth as per-frame lists."""
    if len(frames) == 1:
        return dict(frames[0])
    feed = {}
    for k in ("image_data", "lidar_bv_data", "lidar_fv_data"):
        if k in frames[0]:
            feed[k] = np.concatenate([np.asarray(f[k]) for f in frames], 0)
    feed["im_info"] = np.concatenate([np.asarray(f["im_info"], np.float32).reshape(1, 3) for f in frames], 0)
    feed["calib"] = np.stack([np.asarray(f["calib"], np.float32).reshape(4, 12) for f in frames], 0)
    for k in ("gt_boxes", "gt_boxes_bv", "gt_boxes_3d", "gt_boxes_corners"):
        if k in frames[0]:
            feed[k] = [np.asarray(f[k], np.float32) for f in frames]
    return feed


def group_frames_by_shape(frames):
    """frames of a step -> sub-batches whose images / maps have ONE size each (order kept).  KITTI images come in several
    sizes (375 x 1242, 370 x 1224, 374 x 1238, 376 x 1241) and the reference's data layer neither resizes nor crops
    (lib/roi_data_layer/minibatch_mv3d.py:17-76), so the frames of a step can only be stacked when their sizes agree."""
    groups, index = [], {}
    for f in frames:
        key = tuple(tuple(np.asarray(f[k]).shape) for k in ("image_data", "lidar_bv_data", "lidar_fv_data") if k in f)
        if key not in index:
            index[key] = len(groups)
            groups.append([])
        groups[index[key]].append(f)
    return groups


def snapshot_filename(output_dir, iter):
    """train_mv.py:56-61: <output_dir>/<SNAPSHOT_PREFIX>[_<SNAPSHOT_INFIX>]_iter_<iter+1>.ckpt"""
    infix = ('_' + cfg.TRAIN.SNAPSHOT_INFIX if cfg.TRAIN.SNAPSHOT_INFIX != '' else '')
    return os.path.join(output_dir, cfg.TRAIN.SNAPSHOT_PREFIX + infix + '_iter_{:d}'.format(iter + 1) + '.ckpt')


def save_weights_npy(net, path):
    """The `.npy` dict {layer: {'weights': ..., 'biases': ...}} that network.load() reads (network.py:45-64; the
    reference writes the same structure at test_mv.py:345-372): weights in the TF layouts (HWIO / [in, out]), like
    SolverWrapper.snapshot -- save -> load is the identity."""
    d = {name: {'weights': _tf_layout(w), 'biases': b.detach().cpu().numpy()} for name, (w, b) in net.params.items()}
    np.save(path, d, allow_pickle=True)
    return path


# ---------------------------------------------------------------------------------------------------- training loop
def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class SolverWrapper(object):
    """lib/fast_rcnn/train_mv.py:26-219.  `sess` and `saver` are accepted for signature compatibility and ignored."""

    LEARNING_RATE = 0.00001                                   # train_mv.py:143

    def __init__(self, sess, saver, network, imdb, roidb, output_dir, pretrained_model=None):
        self.net, self.imdb, self.roidb = network, imdb, roidb
        self.output_dir, self.pretrained_model = output_dir, pretrained_model
        self.saver = saver
        self.optimizer = None
        self.log = print

    def snapshot(self, sess, iter):
        """<output_dir>/<SNAPSHOT_PREFIX>[_<INFIX>]_iter_<iter+1>.ckpt (:49-65).  The file holds the `.npy` weight dict
        that `network.load` reads; the optimiser state goes next to it (<name>.optim.pt) so that training can resume."""
        if not os.path.exists(self.output_dir):
            os.makedirs(self.output_dir)
        filename = snapshot_filename(self.output_dir, iter)
        with open(filename, 'wb') as f:
            np.save(f, {name: {'weights': _tf_layout(w), 'biases': b.detach().cpu().numpy()}
                        for name, (w, b) in self.net.params.items()}, allow_pickle=True)
        if self.optimizer is not None:
            torch.save({'iter': iter + 1, 'optimizer': self.optimizer.state_dict()}, filename + '.optim.pt')
        self.log('Wrote snapshot to: {:s}'.format(filename))
        return filename

    def train_model(self, sess, max_iters, frames_per_step=1, start_iter=0, resume=None):
        """Network training loop (:87-219).  Returns the list of per-iteration loss tuples
        (total, rpn_loss_cls, rpn_loss_box, loss_cls, loss_box).  `resume` = a snapshot written by snapshot(): its weights are
        loaded, and -- when the `<snapshot>.optim.pt` written next to it exists -- the Adam state and the iteration counter, so
        that training continues where it stopped (the reference can only restart from weights)."""
        from .. import sharding
        dist = _dist()
        rank = dist.get_rank() if dist is not None else 0
        world = dist.get_world_size() if dist is not None else 1
        shard = list(sharding.frame_shard(len(self.roidb), rank, world)) or [rank % len(self.roidb)]   # (fewer frames than ranks:
        roidb = self.roidb if world == 1 else [self.roidb[i] for i in shard]        # wrap around -- no rank may sit out a collective)
        data_layer = get_data_layer(roidb, self.imdb.num_classes)
        if self.pretrained_model is not None:
            self.log('Loading pretrained model weights from {:s}'.format(self.pretrained_model))
            self.net.load(self.pretrained_model, sess, self.saver, True)
        if cfg.TRAIN.get("MIXED_PRECISION", False) and hasattr(self.net, "mfma_trunk"):
            self.net.mfma_trunk, self.net.amp_dtype = True, torch.bfloat16
            self.log('Mixed precision: bf16 MFMA trunks, fp32 master weights')
        elif cfg.TRAIN.get("MFMA_TRUNK", False) and hasattr(self.net, "mfma_trunk"):
            self.net.mfma_trunk, self.net.amp_dtype = True, None
            self.log('fp32 trunks on the exact-f32 MFMA kernels')
        params = self.net.parameters()
        if dist is not None:                                   # identical replicas: rank 0's weights everywhere
            for p_ in params:
                dist.broadcast(p_.data, src=0)
        lr = self.LEARNING_RATE
        # tf.train.AdamOptimizer(lr) defaults: beta 0.9 / 0.999, eps 1e-8.  fused: ONE launch over all parameters on the device
        # (the foreach form is seven launches and 4 ms of a step for the 214 M parameters of the 3-view graph); same update rule
        from ..optim import Adam                                      # (torch.optim.Adam whose step is ONE launch of mv3d_adam_step)
        self.optimizer = Adam(params, lr=lr) if all(p.is_cuda for p in params) else torch.optim.Adam(params, lr=lr)
        if hasattr(self.net, "attach_optimizer"):
            self.net.attach_optimizer(self.optimizer)
        if resume is not None:
            self.net.load(resume, sess, self.saver, False)
            if os.path.exists(resume + '.optim.pt'):
                state = torch.load(resume + '.optim.pt', map_location=params[0].device)
                self.optimizer.load_state_dict(state['optimizer'])
                start_iter = int(state['iter'])
            self.log('Resumed from {:s} at iteration {:d}'.format(resume, start_iter))
        bucketer = sharding.GradBucketer(params, dist)
        history, last_snapshot_iter, spent = [], -1, 0.0
        it = start_iter - 1
        for it in range(start_iter, max_iters):
            t0 = time.perf_counter()
            bucketer.zero_grad()
            # the frames of a step go through the graph as ONE batch (the hot-path kernels take the frame as blockIdx.y); the
            # four losses are means over the batch's anchors / ROIs, as train_mv.py:92-130 has them for its single frame
            # Frames of different image sizes cannot be stacked: they go through as one sub-batch per size, their gradients
            # accumulated with the weight (frames of the sub-batch / frames of the step) -- the step's loss is the same mean
            # over its frames either way; the all-reduce starts with the LAST sub-batch's backward pass.
            groups = group_frames_by_shape([data_layer.forward() for _ in range(frames_per_step)])   # get one batch (:162), x frames
            vals = np.zeros(4)
            for gi, grp in enumerate(groups):
                feed = stack_blobs(grp)
                feed["keep_prob"] = 0.5                                                   # feed_dict (:165-173)
                layers = self.net.forward(feed)
                loss, parts = total_loss(layers)
                share = len(grp) / float(frames_per_step)
                bucketer.reset()
                bucketer.dist_enabled = (gi == len(groups) - 1)  # the bucketed all-reduce overlaps the last backward pass
                (loss if len(groups) == 1 else loss * share).backward()
                vals += share * np.array([float(v.detach()) for v in parts])              # (ce, box, rpn_ce, rpn_box)
            bucketer.finish()
            self.optimizer.step()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            spent += time.perf_counter() - t0
            if dist is not None and (it + 1) % cfg.TRAIN.DISPLAY == 0:
                # the logged losses are the mean over the ranks' frames (one small all-reduce per DISPLAY)
                t = torch.as_tensor(vals, dtype=torch.float64, device=params[0].device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t)
                vals = (t / world).cpu().numpy()
            loss_cls, loss_box, rpn_loss_cls, rpn_loss_box = vals
            history.append((rpn_loss_cls + rpn_loss_box + loss_cls + loss_box, rpn_loss_cls, rpn_loss_box, loss_cls, loss_box))
            if (it + 1) % cfg.TRAIN.DISPLAY == 0 and rank == 0:
                self.log('iter: %d / %d, total loss: %.4f, rpn_loss_cls: %.4f, rpn_loss_box: %.4f, loss_cls: %.4f, loss_box: %.4f, lr: %f'
                         % (it + 1, max_iters, history[-1][0], rpn_loss_cls, rpn_loss_box, loss_cls, loss_box, lr))
                self.log('speed: {:.3f}s / iter'.format(spent / (it + 1 - start_iter)))
            if (it + 1) % cfg.TRAIN.SNAPSHOT_ITERS == 0:
                last_snapshot_iter = it
                if rank == 0:
                    self.snapshot(sess, it)
        if last_snapshot_iter != it and it >= start_iter and rank == 0:
            self.snapshot(sess, it)
        bucketer.close()
        return history


def _tf_layout(w):
    """torch (out, in, kh, kw) / (out, in) -> the TF layouts `network.load` expects (HWIO / [in, out])"""
    a = w.detach().cpu().numpy()
    return a.transpose(2, 3, 1, 0) if a.ndim == 4 else a.T


def get_training_roidb(imdb):
    """Returns a roidb for use in training (:315-331)."""
    if cfg.TRAIN.USE_FLIPPED:
        print('Appending horizontally-flipped training examples...')
        imdb.append_flipped_images()
        print('done')
    print('Preparing training data...')
    from ..roi_data_layer.roidb import prepare_roidb
    prepare_roidb(imdb)
    print('done')
    return imdb.roidb


def get_data_layer(roidb, num_classes):
    """return a data layer (:333-343; the multi-scale GtDataLayer of the 2-D Faster-RCNN graphs is out of scope)."""
    from ..roi_data_layer.layer import RoIDataLayer
    return RoIDataLayer(roidb, num_classes)


def filter_roidb(roidb):
    """Remove roidb entries that have no usable RoIs (:345-370): an entry stays if it has a foreground box (max overlap >=
    FG_THRESH) or a background box (BG_THRESH_LO <= max overlap < BG_THRESH_HI)."""
    T = cfg.TRAIN

    def usable(entry):
        ov = np.asarray(entry['max_overlaps'])
        return bool(np.any(ov >= T.FG_THRESH) or np.any((ov < T.BG_THRESH_HI) & (ov >= T.BG_THRESH_LO)))

    kept = [e for e in roidb if usable(e)]
    print('Filtered {} roidb entries: {} -> {}'.format(len(roidb) - len(kept), len(roidb), len(kept)))
    return kept


def train_net(network, imdb, roidb, output_dir, pretrained_model=None, max_iters=10000):
    """Train a Fast R-CNN network (:373-383)."""
    roidb = filter_roidb(roidb)
    sw = SolverWrapper(None, None, network, imdb, roidb, output_dir, pretrained_model=pretrained_model)
    print('Solving...')
    history = sw.train_model(None, max_iters)
    print('done solving')
    return history


def bench_train_step(rank, world, dist, steps=None, warmup=2, frames_per_step=2, seed=0, amp=None, views=3, mfma=False, step_hook=None,
                     cast_many=True, seconds=1.0, fused_head=True, kernel_adam=True):
    """Full MV3D training step WITH the dense layers, for bench.py's `with_trunk` key ("also reported with
    VGG16 trunks included"; BASELINE configs[2] at one GPU, configs[3] under torch.distributed): synthetic KITTI-shaped
    frames (608x608x9 BEV, 375x1242x3 image), `frames_per_step` frames per rank and step, forward + four losses + backward +
    bucketed gradient all-reduce + Adam.  The VGG16 convolutions / FC layers run through torch (MIOpen / rocBLAS -- no
    hand-written kernel is claimed for them); the hot-path layers are the HIP kernels of this repository.  Timed over >= `seconds`
    (`steps` = None: a fixed count per precision from the round-5 step times, the same on every rank); the minimum and median step
    (one HIP event per step) are reported beside the mean."""
    from .. import sharding, synth
    from ..utils.timing import step_stats, timed_steps
    if steps is None:
        guess_ms = (12.0 if amp is not None else 43.0) if mfma else (53.0 if amp is None else 25.0)
        steps = max(8, int(np.ceil(seconds * 1e3 / (guess_ms * frames_per_step / 2.0))))
    from ..networks import get_network
    np.random.seed(cfg.RNG_SEED + rank)
    net = get_network("MV3D_train_3view" if views == 3 else "MV3D_train")   # configs[2]: "full 3-view MV3D -- BEV/FV/RGB VGG16"
    net.amp_dtype = amp                                          # None = the reference's fp32; torch.bfloat16: autocast dense layers
    net.mfma_trunk = bool(mfma)                                  # trunks' forward + backward on the bf16 MFMA kernel (trunk_train.py)
    net.cast_many = bool(cast_many)                              # (False: autocast's per-tensor casts -- tools/train_probe.py's A / B)
    net.fused_head = bool(fused_head)                            # (False: the head op by op -- tools/train_ab.py's A / B)
    params = net.parameters()
    from ..optim import Adam
    opt = Adam(params, lr=SolverWrapper.LEARNING_RATE) if kernel_adam else torch.optim.Adam(params, lr=SolverWrapper.LEARNING_RATE, fused=True)
    net.attach_optimizer(opt)                                     # (the kernel Adam also writes the head's 16-bit weight copies)
    bucketer = sharding.GradBucketer(params, dist if world > 1 else None)
    rng = np.random.RandomState(100 + rank)
    frames = []
    for k in range(frames_per_step):
        _, _, info, calib, (gt_bv, gt_3d, gt_cnr) = synth.rpn_head(7000 + 10 * rank + k, 76, 76, "peaky", return_gt=True)
        bev = (rng.random_sample((1, 608, 608, 9)) < 0.03).astype(np.float32) * rng.uniform(0, 2.4, (1, 608, 608, 9)).astype(np.float32)
        img = rng.randint(0, 255, (1, 375, 1242, 3)).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)
        fv = rng.uniform(0, 1, (1, 64, 512, 3)).astype(np.float32)
        frames.append({"lidar_bv_data": bev, "image_data": img.astype(np.float32), "lidar_fv_data": fv, "im_info": info, "calib": calib,
                       "gt_boxes_bv": gt_bv, "gt_boxes_3d": gt_3d, "gt_boxes_corners": gt_cnr})
    feed = stack_blobs(frames)
    if views != 3:
        feed.pop("lidar_fv_data", None)
    for k in ("lidar_bv_data", "image_data", "lidar_fv_data"):
        if k in feed:
            feed[k] = torch.as_tensor(feed[k]).cuda()            # resident inputs
    feed["keep_prob"] = 0.5

    def step():
        bucketer.zero_grad()
        layers = net.forward(feed)                               # the frames of the step as one batch
        loss, _ = total_loss(layers)
        bucketer.reset()
        bucketer.dist_enabled = True
        loss.backward()
        bucketer.finish()
        opt.step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    dt, ms = timed_steps(step, steps, barrier, hook=step_hook)   # (step_hook: tools/train_gap_probe.py brackets steps with a profiler)
    dt = sharding.max_over_ranks(dt, dist if world > 1 else None, device="cuda")
    nparam = sum(p.numel() for p in params)
    dense = (("the trunks' 3x3 convolutions forward + backward on this library's bf16 MFMA kernels (mv3d_tf_amd/trunk_train.py), rpn convs / FC "
              "head through torch autocast bf16, fp32 master weights, f32 hot path") if amp is not None else
             ("fp32 throughout: the trunks' forward, data-gradient and weight-gradient convolutions on this library's exact-f32 MFMA "
              "kernels (v_mfma_f32_32x32x2_f32), FC head through torch (rocBLAS)")) if mfma else (
        "torch (MIOpen / rocBLAS) VGG16 trunks + FC head, " + ("fp32" if amp is None else "autocast to %s (fp32 master weights, f32 hot path)"
                                                                 % str(amp).split(".")[-1]))
    out = {"workload": "MV3D_train%s full step: %d frames / GPU / step, 608x608x9 BEV + 375x1242x3 image%s; %s; Adam (mv3d_adam_step, one launch)"
                       % ("_3view" if views == 3 else "", frames_per_step, " + 64x512x3 front view" if views == 3 else "", dense),
           "frames_per_s": round(steps * frames_per_step * world / dt, 3), "ms_per_step": round(dt / steps * 1e3, 3), "timed_s": round(dt, 3),
           **step_stats(ms),
           "parameters": nparam, "gradient_bytes_per_step": bucketer.total_bytes(), "allreduce_buckets": len(bucketer.buckets),
           "allreduce": "RCCL, 25 MB buckets, last layer first, overlapping backward" if world > 1 else "none (1 GPU)"}
    bucketer.close()
    return out
