"""MV3D graphs (lib/networks/MV3D_test.py:32-123, MV3D_train.py:42-182) as data tables run
by a small executor, with the hot-path layers bound to libmv3d_hip.so:

    proposal_layer_3d                                                     mv3d_tf_amd.rpn_msr (device tensors, any batch)
    anchor_target_layer + proposal_target_layer_3d (TRAIN)                mv3d_tf_amd.train_path.TrainPathStream: the batched
                                                                          C entries, ONE host round trip per step for the draws
    roi_pool of every view (+ gradient)                                   roi_pooling_layer.roi_pool_views: the library's RoiPool pair
                                                                          (one launch forward, index + gather backward)
    proposal_transform                                                    tuple element 0 ('bv') / 1 ('img')

`forward(feed)` takes B >= 1 frames (B > 1: lists of per-frame ground-truth arrays, im_info (B,3), calib (B,4,12)); the ROI
batch column is the frame index, every hot-path kernel runs once for the whole batch (blockIdx.y = frame).  The numpy-contract
callables of mv3d_tf_amd.rpn_msr (the tf.py_func boundary of network.py:221-273) are unchanged and give the same values.

Layer names, shapes and the plumbing of lib/networks/network.py:199-405 are kept (`layers`
dict, `get_output(name)`, NHWC activations, fc on a 4-D input flattens in (c,h,w) order,
`reshape_layer(d)` + pairwise softmax for the RPN scores, `load()` of the `.npy`
{layer: {weights, biases}} dictionary with ignore_missing).  The VGG16 convolutions / fully
connected layers are dense contractions outside this repository's parity contract: they run
through torch (MIOpen / rocBLAS) purely so that the drop-in entry points are runnable end to end.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from ..fast_rcnn.config import cfg
from ..roi_pooling_layer.roi_pooling_op import roi_pool_views
from ..rpn_msr.proposal_layer_tf import proposal_layer_3d, proposal_layer_3d_fixed

n_classes = 2                 # lib/networks/MV3D_train.py:4
_feat_stride = [8, 8]         # :5
anchor_scales = [1.0, 1.0]    # :6 (only len() is used: A = 2 * 2)

# conv name stem -> (out channels, followed by a 2x2/2 VALID max-pool?)   (no pool4: stride stays 8)
_VGG = [("conv1_1", 64, False), ("conv1_2", 64, True), ("conv2_1", 128, False), ("conv2_2", 128, True),
        ("conv3_1", 256, False), ("conv3_2", 256, False), ("conv3_3", 256, True),
        ("conv4_1", 512, False), ("conv4_2", 512, False), ("conv4_3", 512, False),
        ("conv5_1", 512, False), ("conv5_2", 512, False), ("conv5_3", 512, False)]
_INPUTS = ("lidar_bv_data", "image_data", "lidar_fv_data", "im_info", "calib", "gt_boxes", "gt_boxes_bv", "gt_boxes_3d",
           "gt_boxes_corners")


class MV3D:
    """One class for both graphs; `phase` is 'TEST' (MV3D_test) or 'TRAIN' (MV3D_train)."""

    def __init__(self, phase="TEST", trainable=True, device=None, seed=0, views=2):
        """views = 2: the reference's graphs (BEV + RGB towers).  views = 3 adds the front view the MV3D paper has and the
        reference leaves a TODO (`proposal_transform`, network.py:293-315): a third VGG16 trunk on `lidar_fv_data`
        (1, 64, 512, 3), `rois_fv` = mv3d_rois_3d_to_fv(rois_3d), a third RoiPool `pool_5_3` and tower fc6_3 / fc7_3, the
        three towers concatenated (6144 -> cls_score / bbox_pred).  Parity unpinned by construction."""
        self.phase = phase
        self.views = int(views)
        self.trainable = trainable
        self.device = torch.device(device or ("cuda:%d" % cfg.GPU_ID))
        self.layers = {}
        self.keep_prob = 1.0 if phase == "TEST" else 0.5          # train_mv.py:167 / test_mv.py:183
        # optional reduced-precision DENSE layers (torch.float16 / torch.bfloat16 autocast of the VGG16 trunks, RPN convs and FC
        # head: BASELINE configs[4] "fp16 VGG16"); the hot-path layers always get and give f32.  None = the reference's fp32.
        self.amp_dtype = None
        # the 3x3 convolutions on the hand-written MFMA kernel.  TEST graph / no_grad: trunks + rpn_conv/3x3 in f16 (mv3d_tf_amd.trunk,
        # forward only, next to amp_dtype = torch.float16).  TRAIN graph with gradients: the trunks' forward AND backward in bf16
        # with fp32 master weights (mv3d_tf_amd.trunk_train, next to amp_dtype = torch.bfloat16 for the other dense layers).
        self.mfma_trunk = False
        self.fused_head = True           # TRAIN phase: the fusion head as one autograd function (fused_head.py); False = op by op
        self.fixed_rois = False          # TEST phase: B x capacity ROI rows, no host sync (fast_rcnn.test_mv.ServeGraph)
        self._mfma = None
        self._train_pool = None
        self._wcache = {}
        self._step_half = None      # {name: (w, b)} in amp_dtype for the current training step (amp_cast.cast_params)
        # amp training: the dense head's reduced-precision parameter copies of a step from ONE multi-tensor launch (amp_cast.CastMany)
        # instead of one autocast cast per tensor and use; False = autocast's own casts (tools' A / B)
        self.cast_many = True
        self.params = {}
        g = torch.Generator().manual_seed(seed)

        def var(name, shape, std):
            w = (torch.randn(shape, generator=g) * std).to(self.device).requires_grad_(trainable)
            b = torch.zeros(shape[0], device=self.device, requires_grad=trainable)
            self.params[name] = [w, b]

        trunks = (("", 9), ("_2", 3)) + ((("_3", 3),) if self.views == 3 else ())
        for suffix, cin in trunks:                                 # BEV trunk (9 ch), RGB trunk (3 ch) [, FV trunk (3 ch)]
            c = cin
            for stem, cout, _ in _VGG:
                var(stem + suffix, (cout, c, 3, 3), 0.01)
                c = cout
        var("rpn_conv/3x3", (512, 512, 3, 3), 0.01)
        var("rpn_cls_score", (len(anchor_scales) * 2 * 2, 512, 1, 1), 0.01)
        var("rpn_bbox_pred", (len(anchor_scales) * 2 * 6, 512, 1, 1), 0.01)
        towers = ("_1", "_2") + (("_3",) if self.views == 3 else ())
        for t in towers:
            var("fc6" + t, (2048, 7 * 7 * 512), 0.01)
            var("fc7" + t, (2048, 2048), 0.01)
        var("cls_score", (n_classes, 2048 * len(towers)), 0.01)
        var("bbox_pred", (n_classes * 24, 2048 * len(towers)), 0.001)   # network.py:382-384

    # ---- lib/networks/network.py plumbing
    def get_output(self, layer):
        """The hot-path tensors of a TRAIN step (`rpn_rois`, `rpn_data`, `roi_data_3d`, ...) are VIEWS of the path's slot
        buffers: valid until the second-next forward() of the same batch shape; clone what has to live longer."""
        try:
            return self.layers[layer]
        except KeyError:
            raise KeyError("Unknown layer name fed: %s" % layer)

    def parameters(self):
        return [p for wb in self.params.values() for p in wb]

    def load(self, data_path, session=None, saver=None, ignore_missing=False):
        """`.npy` dict {layer: {'weights' (TF HWIO / [in,out]), 'biases'}} (network.py:45-64)."""
        data = np.load(data_path, allow_pickle=True, encoding="latin1").item()
        for key, sub in data.items():
            if key not in self.params:
                if not ignore_missing:
                    raise ValueError("no variable scope %s" % key)
                continue
            # per subkey, as network.py:55-64: a tensor that does not fit its variable (tf.assign's ValueError) is skipped under
            # ignore_missing -- VGG_imagenet.npy's conv1_1 filter is (3,3,3,64), the BEV variable (3,3,9,64): the reference
            # prints "ignore conv1_1", still assigns the biases and trains -- and raises otherwise.  Never reshaped.
            for subkey, want in zip(("weights", "biases"), self.params[key]):
                if subkey not in sub:
                    continue
                t = torch.as_tensor(np.asarray(sub[subkey], np.float32))
                if subkey == "weights":
                    t = t.permute(3, 2, 0, 1) if t.ndim == 4 else (t.t() if t.ndim == 2 else t)
                if tuple(t.shape) != tuple(want.shape):
                    print("ignore " + key)
                    if not ignore_missing:
                        raise ValueError("%s/%s: checkpoint tensor %s does not fit the variable %s"
                                         % (key, subkey, tuple(t.shape), tuple(want.shape)))
                    continue
                with torch.no_grad():
                    want.copy_(t)

    # ---- dense layers (torch; NHWC kept as channels_last NCHW views)
    def _amp(self):
        return torch.autocast(device_type="cuda", dtype=self.amp_dtype or torch.float16, enabled=self.amp_dtype is not None)

    def _conv(self, x, name, relu=True, pad=1):
        w, b = self.params[name]
        with self._amp():
            y = F.conv2d(x, w, b, padding=pad)
            return F.relu(y) if relu else y

    def _trunk(self, x, suffix):
        for stem, _, pool in _VGG:
            x = self._conv(x, stem + suffix)
            self.layers[stem + suffix] = x.permute(0, 2, 3, 1)      # NHWC view, as fetched by callers
            if pool:
                x = F.max_pool2d(x, 2, 2)
        if x.dtype != torch.float32:                                # the feature map the RoiPool layer reads: f32 NHWC
            self.layers[stem + suffix] = x.float().permute(0, 2, 3, 1)
        return x

    def _mfma_trunks(self, L):
        """trunks + RPN head with every 3x3 convolution on mv3d_conv3x3_f16; returns (rpn_cls_score, rpn_bbox_pred) f32 NHWC"""
        if torch.is_grad_enabled() and self.trainable:
            raise RuntimeError("mfma_trunk is the forward-only serving trunk: run it under torch.no_grad() (or trainable=False)")
        half = self.amp_dtype or torch.float32                       # amp_dtype None: the reference's fp32, on the f32 MFMA
        if self._mfma is None or self._mfma.dtype != half:
            from ..trunk import MfmaTrunks
            self._mfma = MfmaTrunks(self, _VGG, dtype=half)
        # all trunks walked together: one launch per depth for the BEV / image / front-view maps (MfmaTrunks.trunks)
        keys = [("", "lidar_bv_data"), ("_2", "image_data")] + ([("_3", "lidar_fv_data")] if self.views == 3 else [])
        maps = self._mfma.trunks([L[k] for _, k in keys], [sfx for sfx, _ in keys], [sfx == "" for sfx, _ in keys])
        bev = maps[0]
        L["conv5_3"] = bev[:, 1:-1, 1:-1].float().contiguous()       # the f32 NHWC map the RoiPool layer reads
        rpn = self._mfma.rpn_conv(bev)                               # (B, H, W, 512) f16
        L["rpn_conv/3x3"] = rpn
        heads = []
        for name in ("rpn_cls_score", "rpn_bbox_pred"):               # 1x1 convolutions = a matmul over the channel axis
            w, b = self.params[name]
            heads.append(F.linear(rpn, w.reshape(w.shape[0], -1).to(half), b.to(half)).float().contiguous())
        return heads

    def _serving_weights(self, name, nhwc_from=None):
        """Inference-only view of a layer's parameters, cached until the fp32 parameter changes: cast once to amp_dtype (autocast
        would re-cast the 100 M fc6 weights on every call) and, for a layer fed by a 4-D NHWC blob, with the input axis
        re-ordered from the reference's (c,h,w) flattening to (h,w,c) so that the blob is used as it lies in memory."""
        w, b = self.params[name]
        key = (name, self.amp_dtype, nhwc_from)
        ver = (w._version, b._version)
        hit = self._wcache.get(key)
        if hit is None or hit[0] != ver:
            wd = w.detach()
            if nhwc_from is not None:
                h_, w_, c_ = nhwc_from
                wd = wd.reshape(w.shape[0], c_, h_, w_).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
            dt = self.amp_dtype or torch.float32
            hit = (ver, wd.to(dt).contiguous(), b.detach().to(dt))
            self._wcache[key] = hit
        return hit[1], hit[2]

    def _fc(self, x, name, relu=True, wb=None):
        if wb is None and not torch.is_grad_enabled():              # serving: cached weights, no flattening copy
            nhwc = tuple(x.shape[1:]) if x.ndim == 4 else None
            w, b = self._serving_weights(name, nhwc)
            # (a plain GEMM: rocBLAS / hipBLASLt.  The convolution kernel run without taps measured 617 vs 943 TFLOP/s on fc6.)
            y = F.linear(x.reshape(x.shape[0], -1).to(w.dtype), w, b)
            return F.relu(y) if relu else y
        if x.ndim == 4:                                             # NHWC -> (c,h,w) flattening (network.py:373-377)
            x = x.permute(0, 3, 1, 2).reshape(x.shape[0], -1)
        w, b = wb if wb is not None else self._half_or_master(name)
        with self._amp():
            y = F.linear(x, w, b)
            return F.relu(y) if relu else y

    def _head_fn(self, pools, P, keep_prob):
        """The fusion head (MV3D_train.py:108-136 / MV3D_test.py:95-123) on the pooled maps: per view fc6 -> (dropout) -> fc7 ->
        (dropout), concatenated, cls_score (+ softmax) and bbox_pred.  P(name) -> (w, b) | None (= this network's own).  Returns
        ([fc7 per view], cls_score f32, cls_prob, bbox_pred f32)."""
        tower = []
        for t, x in zip(("_1", "_2", "_3"), pools):
            x = self._fc(x, "fc6" + t, wb=P("fc6" + t))
            if self.phase == "TRAIN":
                x = F.dropout(x, 1.0 - keep_prob, training=True)
            x = self._fc(x, "fc7" + t, wb=P("fc7" + t))
            if self.phase == "TRAIN":
                x = F.dropout(x, 1.0 - keep_prob, training=True)
            tower.append(x)
        fused = torch.cat(tower, dim=1)
        cls = self._fc(fused, "cls_score", relu=False, wb=P("cls_score")).float()
        prob = ops.softmax_rows(cls) if (cls.is_cuda and not (torch.is_grad_enabled() and cls.requires_grad)) else F.softmax(cls, dim=1)
        return tower, cls, prob, self._fc(fused, "bbox_pred", relu=False, wb=P("bbox_pred")).float()

    _HEAD_LAYERS = ("rpn_cls_score", "rpn_bbox_pred", "fc6_1", "fc7_1", "fc6_2", "fc7_2", "fc6_3", "fc7_3", "cls_score", "bbox_pred")

    def _half_or_master(self, name):
        """(w, b) of a dense head layer for this training step: under amp the copies in amp_dtype that ONE multi-tensor launch made
        for all head layers (amp_cast.CastMany; autocast then finds nothing to cast), otherwise the fp32 parameters themselves"""
        if self._step_half is not None and name in self._step_half:
            return self._step_half[name]
        return self.params[name]

    # ---- hot-path plumbing
    _TP_CACHE = 4                 # TrainPathStream objects kept alive (KITTI has four image sizes -> four sub-batch shapes)

    def _train_path(self, B, H, W, max_gt=1):
        """The batched target-layer path for B frames of an H x W head.  Two slots alternate, so the tensors a step's layers
        dict holds stay valid until the step after the next one OF THE SAME SHAPE starts (no per-step copies).  One C object
        (helper thread, pinned staging, device slots) per (B, H, W, capacity), kept in a small LRU: a step whose frames come in
        several image sizes runs one sub-batch per size (train_mv.group_frames_by_shape), each on its own object, so mixed-size
        steps neither rebuild the path nor overwrite each other's layer tensors.  A frame with more ground-truth
        boxes than the slots were sized for gets a larger object (a dense frame must not abort training) --
        capacities grow in powers of two from 64."""
        from ..train_path import TrainPathStream
        cap = getattr(self, "_tp_max_gt", 64)
        while cap < max_gt:
            cap *= 2
        self._tp_max_gt = cap
        key = (B, H, W, cap)
        cache = self.__dict__.setdefault("_tp_cache", {})
        tp = cache.pop(key, None)
        if tp is None:
            tp = TrainPathStream(B, H, W, self.device, num_classes=n_classes, depth=2, max_gt=cap, want_fv=(self.views == 3))
            while len(cache) >= self._TP_CACHE:                   # least recently used first (dicts keep insertion order)
                cache.pop(next(iter(cache))).close()
        cache[key] = tp                                           # (re-inserted: most recently used last)
        return tp

    @staticmethod
    def _gt_frames(L, B):
        """per frame (gt_boxes_bv, gt_boxes_3d, gt_boxes_corners): the feed holds arrays for one frame, lists for several"""
        keys = ("gt_boxes_bv", "gt_boxes_3d", "gt_boxes_corners")
        if isinstance(L[keys[0]], (list, tuple)):
            frames = list(zip(*[L[k] for k in keys]))
        else:
            frames = [tuple(L[k] for k in keys)]
        if len(frames) != B:
            raise ValueError("ground truth for %d frames, data for %d" % (len(frames), B))
        return frames

    # ---- the fused head's 16-bit weight copies, kept from step to step and written by the optimizer's launch
    def _held_head(self, sfx):
        """(stacked buffers, are they current?) for fused_head, or None outside mixed precision.  The copies are current when the last
        writer of every head parameter was an optimizer step that also wrote its copy (attach_optimizer): the parameter's version
        counter still is what that step saw (any in-place change through torch -- load(), a manual update -- moves it)."""
        if self.amp_dtype is None:
            return None
        h = self.__dict__.get("_head_lowp")
        if h is None or h["dtype"] != self.amp_dtype or h["sfx"] != sfx:
            from ..fused_head import head_buffers
            bufs, pieces = head_buffers(self.params, ["fc6" + t for t in sfx], ["fc7" + t for t in sfx], self.amp_dtype, self.device)
            h = self.__dict__["_head_lowp"] = {"dtype": self.amp_dtype, "sfx": sfx, "bufs": bufs, "pieces": pieces, "stamp": {}}
            opt = self.__dict__.get("_lowp_opt")
            if opt is not None:
                self._register_lowp(opt)
        current = bool(h["stamp"]) and all(h["stamp"].get(id(p)) == p._version for p, _ in h["pieces"])
        return h["bufs"], current

    def _register_lowp(self, opt):
        h = self.__dict__.get("_head_lowp")
        if h is None:
            return
        stamp = h["stamp"]
        for p, dst in h["pieces"]:
            opt.register_lowp(p, dst, lambda q, s=stamp: s.__setitem__(id(q), q._version))

    def attach_optimizer(self, opt):
        """mv3d_tf_amd.optim.Adam only: its step also writes the 16-bit copies of the head's weights the next forward reads (no cast launch,
        no second read of the fp32 masters).  Optional: without it the fused head casts at the top of every step."""
        if hasattr(opt, "register_lowp"):
            self.__dict__["_lowp_opt"] = opt
            self._register_lowp(opt)

    _STAGE_BYTES = 1 << 20

    def _stage_host_inputs(self, feed, L, dev, to_dev):
        """feed -> L on the device.  Tensors go as they are.  The SMALL host arrays of a step (im_info, calib, the ground-truth lists: a
        dozen arrays of a few hundred bytes) are packed into one pinned staging buffer and cross in ONE asynchronous copy -- each on
        its own is a pageable, synchronous hipMemcpy (profiles/r06_train_tail: 7 copy -> copy gaps of 24 us per step).  Two staging
        buffers alternate: the copy of step n may still be reading when step n + 1 packs."""
        small, order = [], []
        for k in _INPUTS:
            if k not in feed or feed[k] is None:
                continue
            v = feed[k]
            items = list(v) if isinstance(v, (list, tuple)) else [v]
            if dev.type == "cuda" and all(not isinstance(a, torch.Tensor) and np.asarray(a).nbytes <= 65536 for a in items):
                arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in items]
                small += arrs
                order.append((k, isinstance(v, (list, tuple)), [a.shape for a in arrs]))
            else:
                L[k] = [to_dev(a) for a in v] if isinstance(v, (list, tuple)) else to_dev(v)
        if not small:
            return
        total = sum(-(-a.size // 4) * 4 for a in small)                  # (16-byte aligned pieces)
        if total * 4 > self._STAGE_BYTES:
            for k, is_list, shapes in order:                            # (larger than the staging buffer: the plain way)
                v = feed[k]
                L[k] = [to_dev(a) for a in v] if is_list else to_dev(v)
            return
        st = self.__dict__.get("_stage")
        if st is None:                                                  # (pinned allocations cost milliseconds: once per network)
            st = self.__dict__["_stage"] = {"pin": [torch.empty(self._STAGE_BYTES // 4, dtype=torch.float32).pin_memory() for _ in range(2)], "n": 0}
        pin = st["pin"][st["n"] & 1]
        st["n"] += 1
        host = pin.numpy()
        off, offs = 0, []
        for a in small:
            host[off:off + a.size] = a.reshape(-1)
            offs.append(off)
            off += -(-a.size // 4) * 4
        devbuf = torch.empty(total, dtype=torch.float32, device=dev)
        devbuf.copy_(pin[:total], non_blocking=True)
        it = iter(zip(offs, small))
        for k, is_list, shapes in order:
            views = []
            for shp in shapes:
                o, a = next(it)
                views.append(devbuf[o:o + a.size].view(shp))
            L[k] = views if is_list else views[0]

    # ---- the graph
    def forward(self, feed):
        """feed: dict with the reference's placeholder names (Appendix C of SURVEY.md); numpy or tensors."""
        L = self.layers
        L.clear()
        dev = self.device
        to_dev = lambda a: a.to(dev) if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, np.float32)).to(dev)
        self._stage_host_inputs(feed, L, dev, to_dev)
        keep_prob = float(feed.get("keep_prob", self.keep_prob))
        self._step_half = None
        if self.cast_many and self.amp_dtype is not None and self.phase == "TRAIN" and torch.is_grad_enabled():
            from ..amp_cast import cast_params
            # (the fused head and RpnHeads cast their own weights; the torch trunks' RPN convolutions still take the step's copies)
            head = ("rpn_cls_score", "rpn_bbox_pred") if self.fused_head else self._HEAD_LAYERS
            if not (self.fused_head and self.mfma_trunk):
                self._step_half = cast_params(self.params, [n for n in head if n in self.params], self.amp_dtype)
        # plain NCHW for the torch / MIOpen convolutions: measured 22.2 ms vs 29.3 ms (channels_last) for fwd + bwd of the two
        # trunks' 26 convolutions of one frame (tools/conv_layout_probe.py); the hot-path layers take NHWC, made at conv5_3
        to_nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()
        if self.mfma_trunk and self.phase == "TRAIN" and torch.is_grad_enabled():
            # mixed-precision training trunks: forward and backward convolutions on the bf16 MFMA kernel (mv3d_tf_amd.trunk_train)
            from ..trunk_train import BufferPool, trunks as _trunks_fn
            tdt = torch.bfloat16 if self.amp_dtype is not None else torch.float32     # amp_dtype None: the reference's fp32, exact-f32 MFMA
            if self._train_pool is None:
                self._train_pool = BufferPool()         # (one forward / backward pair per network in flight)
            # the BEV / image / front-view trunks walk the same layer list: ONE launch per depth for all of them (grouped entries
            # mv3d_conv3x3_views_*, mv3d_maxpool2x2*_views_*, mv3d_conv3x3_wgrad_views_*).  At a training batch of 2 one trunk's
            # launches do not fill the chip, together they do -- on one stream, the same launches with and without data parallelism.
            keys = [("", "lidar_bv_data", "conv5_3"), ("_2", "image_data", "conv5_3_2")] + \
                ([("_3", "lidar_fv_data", "conv5_3_3")] if self.views == 3 else [])
            maps = _trunks_fn(_VGG, [L[k] for _, k, _ in keys], self.params, [sfx for sfx, _, _ in keys], pool=self._train_pool, dtype=tdt)
            for (_, _, out), m in zip(keys, maps):
                L[out] = m
            bev_nhwc = L["conv5_3"]
            from ..trunk_train import conv_relu
            rpn_nhwc = conv_relu(bev_nhwc, *self.params["rpn_conv/3x3"], dtype=tdt)   # (B, H, W, 512) f32, same kernels
            L["rpn_conv/3x3"] = rpn_nhwc
            if self.fused_head:
                # both 1x1 heads as ONE GEMM on the stacked filters (fused_head.RpnHeads)
                from ..fused_head import RpnHeads
                score, L["rpn_bbox_pred"] = RpnHeads.apply(self.amp_dtype or torch.float32, rpn_nhwc, *self.params["rpn_cls_score"], *self.params["rpn_bbox_pred"])
            else:
                heads = []
                for name in ("rpn_cls_score", "rpn_bbox_pred"):                   # 1x1 convolutions = a matmul over the channel axis
                    w, b = self._half_or_master(name)
                    with self._amp():
                        heads.append(F.linear(rpn_nhwc, w.reshape(w.shape[0], -1), b).float().contiguous())
                score, L["rpn_bbox_pred"] = heads
        elif self.mfma_trunk:
            score, L["rpn_bbox_pred"] = self._mfma_trunks(L)
        else:
            bev = self._trunk(to_nchw(L["lidar_bv_data"]), "")
            self._trunk(to_nchw(L["image_data"]), "_2")
            if self.views == 3:
                self._trunk(to_nchw(L["lidar_fv_data"]), "_3")
            # RPN (MV3D_train.py:82-103)
            rpn = self._conv(bev, "rpn_conv/3x3")
            L["rpn_conv/3x3"] = rpn.permute(0, 2, 3, 1)
            score = self._conv(rpn, "rpn_cls_score", relu=False, pad=0).float().permute(0, 2, 3, 1).contiguous()
            L["rpn_bbox_pred"] = self._conv(rpn, "rpn_bbox_pred", relu=False, pad=0).float().permute(0, 2, 3, 1).contiguous()
        L["rpn_cls_score"] = score
        n, h, w, c = score.shape
        L["rpn_cls_score_reshape"] = score.reshape(n, h, -1, 2)                       # reshape_layer(2) (network.py:333-341)
        # :399-403 -- nothing differentiates through rpn_cls_prob (the RPN losses take the logits, the proposal layer is a py_func):
        # on the device the library's forward-only softmax (mv3d_softmax_rows)
        if score.is_cuda and score.dtype == torch.float32:
            L["rpn_cls_prob"] = ops.softmax_rows(L["rpn_cls_score_reshape"]).reshape(n, h, -1, 2)
        else:
            L["rpn_cls_prob"] = F.softmax(L["rpn_cls_score_reshape"].reshape(-1, 2), dim=1).reshape(n, h, -1, 2)
        L["rpn_cls_prob_reshape"] = L["rpn_cls_prob"].reshape(n, h, w, c)
        stride = _feat_stride[0]
        B = int(n)
        info = L["im_info"].reshape(-1, 3)
        cal = L["calib"].reshape(-1, 4, 12)
        if info.shape[0] == 1 and B > 1:
            info = info.expand(B, 3)
        if cal.shape[0] == 1 and B > 1:
            cal = cal.expand(B, 4, 12)
        info, cal = info.contiguous(), cal.contiguous()
        prob = L["rpn_cls_prob_reshape"].detach().contiguous()
        pred = L["rpn_bbox_pred"].detach().contiguous()
        if self.phase == "TRAIN":
            # anchor_target_layer (MV3D_train.py:88) + proposal_layer_3d (:98) + proposal_target_layer_3d (:105), all frames
            # behind one launch per kernel; the subsampling draws come from the numpy global RNG, frame by frame
            gt = [tuple(t.reshape(-1, c).contiguous() for t, c in zip(g, (5, 7, 25))) for g in self._gt_frames(L, B)]
            path = self._train_path(B, h, w, max(int(g[0].shape[0]) for g in gt))
            out = path.finish(path.submit(prob, pred, info, cal, gt))                 # (views of the slot's buffers: valid for two steps)
            L["roi_rows"] = out["S"]
            bvb, imgb, b3b = out["proposals"][:3]
            cnt = out["num_proposals"]
            cat = lambda t: t[0, :cnt[0]] if B == 1 else torch.cat([t[b, :cnt[b]] for b in range(B)], 0)
            rois = (cat(bvb), cat(imgb), cat(b3b))
            rois = rois + (rois[2],)                                                  # network.py:234
            L["rpn_rois"] = rois
            if B == 1:
                m = int(out["n_anchors"][0].item())
                L["rpn_data"] = (out["rpn_labels"][0], out["rpn_targets"][0], out["anchors"][0, :m], out["anchors_3d"][0, :m])
            else:
                L["rpn_data"] = (out["rpn_labels"], out["rpn_targets"], out["anchors"], out["anchors_3d"], out["n_anchors"])
            L["rpn-data"] = L["rpn_data"]                                             # (the Faster-RCNN spelling)
            data = (out["rois"]["bev"], out["rois"]["rgb"], out["labels"], out["bbox_targets"], out["rois_3d"])
            L["roi_data_3d"] = data                                                   # (rois_bv, rois_img, labels, targets, rois_3d)
            L["roi_data_bv"], L["roi_data_img"] = data[0], data[1]                    # proposal_transform (network.py:292-315)
            r3, rois_fv = data[4], out["rois"]["fv"]
        else:
            if getattr(self, "fixed_rois", False):
                # the serving step as a capturable graph (fast_rcnn.test_mv.ServeGraph): B * cap rows, no host sync; num_rois read later
                bv, img, b3, L["num_rois"], L["rois_status"], L["rois_per_frame"] = proposal_layer_3d_fixed(prob, pred, info, cal, self.phase, stride)
            else:
                bv, img, b3 = proposal_layer_3d(prob, pred, info, cal, self.phase, [stride, ], anchor_scales)
            rois = (bv, img, b3, b3)
            L["rois"] = rois
            L["roi_data_bv"], L["roi_data_img"] = rois[0], rois[1]
            r3, rois_fv = rois[2], None
        # RoI pooling of every view in one launch (+ the pair's gradient) and the fusion head (MV3D_test.py:95-123)
        views = [(L["conv5_3"], L["roi_data_bv"]), (L["conv5_3_2"], L["roi_data_img"])]
        names = ["pool_5", "pool_5_2"]
        if self.views == 3:
            if rois_fv is None:
                from ..utils.front_view import rois_3d_to_fv
                rois_fv = rois_3d_to_fv(r3)
            L["roi_data_fv"] = rois_fv
            views.append((L["conv5_3_3"], rois_fv))
            names.append("pool_5_3")
        # (serving in 16-bit mode: the pooled maps in the head's type straight from the pooling launch -- no f32 copy, no cast launch)
        for name, top in zip(names, roi_pool_views([(d.contiguous(), r.contiguous()) for d, r in views], 7, 7, 1.0 / 8, top_dtype=self.amp_dtype)):
            L[name] = top
        if self.fused_head and self.phase == "TRAIN" and torch.is_grad_enabled():
            # the head as ONE autograd function (mv3d_tf_amd/fused_head.py): batched GEMMs over the views, no per-op graph nodes
            from ..fused_head import fused_head
            sfx = ("_1", "_2", "_3")[:len(names)]
            L["cls_score"], L["bbox_pred"], tower = fused_head([L[n] for n in names], self.params, ["fc6" + t for t in sfx], ["fc7" + t for t in sfx],
                                                               keep_prob, self.amp_dtype or torch.float32, held=self._held_head(sfx))
            L["cls_prob"] = ops.softmax_rows(L["cls_score"]) if L["cls_score"].is_cuda else F.softmax(L["cls_score"].detach(), dim=1)   # (fetched, never differentiated: the losses take cls_score)
        else:
            tower, L["cls_score"], L["cls_prob"], L["bbox_pred"] = self._head_fn([L[n] for n in names], lambda n: None, keep_prob)
        for t, x in zip(("_1", "_2", "_3"), tower):
            L["fc7" + t] = x
        self._step_half = None                                                      # (the autograd graph keeps what backward needs)
        return L
