"""The training-side hot path on FRESH inputs, B frames per batch, as the train graph and a data loader drive it
(BASELINE configs[2] / configs[3] per-GPU step):

    mv3d_proposal_3d (TRAIN cfg)                                 lib/rpn_msr/proposal_layer_tf.py:25-202
    mv3d_anchor_target_stage1_batch / stage2_batch               lib/rpn_msr/anchor_target_layer_tf.py:21-250
    mv3d_proposal_target_stage1_batch_devn / stage2_batch_devn   lib/rpn_msr/proposal_target_layer_tf.py:19-94
      (+ the third view's ROIs out of the emit launch)

`hot_path.TrainPathBatch` freezes one batch and replays it; this class takes NEW heads / ground truth every batch.  The
random subsamplings stay what the reference makes them -- draws from the numpy GLOBAL RNG on the host, frame by frame, anchor
targets before proposal targets (`draw_subsamples`, `draw_samples`: the same functions the numpy-contract callables use) --
so a batch costs ONE host round trip, and it is taken off the device's critical path:

    submit(...)   stage 1 of a batch: every launch up to the candidate lists, then the counts (and the foreground flags the
                  third anchor draw needs) travel to PINNED host memory behind an event; the proposals' number never leaves
                  the device (`*_devn` entries).  Returns at once.
    host stage    waits for that event only, draws, uploads the index lists (one pinned buffer, one copy), launches stage 2
                  into fixed-capacity ROI buffers (frame b's rows behind frame b - 1's, S <= B * cfg.TRAIN.BATCH_SIZE).
    finish(slot)  waits until the host stage has ENQUEUED stage 2 and returns the batch's tensors.

With `depth` slots a caller submits batch i + 1 before it finishes batch i: the device works on batch i's stage 2 / RoiPool
(and, in the train graph, the dense layers) while the host draws.  Since round 4 the whole sequence is C
(`mv3d_train_path_*`, csrc/train_stream.hip): submit() and finish() are ONE library call each, the host stage runs on a helper
thread the library owns, and the legacy `RandomState.permutation` shuffles (21 k background anchors per frame, twice) run in C
on numpy's own generator state (`mv3d_draw_training_subsamples`, csrc/legacy_rng.hip).  Between submit() and finish() of a slot
the numpy global RNG belongs to the path (any other draw in that window would interleave with the batch's).
`draw_subsamples_host` / `draw_samples_host` below are the numpy statement of the same draws (tests compare the two, draw for
draw)."""
import ctypes as C

import numpy as np
import numpy.random as npr
import torch

from . import ops
from ._lib import AnchorTargetParams, DrawParams, ProposalTargetParams, TrainPathConfig, TrainPathSlot, check, lib
from .fast_rcnn.config import cfg

_HEAD = 4096                       # bytes of (counts + foreground flags) per frame that travel with the first copy


def draw_subsamples_host(head, n_fg_flags_fetch, N):
    """anchor_target_layer_tf.py:146-183 given stage 1's counts + foreground flags as HOST bytes (`head`, the first _HEAD
    bytes of the frame's report; `n_fg_flags_fetch()` returns all flags in the rare case of more positives than fit).
    Same draws, same order as rpn_msr.anchor_target_layer_tf.draw_subsamples."""
    T = cfg.TRAIN
    n_inside, n_fg, n_bg, n_low = (int(v) for v in head[:16].view(np.int32))
    num_fg = int(T.RPN_FG_FRACTION * T.RPN_BATCHSIZE)
    dis_fg = npr.permutation(n_fg)[:n_fg - num_fg] if n_fg > num_fg else None                 # :146-151
    num_bg = T.RPN_BATCHSIZE - min(n_fg, num_fg)
    dis_bg1 = npr.permutation(n_bg)[:n_bg - num_bg] if n_bg > num_bg else None                # :154-159
    n_pos = 0
    if n_fg:                                                                                  # :176-183
        alive = (head[32:32 + n_fg] if 32 + n_fg <= len(head) else n_fg_flags_fetch()[:n_fg]).astype(bool)
        if dis_fg is not None:
            alive[dis_fg] = False
        n_pos = int(alive.sum())
    num_bg2 = T.RPN_BATCHSIZE - n_pos
    dis_bg2 = npr.permutation(n_low)[:n_low - num_bg2] if n_low > num_bg2 else None
    return dis_fg, dis_bg1, dis_bg2


def draw_samples_host(counts):
    """proposal_target_layer_tf.py:246-269 given stage 1's counts on the host; same draws as
    rpn_msr.proposal_target_layer_tf.draw_samples."""
    T = cfg.TRAIN
    n_fg, n_bg = int(counts[1]), int(counts[2])
    rois_per_image = T.BATCH_SIZE // 1
    fg_n = int(min(np.round(T.FG_FRACTION * rois_per_image), n_fg))
    fg_pick = npr.permutation(n_fg)[:fg_n] if n_fg > 0 else np.zeros(0, np.int64)
    bg_n = int(min(rois_per_image - fg_n, n_bg))
    bg_pick = npr.permutation(n_bg)[:bg_n] if n_bg > 0 else np.zeros(0, np.int64)
    return fg_pick, bg_pick


def _global_mt19937_address():
    """Address of the {uint32 key[624]; int pos} state of the numpy GLOBAL legacy RandomState, which the library draws on in C
    (csrc/legacy_rng.hip).  Refused unless that generator IS an MT19937 (`np.random.set_bit_generator` can replace it by a PCG64
    or anything else, whose state the C code would scribble over: ADVICE r04).  Threading contract: the helper thread draws
    without numpy's generator lock (a threading.Lock it cannot take from C) -- between submit() and finish() of a slot no other
    thread of the process may draw from the global RNG; the path is for one submitting thread per process."""
    bg = npr.mtrand._rand._bit_generator
    if not isinstance(bg, npr.MT19937):
        raise TypeError("TrainPathStream draws on numpy's global legacy MT19937; the global bit generator is a %s" % type(bg).__name__)
    return bg.ctypes.state_address


class _Slot:
    pass


class TrainPathStream:
    """Python face of the C object `mv3d_train_path` (include/mv3d_hip.h, csrc/train_stream.hip): this class allocates the slots'
    buffers (torch tensors: device memory and pinned host memory), hands their addresses over once, and then costs its caller
    ONE C call per submit() and one per finish(); the launches, the copies, the event, the draws and the helper thread live in
    the library.  `async_draws=False` builds the object without its helper thread (finish() then does the host stage itself)."""

    def __init__(self, B, H, W, device, num_classes=2, depth=2, max_gt=64, want_fv=True, stream=None, async_draws=True, streams=None):
        self.B, self.H, self.W, self.dev, self.nc = int(B), int(H), int(W), torch.device(device), int(num_classes)
        self.N = self.H * self.W * 4
        self.want_fv = bool(want_fv)
        self.stream = stream
        # streams: one HIP stream per slot (slot k's launches and copies go to streams[k % len]): the batches in flight then overlap on
        # the device as well -- stage 1 of a batch is a chain of small, latency-bound launches that leaves the chip mostly idle, which
        # on ONE stream sits in front of the previous batch's RoiPool launches.  submit() orders the slot's stream behind the caller's
        # current stream (the inputs); the caller consumes a batch's outputs on out["stream"].
        self.streams = list(streams) if streams else None
        self.max_gt = int(max_gt)
        T = cfg.TRAIN
        self.anchor_cap = max(int(T.RPN_BATCHSIZE), 1) * 2
        self.roi_cap = int(T.BATCH_SIZE)
        L = lib()
        self.pparams = ops.proposal_params(T)
        self.cap = L.mv3d_proposal_3d_capacity(self.H, self.W, C.byref(self.pparams))
        if self.cap < 0:
            check(1, "mv3d_proposal_3d_capacity")
        self._cfg_key, self._config = self._make_config()
        self.slots = [self._make_slot(k) for k in range(int(depth))]
        for k, sl_ in enumerate(self.slots):
            sl_.stream = self.streams[k % len(self.streams)] if self.streams else self.stream
        self._slots_c = (TrainPathSlot * len(self.slots))(*[sl_.c for sl_ in self.slots])
        self._h = C.c_void_p()
        self.async_draws = bool(async_draws)
        with torch.cuda.device(self.dev):
            check(L.mv3d_train_path_create(C.byref(self._config), len(self.slots), self._slots_c, 1 if self.async_draws else 0,
                                           C.byref(self._h)), "mv3d_train_path_create")
        self._next = 0

    # ------------------------------------------------------------------ parameters (cfg may change between batches: cfg_from_file)
    def _make_config(self):
        T = cfg.TRAIN
        rois_per_image = int(T.BATCH_SIZE) // 1
        key = (float(T.FG_THRESH), float(T.BG_THRESH_HI), float(T.BG_THRESH_LO), int(T.RPN_BATCHSIZE), float(T.RPN_FG_FRACTION),
               float(T.FG_FRACTION), rois_per_image, bool(T.RPN_CLOBBER_POSITIVES), float(T.RPN_NEGATIVE_OVERLAP), float(T.RPN_POSITIVE_OVERLAP))
        c = TrainPathConfig()
        c.batch, c.H, c.W, c.num_classes = self.B, self.H, self.W, self.nc
        c.proposal_cap, c.anchor_cap, c.roi_cap, c.max_gt = self.cap, self.anchor_cap, self.roi_cap, self.max_gt
        c.proposal = self.pparams
        c.anchor = AnchorTargetParams(8, 1 if T.RPN_CLOBBER_POSITIVES else 0, float(T.RPN_NEGATIVE_OVERLAP), float(T.RPN_POSITIVE_OVERLAP))
        c.target = ProposalTargetParams(self.nc, 0, float(T.FG_THRESH), float(T.BG_THRESH_HI), float(T.BG_THRESH_LO))
        c.draw = DrawParams(int(T.RPN_BATCHSIZE), int(T.RPN_FG_FRACTION * T.RPN_BATCHSIZE), rois_per_image,
                            int(np.round(T.FG_FRACTION * rois_per_image)))
        return key, c

    # ------------------------------------------------------------------ buffers of one batch in flight
    def _make_slot(self, index):
        B, H, W, N, dev, nc, L = self.B, self.H, self.W, self.N, self.dev, self.nc, lib()
        s = _Slot()
        s.index = index
        e = lambda shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)
        s.pack, s.prop = ops.proposal_3d_outputs(B, self.cap, dev)
        s.tail = s.pack[s.pack.numel() - 2 * B:].view(torch.int32)             # [proposal counts (B) | status words (B)], contiguous
        s.pws = e((max(L.mv3d_proposal_3d_workspace_bytes(B, H, W, C.byref(self.pparams)), 256),), torch.uint8)
        s.rpn_labels, s.rpn_targets = e((B, N)), e((B, N, 6))
        s.anchors, s.anchors_3d = e((B, self.anchor_cap, 5)), e((B, self.anchor_cap, 7))
        s.n_anchors = e((B,), torch.int32)
        # per frame: [counts 32 B | foreground flags N B], rows padded to a multiple of 256 B; the host reads the first _HEAD
        # bytes of every row with one strided copy
        s.row = (32 + N + 255) // 256 * 256
        s.report = e((B, s.row), torch.uint8)
        s.pt_counts = e((B, 4), torch.int32)                                  # (written in place by stage 1, fetched as it lies)
        s.awsz = (max(L.mv3d_anchor_target_workspace_bytes(H, W, self.max_gt), 256) + 255) // 256 * 256
        s.aws = e((B, s.awsz), torch.uint8)
        s.twsz = (max(L.mv3d_proposal_target_workspace_bytes(self.cap, self.max_gt), 256) + 255) // 256 * 256
        s.tws = e((B, s.twsz), torch.uint8)
        S = B * self.roi_cap
        s.rois = {"bev": e((S, 5)), "rgb": e((S, 5)), "fv": e((S, 5))}
        s.rois_3d, s.labels, s.bbox_targets = e((S, 7)), e((S, 1), torch.int32), e((S, 24 * nc))
        # pinned host staging: reports in, index lists out
        head = min(_HEAD, s.row)
        s.h_report = torch.empty((B, head), dtype=torch.uint8).pin_memory()
        s.h_small = torch.empty((B, 4), dtype=torch.int32).pin_memory()           # proposal-target counts
        s.h_tail = torch.empty((2 * B,), dtype=torch.int32).pin_memory()          # proposal counts | status words
        s.list_cap = B * (3 * N + 2 * (self.cap + self.max_gt))                   # every index list at its largest
        s.h_lists = torch.empty((s.list_cap,), dtype=torch.int32).pin_memory()
        s.h_scratch = torch.empty((max(N, self.cap + self.max_gt) + 64,), dtype=torch.int32)      # one permutation at its largest
        s.d_lists = e((s.list_cap,), torch.int32)
        bv, img, b3, num, status = s.prop
        assert s.tail.data_ptr() == num.data_ptr() and status.data_ptr() == num.data_ptr() + 4 * B
        c = s.c = TrainPathSlot()
        c.blob_bv, c.blob_img, c.blob_3d, c.num_proposals = bv.data_ptr(), img.data_ptr(), b3.data_ptr(), s.tail.data_ptr()
        c.proposal_ws, c.proposal_ws_bytes = s.pws.data_ptr(), s.pws.numel()
        c.rpn_labels, c.rpn_targets, c.anchors, c.anchors_3d = (t.data_ptr() for t in (s.rpn_labels, s.rpn_targets, s.anchors, s.anchors_3d))
        c.n_anchors, c.report, c.report_row, c.pt_counts = s.n_anchors.data_ptr(), s.report.data_ptr(), s.row, s.pt_counts.data_ptr()
        c.anchor_ws, c.anchor_ws_bytes, c.target_ws, c.target_ws_bytes = s.aws.data_ptr(), s.awsz, s.tws.data_ptr(), s.twsz
        c.rois_bev, c.rois_rgb = s.rois["bev"].data_ptr(), s.rois["rgb"].data_ptr()
        c.rois_fv = s.rois["fv"].data_ptr() if self.want_fv else None
        c.rois_3d, c.labels, c.bbox_targets = s.rois_3d.data_ptr(), s.labels.data_ptr(), s.bbox_targets.data_ptr()
        c.lists, c.lists_cap = s.d_lists.data_ptr(), s.list_cap
        c.h_report, c.h_report_row, c.h_pt_counts, c.h_num_proposals = s.h_report.data_ptr(), head, s.h_small.data_ptr(), s.h_tail.data_ptr()
        c.h_lists, c.h_scratch, c.scratch_cap = s.h_lists.data_ptr(), s.h_scratch.data_ptr(), s.h_scratch.numel()
        s.rows_c, s.nprop_c, s.sizes_c = (C.c_int32 * B)(), (C.c_int32 * B)(), (C.c_int32 * (5 * B))()
        s.busy = False
        return s

    # ------------------------------------------------------------------ stage 1
    def submit(self, prob, pred, im_info, calib, gt):
        """prob (B,H,W,8), pred (B,H,W,24), im_info (B,3), calib (B,4,12): device f32 tensors; gt: list (len B) of (gt_bv
        (G,5), gt_3d (G,7), gt_corners (G,25)) device tensors.  Returns the slot handle for finish()."""
        B, L = self.B, lib()
        s = self.slots[self._next]
        if s.busy:
            raise RuntimeError("TrainPathStream: every slot is in flight (finish() one first, or raise `depth`)")
        G = [int(g[0].shape[0]) for g in gt]
        if len(G) != B or max(G) > self.max_gt or min(G) <= 0:
            # (no box at all: the reference's anchor_target_layer takes argmax over an empty axis and raises as well --
            # filter_roidb keeps such frames out of training; more than max_gt: build the stream with a larger `max_gt`, as
            # networks/mv3d.py does on demand)
            raise ValueError("TrainPathStream: %d frames with 1 .. max_gt = %d ground-truth boxes each, got %s" % (B, self.max_gt, G))
        key, conf = self._make_config()                                  # cfg's thresholds may have changed since the last batch
        if key != self._cfg_key:
            if any(sl_.busy for sl_ in self.slots):
                raise RuntimeError("TrainPathStream: cfg.TRAIN changed while a batch is in flight")
            check(L.mv3d_train_path_configure(self._h, C.byref(conf)), "mv3d_train_path_configure")
            self._cfg_key, self._config = key, conf
        self._next = (self._next + 1) % len(self.slots)
        if s.stream is not None:
            s.stream.wait_stream(torch.cuda.current_stream())            # the inputs were produced on the caller's stream
        st = (s.stream or torch.cuda.current_stream()).cuda_stream
        s.inputs = (prob, pred, im_info, calib, gt)                              # (kept alive until finish())
        s.G = G
        mt = _global_mt19937_address()                                           # the numpy GLOBAL legacy RandomState's MT19937
        PP = C.c_void_p * B
        check(L.mv3d_train_path_submit(self._h, s.index, prob.data_ptr(), pred.data_ptr(), im_info.data_ptr(), calib.data_ptr(),
                                       PP(*[g[0].data_ptr() for g in gt]), PP(*[g[1].data_ptr() for g in gt]),
                                       PP(*[g[2].data_ptr() for g in gt]), (C.c_int * B)(*G), mt, st), "mv3d_train_path_submit")
        s.busy = True
        return s

    def close(self):
        """destroy the C object: waits for the batches in flight, joins the helper thread (idempotent)"""
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().mv3d_train_path_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def host_seconds(self):
        """(seconds the host stage waited for stage 1, seconds it drew) since the last read"""
        w, d = C.c_double(), C.c_double()
        check(lib().mv3d_train_path_host_seconds(self._h, C.byref(w), C.byref(d)), "mv3d_train_path_host_seconds")
        return w.value, d.value

    # ------------------------------------------------------------------ host draws + stage 2
    def finish(self, s):
        """Returns a dict of device tensors: rpn_labels (B,N), rpn_targets (B,N,6), anchors / anchors_3d / n_anchors,
        rois {bev, rgb, fv} (S,5) with the frame index in column 0, rois_3d (S,7), labels (S,1) i32, bbox_targets (S,24 nc),
        S (list of the frames' row counts), num_proposals (list)."""
        s.busy = False
        check(lib().mv3d_train_path_finish(self._h, s.index, s.rows_c, s.nprop_c, s.sizes_c), "mv3d_train_path_finish")
        S = list(s.rows_c)
        St = sum(S)
        return {"rpn_labels": s.rpn_labels, "rpn_targets": s.rpn_targets, "anchors": s.anchors, "anchors_3d": s.anchors_3d,
                "n_anchors": s.n_anchors, "rois": {k: v[:St] for k, v in s.rois.items()}, "rois_3d": s.rois_3d[:St],
                "labels": s.labels[:St], "bbox_targets": s.bbox_targets[:St], "S": S,
                "num_proposals": list(s.nprop_c), "proposals": s.prop, "stream": s.stream, "draw_sizes": list(s.sizes_c)}
