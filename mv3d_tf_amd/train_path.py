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
    finish(slot)  waits for that event only, draws, uploads the index lists (one pinned buffer, one copy), launches stage 2
                  into fixed-capacity ROI buffers (frame b's rows behind frame b - 1's, S <= B * cfg.TRAIN.BATCH_SIZE).

With `depth` slots a caller submits batch i + 1 before it finishes batch i: the device works on batch i's stage 2 / RoiPool
(and, in the train graph, the dense layers) while the host draws.  What remains is the host's own work: the legacy
`RandomState.permutation` shuffles every candidate (21 k background anchors per frame, twice).  Since round 4 that loop runs
in C on numpy's own generator state (mv3d_draw_training_subsamples, csrc/legacy_rng.hip: one call per batch, lists written
straight into pinned memory, ~2x numpy's own loop and none of its per-call Python); `draw_subsamples_host` /
`draw_samples_host` below are the numpy statement of the same draws (tests compare the two, draw for draw)."""
import ctypes as C
import time

import numpy as np
import numpy.random as npr
import torch

from . import ops
from ._lib import AnchorTargetParams, DrawFrame, DrawParams, ProposalTargetParams, check, lib
from .fast_rcnn.config import cfg

_HEAD = 4096                       # bytes of (counts + foreground flags) per frame that travel with the first copy


def _P(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _ptrs(ts):
    return (C.c_void_p * len(ts))(*[0 if t is None else t.data_ptr() for t in ts])


def _ints(vs):
    return (C.c_int * len(vs))(*[int(v) for v in vs])


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


class _Slot:
    pass


class TrainPathStream:
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
        self.pparams = ops.proposal_params(T)
        self.aparams = AnchorTargetParams(8, 1 if T.RPN_CLOBBER_POSITIVES else 0, float(T.RPN_NEGATIVE_OVERLAP),
                                          float(T.RPN_POSITIVE_OVERLAP))
        self.anchor_cap = max(int(T.RPN_BATCHSIZE), 1) * 2
        self.roi_cap = int(T.BATCH_SIZE)
        L = lib()
        self.cap = L.mv3d_proposal_3d_capacity(self.H, self.W, C.byref(self.pparams))
        if self.cap < 0:
            check(1, "mv3d_proposal_3d_capacity")
        self.slots = [self._make_slot() for _ in range(int(depth))]
        for k, sl_ in enumerate(self.slots):
            sl_.stream = self.streams[k % len(self.streams)] if self.streams else self.stream
        self._next = 0
        self.t_wait = self.t_draw = 0.0                  # host seconds spent waiting for stage 1 / drawing (diagnostics)
        # The host stage of a batch (wait for its reports, draw) runs on ONE helper thread, in submission order: the event wait and
        # the C draws release the GIL, so the submitting thread keeps enqueueing launches meanwhile.  Between submit() and finish()
        # of a slot the numpy global RNG belongs to the path (any other draw in that window would interleave with the batch's).
        self._helper = None
        self.async_draws = bool(async_draws)

    # ------------------------------------------------------------------ buffers of one batch in flight
    def _make_slot(self):
        B, H, W, N, dev, nc, L = self.B, self.H, self.W, self.N, self.dev, self.nc, lib()
        s = _Slot()
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
        awsz = max(L.mv3d_anchor_target_workspace_bytes(H, W, self.max_gt), 256)
        s.aws = [e((awsz,), torch.uint8) for _ in range(B)]
        s.awsz = awsz
        twsz = max(L.mv3d_proposal_target_workspace_bytes(self.cap, self.max_gt), 256)
        s.tws = [e((twsz,), torch.uint8) for _ in range(B)]
        s.twsz = twsz
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
        s.frames_c, s.sizes_c = (DrawFrame * B)(), (C.c_int32 * (5 * B))()
        s.d_lists = e((s.list_cap,), torch.int32)
        s.event = torch.cuda.Event()
        # argument arrays that never change
        s.a_cnt = _ptrs([s.report[b, :32] for b in range(B)])
        s.a_fgh = _ptrs([s.report[b, 32:] for b in range(B)])
        s.a_ws = _ptrs(s.aws)
        s.p_bv = _ptrs([s.prop[0][b] for b in range(B)])
        s.p_b3 = _ptrs([s.prop[2][b] for b in range(B)])
        s.p_cap = _ints([self.cap] * B)
        s.p_num = _ptrs([s.prop[3][b:b + 1] for b in range(B)])
        s.p_cnt = _ptrs([s.pt_counts[b] for b in range(B)])
        s.p_ws = _ptrs(s.tws)
        s.p_wsz = (C.c_size_t * B)(*[twsz] * B)
        s.tpar = (ProposalTargetParams * B)(*[ProposalTargetParams(nc, b, float(cfg.TRAIN.FG_THRESH), float(cfg.TRAIN.BG_THRESH_HI),
                                                                   float(cfg.TRAIN.BG_THRESH_LO)) for b in range(B)])
        s.busy = False
        return s

    def _sid(self, s):
        return C.c_void_p((s.stream or torch.cuda.current_stream()).cuda_stream)

    # ------------------------------------------------------------------ stage 1
    def submit(self, prob, pred, im_info, calib, gt):
        """prob (B,H,W,8), pred (B,H,W,24), im_info (B,3), calib (B,4,12): device f32 tensors; gt: list (len B) of (gt_bv
        (G,5), gt_3d (G,7), gt_corners (G,25)) device tensors.  Returns the slot handle for finish()."""
        B, H, W, L = self.B, self.H, self.W, lib()
        s = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        if s.busy:
            raise RuntimeError("TrainPathStream: every slot is in flight (finish() one first, or raise `depth`)")
        s.busy = True
        if s.stream is not None:
            s.stream.wait_stream(torch.cuda.current_stream())            # the inputs were produced on the caller's stream
        st = self._sid(s)
        G = [int(g[0].shape[0]) for g in gt]
        if max(G) > self.max_gt or min(G) <= 0:
            # (no box at all: the reference's anchor_target_layer takes argmax over an empty axis and raises as well --
            # filter_roidb keeps such frames out of training; more than max_gt: build the stream with a larger `max_gt`, as
            # networks/mv3d.py does on demand)
            s.busy = False
            raise ValueError("TrainPathStream: 1 .. max_gt = %d ground-truth boxes per frame, got %s" % (self.max_gt, G))
        # torch's cfg thresholds may have changed since the slot was built (cfg_from_file): refresh the parameter structs
        for b in range(B):
            s.tpar[b].fg_thresh, s.tpar[b].bg_thresh_hi, s.tpar[b].bg_thresh_lo = (float(cfg.TRAIN.FG_THRESH), float(cfg.TRAIN.BG_THRESH_HI),
                                                                                  float(cfg.TRAIN.BG_THRESH_LO))
        s.inputs = (prob, pred, im_info, calib, gt)                              # (kept alive until finish())
        s.G = G
        bv, img, b3, num, status = s.prop
        check(L.mv3d_proposal_3d(_P(prob), _P(pred), B, H, W, _P(im_info), _P(calib), C.byref(self.pparams), _P(bv), _P(img), _P(b3),
                                 _P(num), _P(status), _P(s.pws), C.c_size_t(s.pws.numel()), st), "mv3d_proposal_3d")
        s.a_gtbv, s.a_gt3d, s.a_gtc = _ptrs([g[0] for g in gt]), _ptrs([g[1] for g in gt]), _ptrs([g[2] for g in gt])
        s.a_G = _ints(G)
        check(L.mv3d_anchor_target_stage1_batch(B, H, W, _P(im_info), s.a_gtbv, s.a_gt3d, s.a_G, C.byref(self.aparams), _P(s.rpn_labels),
                                                _P(s.rpn_targets), s.a_cnt, s.a_fgh, s.a_ws, C.c_size_t(s.awsz), st),
              "mv3d_anchor_target_stage1_batch")
        check(L.mv3d_proposal_target_stage1_batch_devn(B, s.p_bv, s.p_b3, s.p_cap, s.p_num, s.a_gtbv, s.a_gt3d, s.a_G, s.tpar, s.p_cnt,
                                                       s.p_ws, s.p_wsz, st), "mv3d_proposal_target_stage1_batch_devn")
        ctx = torch.cuda.stream(s.stream) if s.stream is not None else _Null()
        with ctx:
            s.h_report.copy_(s.report[:, :s.h_report.shape[1]], non_blocking=True)      # three device-to-host copies, no kernel
            s.h_small.copy_(s.pt_counts, non_blocking=True)
            s.h_tail.copy_(s.tail, non_blocking=True)
            s.event.record()
        s.future = None
        if self.async_draws:
            if self._helper is None:
                from concurrent.futures import ThreadPoolExecutor
                self._helper = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mv3d-draws")
            s.future = self._helper.submit(self._host_stage, s)
        return s

    def close(self):
        """stop the helper thread (idempotent; the stream object stays usable with async_draws falling back to finish())"""
        h, self._helper = self._helper, None
        if h is not None:
            h.shutdown(wait=True)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _host_stage(self, s):
        """wait for the batch's reports, then draw (helper thread, or finish() itself with async_draws = False)"""
        t0 = time.perf_counter()
        s.event.synchronize()                                          # the batch's reports are on the host
        t1 = time.perf_counter()
        small = s.h_small.numpy()
        if int(s.h_tail.numpy()[self.B:].max()) & 1:
            raise ZeroDivisionError("float division")
        res = self._draw(s, small)
        self.t_wait += t1 - t0
        self.t_draw += time.perf_counter() - t1
        return res

    # ------------------------------------------------------------------ the host's draws
    def _draw(self, s, small):
        """Every subsampling draw of the batch -- frame by frame, anchor targets before proposal targets, exactly the draws of
        `draw_subsamples_host` / `draw_samples_host` -- by ONE C call (mv3d_draw_training_subsamples, csrc/legacy_rng.hip) that
        shuffles on the memory of numpy's own global generator and writes the index lists straight into the slot's pinned
        buffer; ctypes releases the GIL for its duration.  Returns (sizes[5 B], offsets[5 B], total)."""
        B, N, T = self.B, self.N, cfg.TRAIN
        head = s.h_report.numpy()
        hsz = head.shape[1]
        keep = []
        for b in range(B):
            n_inside, n_fg, n_bg, n_low = (int(v) for v in head[b, :16].view(np.int32))
            if 32 + n_fg <= hsz:
                flags = s.h_report.data_ptr() + b * hsz + 32
            else:                                                     # (more positives than travel with the first copy: rare)
                full = np.ascontiguousarray(s.report[b, 32:32 + n_fg].cpu().numpy())
                keep.append(full)
                flags = full.ctypes.data
            s.frames_c[b] = DrawFrame(n_fg, n_bg, n_low, int(small[b, 1]), int(small[b, 2]), 0, flags)
        rois_per_image = int(T.BATCH_SIZE) // 1
        par = DrawParams(int(T.RPN_BATCHSIZE), int(T.RPN_FG_FRACTION * T.RPN_BATCHSIZE), rois_per_image,
                         int(np.round(T.FG_FRACTION * rois_per_image)))
        gen = npr.mtrand._rand._bit_generator                          # the numpy GLOBAL legacy RandomState's MT19937
        with gen.lock:
            rc = lib().mv3d_draw_training_subsamples(C.c_void_p(gen.ctypes.state_address), B, s.frames_c, C.byref(par),
                                                     C.c_void_p(s.h_lists.data_ptr()), s.list_cap, s.sizes_c,
                                                     C.c_void_p(s.h_scratch.data_ptr()), s.h_scratch.numel())
        check(rc, "mv3d_draw_training_subsamples")
        sizes = list(s.sizes_c)
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += n
        return sizes, offs, o

    # ------------------------------------------------------------------ host draws + stage 2
    def finish(self, s):
        """Returns a dict of device tensors: rpn_labels (B,N), rpn_targets (B,N,6), anchors / anchors_3d / n_anchors,
        rois {bev, rgb, fv} (S,5) with the frame index in column 0, rois_3d (S,7), labels (S,1) i32, bbox_targets (S,24 nc),
        S (list of the frames' row counts), num_proposals (list)."""
        B, H, W, N, L = self.B, self.H, self.W, self.N, lib()
        try:
            sizes, offs, total = s.future.result() if s.future is not None else self._host_stage(s)
        except Exception:
            s.busy = False
            raise
        ctx = torch.cuda.stream(s.stream) if s.stream is not None else _Null()
        with ctx:
            if total:
                s.d_lists[:total].copy_(s.h_lists[:total], non_blocking=True)
        st = self._sid(s)
        # argument arrays by pointer arithmetic (a torch slice per pointer cost more host time than the launches they feed)
        lists0 = s.d_lists.data_ptr()
        lp = lambda k: (lists0 + 4 * offs[k]) if sizes[k] else 0
        vp = lambda ps: (C.c_void_p * len(ps))(*ps)
        a_d = [vp([lp(5 * b + k) for b in range(B)]) for k in range(3)]
        a_n = [_ints([sizes[5 * b + k] for b in range(B)]) for k in range(3)]
        check(L.mv3d_anchor_target_stage2_batch(B, H, W, C.byref(self.aparams), a_d[0], a_n[0], a_d[1], a_n[1], a_d[2], a_n[2],
                                                _P(s.rpn_labels), _P(s.anchors), _P(s.anchors_3d), _P(s.n_anchors), self.anchor_cap,
                                                s.a_ws, C.c_size_t(s.awsz), st), "mv3d_anchor_target_stage2_batch")
        n_fg = [sizes[5 * b + 3] for b in range(B)]
        n_bg = [sizes[5 * b + 4] for b in range(B)]
        S = [n_fg[b] + n_bg[b] for b in range(B)]
        off = [0]
        for v in S:
            off.append(off[-1] + v)
        rows = lambda t: vp([(t.data_ptr() + off[b] * t.stride(0) * t.element_size()) if S[b] else 0 for b in range(B)])
        prob, pred, im_info, calib, gt = s.inputs
        p_cal = vp([calib.data_ptr() + b * calib.stride(0) * calib.element_size() for b in range(B)])
        p_fg, p_bg = vp([lp(5 * b + 3) for b in range(B)]), vp([lp(5 * b + 4) for b in range(B)])
        outs = [rows(s.rois["bev"]), rows(s.rois["rgb"]), rows(s.labels), rows(s.bbox_targets), rows(s.rois_3d)]
        p_fv = rows(s.rois["fv"]) if self.want_fv else None
        check(L.mv3d_proposal_target_stage2_batch_devn(B, s.p_bv, s.p_b3, s.p_cap, s.p_num, s.a_gtbv, s.a_gt3d, s.a_gtc, s.a_G, p_cal,
                                                       s.tpar, p_fg, _ints(n_fg), p_bg, _ints(n_bg), outs[0], outs[1], outs[2], outs[3],
                                                       outs[4], p_fv, s.p_ws, s.p_wsz, st), "mv3d_proposal_target_stage2_batch_devn")
        St = int(off[-1])
        s.keep = (a_d, a_n, p_cal, p_fg, p_bg, outs, p_fv)                       # (argument arrays outlive the launches)
        s.busy = False
        return {"rpn_labels": s.rpn_labels, "rpn_targets": s.rpn_targets, "anchors": s.anchors, "anchors_3d": s.anchors_3d,
                "n_anchors": s.n_anchors, "rois": {k: v[:St] for k, v in s.rois.items()}, "rois_3d": s.rois_3d[:St],
                "labels": s.labels[:St], "bbox_targets": s.bbox_targets[:St], "S": S,
                "num_proposals": [int(v) for v in s.h_tail.numpy()[:B]], "proposals": s.prop, "stream": s.stream}


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
