"""Frame-level sharding of the hot path across the GPUs of a node.

The reference processes one frame per step in one process; frames are
independent, so the path shards by frame with NO data-path collective: rank r of W takes
frames r, r+W, r+2W, ...  torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo"
in the CPU tests) is used only for (a) the barrier + max-over-ranks timing of bench.py and
(b) gathering the small per-frame results (ROI counts / detections) onto rank 0.
"""
import torch


def frame_shard(num_frames, rank, world_size):
    """Indices of the frames rank `rank` owns (round-robin, so ragged counts differ by <= 1)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, num_frames, world_size))


def owner_of(frame, world_size):
    return frame % world_size


def max_over_ranks(value, dist=None, device="cpu"):
    """max of a python float over all ranks (the bench's step time)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device="cpu"):
    """sum of a python float over all ranks (whole-job rates of per-rank secondary lines)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, dist=None, device="cpu"):
    """a python float of every rank, in rank order, on every rank (per-rank diagnostics of the bench line)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_frame_results(local, num_frames, dist=None, device="cpu"):
    """local: dict {frame index: 1-D int64 tensor} for the frames this rank owns.  Returns on
    every rank the list of per-frame tensors in frame order (None if no process group: local only).
    Small host-side gather (<= 300 detections x a few numbers per frame), never on the hot path."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local[f] for f in range(num_frames)]
    world = dist.get_world_size()
    width = max([int(v.numel()) for v in local.values()] + [0])
    wt = torch.tensor([width], dtype=torch.int64, device=device)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    width = int(wt.item())
    per_rank = (num_frames + world - 1) // world
    buf = torch.full((per_rank, width + 1), -1, dtype=torch.int64, device=device)
    for slot, f in enumerate(frame_shard(num_frames, dist.get_rank(), world)):
        v = local[f].to(device=device, dtype=torch.int64).reshape(-1)
        buf[slot, 0] = v.numel()
        buf[slot, 1:1 + v.numel()] = v
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = []
    for f in range(num_frames):
        row = out[owner_of(f, world)][f // world]
        res.append(row[1:1 + int(row[0])].cpu())
    return res


class GradBucketer:
    """Data-parallel gradient exchange for the training step (one all-reduce of the gradients per step,
    ~143 M fp32 for MV3D, bucketed, in reverse layer order so that it overlaps the backward pass).

    * `params` in forward order; they are packed LAST LAYER FIRST into flat fp32 buffers of ~`bucket_bytes` (25 MB: over
      xGMI a ring all-reduce is bound per link -- 7 links x ~153 GB/s per GPU -- and 25 MB buckets keep every link busy
      without making the first bucket wait for half of the backward pass), and each parameter's `.grad` is made a VIEW
      into its bucket: autograd accumulates straight into the buffer that goes on the wire, no copies.
    * a post-accumulate hook per parameter counts arrivals; when a bucket is complete its all-reduce (sum) is launched
      asynchronously (`async_op=True`: RCCL runs it on its own stream while backward continues).
    * `finish()` waits for the handles and divides by the world size (mean gradient = what a single process would compute
      on the concatenated batch).  With no process group the class is a no-op: `zero_grad()` then simply drops the gradients.

    Pure torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests); no data-path collective of the
    hot path itself is involved -- frames are independent."""

    def __init__(self, params, dist=None, bucket_bytes=25 << 20, average=True, allocate=None):
        self.dist = dist if (dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.world = self.dist.get_world_size() if self.dist is not None else 1
        self.average = average
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []                   # dict(flat, params, pending, handle)
        self._bucket_of = {}
        cur, size = [], 0
        if self.dist is None and not allocate:
            # single process: nothing goes on a wire -- no flat buffers (one f32 copy of every parameter, 856 MB for the 3-view
            # graph, would sit unused: zero_grad() drops the gradients instead of zeroing views).  allocate=True builds the
            # buckets anyway (tests of the packing)
            self._hooks = []
            self.dist_enabled = False
            return
        for p in reversed(self.params):     # last layer first: its gradient is ready first
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._close(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self._close(cur)
        self._hooks = [p.register_post_accumulate_grad_hook(self._arrived) for p in self.params]
        self.dist_enabled = True            # False while earlier frames of an accumulated step run backward
        self.reset()

    def _close(self, plist):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self._bucket_of[id(p)] = len(self.buckets)
        self.buckets.append({"flat": flat, "params": plist, "pending": len(plist), "handle": None, "events": []})

    def reset(self):
        """before every backward pass (gradients are zeroed in place: the views must stay attached)"""
        for b in self.buckets:
            b["pending"], b["handle"], b["events"] = len(b["params"]), None, []
        self._next = 0                      # buckets are launched strictly in index order (see _arrived)

    def zero_grad(self):
        if self.dist is None:
            # single process: nothing goes on a wire, so the gradients need not live in the flat buffers -- dropping them lets
            # autograd STORE each gradient instead of adding it to a zeroed view (one read + one write of every parameter's
            # gradient less per step: ~1 ms of the 14 ms mixed-precision step of the 214 M-parameter graph)
            # (A parameter that receives no gradient in a step then has grad None and Adam SKIPS it -- moments frozen -- whereas
            # under data parallelism its zeroed bucket view makes Adam decay its moments; every parameter of the MV3D graphs
            # receives a gradient every step, so the two paths take the same updates.)
            for p in self.params:
                p.grad = None
            return
        for b in self.buckets:
            b["flat"].zero_()

    def _launch_ready(self):
        # Collectives are matched across ranks by LAUNCH ORDER: bucket i goes out only after buckets 0 .. i - 1 have, on every
        # rank, whatever order the hooks fire in and even if a data-dependent branch left some parameter without a gradient on
        # one rank (its bucket then waits for finish(), and so does everything behind it).
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            b = self.buckets[self._next]
            self._fence(b)
            b["handle"] = self.dist.all_reduce(b["flat"], op=self.dist.ReduceOp.SUM, async_op=True)
            self._next += 1

    def _fence(self, b):
        # the collective is ordered after the stream it is launched from; gradients of the bucket that were accumulated on OTHER
        # streams (a network may run independent branches of its backward pass on side streams) are fenced in by their events
        if b["events"]:
            cur = torch.cuda.current_stream(b["flat"].device)
            for ev in b["events"]:
                cur.wait_event(ev)
            b["events"] = []

    def _arrived(self, p):
        b = self.buckets[self._bucket_of[id(p)]]
        b["pending"] -= 1
        if self.dist is not None and p.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(p.device))       # the stream this gradient was accumulated on
            b["events"].append(ev)
        if b["pending"] == 0 and self.dist is not None and self.dist_enabled:
            self._launch_ready()

    def finish(self):
        """after backward: every bucket reduced (the buckets not launched during backward go out here, in index order)"""
        if self.dist is None:
            return
        for b in self.buckets[self._next:]:
            self._fence(b)
            b["handle"] = self.dist.all_reduce(b["flat"], op=self.dist.ReduceOp.SUM, async_op=True)
        self._next = len(self.buckets)
        for b in self.buckets:
            b["handle"].wait()
            if self.average:
                b["flat"].div_(self.world)

    def total_bytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)

    def close(self):
        for h in self._hooks:
            h.remove()
