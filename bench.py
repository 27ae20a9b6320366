#!/usr/bin/env python3
"""bench.py -- KITTI-shape frames/s through the MV3D hot path on MI355X (BASELINE.json's metric).

Default workload = BASELINE configs[2], path only (the largest single-GPU configuration):

    batch of 2 frames, three views, training step of the path
      mv3d_proposal_3d           TRAIN cfg: 23 104 BEV anchors -> 12 000 pre-NMS -> NMS 0.7 -> 2 000 proposals / frame
      mv3d_anchor_target_*       RPN labels / 6-d targets of every frame
      mv3d_proposal_target_*     <= 128 sampled ROIs / frame (fg first), corner targets, image boxes
      mv3d_rois_3d_to_fv         third (front-view) ROIs
      mv3d_roi_pool_forward_views_pair   BEV 76x76x512 + RGB 46x155x512 + FV 8x64x512, 7x7 (private one-byte argmax plane)
      mv3d_roi_pool_backward_views_pair  the same three layers, RoiPoolGrad (candidate index + zero fill, ordered gather)

A "step" = one pass over `--batches-per-step` such batches (default 1664 batches = 3328 frames: `--steps 20` is a SUSTAINED
~5 s timed region, not a burst), cycling through a ring of `--ring` (default 16) DISTINCT batches per GPU -- distinct frames
and distinct feature maps (1.1 GB of maps + 3.3 GB of outputs per ring), so that nothing is served from the 256 MB Infinity
Cache by re-reading the previous step's frame.

How a batch is driven (`--launch`):
  path (default)  the way a caller drives it: the library's own training-path object (`mv3d_train_path_*`, csrc/train_stream.hip,
      through mv3d_tf_amd.train_path.TrainPathStream) on the ring's inputs -- ONE submit and ONE finish call per batch; the counts of
      the candidate lists travel to the host, the library's helper thread draws the subsamples on numpy's global generator exactly as
      the reference draws them (INSIDE the timed region), uploads the lists and enqueues stage 2 -- then RoiPool forward + backward of
      the three views on the batch's sampled ROIs.  `--streams` (default 8) batches in flight, one HIP stream per slot.  Inputs (RPN
      heads, ground truth, feature maps) are resident in HBM before the timed region; nothing else of a batch is precomputed.
  graph / eager   the frozen-batch replay rounds 1 - 3 headlined (mv3d_tf_amd/hot_path.py): the index lists of the two target layers are
      drawn during set-up (from the numpy RNG, exactly as the reference draws them) and resident, every batch is one hipGraph (or its
      eager launches) on one of `--streams` (default 3) streams, no host stage in the timed region.  With `--launch path` this figure
      is still taken, for three steps, and reported as `secondary.resident_replay`.
The VGG16 trunks / FC head are not part of the path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train|test] [--ring R] [--batches-per-step M]

`--gpus N` with N > 1 and no launcher environment re-executes itself under `python -m torch.distributed.run` with one
rank per GPU (RCCL); under the driver's own torchrun launch the ranks come from RANK / LOCAL_RANK / WORLD_SIZE.  Frames
shard across ranks with no data-path collective ("weak" scaling); the bracketing barrier and the max-over-ranks of the
timed region are the only collectives of the path-only line.

Rank 0 prints ONE JSON line.  `roofline` = the dominant kernel of the step (largest share of GPU time), measured live
with HIP events on the stream it is launched on; `roofline_kernels` lists the RoiPool forward and backward launches
separately; `cpu_baseline` = the C oracle on the same workload on the host cores (bounded sample); `secondary` holds
`fresh_inputs` (the same training path on NEW frames every batch: mv3d_tf_amd.train_path.TrainPathStream, host draws in the
loop, pipelined), the TEST-cfg (configs[4]) line, `config1_latency` (configs[1]: one frame, BEV-only RPN + NMS with the trunk on the
MFMA convolution) and `with_trunk`, the full training step with the torch VGG16 trunks and
the bucketed gradient all-reduce (--no-trunk skips it).
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
METRIC = "KITTI-shape frames/sec (RPN+3-view ROI-pool+NMS) at 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="train", choices=["train", "test"],
                    help="train: BASELINE configs[2] path-only (default); test: configs[4] per-GPU path (batch 16, TEST cfg)")
    ap.add_argument("--batch", type=int, default=0, help="frames per batch per GPU (default: 2 for train, 16 for test)")
    ap.add_argument("--ring", type=int, default=0, help="distinct resident batches per GPU (default: 16 train, 4 test)")
    ap.add_argument("--batches-per-step", type=int, default=0, help="batches per step (default: 1664 train, 208 test)")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams = batches in flight (default: 8 for --launch path, 3 otherwise)")
    ap.add_argument("--launch", default="path", choices=["path", "graph", "eager"],
                    help="path (train workload's default): the product's own driver, mv3d_train_path + RoiPool, on the ring's inputs -- "
                         "nothing of a batch precomputed, the subsampling draws inside the timed region; graph: every ring batch "
                         "(index lists drawn during set-up) is captured once into a hipGraph on its stream and replayed; eager: the "
                         "same frozen batches as plain in-order launches from Python")
    ap.add_argument("--variant", default="peaky", choices=["peaky", "rand"])
    ap.add_argument("--test-argmax", action="store_true", help="--workload test / secondary.test_cfg: RoiPool writes the op's argmax plane too "
                    "(default: the inference graph's call, top only -- nothing reads argmax without a backward pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--with-trunk", action="store_true", help="(default since r03; kept for old command lines)")
    ap.add_argument("--no-trunk", action="store_true", help="skip secondary.with_trunk (the full training step with the torch VGG16 trunks)")
    ap.add_argument("--no-fresh", action="store_true", help="skip secondary.fresh_inputs")
    ap.add_argument("--secondary-seconds", type=float, default=1.0, help="minimum timed window of every with-trunk leg (training / serving step variants)")
    ap.add_argument("--secondary-timeout", type=int, default=1500, help="seconds the secondary legs may take before the line is printed without the rest")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) | gloo (logic tests)")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: run N ranks under torch.distributed.run and relay rank 0's line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Ring:
    """`n` distinct device-resident batches of one rank, batch k bound to stream k % S."""

    def __init__(self, args, rank, workload, batch, n, streams):
        from mv3d_tf_amd import hot_path, synth
        self.slots = []
        self.host_frames_all = []
        dev = torch.device("cuda", torch.cuda.current_device())
        np.random.seed(3 + rank)                                   # cfg.RNG_SEED (tools/train_net.py:78-80), per rank
        for k in range(n):
            st = streams[k % len(streams)]
            seed0 = 100000 * (rank + 1) + k * batch
            frames = [synth.rpn_head(seed0 + b, 76, 76, args.variant, return_gt=True) for b in range(batch)]
            maps = hot_path.synth_maps(batch, seed0, dev)
            if workload == "train":
                slot = hot_path.TrainPathBatch(frames, maps, stream=st, top_diff_seed=k, cold_maps=True)   # ring maps: written long ago
            else:
                slot = hot_path.TestPathBatch([f[:4] for f in frames], maps, stream=st, cold_maps=True,
                                              want_argmax=bool(getattr(args, "test_argmax", False)))
            self.slots.append(slot.setup())
            self.host_frames_all.append(frames)
            if k == 0:
                self.host_frames = frames
        self.cursor = 0
        self.play = [s.run for s in self.slots]
        self.driver = None
        if args.launch == "path" and workload == "train":
            # the PRODUCT's own driver (mv3d_train_path, csrc/train_stream.hip) on the ring's inputs and maps: nothing of a batch is
            # precomputed -- stage 1, the counts' trip to the host, the numpy-global-RNG draws, stage 2, RoiPool forward + backward
            self.driver = PathDriver([(s.prob, s.pred, s.info, s.calib, s.gt) for s in self.slots], [s.maps for s in self.slots],
                                     depth=max(1, args.streams))
        if args.launch == "graph":
            self.make_graphs()

    def make_graphs(self):
        """every ring batch captured once into a hipGraph on its stream; run() then replays (the resident-replay figure)"""
        self.graphs = []
        for s in self.slots:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s.stream):               # the bound launches target s.stream = the capture stream
                s.run()
            self.graphs.append(g)
        torch.cuda.synchronize()

        def player(g, st):
            def play():
                with torch.cuda.stream(st):                           # CUDAGraph.replay() launches on the current stream
                    g.replay()
            return play
        self.play = [player(g, s.stream) for g, s in zip(self.graphs, self.slots)]
        self.driver = None

    def run(self, nbatches):
        if self.driver is not None:
            return self.driver.run(nbatches)
        n = len(self.slots)
        for _ in range(nbatches):
            self.play[self.cursor % n]()
            self.cursor += 1


def masked_streams(n, mode):
    """Experiment ("forward beside company"): the path's streams restricted to HALVES of the chip by
    hipExtStreamCreateWithCUMask -- even streams one half, odd streams the other -- so that two batches' chip-filling RoiPool launches run
    side by side on disjoint compute units instead of interleaving on all of them.  mode "words": compute units 0-127 / 128-255 of the
    mask's numbering; "bits": even / odd compute units.  Returns None (plain streams) for any other mode.  profiles/r06_h_cu_mask_ab.txt."""
    if mode not in ("words", "bits"):
        return None
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    words = (ncu + 31) // 32
    out = []
    for i in range(n):
        half = i & 1
        if mode == "words":
            m = [0xffffffff if (w < words // 2) == (half == 0) else 0 for w in range(words)]
        else:
            m = [0x55555555 if half == 0 else 0xaaaaaaaa for _ in range(words)]
        arr = (ctypes.c_uint32 * words)(*m)
        h = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), arr)
        if rc != 0:
            raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
        out.append(torch.cuda.ExternalStream(h.value))
    return out


class PathDriver:
    """The training path as a caller drives it: `depth` batches in flight through mv3d_tf_amd.train_path.TrainPathStream (one
    submit / finish pair of the C object mv3d_train_path per batch; the library's helper thread draws on numpy's global generator,
    draw for draw the reference's), slot k on its own HIP stream, RoiPool forward + backward of the three views on the batch's
    sampled ROIs behind it on the same stream.  Inputs and maps are the caller's resident tensors (a ring of distinct batches)."""

    def __init__(self, inputs, maps, depth, cold=True, seed=5):
        import ctypes as C
        from mv3d_tf_amd import hot_path
        from mv3d_tf_amd._lib import RoiGradView, RoiView, check, lib
        from mv3d_tf_amd.train_path import TrainPathStream
        self.inputs, self.maps, self.depth = inputs, maps, int(depth)
        dev = inputs[0][0].device
        B = int(inputs[0][0].shape[0])
        self.B = B
        self.streams = masked_streams(self.depth, os.environ.get("MV3D_BENCH_CU_MASK", "")) or [torch.cuda.Stream() for _ in range(self.depth)]
        self.path = TrainPathStream(B, int(inputs[0][0].shape[1]), int(inputs[0][0].shape[2]), dev, depth=self.depth, streams=self.streams)
        cap = self.cap = B * self.path.roi_cap
        g = torch.Generator(device=dev).manual_seed(seed)
        self.bufs = []
        for _ in range(self.depth):
            d = {}
            for v in hot_path.VIEWS:
                H, W, Cc = hot_path.VIEW_MAPS[v]
                d[v] = (torch.empty((cap, 7, 7, Cc), device=dev), torch.empty((cap, 7, 7, Cc), dtype=torch.int32, device=dev),
                        torch.rand((cap, 7, 7, Cc), generator=g, device=dev) * 2.0 - 1.0, torch.empty((B, H, W, Cc), device=dev))
            self.bufs.append(d)
        self.C, self.L, self.check, self.views = C, lib(), check, hot_path.VIEWS
        self.cold = 1 if cold else 0
        self.RoiView, self.RoiGradView = RoiView, RoiGradView
        self.structs = {}
        self.cursor = 0
        self.rows = 0

    def roi(self, out, k, marks=None):
        """The RoiPool pair (forward with the private compact argmax plane; backward = index + fill, gather) on the batch's ROIs: the
        argument structs of a (maps, slot) combination are built once (every buffer has the slots' fixed capacity), a batch only
        sets its row count -- a caller's steady state, no per-batch slicing.  marks: {"fwd": [], "bwd": []} gets a HIP event pair
        per call, recorded on the batch's own stream (the calls' durations with the other batches in flight)."""
        C, L, NV = self.C, self.L, len(self.views)
        St = out["rois"]["bev"].shape[0]
        j = k % self.depth                                       # the slot (and its buffers / stream) this batch went through
        key = (k % len(self.inputs), j)
        hit = self.structs.get(key)
        if hit is None:
            maps, d = self.maps[key[0]], self.bufs[j]
            fwd, bwd = (self.RoiView * NV)(), (self.RoiGradView * NV)()
            for i, v in enumerate(self.views):
                Bm, H, W, Cc = maps[v].shape
                r = out["rois"][v].data_ptr()
                fwd[i] = self.RoiView(maps[v].data_ptr(), r, d[v][0].data_ptr(), d[v][1].data_ptr(), 0.125, Bm, self.cap, H, W, Cc)
                bwd[i] = self.RoiGradView(d[v][3].data_ptr(), r, d[v][2].data_ptr(), d[v][1].data_ptr(), 0.125, Bm, self.cap, H, W, Cc)
            ws = self.bufs[j].get("ws")
            if ws is None:
                wsz = L.mv3d_roi_pool_pair_workspace_bytes(NV, bwd, 7, 7)           # (at the slots' capacity; zeroed once: the entry's contract)
                ws = self.bufs[j]["ws"] = torch.zeros(max(wsz, 256), dtype=torch.uint8, device=maps[self.views[0]].device)
            hit = self.structs[key] = (fwd, bwd, ws)
        fwd, bwd, ws = hit
        for i in range(NV):
            fwd[i].num_rois = St
            bwd[i].num_rois = St
        stream = out["stream"]
        st = C.c_void_p(stream.cuda_stream)
        # RoiPoolGrad without a workspace = ONE launch of LDS map tiles (csrc/roi_grad_tiles.hip): +3 % over index + gather with eight
        # batches in flight (profiles/r05_as_bench_ws_ab.txt); MV3D_BENCH_ROI_GRAD_WS=1 takes the workspace path for an A / B
        wp, wn = (C.c_void_p(ws.data_ptr()), ws.numel()) if os.environ.get("MV3D_BENCH_ROI_GRAD_WS") else (None, 0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if marks is not None else None
        if ev:
            ev[0].record(stream)
        self.check(L.mv3d_roi_pool_forward_views_pair(NV, fwd, 7, 7, self.cold, st), "mv3d_roi_pool_forward_views_pair")
        if ev:
            ev[1].record(stream)
        self.check(L.mv3d_roi_pool_backward_views_pair(NV, bwd, 7, 7, wp, wn, st), "mv3d_roi_pool_backward_views_pair")
        if ev:
            ev[2].record(stream)
            marks["fwd"].append((ev[0], ev[1]))
            marks["bwd"].append((ev[1], ev[2]))
        self.rows += St

    def run(self, nb, capture=None, marks=None):
        path, n, k0 = self.path, len(self.inputs), self.cursor
        flight = [path.submit(*self.inputs[(k0 + j) % n]) for j in range(min(self.depth - 1, nb))]
        for i in range(nb):
            if i + self.depth - 1 < nb:
                flight.append(path.submit(*self.inputs[(k0 + i + self.depth - 1) % n]))
            out = path.finish(flight.pop(0))
            self.roi(out, k0 + i, marks)
            if capture is not None:
                capture.append((k0 + i, out))
        self.cursor = k0 + nb

    def in_flight_us(self, nb=64):
        """average duration of the RoiPool forward / backward calls as the timed loop runs them: `depth` batches in flight, HIP
        event pairs on each batch's own stream around the two calls"""
        marks = {"fwd": [], "bwd": []}
        self.run(2 * self.depth)
        torch.cuda.synchronize()
        self.run(nb, marks=marks)
        torch.cuda.synchronize()
        skip = self.depth                                          # (the pipeline's ramp)
        avg = lambda prs: sum(a.elapsed_time(b) for a, b in prs[skip:]) / max(1, len(prs) - skip) * 1e3
        return {"forward_us": round(avg(marks["fwd"]), 2), "backward_us": round(avg(marks["bwd"]), 2), "batches_in_flight": self.depth,
                "calls_timed": len(marks["fwd"]) - skip}

    def verify(self, host_frames, nbatches=2, seed=1234):
        """What the timed loop computes, checked AFTER it: `nbatches` more batches through this very driver (same slots, buffers,
        argument structs, streams) under a known numpy seed; their ROIs, top / argmax planes and bottom_diff are read back and
        compared, bit for bit, with the oracle run on the same inputs with the same seed (frame by frame, anchor draws before
        proposal draws: the reference's order).  Returns the `verified` object of the bench line."""
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        from mv3d_tf_amd import hot_path
        oracle.build()
        torch.cuda.synchronize()
        nbatches = min(nbatches, self.depth)                      # (every checked batch in its own slot: nothing is overwritten)
        np.random.seed(seed)
        cap = []
        self.run(nbatches, capture=cap)
        torch.cuda.synchronize()
        from mv3d_tf_amd.fast_rcnn.config import cfg
        o_train = dict(oracle.TRAIN, BG_THRESH_LO=float(cfg.TRAIN.BG_THRESH_LO), BG_THRESH_HI=float(cfg.TRAIN.BG_THRESH_HI),
                       FG_THRESH=float(cfg.TRAIN.FG_THRESH))
        np.random.seed(seed)
        bad, rows = [], 0
        for k, out in cap:
            frames = host_frames[k % len(self.inputs)]
            j = k % self.depth
            want = {"bev": [], "rgb": [], "fv": []}
            for b, (prob, pred, info, calib, (gt_bv, gt_3d, gt_cnr)) in enumerate(frames):
                oracle.anchor_target_layer(np.zeros((1, prob.shape[1], prob.shape[2], 8), np.float32), gt_bv, gt_3d, info, [8, ])
                bv, img, b3 = oracle.proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ], cfg={"TRAIN": hot_path.TRAIN_CFG})
                r_bv, r_img, _, _, r_3d = oracle.proposal_target_layer_3d(bv, b3, gt_bv, gt_3d, gt_cnr, calib, 2, train=o_train)
                for a in (r_bv, r_img, r_3d):
                    a[:, 0] = b
                want["bev"].append(r_bv); want["rgb"].append(r_img); want["fv"].append(oracle.rois_3d_to_fv(r_3d))
            St = out["rois"]["bev"].shape[0]
            rows += St
            # the pair keeps its argmax plane as private one-byte codes: decoded to the reference's int32 plane for the comparison
            from mv3d_tf_amd import ops
            mp = self.maps[k % len(self.inputs)]
            fviews = [(mp[v], out["rois"][v], 0.125) for v in self.views]
            with torch.cuda.stream(out["stream"]):
                dec = ops.roi_pool_argmax_decode(fviews, [(self.bufs[j][v][0][:St], self.bufs[j][v][1]) for v in self.views], 7, 7)
            torch.cuda.synchronize()
            for vi, v in enumerate(self.views):
                rois = np.concatenate(want[v])
                if rois.shape[0] != St or not np.array_equal(out["rois"][v].cpu().numpy(), rois):
                    bad.append("batch %d rois_%s" % (k, v))
                    continue
                m = mp[v].cpu().numpy()
                top, _, td, bd = (t.cpu().numpy() for t in self.bufs[j][v])
                am = dec[vi].cpu().numpy()
                o_top, o_am = oracle.roi_pool(m, rois, 7, 7, 0.125)
                o_bd = oracle.roi_pool_grad(m, rois, o_am, td[:St], 7, 7, 0.125)
                for name, a, b_ in (("top", top[:St], o_top), ("argmax", am, o_am), ("bottom_diff", bd, o_bd)):
                    if not np.array_equal(a, b_):
                        bad.append("batch %d %s_%s" % (k, name, v))
        return {"batches": len(cap), "rows": rows, "bit_exact": not bad, "mismatches": bad[:8],
                "what": "after the timed region: %d more batches through the SAME PathDriver (depth %d, its slots / buffers / argument "
                        "structs), numpy seed %d; rois of the 3 views, RoiPool top + argmax, RoiPoolGrad bottom_diff read back and compared "
                        "with oracle/ on the same inputs, seed and draw order" % (len(cap), self.depth, seed)}

    def close(self):
        self.path.close()


def events_ms(stream, fns, rounds):
    """average duration (ms) of one call of the launches in `fns` (cycled), HIP events recorded on `stream`"""
    with torch.cuda.stream(stream):
        for f in fns:
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(rounds):
            for f in fns:
                f()
        e1.record(stream)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (rounds * len(fns))


def pmc_signature(workload, batch, rows, variant, launches):
    """What a PMC pass is keyed by: the workload AND the launch set behind the timed calls -- `launches` names the kernel variant and
    the outputs it writes (train: "pair-tiles-planned" = forward with one-byte codes + planning workgroup, the one-launch RoiPoolGrad on its work list; test: "top-only" /
    "top+argmax").  A table collected for another variant of the same workload must not be quoted (the r03 pass of the
    forward that also wrote the int32 plane was printed next to the top-only kernel)."""
    return "%s/b%d/r%d/%s/%s" % (workload, batch, rows, variant, launches)


def pmc_traffic(kernel, signature):
    """HBM bytes per launch from a committed PMC pass OF THIS EXACT CONFIGURATION (profiles/r0N_pmc_traffic.json: one table per
    signature it was collected with, pmc_signature()); None otherwise -- never a stale number."""
    try:
        table = None
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):     # the newest pass that holds this configuration
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                table = json.load(open(path))["signatures"].get(signature)
            if table:
                break
        if not table:
            return None
        keys = [k for k in table["kernels"] if kernel in k]          # RoiPoolGrad = three kernels behind one call: summed
        return int(sum(table["kernels"][k]["hbm_bytes_per_launch"] for k in keys)) if keys else None
    except Exception:
        return None


def roofline_entries(ring, workload, signature):
    """RoiPool forward / backward calls (all three views each) timed INSIDE the batch's own launch sequence: every batch of
    stream 0 is replayed eagerly with a HIP event pair on that stream around the two calls, over several rounds of the
    stream's ring batches (distinct maps / outputs every time, the ~26 small launches of the batch in between) -- so the
    durations are the ones a kernel trace of the bench shows (profiles/r02_*_kernel_stats.txt), not the back-to-back rate of
    a cache-warm loop."""
    s0 = ring.slots[0].stream
    mine = [s for s in ring.slots if s.stream is s0]
    if workload == "train":
        # the RoiPool pair: forward with one-byte argmax codes; RoiPoolGrad = ONE launch of LDS map tiles (no workspace)
        legs = [("roi_pool_fwd_pair_cold_kernel" if getattr(mine[0], "cold_maps", False) else "roi_pool_fwd_pair_kernel",
                 "mv3d_roi_pool_forward_views_pair", "roi_forward_bytes"),
                ("roi_pair_tiles_kernel", "mv3d_roi_pool_backward_views_pair", "roi_backward_bytes")]
    else:
        legs = [("roi_pool_fwd_xcd_multi%s_kernel" % ("_cold" if getattr(mine[0], "cold_maps", False) else ""),
                 mine[0].fwd_fn.__name__, "roi_forward_bytes")]
    marks = {fn: [] for _, fn, _ in legs}
    torch.cuda.synchronize()
    with torch.cuda.stream(s0):
        for rnd in range(5):
            for s in mine:
                s.bound.run_marked(s0, marks if rnd else {})      # round 0 = warm-up
    torch.cuda.synchronize()
    out = []
    for kname, fn, bytes_meth in legs:
        if not marks[fn]:                                         # (diagnostic replays that drop the call: MV3D_SKIP)
            continue
        ms = sum(a.elapsed_time(b) for a, b in marks[fn]) / len(marks[fn])
        alg = getattr(mine[0], bytes_meth)()
        gbs = alg / (ms * 1e-3) / 1e9
        out.append({"kernel": "%s (BEV 76x76x512 + RGB 46x155x512 + FV 8x64x512 views, R=%d rows each, batch %d)"
                              % (kname, mine[0].num_rois, mine[0].B),
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("roi_pair_" if "roi_pair_" in kname else kname, signature),
                    "alg_bytes_per_launch": int(alg), "avg_launch_us": round(ms * 1e3, 2), "launches_timed": len(marks[fn])})
        moved = getattr(mine[0], bytes_meth.replace("_bytes", "_moved_bytes"), None)
        if moved is not None:
            # (`achieved` prices the reference op's 8 B per pooled value; the pair's own minimum is 5 B -- its real bandwidth)
            out[-1]["moved_bytes_per_launch"] = int(moved())
            out[-1]["moved_frac"] = round(moved() / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return out


def nms_roofline_entries(variant="peaky", reps=30):
    """SURVEY 8(d) for the greedy NMS: "achieved time vs the serial-chain lower bound".  One bench frame's pre-NMS boxes in processing
    order (the product's own decode + rank: proposal_3d with a threshold nothing reaches), TRAIN cfg (12000 -> 2000 @ 0.7) and TEST cfg
    (6000 -> 300): the call's duration from HIP events on the launch stream, and from mv3d_nms_device_trace the number of 64-box blocks
    the chain visited and the cycles of its fastest link -- floor = blocks x fastest link: what the chain would take if every link ran
    at the speed of its best one and nothing else (tile phase, launches of later rounds) took time."""
    import ctypes as C
    from mv3d_tf_amd import ops, synth
    from mv3d_tf_amd._lib import check, lib
    out = []
    # shader clock for cycles -> time: the device's current clock if the runtime reports one, else the part's 2.4 GHz peak engine clock
    clk_hz, clk_src = 2.4e9, "MI355X peak engine clock (assumed)"
    try:
        clk_hz, clk_src = float(torch.cuda.clock_rate()) * 1e6, "torch.cuda.clock_rate()"
    except Exception:
        pass
    prob, pred, info, calib = synth.rpn_head(1000, 76, 76, variant)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    for name, pre, post in (("TRAIN cfg 12000 -> 2000", 12000, 2000), ("TEST cfg 6000 -> 300", 6000, 300)):
        prm = ops.proposal_params(dict(RPN_PRE_NMS_TOP_N=pre, RPN_POST_NMS_TOP_N=pre, RPN_NMS_THRESH=2.0, RPN_MIN_SIZE=5))
        bv, _, _, num, _ = ops.proposal_3d(t(prob), t(pred), t(info), t(calib[None]), prm)
        k = int(num[0])
        d = torch.zeros((k, 5), dtype=torch.float32, device="cuda")
        d[:, :4] = bv[0, :k, 1:5]
        d[:, 4] = torch.linspace(1, 0, k, device="cuda")
        for _ in range(3):
            keep, cnt, _ = ops.nms_device(d, 0.7, post)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        evs[0].record()
        for i in range(reps):
            ops.nms_device(d, 0.7, post)
            evs[i + 1].record()
        torch.cuda.synchronize()
        us = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]) * 1e3
        nb = (k + 63) // 64
        keep = torch.empty((k,), dtype=torch.int32, device="cuda")
        c2 = torch.zeros((2,), dtype=torch.int32, device="cuda")
        tr = torch.zeros((nb * 4 + 8,), dtype=torch.int64, device="cuda")
        ws = torch.empty((lib().mv3d_nms_workspace_bytes(k),), dtype=torch.uint8, device="cuda")
        for _ in range(2):
            check(lib().mv3d_nms_device_trace(C.c_void_p(d.data_ptr()), k, 0.7, int(post), C.c_void_p(keep.data_ptr()), C.c_void_p(c2.data_ptr()),
                                              C.c_void_p(c2[1:].data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), None, C.c_void_p(tr.data_ptr())), "mv3d_nms_device_trace")
            torch.cuda.synchronize()
        tt = tr.cpu().numpy()[:nb * 4].reshape(nb, 4)
        tt = tt[tt[:, 2] != 0]
        # links between consecutive blocks of ONE launch (later rounds are other launches on other compute units: their cycle counters
        # are not comparable, such differences are dropped)
        step = np.diff(tt[:, 0]) if len(tt) > 1 else np.array([1])
        step = step[(step > 0) & (step < 200000)]
        if len(step) == 0:
            step = np.array([1])
        floor_us = len(tt) * float(step.min()) / clk_hz * 1e6
        out.append({"kernel": "greedy NMS, %s @ 0.7, one frame of %d boxes (nms_tiles_kernel + nms_chain_lds_kernel + nms_round_kernel<W>)" % (name, k),
                    "bound": "latency (serial greedy chain)", "measured_us": round(float(np.median(us)), 2), "measured_us_min": round(float(us.min()), 2),
                    "blocks_total": int(nb), "blocks_visited": int(len(tt)), "kept": int(c2[0]),
                    "link_cycles_min": int(step.min()), "link_cycles_median": int(np.median(step)),
                    "chain_cycles": int(step.sum()), "clock_mhz": round(clk_hz / 1e6, 1), "clock_source": clk_src,
                    "chain_floor_us": round(floor_us, 2), "frac": round(floor_us / float(np.median(us)), 4),
                    "note": "floor = blocks visited x the fastest link of the chain (shader cycles / clock); frac = floor / measured call"})
    return out


def usable_cores():
    """host cores this process may really use: the affinity mask, cut by the container's CPU quota (cgroup v2 cpu.max / v1
    cfs_quota_us): 256 worker processes on a 16-CPU quota just time-slice (measured: half the rate of 64 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = ""
    try:
        quota = period = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota, period = (None if q == "max" else float(q)), float(p)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if quota <= 0 else quota
        if quota is not None and period:
            lim = max(1, int(quota / period + 0.5))
            if lim < n:
                note = ", CPU quota of the container: %d of them" % lim
                n = lim
    except (OSError, ValueError):
        pass
    return max(1, n), note


def cpu_baseline(ring, workload, seconds):
    """The C oracle on the same per-frame workload (frame 0 of batch 0), bounded sample, as SURVEY.md section 8(d) defines the
    CPU baseline: one thread first, then ALL host cores -- one worker PROCESS per core (oracle/cpu_worker.py: its own numpy
    global RNG for the subsampling draws, no lock, no shared GIL), every worker processing whole frames between a common
    start and a common stop time."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_worker
    import oracle
    from mv3d_tf_amd import hot_path
    oracle.build()
    prob, pred, info, calib, (gt_bv, gt_3d, gt_cnr) = ring.host_frames[0]
    slot = ring.slots[0]
    key = "TRAIN" if workload == "train" else "TEST"
    cfgd = hot_path.TRAIN_CFG if workload == "train" else hot_path.TEST_CFG
    arrays = dict(prob=prob, pred=pred, info=info, calib=calib, gt_bv=gt_bv, gt_3d=gt_3d, gt_cnr=gt_cnr)
    arrays.update({"map_" + v: m[0:1].cpu().numpy() for v, m in slot.maps.items()})
    arrays.update({"cfg_" + k: np.asarray(v) for k, v in cfgd.items()})
    tmp = tempfile.mkdtemp(prefix="mv3d_cpu_")
    path = os.path.join(tmp, "inputs.npz")
    np.savez(path, **arrays)
    # ---- one thread, in this process
    frame = cpu_worker.make_frame(np.load(path), workload)
    frame()
    n1, t0 = 0, time.perf_counter()
    while True:
        frame()
        n1 += 1
        dt1 = time.perf_counter() - t0
        if dt1 >= seconds / 3 or n1 >= 400:
            break
    # ---- every host core: one process each
    cores, quota_note = usable_cores()
    span = 2 * seconds / 3
    start = time.time() + 3.0 + cores * 0.05              # (interpreter + numpy start-up of `cores` processes)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), path, workload, repr(start),
                               repr(start + span), str(1000 + k)], stdout=subprocess.PIPE, env=env, text=True) for k in range(cores)]
    done, late = [], 0
    for p in procs:
        out, _ = p.communicate(timeout=120 + 10 * seconds)
        try:
            n, t = out.split()
            done.append((int(n), float(t)))
        except ValueError:
            late += 1
    nm = sum(n for n, _ in done)
    dtm = max([t for _, t in done] + [span])
    try:
        os.remove(path)
        os.rmdir(tmp)
    except OSError:
        pass
    return {"value": round(nm / dtm, 3), "unit": "frames/s", "cores": len(done), "kind": "port",
            "one_thread_frames_per_s": round(n1 / dt1, 3),
            "sample": "%d frames by %d worker processes (one per usable host core; %d hardware threads visible%s%s) in %.1f s after %d frames on 1 thread "
                      "in %.1f s; the same per-frame workload (%s cfg path incl. target layers, 3-view RoiPool fwd%s, frame 0 of the "
                      "ring); C restatement oracle/mv3d_oracle.c, gcc -O2 -ffp-contract=off.  Reference as shipped (its Python / "
                      "Cython proposal_layer_3d alone, one thread, survey container, BASELINE.md section 2): 1.24 - 1.31 s per frame = "
                      "0.8 frames/s" % (nm, len(done), os.cpu_count() or 0, quota_note, (", %d workers failed" % late) if late else "", dtm, n1, dt1,
                                        key, "+bwd" if workload == "train" else "")}


def fresh_inputs_line(rank, variant, seconds=1.5, depth=3, nstreams=3):
    """The training path on FRESH inputs: every batch brings new RPN heads / ground truth / feature maps (a pool of 8 resident
    input batches, each processed as new: stage 1 -> counts to the host -> numpy-global-RNG draws -> index lists back -> stage
    2 -> RoiPool forward + backward on its ROIs), batch i + 1 submitted before batch i is finished so that the device never
    waits for the host's draws of the batch it is working on."""
    from mv3d_tf_amd import hot_path, ops, synth
    from mv3d_tf_amd.train_path import TrainPathStream
    dev = torch.device("cuda", torch.cuda.current_device())
    B, POOL = 2, 8
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    pool = []
    for k in range(POOL):
        frames = [synth.rpn_head(300000 * (rank + 1) + k * B + b, 76, 76, variant, return_gt=True) for b in range(B)]
        pool.append(((t(np.concatenate([f[0] for f in frames])), t(np.concatenate([f[1] for f in frames])),
                      t(np.concatenate([f[2] for f in frames])), t(np.stack([f[3] for f in frames])),
                      [tuple(t(a) for a in f[4]) for f in frames]), hot_path.synth_maps(B, 900 + k, dev)))
    # `depth` batches in flight, slot k on stream k % nstreams: a batch is one submit / finish pair of the C object mv3d_train_path
    # (csrc/train_stream.hip) plus the two RoiPool calls, so the host no longer limits the path and the batches in flight overlap on
    # the device the way the resident replay's three streams do
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams else None
    path = TrainPathStream(B, 76, 76, dev, depth=depth, streams=streams)
    cap = B * path.roi_cap
    g = torch.Generator(device=dev).manual_seed(5)
    bufs = []
    for _ in range(depth):
        d = {}
        for v in hot_path.VIEWS:
            H, W, Cc = hot_path.VIEW_MAPS[v]
            d[v] = (torch.empty((cap, 7, 7, Cc), device=dev), torch.empty((cap, 7, 7, Cc), dtype=torch.int32, device=dev),
                    torch.rand((cap, 7, 7, Cc), generator=g, device=dev) * 2.0 - 1.0, torch.empty((B, H, W, Cc), device=dev))
        bufs.append(d)

    # RoiPool forward + backward on the batch's ROIs: the argument structs of a (maps, buffers, slot) combination are built once
    # (every buffer has the slots' fixed capacity), a batch only sets its row count -- a caller's steady state, no per-batch slicing
    import ctypes as C
    from mv3d_tf_amd._lib import RoiGradView, RoiView, check, lib
    L = lib()
    NV = len(hot_path.VIEWS)
    structs = {}

    def roi(out, maps, d, key):
        St = out["rois"]["bev"].shape[0]
        hit = structs.get(key)
        if hit is None or hit[3] != out["rois"]["bev"].data_ptr():
            fwd, bwd = (RoiView * NV)(), (RoiGradView * NV)()
            for k, v in enumerate(hot_path.VIEWS):
                Bm, H, W, Cc = maps[v].shape
                r = out["rois"][v].data_ptr()
                fwd[k] = RoiView(maps[v].data_ptr(), r, d[v][0].data_ptr(), d[v][1].data_ptr(), 0.125, Bm, St, H, W, Cc)
                bwd[k] = RoiGradView(d[v][3].data_ptr(), r, d[v][2].data_ptr(), d[v][1].data_ptr(), 0.125, Bm, St, H, W, Cc)
            wsz = L.mv3d_roi_pool_backward_workspace_bytes(NV, bwd, 7, 7)
            for k in range(NV):
                bwd[k].num_rois = cap
            wsz = max(wsz, L.mv3d_roi_pool_backward_workspace_bytes(NV, bwd, 7, 7))
            ws = torch.zeros(max(wsz, 256), dtype=torch.uint8, device=dev)
            hit = structs[key] = (fwd, bwd, ws, out["rois"]["bev"].data_ptr())
        fwd, bwd, ws, _ = hit
        for k in range(NV):
            fwd[k].num_rois = St
            bwd[k].num_rois = St
        st = C.c_void_p((out["stream"] or torch.cuda.current_stream()).cuda_stream)
        check(L.mv3d_roi_pool_forward_views(NV, fwd, 7, 7, st), "mv3d_roi_pool_forward_views")
        check(L.mv3d_roi_pool_backward_views(NV, bwd, 7, 7, C.c_void_p(ws.data_ptr()), ws.numel(), st), "mv3d_roi_pool_backward_views")

    def run(nb):
        flight = [path.submit(*pool[j % POOL][0]) for j in range(min(depth - 1, nb))]
        for i in range(nb):
            if i + depth - 1 < nb:
                flight.append(path.submit(*pool[(i + depth - 1) % POOL][0]))
            out = path.finish(flight.pop(0))
            roi(out, pool[i % POOL][1], bufs[i % depth], (i % POOL, i % depth))

    np.random.seed(7 + rank)
    run(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(40)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 40
    nb = max(40, int(seconds / per))
    path.host_seconds
    t0 = time.perf_counter()
    run(nb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_wait, t_draw = path.host_seconds
    path.close()
    return {"workload": "the default line's training path on FRESH inputs (mv3d_tf_amd.train_path.TrainPathStream): new heads / "
                        "ground truth / maps every batch of 2 frames, one host round trip per batch for the numpy-global-RNG "
                        "subsampling draws (exactly the reference's, draw for draw) on the library's helper thread, %d batches "
                        "in flight on %d stream(s); RoiPool fwd + bwd of the 3 views on the batch's ROIs; eager launches" % (depth, max(nstreams, 1)),
            "frames_per_s": round(nb * B / dt, 2), "batches_timed": nb,
            "host_draw_ms_per_frame": round(t_draw / (nb * B) * 1e3, 4),
            "host_wait_for_device_ms_per_frame": round(t_wait / (nb * B) * 1e3, 4),
            "bound": "host: the legacy RandomState.permutation of every candidate list (~21 k background anchors per frame, twice) "
                     "is the reference's subsampling contract; one process draws for one GPU"}


def timed(ring, nbatches, steps, warmup, barrier):
    """-> (wall seconds of the timed region, seconds the host spent enqueueing it, CPU seconds of this PROCESS inside it:
    time.process_time() = user + system time of every thread, the submitting thread and the library's helper thread alike)"""
    for _ in range(warmup):
        ring.run(nbatches)
    barrier()
    c0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(steps):
        ring.run(nbatches)
    t_enq = time.perf_counter() - t0
    barrier()
    return time.perf_counter() - t0, t_enq, time.process_time() - c0


def plan_host_threads(world, local_rank, usable, affinity):
    """Where a rank's two host threads (the submitting thread and the library's helper thread, which draws the subsamples) run when
    N ranks share a node, and whether the host can carry `--launch path` at all:

        usable >= 2 * world   every rank gets its own PAIR of cores out of the affinity mask (rank r: cores 2r, 2r + 1 of the sorted
                              mask) -- the ranks' threads then never time-slice one another;
        usable <  2 * world   not enough cores for a submitting + a drawing thread per rank: fall back to `--launch graph` (frozen
                              batches, no host stage in the timed region) and say so in the line.

    world == 1 pins nothing.  Pure function of its arguments (CPU test); returns {"launch", "cores", "note"}."""
    cores = sorted(affinity)
    if world <= 1:
        return {"launch": "path", "cores": None, "note": "1 rank: no pinning (%d usable cores)" % usable}
    if usable < 2 * world or len(cores) < 2 * world:
        return {"launch": "graph", "cores": None,
                "note": "%d usable host cores for %d ranks (< 2 per rank: a submitting and a drawing thread each): --launch graph "
                        "(frozen batches, no host stage in the timed region) instead of --launch path" % (usable, world)}
    pair = [cores[2 * local_rank], cores[2 * local_rank + 1]]
    return {"launch": "path", "cores": pair, "note": "rank-local core pair %s of %d usable cores" % (pair, usable)}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    ndev = torch.cuda.device_count()
    if local >= ndev and args.dist_backend == "nccl":
        raise SystemExit("rank %d has no GPU (LOCAL_RANK %d, %d visible)" % (rank, local, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    from mv3d_tf_amd import build, sharding
    from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml
    # the configuration the reference trains / tests MV3D with (experiments/scripts/mv3d.sh --cfg
    # experiments/cfgs/faster_rcnn_end2end.yml): 128 sampled ROIs per frame, bg = IoU in [0, 0.5), fg >= 0.7
    apply_end2end_yml()
    if rank == 0:
        build.build()
    if dist is not None:
        dist.barrier()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    wl = args.workload
    batch = args.batch or (2 if wl == "train" else 16)
    nb = args.batches_per_step or (1664 if wl == "train" else 208)
    ring_n = args.ring or (16 if wl == "train" else 4)
    if wl != "train" and args.launch == "path":
        args.launch = "graph"                          # (the TEST-cfg path has no host stage: the frozen batch IS the path)
    ncores, quota_note = usable_cores()
    host_plan = plan_host_threads(world, local, ncores, os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else range(os.cpu_count() or 1))
    if wl == "train" and args.launch == "path":
        if host_plan["launch"] != "path":
            args.launch = host_plan["launch"]
        elif host_plan["cores"] and hasattr(os, "sched_setaffinity"):
            # this thread and the threads it creates from here on (the library's helper thread inherits the mask)
            os.sched_setaffinity(0, host_plan["cores"])
    if args.streams <= 0:
        # (8: the runtime multiplexes streams onto 4 hardware queues -- 4 / 6 / 10 streams measured 12.2 - 12.7 k frames/s, 8 / 12 / 16
        # 13.6 - 13.75 k; GPU_MAX_HW_QUEUES=8 is slower)
        args.streams = 8 if args.launch == "path" else 3
    streams = [torch.cuda.Stream() for _ in range(max(1, min(args.streams, 3) if args.launch == "path" else args.streams))]
    ring = Ring(args, rank, wl, batch, ring_n, streams)
    dt, t_enq, cpu_s = timed(ring, nb, args.steps, args.warmup, barrier)
    host_draws = ring.driver.path.host_seconds if ring.driver is not None else (0.0, 0.0)      # (helper thread: waited, drew) since set-up
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    dt = sharding.max_over_ranks(dt, dist, device=red_dev)
    cpu_all = sharding.gather_over_ranks(cpu_s / args.steps, dist, device=red_dev)          # host CPU seconds per step, every rank

    if rank == 0:
        frames = args.steps * nb * batch * world
        s0 = ring.slots[0]
        if wl == "train":
            desc = ("BASELINE configs[2] path-only: batch %d, 3 views; proposal_layer_3d TRAIN cfg (23104 BEV anchors, "
                    "pre/post-NMS 12000/2000, NMS 0.7) + anchor_target + proposal_target (%d sampled ROIs in batch 0) + FV ROIs "
                    "+ RoiPool 7x7 fwd+bwd on BEV 76x76x512 / RGB 46x155x512 / FV 8x64x512; %s scores; %s"
                    % (batch, s0.num_rois, args.variant,
                       "driven by the library's own mv3d_train_path (submit -> counts to the host -> numpy-global-RNG draws on the helper "
                       "thread -> stage 2) + RoiPool on the sampled ROIs, %d batches in flight, one HIP stream each" % args.streams
                       if args.launch == "path" else "frozen batches (index lists drawn during set-up), %s launches" % args.launch))
        else:
            desc = ("BASELINE configs[4] per-GPU path: batch %d, TEST cfg (pre/post-NMS 6000/300, NMS 0.7) proposal_layer_3d + FV "
                    "ROIs + RoiPool 7x7 fwd on 3 views, R=%d rows; %s scores" % (batch, s0.num_rois, args.variant))
        signature = pmc_signature(wl, batch, s0.num_rois, args.variant,
                                  "pair-tiles-planned" if wl == "train" else ("top+argmax" if getattr(args, "test_argmax", False) else "top-only"))
        res = {
            "metric": METRIC, "value": round(frames / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "batch_per_gpu": batch, "batches_per_step": nb, "frames_per_step_per_gpu": nb * batch,
                       "ring_batches": ring_n, "streams": args.streams, "launch": args.launch, "hipgraph": args.launch == "graph",
                       "host_draws_in_timed_region": args.launch == "path", "timed_region_s": round(dt, 3),
                       "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4), "parallelism": "frames/%d" % world,
                       # CPU seconds (user + system, all threads: submitting thread + the library's helper) a rank's process spent
                       # per step of the timed region; next to ms_per_step it tells host-bound from device-bound at any N
                       "host_cpu_s_per_step": round(cpu_all[0], 5), "host_cpu_s_per_step_per_rank": [round(c, 5) for c in cpu_all],
                       "host_cores": {"usable": ncores, "note": ("%d hardware threads visible%s" % (os.cpu_count() or 0, quota_note)),
                                      "pinned": host_plan["cores"], "plan": host_plan["note"]}},
        }
        res["config"]["pmc_signature"] = signature
        # one batch alone on one stream: the latency of the path
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            s0.run()
        torch.cuda.synchronize()
        res["config"]["one_batch_latency_ms"] = round((time.perf_counter() - t1) / 20 * 1e3, 4)
        entries = roofline_entries(ring, wl, signature)
        dom = max(entries, key=lambda e: e["avg_launch_us"]) if entries else {}
        res["roofline"] = dict(dom, note="dominant launch of the step (RoiPoolGrad = one launch of LDS map tiles over the work list its forward launch planned, no workspace; algorithmic "
                                         "bytes as SURVEY 8(d) defines them -- 8 B per pooled value -- while the pair moves 5: its argmax plane holds one-byte codes); "
                                         "HIP event pairs on the launch stream around the call inside the batch's eager launch "
                                         "sequence, all stream-0 ring batches x 4 rounds; traffic = PMC pass of this exact "
                                         "configuration or null")
        res["roofline_kernels"] = entries
        if wl == "train":
            try:
                res["roofline_kernels"] = entries + nms_roofline_entries(args.variant)
            except Exception as e:                                   # (a diagnostic leg must not cost the line)
                res["roofline_kernels"] = entries + [{"kernel": "greedy NMS", "error": "%s: %s" % (type(e).__name__, str(e)[:200])}]
        if ring.driver is not None:
            # the same two calls as the TIMED loop runs them (8 batches in flight), next to the isolated figures above
            fl = ring.driver.in_flight_us()
            res["roofline"]["in_flight"] = dict(fl, note="HIP event pairs on each batch's own stream around the RoiPool calls while the "
                                                         "path driver keeps its batches in flight (the mode `value` is measured in)")
            for e in entries:
                us = fl["backward_us"] if "roi_pair_" in e["kernel"] else fl["forward_us"]
                e["in_flight_us"] = us
                e["in_flight_frac"] = round(e["alg_bytes_per_launch"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            res["verified"] = ring.driver.verify(ring.host_frames_all)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(ring, wl, args.cpu_seconds)
    failed_here = False
    if not args.no_secondary and wl == "train":
        sec = {}
        # The secondaries must never cost the line: an exception in one of them (they hold the only RCCL collectives of the bench at
        # N > 1) is recorded in secondary.error, and a watchdog prints the line without the unfinished part and leaves if they hang
        # (a rank that failed inside a collective leaves the others waiting).
        done = threading.Lock()

        def emit_and_leave():
            if done.acquire(blocking=False):
                sec.setdefault("error", "secondary legs still running after %d s: line printed without the unfinished ones" % args.secondary_timeout)
                if rank == 0:
                    res["secondary"] = sec
                    print(json.dumps(res), flush=True)
                os._exit(0)

        watchdog = threading.Timer(args.secondary_timeout, emit_and_leave)
        watchdog.daemon = True
        watchdog.start()
        try:
            if not args.no_fresh and args.launch == "path":
                # the figure rounds 1-4 headlined: the same ring with every batch's index lists drawn during set-up, each batch one
                # hipGraph replayed on three streams (no host stage inside the timed region)
                host = host_draws
                ring.driver.close()
                ring.make_graphs()
                dt_r, _, _ = timed(ring, nb, 3, 1, barrier)
                rep = sharding.sum_over_ranks(3 * nb * batch / dt_r, dist, device="cuda" if args.dist_backend == "nccl" else "cpu")
                if rank == 0:
                    sec["resident_replay"] = {"frames_per_s": round(rep, 2), "launch": "hipGraph replay of frozen batches, 3 streams",
                                              "note": "rounds 1-4 reported this as `value`: index lists of the two target layers drawn during "
                                                      "set-up and resident, so no host stage in the timed region"}
                    nfr = (args.steps + args.warmup) * nb * batch
                    sec["fresh_inputs"] = {"frames_per_s": res["value"], "fraction_of_resident_replay": round(res["value"] / rep, 4),
                                           "host_draw_ms_per_frame": round(host[1] / nfr * 1e3, 4),
                                           "host_wait_for_device_ms_per_frame": round(host[0] / nfr * 1e3, 4),
                                           "note": "= the headline since round 4 (the path on new inputs, draws in the loop); host_* = the "
                                                   "library's helper thread on rank 0: drawing / waiting for a batch's stage 1"}
            elif not args.no_fresh:
                fr = fresh_inputs_line(rank, args.variant)
                fr["frames_per_s"] = round(sharding.sum_over_ranks(fr["frames_per_s"], dist, device="cuda" if args.dist_backend == "nccl" else "cpu"), 2)
                if rank == 0:
                    fr["fraction_of_resident_replay"] = round(fr["frames_per_s"] / res["value"], 4)
                    sec["fresh_inputs"] = fr
                torch.cuda.empty_cache()
            if ring.driver is not None:
                ring.driver.close()
            del ring
            torch.cuda.empty_cache()
            args2 = argparse.Namespace(**vars(args))
            if args2.launch == "path":
                args2.launch = "graph"
            r2 = Ring(args2, rank, "test", 16, 3, streams[:3])
            nb2, st2 = 48, max(2, args.steps // 2)
            dt2, _, _ = timed(r2, nb2, st2, 1, barrier)
            dt2 = sharding.max_over_ranks(dt2, dist, device="cuda" if args.dist_backend == "nccl" else "cpu")
            if rank == 0:
                sec["test_cfg"] = {"workload": "BASELINE configs[4] per-GPU path: batch 16, TEST cfg 6000->300, FV ROIs, RoiPool fwd x3 "
                                               "views (the inference call: top only%s), ring of 3 batches on the step's streams (%s launches)"
                                               % ("" if not getattr(args2, "test_argmax", False) else " + argmax", args2.launch),
                                   "frames_per_s": round(st2 * nb2 * 16 * world / dt2, 2), "timed_s": round(dt2, 3),
                                   "roofline_kernels": roofline_entries(r2, "test", pmc_signature("test", 16, 4800, args.variant, "top+argmax" if getattr(args2, "test_argmax", False) else "top-only"))}
            del r2
            torch.cuda.empty_cache()
            if not args.no_trunk:
                if rank == 0:
                    # BASELINE configs[1]: one frame, BEV-only RPN + HIP NMS with the VGG16 trunk on the MFMA convolution, batch 1
                    from mv3d_tf_amd.fast_rcnn import test_mv as _tm
                    sec["config1_latency"] = _tm.bench_config1_latency()
                    torch.cuda.empty_cache()
                from mv3d_tf_amd.fast_rcnn import train_mv
                sec["with_trunk"] = train_mv.bench_train_step(rank, world, dist, seconds=args.secondary_seconds)
                torch.cuda.empty_cache()
                # the same step with the trunks' forward AND backward convolutions on this library's bf16 MFMA kernels (mixed precision:
                # a lower precision than the reference's fp32 training, reported next to it)
                mp = train_mv.bench_train_step(rank, world, dist, amp=torch.bfloat16, mfma=True, seconds=args.secondary_seconds)
                torch.cuda.empty_cache()
                # ... and in the reference's fp32 with the trunks' forward / data-gradient convolutions on the exact-f32 MFMA kernel
                fp = train_mv.bench_train_step(rank, world, dist, amp=None, mfma=True, seconds=args.secondary_seconds)
                torch.cuda.empty_cache()
                if rank == 0:
                    from mv3d_tf_amd import trunk_train
                    from mv3d_tf_amd.networks.mv3d import _VGG as vgg_layers
                    sec["with_trunk"]["fp32_mfma_trunk"] = {"workload": fp["workload"], "frames_per_s": fp["frames_per_s"], "ms_per_step": fp["ms_per_step"],
                                                             "ms_per_step_min": fp["ms_per_step_min"], "ms_per_step_median": fp["ms_per_step_median"], "steps_timed": fp["steps_timed"],
                                                             "roofline_kernels": [trunk_train.bench_wgrad_layers(vgg_layers, dtype=torch.float32)]}
                    sec["with_trunk"]["bf16_mfma_trunk"] = {"workload": mp["workload"], "frames_per_s": mp["frames_per_s"], "ms_per_step": mp["ms_per_step"],
                                                             "ms_per_step_min": mp["ms_per_step_min"], "ms_per_step_median": mp["ms_per_step_median"], "steps_timed": mp["steps_timed"],
                                                             "roofline_kernels": [trunk_train.bench_wgrad_layers(vgg_layers)]}
                if dist is not None:
                    dist.barrier()
                torch.cuda.empty_cache()
                from mv3d_tf_amd.fast_rcnn import test_mv
                sec["serving_with_trunk"] = test_mv.bench_serve_step(rank, world, dist, reduce_device="cuda" if args.dist_backend == "nccl" else "cpu", seconds=args.secondary_seconds)
                torch.cuda.empty_cache()
                if rank == 0:
                    # the hand-written MFMA convolution against the dense f16 matrix-core peak (rank 0 only: a per-kernel figure)
                    from mv3d_tf_amd import trunk
                    from mv3d_tf_amd.networks.mv3d import _VGG
                    import torch as _t
                    sec["serving_with_trunk"]["roofline_kernels"] = [trunk.bench_conv_layers(_VGG), trunk.bench_conv_layers(_VGG, dtype=_t.float32)]
                if dist is not None:
                    dist.barrier()
        except Exception as e:                                   # (KeyboardInterrupt / SystemExit pass)
            sec["error"] = "%s: %s" % (type(e).__name__, str(e)[:500])
            failed_here = True
        watchdog.cancel()
        if not done.acquire(blocking=False):                     # the watchdog is printing the line
            time.sleep(3600)
        if rank == 0:
            res["secondary"] = sec
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None and not failed_here:                      # (a rank whose secondaries raised does not wait for the others)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and "verified" in res and not res["verified"]["bit_exact"]:
        raise SystemExit("bench.py: the path driver's outputs differ from the oracle: %s" % res["verified"]["mismatches"])


if __name__ == "__main__":
    main()
