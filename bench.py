#!/usr/bin/env python3
"""bench.py -- KITTI-shape frames/s through the MV3D hot path on MI355X.

A "step" = one pass of the hot path over one batch of synthetic KITTI-shaped frames whose
inputs are already resident in HBM:

    proposal_layer_3d (decode + project + clip/filter + score sort + NMS + top-N)
      -> RoiPool 7x7 on the BEV feature map (76x76x512) with rois_bv
      -> RoiPool 7x7 on the RGB feature map (46x155x512) with rois_img
    [--train adds anchor_target stage-1, and RoiPoolGrad on both views]

i.e. BASELINE.json configs[1] (1 GPU, BEV RPN with 76x76x4 = 23 104 anchors + HIP NMS,
batch 1) widened by the two RoiPool views the reference has; the VGG16 trunks are not part
of the path.  One process per GPU; frames shard across ranks with no
data-path collective ("weak" scaling: per-GPU work fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--cfg TEST|TRAIN]
                    [--variant peaky|rand] [--graph] [--no-cpu-baseline]

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel, measured live with HIP events on the launch stream) and `cpu_baseline` (the C
oracle, single thread, bounded sample) objects.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

CFGS = {
    # experiments/cfgs/faster_rcnn_end2end.yml:15-20 (what experiments/scripts/mv3d.sh tests with)
    "TEST": dict(RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300, RPN_NMS_THRESH=0.7, RPN_MIN_SIZE=5),
    # lib/fast_rcnn/config.py:126-148
    "TRAIN": dict(RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000, RPN_NMS_THRESH=0.7, RPN_MIN_SIZE=5),
}
BEV_MAP = (76, 76, 512)      # conv5_3 of the 608x608 BEV, stride 8
RGB_MAP = (46, 155, 512)     # conv5_3 of the 375x1242 image, stride 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="frames per step per GPU")
    ap.add_argument("--cfg", default="TEST", choices=list(CFGS))
    ap.add_argument("--variant", default="peaky", choices=["peaky", "rand"])
    ap.add_argument("--graph", action="store_true",
                    help="replay a captured hipGraph instead of launching eagerly (measured ~5 %% slower at batch 1: "
                         "the 7 launches of a step are enqueued ahead of the GPU either way, and graph nodes carry more "
                         "per-node overhead than in-order stream launches)")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--streams", type=int, default=0,
                    help="frames in flight: step i runs on stream i %% S with its own outputs and workspace, so the "
                         "latency-bound kernels of one frame overlap the RoiPool of another (each step is still one batch). "
                         "0 = auto: 3 up to batch 4 (batch 1: 16.3k / 24.5k / 28.8k / 22.7k frames/s for 1-4 streams; batch 4: 27.9k / "
                         "34.4k / 35.2k for 1-3), 1 above (batch 16: 31k alone, 24k with 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--only", default="", choices=["", "proposal", "roi"], help="diagnostics: run only one half of the step")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) | gloo (logic tests on one GPU)")
    return ap.parse_args()


class Frames:
    """Per-rank device-resident inputs and the step closure."""

    def __init__(self, args, rank, stream=None):
        from mv3d_tf_amd import ops, synth
        self.ops, self.args = ops, args
        self.stream = stream                          # None: whatever stream is current when step() runs
        B = args.batch
        heads = [synth.rpn_head(1000 + rank * 64 + b, 76, 76, args.variant) for b in range(B)]
        self.host_frame0 = heads[0]
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
        self.prob = t(np.concatenate([h[0] for h in heads]))
        self.pred = t(np.concatenate([h[1] for h in heads]))
        self.info = t(np.concatenate([h[2] for h in heads]))
        self.calib = t(np.stack([h[3] for h in heads]))
        self.bev = t(synth.feature_map(7, *BEV_MAP[:2], BEV_MAP[2], B))
        self.rgb = t(synth.feature_map(8, *RGB_MAP[:2], RGB_MAP[2], B))
        self.params = ops.proposal_params(CFGS[args.cfg])
        self.out = None
        self.step()                                   # allocates outputs / workspace
        torch.cuda.synchronize()

    def step(self):
        """one pass of the hot path over the batch: mv3d_proposal_3d (6 launches) + both RoiPool views (1 launch).
        After the first call the two C entry points are called with pre-built arguments (nothing is allocated,
        looked up or converted per step; the host side of a step is two ctypes calls)."""
        sid = self.stream.cuda_stream if self.stream is not None else torch.cuda.current_stream().cuda_stream
        bound = self._bound.get(sid) if hasattr(self, "_bound") else None
        if bound is None:
            bound = self._bind(sid)
        only = self.args.only
        rc = bound[0](*bound[1]) if only != "roi" else 0
        if rc == 0 and only != "proposal":
            rc = bound[2](*bound[3])
        if rc != 0:
            from mv3d_tf_amd._lib import check
            check(rc, "bench step")
        return self.out[0].shape[1]

    def _bind(self, sid):
        import ctypes as C
        from mv3d_tf_amd import _lib
        from mv3d_tf_amd._lib import RoiView, lib
        o = self.ops
        if self.out is None:                                                # first call: outputs (through the wrappers)
            with torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext():
                self.out = o.proposal_3d(self.prob, self.pred, self.info, self.calib, self.params)
            R = self.out[0].shape[0] * self.out[0].shape[1]
            mk = lambda c, dt: torch.empty((R, 7, 7, c), dtype=dt, device="cuda")
            self.tops = (mk(BEV_MAP[2], torch.float32), mk(BEV_MAP[2], torch.int32),
                         mk(RGB_MAP[2], torch.float32), mk(RGB_MAP[2], torch.int32))
        bv, img, b3, num, status = self.out
        rois_bv, rois_img = bv.view(-1, 5), img.view(-1, 5)                 # (B*cap, 5), column 0 = frame index
        B, H, W, _ = self.prob.shape
        P = lambda t: C.c_void_p(t.data_ptr())
        nbytes = lib().mv3d_proposal_3d_workspace_bytes(B, H, W, C.byref(self.params))
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.prob.device)     # one per (Frames, stream)
        st = C.c_void_p(sid)
        a1 = (P(self.prob), P(self.pred), B, H, W, P(self.info), P(self.calib), C.byref(self.params), P(bv), P(img), P(b3),
              P(num), P(status), P(ws), C.c_size_t(ws.numel()), st)
        arr = (RoiView * 2)()
        for k, (data, rois, top, am) in enumerate(((self.bev, rois_bv, self.tops[0], self.tops[1]),
                                                   (self.rgb, rois_img, self.tops[2], self.tops[3]))):
            Bd, Hd, Wd, Cd = data.shape
            arr[k] = RoiView(data.data_ptr(), rois.data_ptr(), top.data_ptr(), am.data_ptr(), 0.125, Bd, rois.shape[0], Hd, Wd, Cd)
        a2 = (2, arr, 7, 7, st)
        if not hasattr(self, "_bound"):
            self._bound = {}
        self._keep = getattr(self, "_keep", []) + [ws, arr]
        self._bound[sid] = (lib().mv3d_proposal_3d, a1, lib().mv3d_roi_pool_forward_views, a2)
        return self._bound[sid]

    def _roi_views(self, rois_bv, rois_img):
        """both RoiPool layers of the step (MV3D_test.py:95-107) in one launch"""
        from mv3d_tf_amd import ops as o
        o.roi_pool_forward_views([(self.bev, rois_bv, 0.125), (self.rgb, rois_img, 0.125)], 7, 7,
                                 outs=[(self.tops[0], self.tops[1]), (self.tops[2], self.tops[3])])

    def _roi(self, data, rois, top, argmax):
        import ctypes as C
        from mv3d_tf_amd._lib import check, lib
        B, H, W, Cc = data.shape
        rc = lib().mv3d_roi_pool_forward(C.c_void_p(data.data_ptr()), C.c_float(0.125), B, rois.shape[0], H, W, Cc,
                                         7, 7, C.c_void_p(rois.data_ptr()), C.c_void_p(top.data_ptr()),
                                         C.c_void_p(argmax.data_ptr()),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
        check(rc, "mv3d_roi_pool_forward")


def time_kernel_events(fn, iters=50):
    """average duration (ms) of fn()'s launches with HIP events on the current stream"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed PMC passes (tools/gpu_pmc.sh + tools/pmc_summary.py:
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes); None if no profile is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        table = json.load(open(path))
        key = next(k for k in table if kernel in k)       # template instances carry a "void ...<N>" decoration
        return int(table[key]["hbm_bytes_per_launch"])
    except Exception:
        return None


def roofline(fr):
    """Dominant kernel = the RoiPool forward launch (BEV + RGB views; largest share of the step and of
    its traffic).  Algorithmic bytes per launch: each feature map once + rois + (top f32 +
    argmax i32) outputs of both views."""
    B = fr.args.batch
    R = fr.out[0].shape[0] * fr.out[0].shape[1]
    alg = 0
    for (H, W, C) in (BEV_MAP, RGB_MAP):
        alg += B * H * W * C * 4 + R * 20 + R * 49 * C * 8
    rois_bv, rois_img = fr.out[0].view(-1, 5), fr.out[1].view(-1, 5)
    ms = time_kernel_events(lambda: fr._roi_views(rois_bv, rois_img))
    gbs = alg / (ms * 1e-3) / 1e9
    name = "roi_pool_fwd_xcd_multi_kernel"
    return {"kernel": "%s (BEV 76x76x512 + RGB 46x155x512 views, R=%d rows each)" % (name, R), "bound": "hbm",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic(name),
            "alg_bytes_per_launch": alg, "avg_launch_us": round(ms * 1e3, 2),
            "note": "HIP events over 50 back-to-back launches on the launch stream; traffic = PMC FETCH_SIZE*2+WRITE_SIZE "
                    "per launch (profiles/r01_pmc_traffic.txt); write-only fill ceiling on this box 5.8-6.0 TB/s (profiles/r01_hbm_probe.txt)"}


def cpu_baseline(fr, seconds):
    """The C oracle on the same frame-0 workload, bounded sample: first one thread (the scalar port, ~1/3 of the
    time), then `cores` threads that each process whole frames (ctypes releases the GIL inside the C calls) --
    the frame-parallel way a CPU deployment of the reference would use the host."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from mv3d_tf_amd import synth
    prob, pred, info, calib = fr.host_frame0
    bev = synth.feature_map(7, *BEV_MAP[:2], BEV_MAP[2], 1)
    rgb = synth.feature_map(8, *RGB_MAP[:2], RGB_MAP[2], 1)
    cfg = {fr.args.cfg: CFGS[fr.args.cfg]}

    def frame():
        bv, img, b3 = oracle.proposal_layer_3d(prob, pred, info, calib, fr.args.cfg, [8, ], cfg=cfg)
        oracle.roi_pool(bev, bv, 7, 7, 0.125)
        oracle.roi_pool(rgb, img, 7, 7, 0.125)

    n1, t0 = 0, time.perf_counter()
    while True:
        frame()
        n1 += 1
        dt1 = time.perf_counter() - t0
        if dt1 >= seconds / 3 or n1 >= 400:
            break
    cores = max(1, min(os.cpu_count() or 1, 64))
    done = [0] * cores
    stop_at = time.perf_counter() + 2 * seconds / 3

    def worker(k):
        while time.perf_counter() < stop_at:
            frame()
            done[k] += 1

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dtm = time.perf_counter() - t0
    nm = sum(done)
    return {"value": round(nm / dtm, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "one_thread_frames_per_s": round(n1 / dt1, 3),
            "sample": "%d frames on %d threads in %.1f s (frame-parallel) after %d frames on 1 thread in %.1f s; same workload "
                      "(frame 0, %s cfg, %s scores); C restatement oracle/mv3d_oracle.c, gcc -O2; host has %d cores"
                      % (nm, cores, dtm, n1, dt1, fr.args.cfg, fr.args.variant, os.cpu_count())}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    ndev = torch.cuda.device_count()
    if local >= ndev and args.dist_backend == "nccl":
        raise SystemExit("rank %d has no GPU (LOCAL_RANK %d, %d visible)" % (rank, local, ndev))
    local = local % ndev                               # gloo logic tests may share one GPU
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    from mv3d_tf_amd import build
    build.build()

    nstreams = args.streams if args.streams > 0 else (3 if args.batch <= 4 and not args.graph else 1)
    if nstreams == 1:
        frs = [Frames(args, rank)]
    else:
        frs = [Frames(args, rank, torch.cuda.Stream()) for _ in range(nstreams)]
    fr = frs[0]
    graph = None
    if args.graph and not args.no_graph and nstreams == 1:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fr.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fr.step()
        run = graph.replay
    elif nstreams == 1:
        run = fr.step
    else:
        turn = [0]

        def run():
            frs[turn[0] % nstreams].step()
            turn[0] += 1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    t_enq = time.perf_counter() - t0                    # host time to enqueue the steps (eager: must stay < dt)
    barrier()
    dt = time.perf_counter() - t0
    lat = None
    if nstreams > 1:                                   # the same step alone on one stream: its latency
        for _ in range(10):
            fr.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(100):
            fr.step()
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t1) / 100
    from mv3d_tf_amd import sharding
    dt = sharding.max_over_ranks(dt, dist, device="cuda" if args.dist_backend == "nccl" else "cpu")

    if rank == 0:
        frames = args.steps * args.batch * world
        res = {
            "metric": "KITTI-shape frames/sec (RPN+ROI-pool+NMS hot path)",
            "value": round(frames / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] + RoiPool views: proposal_layer_3d (76x76x4=23104 BEV anchors, "
                                   "%s cfg pre/post-NMS %d/%d, NMS 0.7) -> RoiPool 7x7 BEV 76x76x512 + RGB 46x155x512, "
                                   "R=%d rows/frame; %s scores" % (args.cfg, CFGS[args.cfg]["RPN_PRE_NMS_TOP_N"],
                                                                   CFGS[args.cfg]["RPN_POST_NMS_TOP_N"],
                                                                   fr.out[0].shape[1], args.variant),
                       "batch_per_gpu": args.batch, "hipgraph": graph is not None,
                       "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4), "streams": nstreams,
                       "one_stream_ms_per_step": None if lat is None else round(lat * 1e3, 4), "parallelism": "frames/%d" % world},
            "kept_rois_frame0": int(fr.out[3][0].item()),
        }
        res["roofline"] = roofline(fr)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(fr, args.cpu_seconds)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
