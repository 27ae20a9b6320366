"""Training entry points (SURVEY.md §8(b) "Entry points kept", §8(e) multi-GPU): CPU tests of the data-parallel gradient
exchange (world_size-2 gloo) and of the host logic; `gpu` tests of train_net / SolverWrapper on a synthetic imdb."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_model(seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.3).requires_grad_(True)
    return [mk(8, 3, 3, 3), mk(8), mk(16, 8, 3, 3), mk(16), mk(5, 16), mk(5)]


def _tiny_loss(params, x, y):
    import torch.nn.functional as F
    w1, b1, w2, b2, w3, b3 = params
    h = F.relu(F.conv2d(x, w1, b1, padding=1))
    h = F.relu(F.conv2d(h, w2, b2, padding=1)).mean(dim=(2, 3))
    return F.cross_entropy(F.linear(h, w3, b3), y)


def _frame(k):
    g = torch.Generator().manual_seed(100 + k)
    return torch.randn(1, 3, 12, 12, generator=g), torch.randint(0, 5, (1,), generator=g)


def _dp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mv3d_tf_amd import sharding
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = _tiny_model(1)                                   # identical replicas
    b = sharding.GradBucketer(params, dist, bucket_bytes=1500)        # several buckets
    assert len(b.buckets) >= 3 and b.buckets[0]["params"][0] is params[-1]       # last layer first
    frames = sharding.frame_shard(4, rank, world)             # 4 frames over 2 ranks: 2 accumulated frames per rank and step
    b.zero_grad()
    for k, f in enumerate(frames):
        b.reset()
        b.dist_enabled = (k == len(frames) - 1)
        (_tiny_loss(params, *_frame(f)) / len(frames)).backward()
    b.finish()
    torch.save([p.grad.clone() for p in params], os.path.join(out_dir, "g%d.pt" % rank))
    owned = torch.tensor(frames)
    gathered = [torch.zeros_like(owned) for _ in range(world)]
    dist.all_gather(gathered, owned)
    if rank == 0:
        torch.save(torch.cat(gathered), os.path.join(out_dir, "frames.pt"))
    b.close()
    dist.destroy_process_group()


def test_data_parallel_gradients_equal_single_process_mean(tmp_path):
    """2 ranks x 2 frames, bucketed all-reduce (gloo) == one process on the 4-frame mean loss; the shards cover the
    frames exactly once."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    params = _tiny_model(1)
    loss = sum(_tiny_loss(params, *_frame(f)) for f in range(4)) / 4
    want = torch.autograd.grad(loss, params)
    for a, b, w in zip(g0, g1, want):
        assert torch.equal(a, b)                              # every rank holds the same reduced gradient
        assert torch.allclose(a, w, rtol=1e-5, atol=1e-7)
    assert sorted(torch.load(tmp_path / "frames.pt").tolist()) == [0, 1, 2, 3]


def _skip_worker(rank, world, port, out_dir):
    """rank 1's loss does not touch the LAST layer (the first bucket): its hooks never fire there, so on that rank the first bucket -- and
    everything behind it -- is launched by finish(); rank 0 launches every bucket during backward.  The collectives must still pair up
    bucket by bucket (sharding.GradBucketer._launch_ready: strictly in index order on every rank)."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import torch.nn.functional as F
    from mv3d_tf_amd import sharding
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = _tiny_model(1)
    b = sharding.GradBucketer(params, dist, bucket_bytes=1500)
    launched = []
    real = dist.all_reduce

    class Spy:                                                        # records which bucket goes out when (during backward / in finish)
        ReduceOp = dist.ReduceOp

        @staticmethod
        def all_reduce(t, op=None, async_op=False):
            launched.append((next(i for i, x in enumerate(b.buckets) if x["flat"] is t), phase[0]))
            return real(t, op=op, async_op=async_op)
    b.dist = Spy
    phase = ["backward"]
    b.zero_grad()
    b.reset()
    x, y = _frame(rank)
    if rank == 0:
        loss = _tiny_loss(params, x, y)
    else:
        w1, b1, w2, b2, w3, b3 = params                               # (w3, b3 = the classifier = bucket 0: unused on this rank)
        h = F.relu(F.conv2d(x, w1, b1, padding=1))
        loss = F.relu(F.conv2d(h, w2, b2, padding=1)).mean()
    loss.backward()
    phase[0] = "finish"
    b.finish()
    torch.save({"grads": [p.grad.clone() for p in params], "launched": launched}, os.path.join(out_dir, "s%d.pt" % rank))
    b.close()
    dist.destroy_process_group()


def test_bucket_order_when_one_rank_skips_a_parameter(tmp_path):
    """the ordering rule of GradBucketer._launch_ready: a parameter without a gradient on ONE rank holds its bucket -- and
    every later one -- back until finish() on that rank only; both ranks still launch the same buckets in the same order, and the reduced
    gradients are the mean of the two ranks' (zero where a rank had none)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_skip_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    order0, order1 = [i for i, _ in r0["launched"]], [i for i, _ in r1["launched"]]
    assert order0 == order1 == list(range(len(order0))) and len(order0) >= 3
    assert all(ph == "backward" for _, ph in r0["launched"])          # rank 0: every bucket went out under backward
    assert all(ph == "finish" for _, ph in r1["launched"])            # rank 1: bucket 0 never completed, so nothing could go first
    import torch.nn.functional as F
    params = _tiny_model(1)
    g0 = torch.autograd.grad(_tiny_loss(params, *_frame(0)), params)
    w1, b1, w2, b2, w3, b3 = params
    x1, _ = _frame(1)
    l1 = F.relu(F.conv2d(F.relu(F.conv2d(x1, w1, b1, padding=1)), w2, b2, padding=1)).mean()
    g1 = list(torch.autograd.grad(l1, [w1, b1, w2, b2])) + [torch.zeros_like(w3), torch.zeros_like(b3)]
    for a, b_, u, v in zip(r0["grads"], r1["grads"], g0, g1):
        assert torch.equal(a, b_) and torch.allclose(a, (u + v) / 2, rtol=1e-5, atol=1e-7)


def test_bucketing_of_the_mv3d_parameter_list():
    """the real parameter list (shapes of networks/mv3d.py, ~143 M fp32 = 573 MB): 25 MB buckets, reverse layer order,
    gradients are views of the flat buffers"""
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd import sharding
    shapes = []
    for cin in (9, 3):
        c = cin
        for cout in (64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512):
            shapes += [(cout, c, 3, 3), (cout,)]
            c = cout
    shapes += [(512, 512, 3, 3), (512,), (8, 512, 1, 1), (8,), (24, 512, 1, 1), (24,)]
    for _ in range(2):
        shapes += [(2048, 7 * 7 * 512), (2048,), (2048, 2048), (2048,)]
    shapes += [(2, 4096), (2,), (48, 4096), (48,)]
    params = [torch.empty(s, device="meta", requires_grad=True) for s in shapes]
    nbytes = sum(int(np.prod(s)) for s in shapes) * 4
    assert 130e6 * 4 < nbytes < 150e6 * 4                     # ~143 M fp32
    b = sharding.GradBucketer(params, None, allocate=True)
    assert b.total_bytes() == nbytes
    assert b.buckets[0]["params"][0] is params[-1] and b.buckets[-1]["params"][-1] is params[0]
    small = [x for x in b.buckets if x["flat"].numel() * 4 <= (25 << 20)]
    assert len(small) >= len(b.buckets) - 2                   # only the two fc6 matrices (205 MB each) exceed a bucket
    for x in b.buckets:
        off = 0
        for p in x["params"]:
            assert p.grad.shape == p.shape
            off += p.numel()
        assert off == x["flat"].numel()
    b.close()


def test_data_layer_and_roidb_host_logic():
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd.fast_rcnn import train_mv
    from mv3d_tf_amd.fast_rcnn.config import cfg
    from mv3d_tf_amd.roi_data_layer import RoIDataLayer
    from mv3d_tf_amd import synth
    roidb = []
    for i in range(5):
        gt_bv, gt_3d, gt_cnr = synth.gt_cars(np.random.RandomState(i), 3)
        roidb.append({"image": np.full((6, 8, 3), 100, np.uint8), "lidar_bv": np.zeros((16, 16, 9), np.float32),
                      "calib": synth.KITTI_CALIB.astype(np.float64), "gt_classes": np.array([1, 1, 0], np.int32),
                      "boxes": np.zeros((3, 4), np.float32), "boxes_bv": gt_bv[:, :4], "boxes_3D": gt_3d[:, :6],
                      "boxes_corners": gt_cnr[:, :24], "max_overlaps": np.array([1.0, 1.0, 0.0]), "id": i})
    saved = cfg.TRAIN.IMS_PER_BATCH
    cfg.TRAIN.IMS_PER_BATCH = 1
    try:
        np.random.seed(3)
        layer = RoIDataLayer(roidb, 2)
        np.random.seed(3)
        perm = np.random.permutation(np.arange(5))
        assert np.array_equal(layer._perm, perm)              # the reference's draw (layer.py:28)
        blobs = layer.forward()
        assert blobs["image_data"].shape == (1, 6, 8, 3) and blobs["image_data"].dtype == np.float32
        assert np.allclose(blobs["image_data"][0, 0, 0], 100 - cfg.PIXEL_MEANS[0, 0])
        assert blobs["lidar_bv_data"].shape == (1, 16, 16, 9) and blobs["gt_boxes_3d"].shape == (2, 7)
        assert np.array_equal(blobs["im_info"], np.array([[16, 16, 1]], np.float32)) and blobs["gt_boxes_corners"].shape == (2, 25)
        seen = [int(perm[0])]
        for _ in range(3):
            layer.forward(); seen.append(int(layer._perm[layer._cur - 1]))
        assert sorted(seen) == sorted(perm[:4].tolist())
        layer.forward()                                        # cur + 1 >= 5: reshuffled before the draw
        assert layer._cur == 1
    finally:
        cfg.TRAIN.IMS_PER_BATCH = saved
    kept = train_mv.filter_roidb(roidb + [dict(roidb[0], max_overlaps=np.array([0.05]))])
    assert len(kept) == 5                                      # 0.05 is neither fg nor in [BG_THRESH_LO, BG_THRESH_HI)
    assert train_mv.snapshot_filename("/x", 9).endswith("_iter_10.ckpt")


@pytest.mark.gpu
@pytest.mark.parametrize("mixed", [False, True, "fp32_mfma"])
def test_train_net_two_iterations_and_snapshot(tmp_path, capsys, mixed):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd import build, synth
    build.build()
    from mv3d_tf_amd.fast_rcnn import train_mv
    from mv3d_tf_amd.fast_rcnn.config import cfg
    from mv3d_tf_amd.networks import get_network

    class Imdb:
        num_classes = 2
        name = "synthetic"

    rng = np.random.RandomState(0)
    roidb = []
    for i in range(3):
        _, _, _, calib, (gt_bv, gt_3d, gt_cnr) = synth.rpn_head(40 + i, 76, 76, "peaky", return_gt=True)
        roidb.append({"image": rng.randint(0, 255, (96, 320, 3)).astype(np.uint8),
                      "lidar_bv": (rng.random_sample((608, 608, 9)) < 0.02).astype(np.float32), "calib": calib.astype(np.float64),
                      "gt_classes": np.ones(len(gt_bv), np.int32), "boxes": np.zeros((len(gt_bv), 4), np.float32),
                      "boxes_bv": gt_bv[:, :4], "boxes_3D": gt_3d[:, :6], "boxes_corners": gt_cnr[:, :24],
                      "max_overlaps": np.ones(len(gt_bv))})
    saved = (cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS)
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS = 1, 1, 2
    cfg.TRAIN.MIXED_PRECISION = mixed is True                  # True: the trunks on the bf16 MFMA kernels (trunk_train.py)
    cfg.TRAIN.MFMA_TRUNK = mixed == "fp32_mfma"                # the reference's fp32 on the exact-f32 MFMA kernels
    try:
        np.random.seed(cfg.RNG_SEED)
        net = get_network("MV3D_train")
        before = net.params["rpn_bbox_pred"][0].detach().clone()
        hist = train_mv.train_net(net, Imdb(), roidb, str(tmp_path), max_iters=3)
    finally:
        cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS = saved
        cfg.TRAIN.MIXED_PRECISION = cfg.TRAIN.MFMA_TRUNK = False
    assert net.mfma_trunk == bool(mixed)
    assert len(hist) == 3 and all(np.isfinite(h[0]) for h in hist)
    assert not torch.equal(before, net.params["rpn_bbox_pred"][0].detach())     # Adam moved the weights
    out = capsys.readouterr().out
    assert ("Mixed precision" in out) == (mixed is True) and ("exact-f32" in out) == (mixed == "fp32_mfma")
    assert "iter: 1 / 3, total loss: " in out and "rpn_loss_cls: " in out and ", lr: 0.000010" in out
    assert "speed: " in out and "s / iter" in out and "Wrote snapshot to: " in out and "done solving" in out
    files = sorted(os.listdir(tmp_path))
    pref = cfg.TRAIN.SNAPSHOT_PREFIX
    assert pref + "_iter_2.ckpt" in files and pref + "_iter_3.ckpt" in files and pref + "_iter_3.ckpt.optim.pt" in files
    # a snapshot is the .npy weight dict network.load reads (TF layouts)
    net2 = get_network("MV3D_train")
    net2.load(os.path.join(tmp_path, pref + "_iter_3.ckpt"))
    for k in ("conv5_3", "rpn_bbox_pred", "fc6_1", "bbox_pred"):
        assert torch.equal(net2.params[k][0], net.params[k][0].detach()) and torch.equal(net2.params[k][1], net.params[k][1].detach())


# ---------------------------------------------------------------------------------------------------- SolverWrapper under DP
class _ToyNet:
    """the network face SolverWrapper uses (params dict, parameters(), forward(feed) -> layers, load()) on CPU tensors"""

    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        mk = lambda *s: (torch.randn(*s, generator=g) * 0.3).requires_grad_(True)
        self.params = {"conv": [mk(4, 3, 3, 3), mk(4)], "fc": [mk(3, 4), mk(3)], "unused": [mk(2, 2), mk(2)]}
        self.loaded = None

    def parameters(self):
        return [p for wb in self.params.values() for p in wb]

    def forward(self, feed):
        import torch.nn.functional as F
        x = torch.as_tensor(np.asarray(feed["image_data"], np.float32)).permute(0, 3, 1, 2)
        h = F.relu(F.conv2d(x, *self.params["conv"], padding=1)).mean(dim=(2, 3))
        return {"logits": F.linear(h, *self.params["fc"]), "n": x.shape[0]}      # ("unused" never gets a gradient)

    def load(self, path, *a):
        self.loaded = path


class _ToyImdb:
    num_classes = 2


def _toy_roidb(n):
    rng = np.random.RandomState(0)
    return [{"image": rng.randint(0, 255, (6, 8, 3)).astype(np.uint8), "lidar_bv": np.zeros((8, 8, 9), np.float32),
             "calib": np.zeros((4, 12), np.float32), "boxes": np.zeros((1, 4)), "k": k,
             "max_overlaps": np.array([1.0])} for k in range(n)]


def _solver_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mv3d_tf_amd.fast_rcnn import train_mv
    from mv3d_tf_amd.fast_rcnn.config import cfg
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS, cfg.TRAIN.IMS_PER_BATCH = 1, 2, 1
    np.random.seed(5)

    def toy_total_loss(layers, sigma=3.0):                      # (the real losses are device kernels: not under test here)
        loss = (layers["logits"] ** 2).mean()
        z = loss.detach() * 0
        return loss, (loss, z, z, z)

    train_mv.total_loss = toy_total_loss
    train_mv.get_data_layer = lambda roidb, nc: _ToyLayer(roidb)
    net = _ToyNet(10 + rank)                                    # DIFFERENT initial weights per rank: rank 0's must win
    sw = train_mv.SolverWrapper(None, None, net, _ToyImdb(), _toy_roidb(5), os.path.join(out_dir, "snap"))
    lines = []
    sw.log = lines.append
    hist = sw.train_model(None, 3, frames_per_step=2)
    torch.save({"w": [p.detach().clone() for p in net.parameters()], "hist": hist, "lines": lines}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


class _ToyLayer:
    def __init__(self, roidb):
        self.roidb, self.i = roidb, 0

    def forward(self):
        e = self.roidb[self.i % len(self.roidb)]
        self.i += 1
        return {"image_data": e["image"][None].astype(np.float32), "lidar_bv_data": e["lidar_bv"][None], "calib": e["calib"],
                "im_info": np.array([[8, 8, 1]], np.float32), "gt_boxes_bv": np.zeros((1, 5), np.float32),
                "gt_boxes_3d": np.zeros((1, 7), np.float32), "gt_boxes_corners": np.zeros((1, 25), np.float32)}


def test_solver_wrapper_train_model_under_gloo_world2(tmp_path):
    """SolverWrapper.train_model itself under torch.distributed (gloo, 2 ranks, 2 frames per rank and step as ONE batch):
    rank 0's initial weights are broadcast, the weights are identical on both ranks after 3 iterations (a parameter that never
    gets a gradient does not desynchronise the bucket order), only rank 0 logs and snapshots, the logged losses are the mean
    over the ranks (DISPLAY all-reduce), the snapshot + .optim.pt pair is written."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_solver_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt", weights_only=False), torch.load(tmp_path / "r1.pt", weights_only=False)
    for a, b in zip(r0["w"], r1["w"]):
        assert torch.equal(a, b)
    assert len(r0["hist"]) == 3 and np.allclose(np.array(r0["hist"]), np.array(r1["hist"]))     # all-reduced display values
    assert any(l.startswith("iter: 3 / 3") for l in r0["lines"]) and any("speed:" in l for l in r0["lines"])
    assert not r1["lines"]                                                                     # rank 0 only
    snaps = sorted(os.listdir(tmp_path / "snap"))
    assert any(n.endswith("_iter_2.ckpt") for n in snaps) and any(n.endswith("_iter_3.ckpt") for n in snaps)
    assert any(n.endswith("_iter_3.ckpt.optim.pt") for n in snaps)


def test_stack_blobs_batches_frames():
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd.fast_rcnn.train_mv import stack_blobs
    a, b = _ToyLayer(_toy_roidb(2)).forward(), _ToyLayer(_toy_roidb(2)).forward()
    one = stack_blobs([a])
    assert one["image_data"].shape == (1, 6, 8, 3) and one["gt_boxes_bv"].shape == (1, 5)      # one frame: the reference's feed
    two = stack_blobs([a, b])
    assert two["image_data"].shape == (2, 6, 8, 3) and two["im_info"].shape == (2, 3) and two["calib"].shape == (2, 4, 12)
    assert isinstance(two["gt_boxes_3d"], list) and len(two["gt_boxes_3d"]) == 2


@pytest.mark.gpu
def test_data_parallel_mixed_precision_step_is_deterministic_and_agrees_across_ranks():
    """two gloo ranks sharing the GPU: the mixed-precision training graph (grouped trunk launches on one stream, the GradBucketer's
    all-reduce overlapping backward) ends with the same averaged gradient bits on both ranks, and the same bits when run again"""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29633", os.path.join(ROOT, "tools", "dp_grad_probe.py"), "--no-timing"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "averaged gradients bit-identical: True" in out.stdout, out.stdout[-1000:]


def test_train_model_with_mixed_image_sizes_in_one_step(tmp_path):
    """KITTI images come in several sizes and the data layer does not resize, so the frames of a step cannot always
    be stacked.  train_model(frames_per_step=2) then runs one sub-batch per size and accumulates: the update equals one Adam
    step on the mean of the two frames' losses."""
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd.fast_rcnn import train_mv
    from mv3d_tf_amd.fast_rcnn.config import cfg

    class MixedLayer(_ToyLayer):
        def forward(self):
            out = _ToyLayer.forward(self)
            if self.i % 2 == 0:                                   # every second frame: a smaller image
                out["image_data"] = np.ascontiguousarray(out["image_data"][:, :5, :7])
            return out

    def toy_total_loss(layers, sigma=3.0):
        loss = (layers["logits"] ** 2).mean()
        z = loss.detach() * 0
        return loss, (loss, z, z, z)

    old = (train_mv.total_loss, train_mv.get_data_layer, cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS)
    try:
        train_mv.total_loss = toy_total_loss
        train_mv.get_data_layer = lambda roidb, nc: MixedLayer(roidb)
        cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS = 1, 100
        assert len(train_mv.group_frames_by_shape([MixedLayer(_toy_roidb(2)).forward() for _ in range(1)])) == 1
        net, ref = _ToyNet(3), _ToyNet(3)
        sw = train_mv.SolverWrapper(None, None, net, _ToyImdb(), _toy_roidb(4), str(tmp_path / "snap"))
        sw.log = lambda *_: None
        hist = sw.train_model(None, 1, frames_per_step=2)
        layer = MixedLayer(_toy_roidb(4))
        frames = [layer.forward(), layer.forward()]
        assert frames[0]["image_data"].shape != frames[1]["image_data"].shape
        losses = [toy_total_loss(ref.forward(f))[0] for f in frames]
        want = (losses[0] + losses[1]) / 2
        opt = torch.optim.Adam(ref.parameters(), lr=sw.LEARNING_RATE)
        want.backward()
        opt.step()
        assert abs(hist[0][0] - float(want)) < 1e-6 * abs(float(want))
        for a, b in zip(net.parameters(), ref.parameters()):
            assert torch.allclose(a.detach(), b.detach(), atol=1e-7)
    finally:
        train_mv.total_loss, train_mv.get_data_layer, cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS = old


# ---------------------------------------------------------------------------------------------------- network.load (CPU)
def test_load_skips_a_misfitting_tensor_like_the_reference_does(tmp_path, capsys):
    """network.py:52-64: the ValueError of a tensor that does not fit its variable is caught PER SUBKEY and swallowed under
    ignore_missing (VGG_imagenet.npy's conv1_1 filter is (3,3,3,64), the BEV variable (3,3,9,64)): the biases still load."""
    import types
    from mv3d_tf_amd.networks.mv3d import MV3D
    net = types.SimpleNamespace(params={"conv1_1": [torch.zeros(64, 9, 3, 3), torch.zeros(64)],
                                        "conv1_2": [torch.zeros(64, 64, 3, 3), torch.zeros(64)],
                                        "fc6_1": [torch.zeros(8, 6), torch.zeros(8)]})
    rng = np.random.RandomState(0)
    vgg = {"conv1_1": {"weights": rng.randn(3, 3, 3, 64).astype(np.float32), "biases": rng.randn(64).astype(np.float32)},
           "conv1_2": {"weights": rng.randn(3, 3, 64, 64).astype(np.float32), "biases": rng.randn(64).astype(np.float32)},
           "fc6_1": {"weights": rng.randn(6, 8).astype(np.float32), "biases": rng.randn(8).astype(np.float32)},
           "fc8": {"weights": rng.randn(4, 4).astype(np.float32), "biases": rng.randn(4).astype(np.float32)}}
    path = str(tmp_path / "VGG_imagenet.npy")
    np.save(path, vgg, allow_pickle=True)
    MV3D.load(net, path, ignore_missing=True)
    assert "ignore conv1_1" in capsys.readouterr().out
    assert torch.count_nonzero(net.params["conv1_1"][0]) == 0                          # the misfitting filter is left alone
    assert torch.equal(net.params["conv1_1"][1], torch.as_tensor(vgg["conv1_1"]["biases"]))       # ... its biases load
    assert torch.equal(net.params["conv1_2"][0], torch.as_tensor(vgg["conv1_2"]["weights"]).permute(3, 2, 0, 1))
    assert torch.equal(net.params["fc6_1"][0], torch.as_tensor(vgg["fc6_1"]["weights"]).t())
    with pytest.raises(ValueError):                                                     # without ignore_missing: raise
        MV3D.load(net, path, ignore_missing=False)


@pytest.mark.gpu
def test_adam_step_kernel_equals_torch_adam():
    """mv3d_tf_amd.optim.Adam (ONE launch of mv3d_adam_step for all tensors; lib/fast_rcnn/train_mv.py:138-146's AdamOptimizer) against
    torch.optim.Adam over several steps: tensors of odd sizes (a last partial chunk, a bias of 3), a parameter that gets its first gradient
    later (its own step count / bias corrections), state_dict interchange in both directions."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd import build, optim
    build.build()
    g = torch.Generator().manual_seed(0)
    shapes = [(3,), (64, 9, 3, 3), (4096 * 3 + 17,), (513, 257), (1,)]
    base = [torch.randn(s, generator=g) for s in shapes]
    pa = [b.clone().cuda().requires_grad_(True) for b in base]
    pb = [b.clone().cuda().requires_grad_(True) for b in base]
    oa, ob = optim.Adam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    for it in range(6):
        for k, (x, y) in enumerate(zip(pa, pb)):
            if k == 4 and it < 2:                                     # the last tensor joins at the third step
                x.grad = y.grad = None
                continue
            gr = torch.randn(x.shape, generator=g).cuda() * (10.0 ** (k - 2))
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        for x, y in zip(pa, pb):
            assert torch.allclose(x, y, rtol=1e-6, atol=1e-6), (it, x.shape)      # (a few ulp of the parameter: both sit equally far from an f64 step)
    sa, sb = oa.state_dict(), ob.state_dict()
    for k in sb["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
        assert torch.allclose(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    ob2 = optim.Adam(pb, lr=1e-3)
    ob2.load_state_dict(sb)                                            # torch's state continues on the kernel ...
    oa2 = torch.optim.Adam(pa, lr=1e-3)
    oa2.load_state_dict(sa)                                            # ... and the kernel's on torch
    for x, y in zip(pa, pb):
        gr = torch.randn(x.shape, generator=g).cuda()
        x.grad, y.grad = gr.clone(), gr.clone()
    oa2.step(); ob2.step()
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_adam_step_writes_the_16_bit_copies_and_the_head_uses_them():
    """optim.Adam.register_lowp: the launch that updates a parameter also writes it rounded to bf16 (what a cast at the top of the next step
    would write); the training network keeps the fused head's stacked weight buffers and finds them current after such a step, stale after
    any other change of a head parameter."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd import build, optim
    build.build()
    g = torch.Generator().manual_seed(3)
    ps = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in ((4096 * 2 + 5,), (7, 33), (3,))]
    opt = optim.Adam(ps, lr=1e-2)
    stamped = []
    copies = [torch.zeros(p.numel(), dtype=torch.bfloat16, device="cuda") for p in ps[:2]]
    for p, c in zip(ps, copies):
        opt.register_lowp(p, c, stamped.append)
    for it in range(3):
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g).cuda()
        opt.step()
        for p, c in zip(ps, copies):
            assert torch.equal(c, p.detach().reshape(-1).to(torch.bfloat16)), it
    assert len(stamped) == 6
    # the network: copies current after an attached optimizer's step, stale after a foreign write
    from mv3d_tf_amd.networks import get_network
    net = get_network("MV3D_train_3view")
    net.amp_dtype, net.mfma_trunk = torch.bfloat16, True
    o2 = optim.Adam(net.parameters(), lr=1e-5)
    net.attach_optimizer(o2)
    sfx = ("_1", "_2", "_3")
    bufs, cur = net._held_head(sfx)
    assert cur is False
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    o2.step()
    bufs, cur = net._held_head(sfx)
    assert cur is True and torch.equal(bufs[0][1], net.params["fc6_2"][0].detach().to(torch.bfloat16))
    assert torch.equal(bufs[4][:2], net.params["cls_score"][0].detach().to(torch.bfloat16))
    with torch.no_grad():
        net.params["fc7_1"][1].add_(1.0)
    assert net._held_head(sfx)[1] is False


def test_fused_head_equals_the_op_by_op_head():
    """mv3d_tf_amd.fused_head.FusedHead (the fusion head of MV3D_train.py:159-182 as one autograd function: stacked weights, one batched GEMM
    per layer for all views) against the same head written op by op: outputs and every gradient (pooled maps, fc6 / fc7 of each view,
    cls_score, bbox_pred), fp32 on CPU; with dropout, the mask statistics and the inverted scaling."""
    import torch.nn.functional as F
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd.fused_head import fused_head
    torch.manual_seed(0)
    V, R, C, N = 3, 6, 4, 8
    pools = [torch.randn(R, 7, 7, C, requires_grad=True) for _ in range(V)]
    params = {}
    for t in ("_1", "_2", "_3"):
        params["fc6" + t] = [(torch.randn(N, C * 49) * 0.1).requires_grad_(True), torch.randn(N).requires_grad_(True)]
        params["fc7" + t] = [(torch.randn(N, N) * 0.1).requires_grad_(True), torch.randn(N).requires_grad_(True)]
    params["cls_score"] = [torch.randn(2, V * N).requires_grad_(True), torch.randn(2).requires_grad_(True)]
    params["bbox_pred"] = [torch.randn(48, V * N).requires_grad_(True), torch.randn(48).requires_grad_(True)]
    leaves = pools + [q for k in params for q in params[k]]
    n6, n7 = ["fc6_1", "fc6_2", "fc6_3"], ["fc7_1", "fc7_2", "fc7_3"]
    cls, box, tow = fused_head(pools, params, n6, n7, 1.0, torch.float32)
    ((cls ** 2).sum() + box.sum() * 0.3).backward()
    got = [p.grad.clone() for p in leaves]
    for p in leaves:
        p.grad = None
    tw = []
    for v, t in enumerate(("_1", "_2", "_3")):
        x = pools[v].permute(0, 3, 1, 2).reshape(R, -1)                # NHWC blob -> the reference's (c, h, w) flattening
        x = F.relu(F.linear(x, *params["fc6" + t]))
        tw.append(F.relu(F.linear(x, *params["fc7" + t])))
    f = torch.cat(tw, 1)
    c2, b2 = F.linear(f, *params["cls_score"]), F.linear(f, *params["bbox_pred"])
    ((c2 ** 2).sum() + b2.sum() * 0.3).backward()
    assert torch.allclose(cls, c2, atol=1e-5) and torch.allclose(box, b2, atol=1e-5)
    assert all(torch.allclose(t1, t2, atol=1e-6) for t1, t2 in zip(tow, tw))
    for a, p in zip(got, leaves):
        assert torch.allclose(a, p.grad, atol=1e-4, rtol=1e-4)
    # dropout: about keep_prob of a tower survives, scaled by 1 / keep_prob; the gradient flows only through the survivors
    for p in leaves:
        p.grad = None
    big = [torch.randn(64, 7, 7, C, requires_grad=True) for _ in range(V)]
    cls, box, tow = fused_head(big, params, n6, n7, 0.5, torch.float32)
    box.sum().backward()
    nz = torch.cat([(t != 0).float().mean().reshape(1) for t in tow])
    assert (nz < 0.45).all() and (nz > 0.05).all()                     # ReLU zeroes about half, dropout half of the rest
    assert all(b.grad is not None and torch.isfinite(b.grad).all() for b in big)


def test_rpn_heads_as_one_gemm_equal_the_two_linears():
    """fused_head.RpnHeads (rpn_cls_score + rpn_bbox_pred of MV3D_train.py:88-97 as one GEMM on the stacked filters) against the two
    F.linear calls: outputs and the gradients of the map, both filters and both biases (fp32, CPU)."""
    import torch.nn.functional as F
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd.fused_head import RpnHeads
    torch.manual_seed(1)
    rpn = torch.randn(2, 5, 6, 16, requires_grad=True)
    wc, bc = torch.randn(8, 16, 1, 1, requires_grad=True), torch.randn(8, requires_grad=True)
    wb, bb = torch.randn(24, 16, 1, 1, requires_grad=True), torch.randn(24, requires_grad=True)
    leaves = (rpn, wc, bc, wb, bb)
    s, p = RpnHeads.apply(torch.float32, rpn, wc, bc, wb, bb)
    ((s ** 2).sum() + p.sum()).backward()
    got = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    s2, p2 = F.linear(rpn, wc.reshape(8, -1), bc), F.linear(rpn, wb.reshape(24, -1), bb)
    ((s2 ** 2).sum() + p2.sum()).backward()
    assert s.shape == (2, 5, 6, 8) and p.shape == (2, 5, 6, 24) and torch.allclose(s, s2, atol=1e-5) and torch.allclose(p, p2, atol=1e-5)
    assert all(torch.allclose(a, t.grad, atol=1e-4) for a, t in zip(got, leaves))
