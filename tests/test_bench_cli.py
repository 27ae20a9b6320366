"""bench.py host logic without a GPU: defaults of the driver contract, `--gpus N` re-execution under torch.distributed.run,
the loud failure when no MI355X is visible, and the signature check of the PMC traffic lookup."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    return importlib.import_module("bench")


def test_defaults_match_the_driver_contract(bench, monkeypatch):
    a = bench.parse()
    assert (a.gpus, a.workload, a.launch, a.streams) == (1, "train", "path", 0) and a.steps > 0 and a.warmup >= 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
    assert "3-view" in bench.METRIC and bench.HBM_PEAK_GBS == 8000.0


def test_gpus_n_respawns_one_rank_per_gpu(bench, monkeypatch):
    seen = {}
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    rc = bench.respawn_under_torchrun(bench.parse())
    cmd = seen["cmd"]
    assert rc == 0 and cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert cmd[cmd.index("--master-port") + 2].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_fails_loudly_without_a_gpu(bench, monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "MI355X" in str(e.value)


def test_pmc_traffic_only_for_the_profiled_configuration(bench, tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r03_pmc_traffic.json").write_text(json.dumps({"signatures": {"train/b2/r256/peaky": {"kernels": {
        "void roi_bwd_gather_kernel<1>": {"hbm_bytes_per_launch": 100}, "void roi_bwd_index_kernel<false>": {"hbm_bytes_per_launch": 10},
        "void roi_pool_fwd_xcd_multi_kernel<2>": {"hbm_bytes_per_launch": 7}}}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("roi_bwd_", "train/b2/r256/peaky") == 110            # RoiPoolGrad = its kernels summed
    # the signature names the launch set: a pass of another kernel variant of the same workload is not quoted
    sig = bench.pmc_signature("test", 16, 4800, "peaky", "top-only")
    assert sig == "test/b16/r4800/peaky/top-only" and bench.pmc_traffic("roi_pool_fwd", sig) is None
    (prof / "r06_pmc_traffic.json").write_text(json.dumps({"signatures": {sig: {"kernels": {"void roi_pool_fwd_xcd_multi_kernel<2>": {"hbm_bytes_per_launch": 9}}}}}))
    assert bench.pmc_traffic("roi_pool_fwd", sig) == 9 and bench.pmc_traffic("roi_pool_fwd", bench.pmc_signature("test", 16, 4800, "peaky", "top+argmax")) is None
    assert bench.pmc_traffic("roi_pool_fwd_xcd_multi_kernel", "train/b2/r256/peaky") == 7
    assert bench.pmc_traffic("roi_bwd_", "train/b2/r128/peaky") is None            # another configuration: never a stale number
    assert bench.pmc_traffic("no_such_kernel", "train/b2/r256/peaky") is None


@pytest.mark.gpu
def test_bench_line_contract_on_the_gpu():
    """a short run of the default workload prints ONE JSON line with the driver's keys, the roofline and the CPU baseline"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--ring", "6",
                          "--batches-per-step", "6", "--cpu-seconds", "2", "--no-secondary"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "frames/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 2 * 6 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01      # frames / time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / r["avg_launch_us"] / 1e3) / r["achieved"] < 0.01
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert "workload" in d["config"] and "model" not in d["config"]
    # the line verifies what it timed and reports the roofline in the mode it timed
    v = d["verified"]
    assert v["bit_exact"] is True and v["batches"] == 2 and v["rows"] > 0 and v["mismatches"] == []
    fl = r["in_flight"]
    assert fl["batches_in_flight"] == 8 and fl["forward_us"] > 0 and fl["backward_us"] > 0
    assert d["config"]["host_cpu_s_per_step"] > 0 and d["config"]["host_cores"]["usable"] >= 1 and d["config"]["host_cores"]["pinned"] is None
    assert all("in_flight_us" in e for e in d["roofline_kernels"] if e["bound"] == "hbm")
    nms = [e for e in d["roofline_kernels"] if "greedy NMS" in e["kernel"]]          # achieved time vs the serial-chain lower bound
    assert len(nms) == 2 and all("error" not in e and 0 < e["frac"] <= 1 and e["blocks_visited"] <= e["blocks_total"] for e in nms)


@pytest.mark.gpu
def test_path_driver_depth8_equals_the_oracle(bench):
    """the headline's own driver (bench.PathDriver, depth 8: eight batches in flight on eight streams, argument structs at capacity
    with the row count set per batch) against the oracle, after it has been run the way the timed loop runs it"""
    import torch
    from mv3d_tf_amd import build, hot_path, synth
    from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml, cfg
    build.build()
    saved = {k: cfg.TRAIN[k] for k in ("BG_THRESH_LO", "BG_THRESH_HI", "FG_THRESH")}
    apply_end2end_yml()
    try:
        dev = torch.device("cuda")
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
        host, inputs, maps = [], [], []
        for k in range(5):
            frames = [synth.rpn_head(7700 + 2 * k + b, 76, 76, "peaky", return_gt=True) for b in range(2)]
            host.append(frames)
            inputs.append((t(np.concatenate([f[0] for f in frames])), t(np.concatenate([f[1] for f in frames])),
                           t(np.concatenate([f[2] for f in frames])), t(np.stack([f[3] for f in frames])),
                           [tuple(t(a) for a in f[4]) for f in frames]))
            maps.append(hot_path.synth_maps(2, 60 + k, dev))
        drv = bench.PathDriver(inputs, maps, depth=8)
        np.random.seed(11)
        drv.run(37)                                                  # (every slot reused several times, cursor not a multiple of anything)
        v = drv.verify(host, nbatches=5, seed=77)
        assert v["bit_exact"], v["mismatches"]
        assert v["batches"] == 5 and v["rows"] > 5 * 2 * 64
        fl = drv.in_flight_us(nb=24)
        assert fl["forward_us"] > 0 and fl["backward_us"] > 0 and fl["calls_timed"] == 16
        v2 = drv.verify(host, nbatches=2, seed=78)                   # ... and still right after the event-marked run
        assert v2["bit_exact"], v2["mismatches"]
        drv.close()
    finally:
        for k, val in saved.items():
            cfg.TRAIN[k] = val


@pytest.mark.gpu
def test_secondary_lines_and_two_gloo_ranks_with_trunk_on_one_gpu():
    """`bench.py --gpus 2 --dist-backend gloo` on the 1-GPU box (both ranks share the device): the sharded path-only line, the
    fresh-input line and the full training step WITH the trunks -- i.e. the bucketed gradient all-reduce overlapping a real
    backward pass of the 143 M-parameter graph has run (gloo here; RCCL needs one GPU per rank)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
                          "--ring", "3", "--batches-per-step", "6", "--no-cpu-baseline", "--secondary-seconds", "0.25"], capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    c = d["config"]                                                  # the N > 1 host plan as the line states it
    assert len(c["host_cpu_s_per_step_per_rank"]) == 2 and all(x > 0 for x in c["host_cpu_s_per_step_per_rank"])
    assert c["host_cores"]["usable"] >= 1 and isinstance(c["host_cores"]["plan"], str) and c["host_cores"]["plan"]
    assert c["parallelism"] == "frames/2" and c["launch"] in ("path", "graph") and c["pmc_signature"].endswith("/pair-tiles-planned")
    sec = d["secondary"]
    fr = sec["fresh_inputs"]
    assert fr["frames_per_s"] > 0 and fr["host_draw_ms_per_frame"] > 0 and 0 < fr["fraction_of_resident_replay"]
    wt = sec["with_trunk"]
    assert wt["frames_per_s"] > 0 and wt["allreduce_buckets"] >= 8 and wt["gradient_bytes_per_step"] > 500e6
    assert sec["test_cfg"]["frames_per_s"] > 0
    assert wt["fp32_mfma_trunk"]["frames_per_s"] > 0
    mp = wt["bf16_mfma_trunk"]
    # (no speed relation asserted: two ranks time-share ONE GPU here)
    assert mp["frames_per_s"] > 0 and mp["roofline_kernels"][0]["bound"] == "mfma" and 0 < mp["roofline_kernels"][0]["frac"] < 1
    sv = sec["serving_with_trunk"]
    assert sv["fp32"]["frames_per_s"] > 0 and sv["fp16_mfma"]["frames_per_s"] > 0 and sv["fp16_mfma"]["rois_per_step"] > 0
    assert sv["fp32_mfma"]["frames_per_s"] > 0
    g = sv["fp16_mfma_graph"]                                        # the whole serving step as one hipGraph replay, counts read after the window
    assert g["frames_per_s"] > 0 and g["rois_per_step"] > 0 and g["steps_timed"] >= 8 and g["ms_per_step_min"] <= g["ms_per_step_median"]
    assert wt["steps_timed"] >= 8 and wt["ms_per_step_min"] <= wt["ms_per_step_median"]
    nms = [e for e in d["roofline_kernels"] if "greedy NMS" in e["kernel"]]
    assert len(nms) == 2 and all(0 < e["frac"] <= 1 and e["blocks_visited"] <= e["blocks_total"] for e in nms)
    rk = sv["roofline_kernels"][0]
    assert rk["bound"] == "mfma" and rk["peak"] == 2500.0 and 0 < rk["frac"] < 1 and abs(rk["frac"] - rk["achieved"] / rk["peak"]) < 1e-3
    r32 = sv["roofline_kernels"][1]
    assert r32["peak"] == 157.3 and 0 < r32["frac"] < 1


def test_conv_roofline_accounting():
    """the layer table behind bench.py's MFMA roofline entries (host logic): 40 3x3 convolutions in the 3-view serving graph, the
    reference's shapes (lib/networks/MV3D_train.py:44-84), 11.2 TFLOP algorithmic at batch 16"""
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd.networks.mv3d import _VGG
    from mv3d_tf_amd.trunk import serving_layers
    rows = serving_layers(_VGG)
    assert len(rows) == 40 and sum(1 for r in rows if r[0] == "rpn_conv/3x3") == 1
    by = {r[0]: r for r in rows}
    assert by["conv1_1"][1:] == (608, 608, 9, 64) and by["conv5_3"][1:] == (76, 76, 512, 512) and by["rpn_conv/3x3"][1:] == (76, 76, 512, 512)
    assert by["conv5_3_2"][1:] == (46, 155, 512, 512) and by["conv4_1_3"][1:] == (8, 64, 256, 512) and by["conv1_2_3"][1:] == (64, 512, 64, 64)
    flop = sum(2.0 * 16 * H * W * cout * 9 * cin for _, H, W, cin, cout in rows)
    assert abs(flop - 11.2e12) < 0.05e12


def test_cpu_baseline_worker_and_usable_cores(tmp_path):
    """bench.py's all-core CPU baseline: one worker PROCESS per usable core runs the C oracle on whole frames between a
    common start and stop time (oracle/cpu_worker.py; its own numpy global RNG, no lock); `usable_cores` = affinity cut by the
    container's CPU quota.  Small frame here (24 x 24 head, small maps): the worker reports frames done and its elapsed time."""
    import importlib.util
    import time
    sys.path.insert(0, ROOT)
    from mv3d_tf_amd import hot_path, synth
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    n, note = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and isinstance(note, str)
    prob, pred, info, calib, (gt_bv, gt_3d, gt_cnr) = synth.rpn_head(5, 24, 24, "peaky", return_gt=True)
    rng = np.random.RandomState(0)
    arrays = dict(prob=prob, pred=pred, info=info, calib=calib, gt_bv=gt_bv, gt_3d=gt_3d, gt_cnr=gt_cnr)
    for v, (H, W, Cc) in zip(("bev", "rgb", "fv"), ((24, 24, 64), (12, 40, 64), (8, 16, 64))):
        arrays["map_" + v] = rng.rand(1, H, W, Cc).astype(np.float32)
    arrays.update({"cfg_" + k: np.asarray(v) for k, v in hot_path.TRAIN_CFG.items()})
    path = str(tmp_path / "in.npz")
    np.savez(path, **arrays)
    start = time.time() + 1.5
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), path, "train", repr(start), repr(start + 0.5), str(k)],
                              stdout=subprocess.PIPE, text=True) for k in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=120)
        frames, secs = out.split()
        assert int(frames) >= 1 and 0.0 < float(secs) < 30.0


def test_host_thread_plan_pins_core_pairs_or_falls_back(bench):
    """VERDICT r04 #6b: N ranks share a node's cores -- every rank's submitting + drawing thread get their own core pair, or the
    bench says that the host cannot carry `--launch path` and times the frozen-batch replay instead"""
    one = bench.plan_host_threads(1, 0, 16, range(16))
    assert one["launch"] == "path" and one["cores"] is None
    mask = [3, 4, 5, 6, 7, 8, 9, 10, 40, 41, 42, 43, 44, 45, 46, 47]            # a 16-core affinity mask, not starting at 0
    plans = [bench.plan_host_threads(8, r, 16, mask) for r in range(8)]
    assert all(p["launch"] == "path" for p in plans)
    pairs = [tuple(p["cores"]) for p in plans]
    assert pairs[0] == (3, 4) and pairs[7] == (46, 47) and len({c for pr in pairs for c in pr}) == 16      # disjoint pairs
    few = bench.plan_host_threads(8, 2, 12, mask)                               # quota 12 < 2 x 8
    assert few["launch"] == "graph" and few["cores"] is None and "12 usable host cores for 8 ranks" in few["note"]
    assert bench.plan_host_threads(8, 2, 16, mask[:10])["launch"] == "graph"      # the mask itself is too small
    n, note = bench.usable_cores()
    assert n >= 1 and isinstance(note, str)
