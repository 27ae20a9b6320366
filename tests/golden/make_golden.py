#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE itself.

Runs only in the build container (needs /root/reference, Cython, gcc); the outputs
(*.npz: inputs or input seeds + the reference's outputs) are committed, the reference
never is.  What it does:

  1. copies the hot-path Python/Cython files of /root/reference/lib into a temp dir,
  2. runs lib2to3 over them (print / xrange / reduce ...), applies the three
     Python-2 integer-division fixes and numpy-2 spelling fixes listed below,
  3. cythonizes cpu_nms.pyx / bbox.pyx / nms.pyx (keeping `np.float thresh`, i.e. the
     Python-float compare that defines the reference's NMS semantics),
  4. calls the reference functions on seeded inputs (mv3d_tf_amd.synth) and stores
     what they return.

Usage:  python tests/golden/make_golden.py  [--keep-scratch]
numpy version and the scratch patches are recorded inside every fixture.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference/lib"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from mv3d_tf_amd import synth  # noqa: E402

PATCH_NOTE = ("lib2to3; BATCH_SIZE//num_images; deltas.shape[1]//24; corners.shape[1]//24; "
              "yaml.safe_load; np.int_t->np.intp_t; dtype=np.int->np.intp; DTYPE=np.float64; "
              "np.float=float, np.int=int aliases; easydict shim; USE_GPU_NMS=False")

EASYDICT = '''
class EasyDict(dict):
    def __init__(self, d=None, **kw):
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)
    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        dict.__setitem__(self, k, v)
    __setitem__ = __setattr__
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
'''

SETUP = '''
from setuptools import setup, Extension
from Cython.Build import cythonize
import numpy as np
a = ["-O2", "-Wno-cpp", "-Wno-unused-function"]
exts = [Extension("nms.cpu_nms", ["lib/nms/cpu_nms.pyx"], include_dirs=[np.get_include()], extra_compile_args=a),
        Extension("utils.cython_bbox", ["lib/utils/cython_bbox.pyx"], include_dirs=[np.get_include()], extra_compile_args=a),
        Extension("utils.cython_nms", ["lib/utils/cython_nms.pyx"], include_dirs=[np.get_include()], extra_compile_args=a)]
setup(ext_modules=cythonize(exts, language_level=2), script_args=["build_ext", "--build-lib", "lib"])
'''


def sub(path, pat, rep, count=0):
    s = open(path).read()
    s2, n = re.subn(pat, rep, s, count=count)
    assert n > 0, (path, pat)
    open(path, "w").write(s2)


def build_scratch(d):
    L = os.path.join(d, "lib")
    for p in ("rpn_msr", "fast_rcnn", "utils", "nms"):
        os.makedirs(os.path.join(L, p))
    os.makedirs(os.path.join(d, "shim"))
    for f in ("__init__", "generate_anchors", "proposal_layer_tf", "anchor_target_layer_tf",
              "proposal_target_layer_tf"):
        shutil.copy(f"{REF}/rpn_msr/{f}.py", f"{L}/rpn_msr/")
    for f in ("bbox_transform", "config", "nms_wrapper"):
        shutil.copy(f"{REF}/fast_rcnn/{f}.py", f"{L}/fast_rcnn/")
    shutil.copy(f"{REF}/utils/transform.py", f"{L}/utils/")
    shutil.copy(f"{REF}/utils/read_lidar.py", f"{L}/utils/")
    os.makedirs(os.path.join(d, "shim", "matplotlib"))
    open(f"{d}/shim/matplotlib/__init__.py", "w").close()
    open(f"{d}/shim/matplotlib/pyplot.py", "w").close()
    for p in ("fast_rcnn", "utils", "nms"):
        open(f"{L}/{p}/__init__.py", "w").close()
    shutil.copy(f"{REF}/nms/cpu_nms.pyx", f"{L}/nms/cpu_nms.pyx")
    shutil.copy(f"{REF}/utils/bbox.pyx", f"{L}/utils/cython_bbox.pyx")
    shutil.copy(f"{REF}/utils/nms.pyx", f"{L}/utils/cython_nms.pyx")
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", L],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sub(f"{L}/rpn_msr/proposal_target_layer_tf.py", r"cfg\.TRAIN\.BATCH_SIZE / num_images",
        "cfg.TRAIN.BATCH_SIZE // num_images")
    sub(f"{L}/fast_rcnn/bbox_transform.py", r"deltas\.shape\[1\]/24", "deltas.shape[1]//24")
    sub(f"{L}/utils/transform.py", r"num_class = corners\.shape\[1\] / 24", "num_class = corners.shape[1] // 24")
    sub(f"{L}/fast_rcnn/config.py", r"yaml\.load\(f\)", "yaml.safe_load(f)")
    for f in (f"{L}/nms/cpu_nms.pyx", f"{L}/utils/cython_nms.pyx"):
        sub(f, r"np\.int_t", "np.intp_t")
        sub(f, r"dtype=np\.int\)", "dtype=np.intp)")
    sub(f"{L}/utils/cython_bbox.pyx", r"(?m)^DTYPE = np\.float$", "DTYPE = np.float64")
    open(f"{d}/shim/easydict.py", "w").write(EASYDICT)
    open(f"{d}/setup_nat.py", "w").write(SETUP)
    subprocess.check_call([sys.executable, "setup_nat.py"], cwd=d, stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    np.float = float
    np.int = int
    for p in ("lib", "lib/rpn_msr", "shim"):
        sys.path.insert(0, os.path.join(d, p))


KITTI_CALIB_TXT = """P0: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 0.000000000000e+00 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
P1: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 -3.875744000000e+02 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
P2: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 4.485728000000e+01 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 2.163791000000e-01 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 2.745884000000e-03
P3: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 -3.395242000000e+02 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 2.199936000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 2.729905000000e-03
R0_rect: 9.999239000000e-01 9.837760000000e-03 -7.445048000000e-03 -9.869795000000e-03 9.999421000000e-01 -4.278459000000e-03 7.402527000000e-03 4.351614000000e-03 9.999631000000e-01
Tr_velo_to_cam: 7.533745000000e-03 -9.999714000000e-01 -6.166020000000e-04 -4.069766000000e-03 1.480249000000e-02 7.280733000000e-04 -9.998902000000e-01 -7.631618000000e-02 9.998621000000e-01 7.523790000000e-03 1.480755000000e-02 -2.717806000000e-01
Tr_imu_to_velo: 9.999976000000e-01 7.553071000000e-04 -2.035826000000e-03 -8.086759000000e-01 -7.854027000000e-04 9.998898000000e-01 -1.482298000000e-02 3.195559000000e-01 2.024406000000e-03 1.482454000000e-02 9.998881000000e-01 -7.997231000000e-01
"""


def gen_kitti(d):
    """KITTI calib / label files -> GT encodings, through the reference's own loader
    (lib/datasets/kitti_mv3d.py, imported with a stub `datasets` package: the real one drags in every dataset) and
    the gt part of lib/roi_data_layer/minibatch_mv3d.py:get_minibatch (cv2.imread stubbed to np.load)."""
    L = os.path.join(d, "lib")
    os.makedirs(f"{L}/datasets", exist_ok=True)
    os.makedirs(f"{L}/roi_data_layer", exist_ok=True)
    shutil.copy(f"{REF}/datasets/kitti_mv3d.py", f"{L}/datasets/kitti_mv3d.py")
    shutil.copy(f"{REF}/roi_data_layer/minibatch_mv3d.py", f"{L}/roi_data_layer/minibatch_mv3d.py")
    shutil.copy(f"{REF}/utils/boxes_grid.py", f"{L}/utils/boxes_grid.py")
    shutil.copy(f"{REF}/utils/blob.py", f"{L}/utils/blob.py")
    open(f"{L}/roi_data_layer/__init__.py", "w").close()
    open(f"{L}/datasets/__init__.py", "w").write("import os.path as osp\nROOT_DIR = osp.dirname(__file__)\nfrom .imdb import imdb\n")
    open(f"{L}/datasets/imdb.py", "w").write("class imdb(object):\n    def __init__(self, name):\n        self._name = name\n"
                                               "    name = property(lambda s: s._name)\n    classes = property(lambda s: s._classes)\n"
                                               "    num_classes = property(lambda s: len(s._classes))\n"
                                               "    image_index = property(lambda s: s._image_index)\n")
    open(f"{d}/shim/cv2.py", "w").write("import numpy as np\ndef imread(p):\n    return np.load(p + '.npy')\n")
    for f in (f"{L}/datasets/kitti_mv3d.py", f"{L}/roi_data_layer/minibatch_mv3d.py", f"{L}/utils/boxes_grid.py", f"{L}/utils/blob.py"):
        subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", f], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from datasets.kitti_mv3d import kitti_mv3d
    from roi_data_layer.minibatch_mv3d import get_minibatch
    rng = np.random.RandomState(71)
    root = os.path.join(d, "KITTI")
    for sub_ in ("ImageSets", "object/training/calib", "object/training/label_2", "object/training/image_2", "object/training/lidar_bv"):
        os.makedirs(os.path.join(root, sub_))
    types = ["Car", "Van", "Car", "Pedestrian", "DontCare", "Car", "Cyclist", "Car", "Truck", "Car", "Car", "Misc", "Car"]
    frames, calib_txt = [], []
    for fi in range(3):
        lines = []
        for t in (types if fi < 2 else ["DontCare", "Van"]):
            tx, tz = rng.uniform(-20, 20), rng.uniform(4, 58)
            ty = rng.uniform(1.2, 2.0)
            h, w, l = rng.uniform(1.3, 1.9), rng.uniform(1.4, 1.9), rng.uniform(3.0, 4.8)
            ry = rng.uniform(-3.14, 3.14)
            x1, y1 = rng.uniform(0, 1100), rng.uniform(100, 300)
            vals = [rng.choice([0.0, 0.25, 0.6]), int(rng.randint(0, 3)), rng.uniform(-3.14, 3.14), x1, y1, x1 + rng.uniform(20, 140),
                    y1 + rng.uniform(15, 70), h, w, l, tx, ty, tz, ry]
            if t == "DontCare":
                vals[7:14] = [-1, -1, -1, -1000, -1000, -1000, -10]
            lines.append(t + " " + " ".join(("%d" % v) if isinstance(v, int) else ("%.2f" % v) for v in vals))
        txt = "\n".join(lines) + "\n"
        # per-frame calibration: frame 0 = KITTI 000000 values, the others perturbed in the printed digits
        ctxt = KITTI_CALIB_TXT
        if fi:
            rows = []
            for row in KITTI_CALIB_TXT.strip().split("\n"):
                k, v = row.split(": ")
                v = [float(x) * (1 + 1e-3 * rng.uniform(-1, 1)) for x in v.split(" ")]
                rows.append(k + ": " + " ".join("%.12e" % x for x in v))
            ctxt = "\n".join(rows) + "\n"
        idx = "%06d" % fi
        open(os.path.join(root, "object/training/label_2", idx + ".txt"), "w").write(txt)
        open(os.path.join(root, "object/training/calib", idx + ".txt"), "w").write(ctxt)
        np.save(os.path.join(root, "object/training/image_2", idx + ".png.npy"), rng.randint(0, 255, (12, 40, 3)).astype(np.uint8))
        open(os.path.join(root, "object/training/image_2", idx + ".png"), "w").close()
        np.save(os.path.join(root, "object/training/lidar_bv", idx + ".npy"), rng.random_sample((601, 601, 9)).astype(np.float32)[:8, :9])
        frames.append(txt); calib_txt.append(ctxt)
    open(os.path.join(root, "ImageSets", "train.txt"), "w").write("000000\n000001\n000002\n")
    db = kitti_mv3d("train", root)
    kw = dict(n_frames=3)
    from fast_rcnn.config import cfg
    for fi in range(3):
        idx = "%06d" % fi
        ann = db._load_kitti_annotation(idx)
        cal = db.calib_at(fi)
        kw["labels_txt_%d" % fi] = np.array(frames[fi]); kw["calib_txt_%d" % fi] = np.array(calib_txt[fi])
        kw["calib_%d" % fi] = cal
        for k_, v in ann.items():
            if k_ == "gt_overlaps":
                v = v.toarray()
            kw["ann%d_%s" % (fi, k_)] = np.asarray(v)
        entry = dict(ann, image_path=db.image_path_at(fi), lidar_bv_path=db.lidar_path_at(fi), calib=cal)
        blobs = get_minibatch([entry], db.num_classes)
        for k_ in ("gt_boxes", "gt_boxes_bv", "gt_boxes_3d", "gt_boxes_corners", "im_info", "calib"):
            kw["blob%d_%s" % (fi, k_)] = np.asarray(blobs[k_])
    save("kitti_label", **kw)


ONLY_NAMES = None          # --only extra: write just these fixtures (the others are computed and left untouched on disk)
EXTRA_R03 = ("proposal3d_75_TRAIN_peaky", "proposal3d_75_TEST_rand", "anchor_target_76_no_gt_overlap")


def save(name, **kw):
    if ONLY_NAMES is not None and name not in ONLY_NAMES:
        return
    kw["numpy_version"] = np.__version__
    kw["scratch_patches"] = PATCH_NOTE
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"{name:42s} {os.path.getsize(path) / 1024:8.1f} KB")


def gen_bev_ranges():
    """point_cloud_2_top with its own parameters (lib/utils/read_lidar.py:10-16) on clouds with points ON the limits (round 5)"""
    from utils.read_lidar import point_cloud_2_top
    for k, case in enumerate(sorted(synth.BEV_RANGE_CASES)):
        res, zres, side, fwd, hr = synth.BEV_RANGE_CASES[case]
        pts = synth.point_cloud_ranges(70 + k, 30000, case)
        top = point_cloud_2_top(pts, res=res, zres=zres, side_range=side, fwd_range=fwd, height_range=hr)
        nz = np.flatnonzero(top)
        save(f"point_cloud_top_ranges_{case}", seed=70 + k, P=30000, sha_points=synth.sha256(pts), shape=np.array(top.shape),
             nz_index=nz.astype(np.int32), nz_value=top.ravel()[nz], sha_top=synth.sha256(top))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep-scratch", action="store_true")
    ap.add_argument("--only", default="", help="'kitti': regenerate only the KITTI label / calib fixture; 'bev': only the rasteriser's parameter cases; 'extra': write only "
                                               "the fixtures added in round 3 (EXTRA_R03)")
    args = ap.parse_args()
    d = tempfile.mkdtemp(prefix="mv3d_ref_")
    build_scratch(d)
    if args.only == "kitti":
        gen_kitti(d)
        if not args.keep_scratch:
            shutil.rmtree(d, ignore_errors=True)
        return
    if args.only == "bev":
        gen_bev_ranges()
        if not args.keep_scratch:
            shutil.rmtree(d, ignore_errors=True)
        return
    if args.only == "extra":
        global ONLY_NAMES
        ONLY_NAMES = EXTRA_R03
    from fast_rcnn.config import cfg
    cfg.USE_GPU_NMS = False
    from rpn_msr.proposal_layer_tf import proposal_layer_3d
    from rpn_msr.anchor_target_layer_tf import anchor_target_layer
    from rpn_msr.proposal_target_layer_tf import proposal_target_layer_3d
    from nms.cpu_nms import cpu_nms
    from utils.cython_bbox import bbox_overlaps
    from utils.cython_nms import nms as cython_nms
    import utils.transform as T
    import fast_rcnn.bbox_transform as BT
    from generate_anchors import generate_anchors_bv

    # ---- a1 / constants
    save("anchors_bv", base=generate_anchors_bv(), Xn=np.int64(T.Xn), Yn=np.int64(T.Yn))

    # ---- numpy arithmetic probes the restatement leans on
    rng = np.random.RandomState(11)
    a = np.concatenate([rng.uniform(-5, 70, 20000), np.arange(0, 700) * 0.1, np.arange(0, 700) * 0.1 + 1e-9,
                        np.arange(1, 700) * 0.1 - 1e-9, (np.arange(0, 7000) * np.float32(0.01)).astype(np.float32).astype(np.float64)])
    save("floor_divide", a=a, q=a // 0.1)
    x32 = np.concatenate([rng.uniform(-3, 3, 20000), rng.uniform(-80, 80, 2000), [0.0, -0.0, 88.0, -100.0]]).astype(np.float32)
    x64 = rng.uniform(0.05, 20, 20000)
    save("exp_log", x32=x32, exp32=np.exp(x32), x64=x64, log64=np.log(x64))

    # ---- a4
    b_int = np.floor(rng.uniform(0, 600, (400, 2)))
    b_int = np.hstack([b_int, b_int + np.floor(rng.uniform(0, 60, (400, 2)))])
    q_int = np.floor(rng.uniform(0, 600, (9, 2)))
    q_int = np.hstack([q_int, q_int + np.floor(rng.uniform(5, 60, (9, 2)))])
    q_int[0] = b_int[0]
    save("bbox_overlaps_int", boxes=b_int, query=q_int, overlaps=bbox_overlaps(b_int, q_int))
    b_fr = rng.uniform(0, 100, (300, 2)); b_fr = np.hstack([b_fr, b_fr + rng.uniform(-2, 30, (300, 2))])
    q_fr = rng.uniform(0, 100, (7, 2)); q_fr = np.hstack([q_fr, q_fr + rng.uniform(0, 30, (7, 2))])
    save("bbox_overlaps_frac", boxes=b_fr, query=q_fr, overlaps=bbox_overlaps(b_fr, q_fr))

    # ---- a15 NMS
    for n, var, thr, seed in ((6000, "rand", 0.7, 21), (6000, "clustered", 0.7, 22), (12000, "rand", 0.7, 23),
                              (12000, "clustered", 0.7, 24), (300, "clustered", 0.1, 25), (300, "rand", 0.5, 26),
                              (1000, "clustered", 0.7, 27), (65, "clustered", 0.5, 28), (1, "rand", 0.7, 29)):
        dets = synth.nms_dets(seed, n, var, integer=(seed != 27))
        keep = cpu_nms(dets, thr)
        assert keep == cython_nms(dets, thr)
        order = np.argsort(-dets[:, 4], kind="stable")
        keep_sorted = cpu_nms(np.ascontiguousarray(dets[order]), thr)
        save(f"nms_{n}_{var}", seed=seed, variant=var, integer=(seed != 27), thresh=thr, sha=synth.sha256(dets),
             dets=(dets if n <= 1000 else np.zeros(0)), keep=np.array(keep, np.int32),
             keep_presorted=np.array(keep_sorted, np.int32))
    # exact-rational IoU pairs that separate the double compare from a float compare
    ex = [(np.array([[0, 0, 9, 9, .9], [0, 0, 9, 6, .8]], np.float32), 0.7),
          (np.array([[0, 0, 9, 9, .9], [0, 0, 9, 4, .8]], np.float32), 0.5),
          (np.array([[0, 0, 9, 9, .9], [0, 0, 9, 0, .8]], np.float32), 0.1),
          (np.array([[0, 0, 9, 9, .9], [0, 0, 9, 9, .8], [20, 20, 30, 30, .7], [20, 20, 30, 30, .95]], np.float32), 0.7)]
    save("nms_exact_iou", **{f"dets{i}": e[0] for i, e in enumerate(ex)},
         **{f"thresh{i}": e[1] for i, e in enumerate(ex)},
         **{f"keep{i}": np.array(cpu_nms(e[0], e[1]), np.int32) for i, e in enumerate(ex)})
    deg = np.array([[5, 5, 4, 4, .9], [5, 5, 4, 4, .8]], np.float32)   # zero areas -> union 0
    try:
        cpu_nms(deg, 0.7); zde = 0
    except ZeroDivisionError:
        zde = 1
    save("nms_degenerate", dets=deg, thresh=0.7, raises_zero_division=zde)

    # ---- a7 proposal_layer_3d (+ intermediates for two cases)
    for H, key, var, seed, full in ((76, "TRAIN", "rand", 3, True), (76, "TEST", "peaky", 4, False),
                                    (76, "TRAIN", "peaky", 5, False), (75, "TEST", "peaky", 6, True),
                                    (75, "TRAIN", "rand", 7, False), (76, "TEST", "rand", 8, False),
                                    (20, "TRAIN", "peaky", 9, True),
                                    (75, "TRAIN", "peaky", 10, False), (75, "TEST", "rand", 12, False)):   # (Appendix D leftovers, r03)
        if key == "TEST":   # experiments/cfgs/faster_rcnn_end2end.yml:15-20
            cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 6000, 300
        prob, pred, im_info, calib = synth.rpn_head(seed, H, H, var)
        bv, img, b3 = proposal_layer_3d(prob, pred, im_info, calib, key, [8, ], [1.0, 1.0])
        kw = dict(seed=seed, H=H, W=H, variant=var, cfg_key=key, sha=synth.sha256(prob, pred, im_info, calib),
                  pre=cfg[key].RPN_PRE_NMS_TOP_N, post=cfg[key].RPN_POST_NMS_TOP_N, thresh=cfg[key].RPN_NMS_THRESH,
                  min_size=cfg[key].RPN_MIN_SIZE, blob_bv=bv, blob_img=img, blob_3d=b3)
        if full:
            # intermediates re-derived with the reference's own helpers (same call sequence as :79-147)
            A = generate_anchors_bv()
            sx, sy = np.meshgrid(np.arange(0, H) * 8, np.arange(0, H) * 8)
            shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
            anchors = (A.reshape((1, 4, 4)) + shifts.reshape((1, -1, 4)).transpose((1, 0, 2))).reshape((-1, 4))
            a3 = T.bv_anchor_to_lidar(anchors)
            p3 = BT.bbox_transform_inv_3d(a3, pred.reshape((-1, 6)))
            pbv = T.lidar_3d_to_bv(p3)
            cnr = T.lidar_3d_to_corners(p3)
            pim = T.lidar_cnr_to_img(cnr, calib[3], calib[2], calib[0])
            assert np.abs(pbv).max() < 30000
            kw.update(anchors3d=a3, props3d=p3, bv_raw=pbv.astype(np.int16), bv_raw_is_integral=bool((pbv == np.round(pbv)).all()),
                      img=pim)
        save(f"proposal3d_{H}_{key}_{var}", **kw)

    # ---- a11 edge cases: boxes straddling / behind the camera plane, huge boxes, NaN
    edge = np.array([[0.2, 0.0, -0.9, 3.9, 1.6, 1.5], [-5.0, 2.0, -0.9, 4.0, 1.6, 1.5], [0.27, 0.0, -1.0, 0.0, 0.0, 0.0],
                     [30.0, 0.0, -1.0, 200.0, 200.0, 3.0], [1e-3, 0.0, 0.0, 1.0, 1.0, 1.0], [10.0, -3.0, -1.0, 4.0, 1.7, 1.5],
                     [np.nan, 0.0, 0.0, 1.0, 1.0, 1.0], [1e30, 0, 0, 1, 1, 1], [60.0, 30.0, 0.4, 4.5, 1.8, 1.7]], np.float32)
    cn = T.lidar_3d_to_corners(edge)
    with np.errstate(all="ignore"):
        im = T.lidar_cnr_to_img(cn, synth.KITTI_CALIB[3], synth.KITTI_CALIB[2], synth.KITTI_CALIB[0])
    save("project_edge", boxes3d=edge, corners=cn, img=im, calib=synth.KITTI_CALIB)
    # projection-matrix probe on perturbed calibrations
    calibs = []; mats = []
    for t in range(64):
        c = synth.KITTI_CALIB.copy()
        if t:
            c[0] = (c[0] * (1 + 0.01 * rng.uniform(-1, 1, 12))).astype(np.float32)
            c[2] = (c[2] + 0.001 * rng.uniform(-1, 1, 12)).astype(np.float32); c[2, 9:] = 0
            c[3] = (c[3] + 0.01 * rng.uniform(-1, 1, 12)).astype(np.float32)
        calibs.append(c)
        mats.append(np.dot(np.dot(c[0].reshape(3, 4), c[2].reshape(4, 3)), c[3].reshape(3, 4)))
    save("proj_matrix", calibs=np.array(calibs), mats=np.array(mats))

    # ---- a3 anchor_target_layer
    def at_case(name, H, gtbv, gt3d, seed):
        size = synth.BEV_SIZE[H]
        im_info = np.array([[size, size, 1]], np.float32)
        score = np.zeros((1, H, H, 8), np.float32)
        np.random.seed(seed)
        lab, tg, anc, anc3 = anchor_target_layer(score, gtbv, gt3d, im_info, [8, ], [1.0, 1.0])
        sel = np.where(lab != -1)[0]
        rs = np.random.RandomState(0).permutation(lab.shape[0])[:3000]
        rows = np.union1d(sel, rs)
        save(f"anchor_target_{H}_{name}", H=H, gt_bv=gtbv, gt_3d=gt3d, im_info=im_info, np_seed=seed,
             labels=lab.astype(np.int8), target_rows=rows.astype(np.int32), targets=tg[rows],
             targets_nonzero_rows=np.where(np.any(tg != 0, 1))[0].astype(np.int32),
             anchors=anc, anchors_3d=anc3)

    r = np.random.RandomState(31)
    for H in (76, 75):
        gtbv, gt3d, _ = synth.gt_cars(r, 3)
        at_case("normal", H, gtbv, gt3d, 3)
        gtbv2, gt3d2 = gtbv.copy(), gt3d.copy()
        gtbv2[1, :4] = [700, 650, 716, 690]      # GT outside the BEV map -> zero-overlap flood
        at_case("gt_outside", H, gtbv2[:2], gt3d2[:2], 4)
        gtbv3, gt3d3, _ = synth.gt_cars(r, 20)
        at_case("many_gt", H, gtbv3, gt3d3, 5)
    gtb = np.array([[100, 100, 103, 102, 1]], np.float32)  # tiny GT: low IoU everywhere
    at_case("tiny_gt", 76, gtb, np.array([[49.8, 19.8, -0.95, 0.3, 0.4, 1.5, 1]], np.float32), 6)
    # a ground-truth box INSIDE the map that no inside-image anchor overlaps (the map corner: every anchor that reaches it
    # crosses the border and is filtered): its gt-argmax row is all zeros -> the zero-overlap flood, next to a normal car
    gtc_ = np.array([[0, 0, 5, 3, 1], [300, 280, 339, 296, 1]], np.float32)
    at_case("no_gt_overlap", 76, gtc_, np.array([[59.9, 29.9, -0.95, 0.4, 0.6, 1.5, 1], [32.0, -1.9, -0.95, 3.9, 1.6, 1.56, 1]], np.float32), 8)

    # ---- a17 proposal_target_layer_3d
    for name, seed, ngt in (("few", 41, 2), ("many", 42, 12)):
        prob, pred, im_info, calib = synth.rpn_head(seed, 76, 76, "peaky")
        r = np.random.RandomState(seed)
        gtbv, gt3d, gtc = synth.gt_cars(r, ngt)
        cfg.TRAIN.RPN_POST_NMS_TOP_N = 2000
        bv, img, b3 = proposal_layer_3d(prob, pred, im_info, calib, "TRAIN", [8, ], [1.0, 1.0])
        # make some proposals overlap GT strongly
        k = min(len(bv), 40)
        for i in range(k):
            g = i % ngt
            bv[i, 1:] = gtbv[g, :4] + np.floor(r.uniform(-3, 3, 4))
        np.random.seed(seed)
        out = proposal_target_layer_3d(bv, b3, gtbv, gt3d, gtc, calib, 2)
        save(f"proposal_target_{name}", rois_bv_in=bv, rois_3d_in=b3, gt_bv=gtbv, gt_3d=gt3d, gt_cnr=gtc, calib=calib,
             np_seed=seed, rois_bv=out[0], rois_img=out[1], labels=out[2], bbox_targets=out[3], rois_3d=out[4])

    # ---- §8(f) rank 2: test-time tail of box_detect (lib/fast_rcnn/test_mv.py:240-261)
    r = np.random.RandomState(51)
    b3 = np.stack([r.uniform(0, 60, 300), r.uniform(-30, 30, 300), r.uniform(-1.5, 0, 300), r.uniform(1, 5, 300),
                   r.uniform(0.5, 2, 300), r.uniform(1, 2, 300)], 1).astype(np.float32)
    b3[0] = [59.95, -29.95, -1, 4, 1.6, 1.5]; b3[1] = [0.05, 29.95, -1, 3.9, 1.6, 1.5]   # map corners
    deltas = (r.uniform(-0.2, 0.2, (300, 48))).astype(np.float32)
    cnr = T.lidar_3d_to_corners(b3)
    pred_cnr = np.hstack((cnr, cnr))
    save("box_tail", boxes_3d=b3, deltas=deltas, corners=cnr, pred_cnr_r=BT.bbox_transform_inv_cnr(cnr, deltas),
         pred_bv=T.corners_to_bv(pred_cnr), pred_bv_r=T.corners_to_bv(BT.bbox_transform_inv_cnr(cnr, deltas)))

    # ---- §8(f) rank 1: BEV rasteriser (lib/utils/read_lidar.py:10-115 == tools/read_lidar.py)
    from utils.read_lidar import point_cloud_2_top
    for name, seed, P in (("small", 61, 20000), ("kitti", 62, 120000)):
        pts = synth.point_cloud(seed, P)
        top = point_cloud_2_top(pts, res=0.1, zres=0.3, side_range=(-30., 30.), fwd_range=(0., 60), height_range=(-2, 0.4))
        nz = np.flatnonzero(top)
        save(f"point_cloud_top_{name}", seed=seed, P=P, sha_points=synth.sha256(pts), shape=np.array(top.shape),
             nz_index=nz.astype(np.int32), nz_value=top.ravel()[nz], sha_top=synth.sha256(top))

    gen_bev_ranges()
    gen_kitti(d)

    if args.keep_scratch:
        print("scratch kept at", d)
    else:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
