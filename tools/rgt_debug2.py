import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from mv3d_tf_amd import build, ops
from oracle import oracle
from test_gpu_configs import dev
build.build()


def case(C, no_ws):
    rs = np.random.RandomState(C)
    B, H, W = 2, 70, 130
    m = rs.uniform(-1, 1, (B, H, W, C)).astype(np.float32)
    m += (np.arange(W, dtype=np.float32)[None, None, :, None] + np.arange(H, dtype=np.float32)[None, :, None, None]) * np.float32(4)
    rois = [[0, 0, 0, 448, 100], [0, 100, 50, 40, 80], [1, 100, 50, 140, 20], [1, 40, 80, 30, 10],
            [0, 8, 8, 8 + 113 * 8, 8 + 56 * 8], [1, 16, 0, 16 + 120 * 8, 456], [0, 0, 16, 500, 16 + 56 * 8], [1, 24, 24, 24 + 56 * 8, 24 + 56 * 8],
            [0, -20000, -20000, 20000, 20000], [1, -100, -40000, 300, 40000]]
    for _ in range(60):
        x1, y1 = rs.randint(-40, W * 8), rs.randint(-40, H * 8)
        rois.append([rs.randint(0, B), x1, y1, x1 + rs.choice([56 * 8, 113 * 8, 120 * 8, -30, 200]), y1 + rs.choice([56 * 8, -20, 90])])
    rois = np.asarray(rois, np.float32)
    d, r = dev(torch, m), dev(torch, rois)
    res = ops.roi_pool_forward_views_pair([(d, r, 0.125)], 7, 7)
    dec, = ops.roi_pool_argmax_decode([(d, r, 0.125)], res, 7, 7)
    o_top, o_am = oracle.roi_pool(m, rois, 7, 7, 0.125)
    print("fwd equal", np.array_equal(res[0][0].cpu().numpy(), o_top), np.array_equal(dec.cpu().numpy(), o_am))
    g = rs.uniform(-1, 1, o_top.shape).astype(np.float32)
    want = oracle.roi_pool_grad(m, rois, o_am, g, 7, 7, 0.125)
    bd, = ops.roi_pool_backward_views_pair([(dev(torch, g), r, res[0][1], m.shape, 0.125)], 7, 7, workspace=False if no_ws else None)
    got = bd.cpu().numpy()
    bad = np.argwhere(got != want)
    print("C", C, "no_ws", no_ws, "differ", len(bad), "nan in got", int(np.isnan(got).sum()))
    if len(bad):
        px = np.unique(bad[:, :3], axis=0)
        print("   pixels:", len(px), px[:12].tolist(), "channels:", np.unique(bad[:, 3])[:16].tolist(), len(np.unique(bad[:, 3])))
        for b in bad[:4]:
            print("   at %s got %r want %r diff %r" % (b.tolist(), got[tuple(b)], want[tuple(b)], got[tuple(b)] - want[tuple(b)]))
        b = bad[0]; n, h, w, c = b
        cand = [(rr_, ph, pw, float(g[rr_, ph, pw, c])) for rr_ in range(len(rois)) for ph in range(7) for pw in range(7)
                if int(rois[rr_][0]) == n and o_am[rr_, ph, pw, c] == (h * W + w) * C + c]
        print("   contributions (argmax names the pixel):", cand)
        for rr_, ph, pw, _ in cand:
            print("      roi", rr_, rois[rr_].tolist())


for C in (256, 512):
    for no_ws in (False, True):
        case(C, no_ws)
