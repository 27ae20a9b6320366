"""CPU facts about the RoiPool bin arithmetic that the one-launch RoiPoolGrad (csrc/roi_grad_tiles.hip) relies on.

The forward pools bin ph over the rows [floor(ph * b), ceil((ph + 1) * b)) of the ROI (b = f32(extent) / f32(P), every product one f32
rounding: roi_pooling_op.cc:148-162); the reference's backward lists, for a pixel at offset x inside the rounded ROI, the bins
[floor(x / b), ceil((x + 1) / b)) (f32 divides, :423-431).  The tile kernel routes a gradient by its argmax code alone, so it needs:
inside the ROI (0 <= x < extent) every bin whose forward rows hold x is in the backward's list.  Outside (x >= extent, reachable
only because f32 7 * (57 / 7) > 57) the backward's containment test (:401-404) drops the gradient; the kernel cuts the rectangle there."""
import numpy as np

f32 = np.float32
EXACT_MAX = 2048          # RGT_EXACT_MAX of csrc/roi_grad_tiles.hip: larger ROIs evaluate the reference's expressions per pixel


def forward_and_backward(extent, P):
    b = f32(extent) / f32(P)
    p = np.arange(P, dtype=np.float32)
    lo = np.floor(p * b).astype(np.int64)
    hi = np.ceil((p + f32(1)) * b).astype(np.int64)
    x = np.arange(extent, dtype=np.float32)
    s = np.clip(np.floor(x / b).astype(np.int64), 0, P)
    e = np.clip(np.ceil((x + f32(1)) / b).astype(np.int64), 0, P)
    xi, pi = np.arange(extent)[:, None], np.arange(P)[None, :]
    fwd = (lo[None, :] <= xi) & (xi < hi[None, :])
    bwd = (s[:, None] <= pi) & (pi < e[:, None])
    return fwd, bwd, int(hi.max())


def test_forward_rectangles_inside_backward_ranges():
    overhang = {}
    for P in range(1, 16):
        for extent in range(1, EXACT_MAX + 1):
            fwd, bwd, top = forward_and_backward(extent, P)
            assert not (fwd & ~bwd).any(), (P, extent)
            if top > extent:
                overhang.setdefault(P, []).append(extent)
    # the overhang exists (the case the kernel cuts): 7 bins over 57 pixels reach pixel 57
    assert overhang[7][:3] == [57, 114, 121]
    assert set(overhang) == {7, 11, 13, 14, 15}
