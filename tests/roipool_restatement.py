"""Second, independent witness for RoiPool / RoiPoolGrad (SURVEY.md §8(c), row "ROI pool").

The reference's op cannot be built here (TensorFlow headers), and its own test holds no
expected values, so parity of a18 / a19 is pinned by AGREEMENT OF TWO INDEPENDENT
RESTATEMENTS of the reference's two arithmetically identical variants:

  * oracle/mv3d_oracle.c follows the CPU op   lib/roi_pooling_layer/roi_pooling_op.cc:127-181, :373-443
  * this file follows the CUDA kernels        lib/roi_pooling_layer/roi_pooling_op_gpu.cu.cc:27-84, :121-189

This file was written from the .cu.cc alone (per-output-element forward, per-input-element
backward that scans every ROI), in numpy, vectorised over the channel axis only.  It shares
no code with the oracle.  Test infrastructure: imported by tests/ and by
tests/golden/make_roipool_golden.py, never by the product.
"""
import numpy as np

F = np.float32
FLT_MAX = np.finfo(np.float32).max


def _round_half_away(x32):
    """C round() of a float, then the int conversion of `int v = round(..)` (.cu.cc:37-40)."""
    x = float(x32)                      # exact widening
    return int(np.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def _roi_ints(roi, spatial_scale):
    s = F(spatial_scale)
    return (int(roi[0]), _round_half_away(F(roi[1]) * s), _round_half_away(F(roi[2]) * s),
            _round_half_away(F(roi[3]) * s), _round_half_away(F(roi[4]) * s))


def forward(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale):
    """ROIPoolForward (.cu.cc:20-85): every (n, ph, pw) bin, all channels at once.
    Returns (top_data f32, argmax i32), both (R, PH, PW, C)."""
    data = np.ascontiguousarray(bottom_data, F)
    rois = np.ascontiguousarray(bottom_rois, F).reshape(-1, 5)
    _, height, width, channels = data.shape
    R = rois.shape[0]
    top = np.zeros((R, pooled_height, pooled_width, channels), F)
    amax = np.full((R, pooled_height, pooled_width, channels), -1, np.int32)
    chan = np.arange(channels, dtype=np.int64)
    for n in range(R):
        b, sw, sh, ew, eh = _roi_ints(rois[n], spatial_scale)
        roi_width = max(ew - sw + 1, 1)                       # :43-44
        roi_height = max(eh - sh + 1, 1)
        bin_h = F(roi_height) / F(pooled_height)              # :45-48
        bin_w = F(roi_width) / F(pooled_width)
        frame = data[b]                                       # bottom_data += roi_batch_ind * C*H*W  (:72)
        for ph in range(pooled_height):
            hstart = int(np.floor(F(ph) * bin_h))             # :50-57
            hend = int(np.ceil(F(ph + 1) * bin_h))
            hstart = min(max(hstart + sh, 0), height)         # :60-63
            hend = min(max(hend + sh, 0), height)
            for pw in range(pooled_width):
                wstart = int(np.floor(F(pw) * bin_w))
                wend = int(np.ceil(F(pw + 1) * bin_w))
                wstart = min(max(wstart + sw, 0), width)
                wend = min(max(wend + sw, 0), width)
                is_empty = (hend <= hstart) or (wend <= wstart)
                maxval = np.full(channels, 0 if is_empty else -FLT_MAX, F)     # :67
                maxidx = np.full(channels, -1, np.int64)                      # :69
                for h in range(hstart, hend):
                    for w in range(wstart, wend):
                        v = frame[h, w]
                        better = v > maxval                    # strict >: first max wins, NaN never wins (:75)
                        maxval = np.where(better, v, maxval)
                        maxidx = np.where(better, (h * width + w) * channels + chan, maxidx)
                top[n, ph, pw] = maxval
                amax[n, ph, pw] = maxidx.astype(np.int32)
    return top, amax


def backward(top_diff, argmax, bottom_rois, batch_size, height, width, pooled_height, pooled_width, spatial_scale):
    """ROIPoolBackward (.cu.cc:113-190): every input position (n, h, w), all channels at once, ROIs ascending,
    candidate bins (ph, pw) ascending, f32 accumulation in exactly that order.  Returns (B, H, W, C) f32."""
    top_diff = np.ascontiguousarray(top_diff, F)
    argmax = np.ascontiguousarray(argmax, np.int32)
    rois = np.ascontiguousarray(bottom_rois, F).reshape(-1, 5)
    R, _, _, channels = top_diff.shape
    geom = [_roi_ints(rois[r], spatial_scale) for r in range(R)]
    gb = np.array([g[0] for g in geom], np.int64).reshape(-1)
    gsw = np.array([g[1] for g in geom], np.int64).reshape(-1)
    gsh = np.array([g[2] for g in geom], np.int64).reshape(-1)
    gew = np.array([g[3] for g in geom], np.int64).reshape(-1)
    geh = np.array([g[4] for g in geom], np.int64).reshape(-1)
    chan = np.arange(channels, dtype=np.int64)
    out = np.zeros((batch_size, height, width, channels), F)
    for n in range(batch_size):
        for h in range(height):
            for w in range(width):
                # :137-153 -- batch match and containment on the rounded (unclamped) ROI
                hit = np.nonzero((gb == n) & (w >= gsw) & (w <= gew) & (h >= gsh) & (h <= geh))[0]
                if hit.size == 0:
                    continue
                gradient = np.zeros(channels, F)
                want = (h * width + w) * channels + chan
                for r in hit:                                             # ascending roi_n
                    sw, sh, ew, eh = int(gsw[r]), int(gsh[r]), int(gew[r]), int(geh[r])
                    roi_width = max(ew - sw + 1, 1)                       # :163-164
                    roi_height = max(eh - sh + 1, 1)
                    bin_h = F(roi_height) / F(pooled_height)
                    bin_w = F(roi_width) / F(pooled_width)
                    phstart = int(np.floor(F(h - sh) / bin_h))            # :171-174
                    phend = int(np.ceil(F(h - sh + 1) / bin_h))
                    pwstart = int(np.floor(F(w - sw) / bin_w))
                    pwend = int(np.ceil(F(w - sw + 1) / bin_w))
                    phstart = min(max(phstart, 0), pooled_height)         # :176-179
                    phend = min(max(phend, 0), pooled_height)
                    pwstart = min(max(pwstart, 0), pooled_width)
                    pwend = min(max(pwend, 0), pooled_width)
                    for ph in range(phstart, phend):
                        for pw in range(pwstart, pwend):
                            m = argmax[r, ph, pw] == want                 # :183
                            if m.any():
                                gradient[m] = gradient[m] + top_diff[r, ph, pw][m]
                out[n, h, w] = gradient
    return out


def finite_difference_grad(fwd, data, rois, top_weight, pooled_height, pooled_width, spatial_scale, eps, positions):
    """d/d data[pos] of sum(top * top_weight), by one-sided differences in f64 bookkeeping, for the listed
    (b, h, w, c) positions.  Valid on tie-free maps whose gaps between values are > eps (the arg-max does not move)."""
    base, _ = fwd(data, rois, pooled_height, pooled_width, spatial_scale)
    f0 = float(np.sum(base.astype(np.float64) * top_weight.astype(np.float64)))
    out = []
    for pos in positions:
        d = data.copy()
        d[pos] = d[pos] + F(eps)
        t, _ = fwd(d, rois, pooled_height, pooled_width, spatial_scale)
        f1 = float(np.sum(t.astype(np.float64) * top_weight.astype(np.float64)))
        out.append((f1 - f0) / float(F(d[pos]) - F(data[pos])))
    return np.array(out)
