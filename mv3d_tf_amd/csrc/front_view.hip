// Third (front-view) ROI: 3D proposal -> box on the 64 x 512 cylindrical front-view map of the MV3D paper
// (Chen et al., CVPR 2017, §3.1: c = floor(atan2(y, x) / d_theta), r = floor(atan2(z, sqrt(x^2 + y^2)) / d_phi)).
//
// PARITY UNPINNED BY CONSTRUCTION: the reference has no such code -- its `proposal_transform` handles 'bv' and
// 'img' only and leaves the front view as a TODO that returns None (lib/networks/network.py:293-315).  The
// projection below is this repository's definition; oracle/mv3d_oracle.c restates it and the two agree bit for bit
// because every step is an IEEE basic operation: f64 +, -, *, /, sqrt, fma and floor, with a table-driven atan
// written out here instead of a libm call.
//
//   window   azimuth  theta in [-45 deg, +45 deg]  -> 512 columns, column 0 = +45 deg (left, +y), d_theta = 90/512 deg
//            elevation phi  in [-24.9 deg, +2 deg] -> 64 rows, row 0 = +2 deg (HDL-64E field of view), d_phi = 26.9/64 deg
//   box      the 8 corners of lidar_3d_to_corners (lib/utils/transform.py:290-315, f32), each projected in f64;
//            [x1, y1, x2, y2] = [min col, min row, max col, max row], clipped to the map like clip_boxes does for
//            the BEV boxes (lib/fast_rcnn/bbox_transform.py:178-191); NaN -> 0.
#include "front_view.h"

__global__ __launch_bounds__(128) void rois_3d_to_fv_kernel(const float *__restrict__ rois_3d, int R, float *__restrict__ rois_fv)
{
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= R) return;
    const float *P = rois_3d + 7 * (long long)i + 1;
    const float hl = P[3] / 2.0f, hw = P[4] / 2.0f, hh = P[5] / 2.0f;     // transform.py:296-313
    double cmin = 0, cmax = 0, rmin = 0, rmax = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double x = (double)(((k & 2) ? -hl : hl) + P[0]);
        const double y = (double)((((k + 1) & 2) ? -hw : hw) + P[1]);
        const double z = (double)(((k & 4) ? hh : -hh) + P[2]);
        double col, row;
        fv_point(x, y, z, col, row);
        bad |= (col != col) || (row != row);
        if (k == 0) { cmin = cmax = col; rmin = rmax = row; }
        else {
            if (col < cmin) cmin = col;
            if (col > cmax) cmax = col;
            if (row < rmin) rmin = row;
            if (row > rmax) rmax = row;
        }
    }
    if (bad) cmin = cmax = rmin = rmax = NAN;
    float *o = rois_fv + 5 * (long long)i;
    o[0] = rois_3d[7 * (long long)i];
    o[1] = fv_clip(cmin, FV_W - 1.0); o[2] = fv_clip(rmin, FV_H - 1.0);
    o[3] = fv_clip(cmax, FV_W - 1.0); o[4] = fv_clip(rmax, FV_H - 1.0);
}

extern "C" int mv3d_rois_3d_to_fv(const float *rois_3d_dev, int num_rois, float *rois_fv_dev, void *stream)
{
    if (num_rois < 0) return MV3D_ERR_INVALID_ARG;
    if (num_rois == 0) return MV3D_OK;
    if (!rois_3d_dev || !rois_fv_dev) return MV3D_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rois_3d_to_fv_kernel, dim3((num_rois + 127) / 128), dim3(128), 0, (hipStream_t)stream, rois_3d_dev,
                       num_rois, rois_fv_dev);
    return mv3d_launch_status();
}
