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
#include "common.h"

#define FV_W 512
#define FV_H 64
#define FV_THETA_MAX 0x1.921fb54442d18p-1      /* +45 deg */
#define FV_DTHETA 0x1.921fb54442d18p-9         /* (pi/2) / 512 */
#define FV_PHI_TOP 0x1.1df46a2529d39p-5        /* +2 deg */
#define FV_DPHI 0x1.e0c2ec0e7b1eep-8           /* 26.9 deg / 64 */
#define FV_PI 0x1.921fb54442d18p+1
#define FV_PI_2 0x1.921fb54442d18p+0

__constant__ double c_atan16[17] = {   // atan(j / 16), j = 0..16, round-to-nearest f64
    0x0.0p+0, 0x1.ff55bb72cfdeap-5, 0x1.fd5ba9aac2f6ep-4, 0x1.7b97b4bce5b02p-3, 0x1.f5b75f92c80ddp-3,
    0x1.362773707ebccp-2, 0x1.6f61941e4def1p-2, 0x1.a64eec3cc23fdp-2, 0x1.dac670561bb4fp-2, 0x1.0657e94db30d0p-1,
    0x1.1e00babdefeb4p-1, 0x1.345f01cce37bbp-1, 0x1.4978fa3269ee1p-1, 0x1.5d58987169b18p-1, 0x1.700a7c5784634p-1,
    0x1.819d0b7158a4dp-1, 0x1.921fb54442d18p-1};

// atan(a) for a >= 0 (NaN propagates): a > 1 -> pi/2 - atan(1/a); then a = k/16 + rest with k = nearest sixteenth:
// atan(a) = atan(k/16) + atan(t), t = (a - k/16) / (1 + a k/16), |t| <= 1/32: odd series to t^11 (next term < 3e-21).
__device__ __forceinline__ double fv_atan_pos(double a)
{
    const bool inv = a > 1.0;
    if (inv) a = 1.0 / a;                       // inf -> 0
    const double kf = floor(fma(a, 16.0, 0.5));
    const int k = (a == a) ? (int)kf : 0;
    const double r = kf * 0.0625;
    const double t = (a - r) / fma(a, r, 1.0);
    const double s = t * t;
    double p = -1.0 / 11.0;
    p = fma(p, s, 1.0 / 9.0);
    p = fma(p, s, -1.0 / 7.0);
    p = fma(p, s, 1.0 / 5.0);
    p = fma(p, s, -1.0 / 3.0);
    const double at = fma(t * s, p, t);
    const double v = c_atan16[k] + at;
    return inv ? FV_PI_2 - v : v;
}

__device__ __forceinline__ double fv_atan2(double y, double x)
{
    if (x != x || y != y) return NAN;
    if (x == 0.0) return y > 0.0 ? FV_PI_2 : (y < 0.0 ? -FV_PI_2 : 0.0);
    const double q = fv_atan_pos(fabs(y / x));            // [0, pi/2]
    const double w = x > 0.0 ? q : FV_PI - q;
    return y < 0.0 ? -w : w;
}

__device__ __forceinline__ float fv_clip(double v, double hi)
{
    v = (v >= 0.0) ? v : 0.0;                              // also NaN -> 0
    v = (v <= hi) ? v : hi;
    return (float)v;
}

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
        const double theta = fv_atan2(y, x);
        const double rho = __dsqrt_rn(fma(x, x, y * y));
        const double phi = fv_atan2(z, rho);
        const double col = floor((FV_THETA_MAX - theta) / FV_DTHETA);
        const double row = floor((FV_PHI_TOP - phi) / FV_DPHI);
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
