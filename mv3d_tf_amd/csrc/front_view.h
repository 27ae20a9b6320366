// Front-view projection of one 3D point (device functions shared by front_view.hip and the fused emit of
// proposal_target.hip); see front_view.hip for the definition and its parity status.
#pragma once
#include "common.h"

#define FV_W 512
#define FV_H 64
#define FV_THETA_MAX 0x1.921fb54442d18p-1      /* +45 deg */
#define FV_DTHETA 0x1.921fb54442d18p-9         /* (pi/2) / 512 */
#define FV_PHI_TOP 0x1.1df46a2529d39p-5        /* +2 deg */
#define FV_DPHI 0x1.e0c2ec0e7b1eep-8           /* 26.9 deg / 64 */
#define FV_PI 0x1.921fb54442d18p+1
#define FV_PI_2 0x1.921fb54442d18p+0

static __constant__ double c_atan16[17] = {   // atan(j / 16), j = 0..16, round-to-nearest f64
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


// column / row of one point on the front-view map (before clipping; NaN if the point has a NaN coordinate)
__device__ __forceinline__ void fv_point(double x, double y, double z, double &col, double &row)
{
    const double theta = fv_atan2(y, x);
    const double rho = __dsqrt_rn(fma(x, x, y * y));
    const double phi = fv_atan2(z, rho);
    col = floor((FV_THETA_MAX - theta) / FV_DTHETA);
    row = floor((FV_PHI_TOP - phi) / FV_DPHI);
}
