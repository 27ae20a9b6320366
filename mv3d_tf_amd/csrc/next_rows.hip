// SURVEY §8(f) "next" rows, the steps either side of the hot path:
//   rank 1  point_cloud_2_top  (lib/utils/read_lidar.py:10-115 == tools/read_lidar.py:10-115, ranges
//           tools/read_lidar.py:121-123): Velodyne points -> 601x601x9 BEV (8 height slices + reflectance)
//   rank 2  test-time tail of box_detect (lib/fast_rcnn/test_mv.py:240-261): lidar_3d_to_corners,
//           bbox_transform_inv_cnr (lib/fast_rcnn/bbox_transform.py:157-176), corners_to_bv
//           (lib/utils/transform.py:342-366)
#include <math.h>
#include "geometry.h"

// ---------------------------------------------------------------------------- BEV rasteriser
// The reference assigns top[y, x, i] = z - height_range[0] and top[y, x, z_max] = reflectance slice by slice with numpy fancy
// indexing: among the points that hit the same (cell, slice) the LAST one in point order wins, and for the reflectance channel the
// last slice then the last point.  Deterministic here via atomicMax on keys written into the output buffer itself (as u32),
// converted in place afterwards:
//   channels < z_max : key = point index + 1
//   channel z_max    : key = slice << 27 | (point index + 1)          (P < 2^27, at most 32 slices)
// (a height written to channel z_max -- possible when (h1 - h0) / zres is not an integer -- is always overwritten by the same
// iteration's reflectance, read_lidar.py:109-112, so that channel only ever shows the reflectance).
// HBM-bound: one clear of the map, P scattered atomics, one read-modify-write of the map.
// The dtypes numpy gives every step (f32 range tests against Python floats, f64 slice tests against np.arange's values, which are
// h0, h0 + zres, h0 + i * ((h0 + zres) - h0), ...) are spelled out in oracle/mv3d_oracle.c:mv3d_ref_point_cloud_2_top_ranges.
struct BevParams {
    float fwd0f, fwd1f, ylo, yhi, resf, h0f;
    double h0, next, delta, zres;
    int xoff, yoff, Hd, Wd, Cd, nslice, zmax;
};

// zero keys: four 16-byte stores per thread, a workgroup clears 16 KB contiguous (hipMemsetAsync splits the 13 MB
// clear into two 6 us fills)
__global__ __launch_bounds__(256) void bev_clear_kernel(float *top, long long n, int *status)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;       // float4 index
    if (status && blockIdx.x == 0 && threadIdx.x == 0) *status = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long q = base + u * 256, e = q * 4;
        if (e + 4 <= n) reinterpret_cast<float4 *>(top)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        else for (long long t = e; t < n; ++t) top[t] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void bev_scatter_kernel(const float *__restrict__ pts, int P, unsigned *keys, const BevParams q, int *status)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float4 v = reinterpret_cast<const float4 *>(pts)[p];
    if (!(v.x > q.fwd0f && v.x < q.fwd1f)) return;               // f_filt (read_lidar.py:58-59): f32 against the f32-rounded bounds
    if (!(v.y > q.ylo && v.y < q.yhi)) return;                   // s_filt (:60-61)
    int xi = (int)(-v.y / q.resf), yi = (int)(-v.x / q.resf);   // f32 divide, astype(int32) truncates (:96-97)
    xi -= q.xoff;                                                // int(np.floor(side_range[0] / res)) (:102)
    yi += q.yoff;                                                // int(np.floor(fwd_range[1] / res))  (:103)
    if (xi < 0) xi += q.Wd;                                      // numpy: a negative index counts from the end ...
    if (yi < 0) yi += q.Hd;
    const bool out = xi < 0 || xi >= q.Wd || yi < 0 || yi >= q.Hd;      // ... anything else outside the map is its IndexError
    const double z = (double)v.z;
    unsigned *cell = keys + ((long long)yi * q.Wd + xi) * q.Cd;
    for (int i = 0; i < q.nslice; ++i) {                         // np.arange(h0, h1, zres) (:80), compared in f64 (:82-83)
        const double height = i == 0 ? q.h0 : (i == 1 ? q.next : q.h0 + (double)i * q.delta);
        if (z >= height && z < height + q.zres) {
            if (out) { if (status) atomicOr(status, 1); return; }
            if (i < q.zmax) atomicMax(&cell[i], (unsigned)p + 1u);
            atomicMax(&cell[q.zmax], ((unsigned)i << 27) | ((unsigned)p + 1u));
        }
    }
}

__global__ __launch_bounds__(256) void bev_resolve_kernel(const float *__restrict__ pts, float *top, long long n, const BevParams q)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const unsigned key = reinterpret_cast<const unsigned *>(top)[e];
    float out = 0.0f;
    if (key) {
        const int c = (int)(e % q.Cd);
        const unsigned p = (c == q.zmax ? (key & 0x7FFFFFFu) : key) - 1u;
        out = (c == q.zmax) ? pts[4 * (long long)p + 3] : (pts[4 * (long long)p + 2] - q.h0f);   // z - height_range[0] in f32 (:106)
    }
    top[e] = out;
}

static bool bev_params(double res, double zres, double side0, double side1, double fwd0, double fwd1, double h0, double h1, BevParams &q)
{
    if (!(res > 0.0) || !(zres > 0.0) || !(side1 > side0) || !(fwd1 > fwd0) || !(h1 >= h0)) return false;
    const double nx = (side1 - side0) / res, ny = (fwd1 - fwd0) / res, nz = (h1 - h0) / zres;
    if (!(nx < 32768.0) || !(ny < 32768.0) || !(nz < 31.0)) return false;       // (the key of the last channel holds 5 bits of slice)
    const int x_max = (int)nx, y_max = (int)ny, z_max = (int)nz;                 // read_lidar.py:49-51
    q.Wd = x_max + 1; q.Hd = y_max + 1; q.Cd = z_max + 1; q.zmax = z_max;
    q.xoff = (int)floor(side0 / res); q.yoff = (int)floor(fwd1 / res);
    q.nslice = (int)ceil(nz);
    q.h0 = h0; q.next = h0 + zres; q.delta = q.next - h0; q.zres = zres;
    q.fwd0f = (float)fwd0; q.fwd1f = (float)fwd1; q.ylo = (float)(-side1); q.yhi = (float)(-side0); q.resf = (float)res; q.h0f = (float)h0;
    return true;
}

static int bev_launch(const float *points_dev, int num_points, const BevParams &q, float *top_dev, int *status_dev, hipStream_t s)
{
    if (num_points < 0 || num_points >= (1 << 27) || !top_dev || (num_points > 0 && !points_dev)) return MV3D_ERR_INVALID_ARG;
    if (((uintptr_t)points_dev & 15) != 0 || ((uintptr_t)top_dev & 15) != 0) return MV3D_ERR_INVALID_ARG;
    const long long n = (long long)q.Hd * q.Wd * q.Cd;
    hipLaunchKernelGGL(bev_clear_kernel, dim3((unsigned)((n / 4 + 1024) / 1024)), dim3(256), 0, s, top_dev, n, status_dev);
    if (num_points > 0) {
        hipLaunchKernelGGL(bev_scatter_kernel, dim3((num_points + 255) / 256), dim3(256), 0, s, points_dev, num_points,
                           reinterpret_cast<unsigned *>(top_dev), q, status_dev);
        hipLaunchKernelGGL(bev_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, points_dev, top_dev, n, q);
    }
    return mv3d_launch_status();
}

extern "C" int mv3d_point_cloud_2_top(const float *points_dev, int num_points, float *top_dev, void *stream)
{
    BevParams q;
    bev_params(0.1, 0.3, -30.0, 30.0, 0.0, 60.0, -2.0, 0.4, q);      // tools/read_lidar.py:121-133 -> (601, 601, 9); no cell can fall outside
    return bev_launch(points_dev, num_points, q, top_dev, nullptr, (hipStream_t)stream);
}

extern "C" int mv3d_point_cloud_2_top_shape(double res, double zres, double side_lo, double side_hi, double fwd_lo, double fwd_hi,
                                            double height_lo, double height_hi, int *dims)
{
    BevParams q;
    if (!dims || !bev_params(res, zres, side_lo, side_hi, fwd_lo, fwd_hi, height_lo, height_hi, q)) return MV3D_ERR_INVALID_ARG;
    dims[0] = q.Hd; dims[1] = q.Wd; dims[2] = q.Cd;
    return MV3D_OK;
}

extern "C" int mv3d_point_cloud_2_top_ranges(const float *points_dev, int num_points, double res, double zres, double side_lo, double side_hi,
                                             double fwd_lo, double fwd_hi, double height_lo, double height_hi, float *top_dev,
                                             int *status_dev, void *stream)
{
    BevParams q;
    if (!status_dev || !bev_params(res, zres, side_lo, side_hi, fwd_lo, fwd_hi, height_lo, height_hi, q)) return MV3D_ERR_INVALID_ARG;
    return bev_launch(points_dev, num_points, q, top_dev, status_dev, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------- box_detect tail
// numpy npy_divmodf -> floor_divide in f32: corners_to_bv runs _lidar_to_bv_coord on f32 arrays, so
// RES = 0.1 becomes the f32 0.1 and the whole floor-division is single precision.
__device__ __forceinline__ float np_floor_dividef(float a, float b)
{
    if (b == 0.0f) return a / b;
    float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.0f) {
        if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0f; }
    }
    float fd;
    if (div != 0.0f) {
        fd = floorf(div);
        if (div - fd > 0.5f) fd += 1.0f;
    } else {
        fd = copysignf(0.0f, a / b);
    }
    return fd;
}

__device__ __forceinline__ void corners_to_bv_one(const float *c, float *out)
{
    float xmin = c[0], xmax = c[0], ymin = c[8], ymax = c[8];
    bool nx = c[0] != c[0], ny = c[8] != c[8];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        if (c[k] < xmin) xmin = c[k];
        if (c[k] > xmax) xmax = c[k];
        if (c[8 + k] < ymin) ymin = c[8 + k];
        if (c[8 + k] > ymax) ymax = c[8 + k];
        nx |= c[k] != c[k]; ny |= c[8 + k] != c[8 + k];
    }
    if (nx) xmin = xmax = NAN;
    if (ny) ymin = ymax = NAN;
    const float res = (float)BV_RES;
    out[0] = (float)BV_YN - np_floor_dividef(ymax - (float)TOP_Y_MIN_D, res);     // transform.py:17-18, 352-355
    out[1] = (float)BV_XN - np_floor_dividef(xmax - (float)TOP_X_MIN_D, res);
    out[2] = (float)BV_YN - np_floor_dividef(ymin - (float)TOP_Y_MIN_D, res);
    out[3] = (float)BV_XN - np_floor_dividef(xmin - (float)TOP_X_MIN_D, res);
}

__global__ __launch_bounds__(128) void box_tail_kernel(const float *__restrict__ rois_3d, const float *__restrict__ deltas,
                                                       int R, int nc, float *corners, float *pred_cnr_r, float *bv, float *bv_r)
{
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= R) return;
    const float *P = rois_3d + 7 * (long long)i + 1;            // rois[2][:, 1:7] (test_mv.py:240)
    const float hl = P[3] / 2.0f, hw = P[4] / 2.0f, hh = P[5] / 2.0f;
    float c[24];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                                // transform.py:296-313
        c[k] = ((k & 2) ? -hl : hl) + P[0];
        c[8 + k] = (((k + 1) & 2) ? -hw : hw) + P[1];
        c[16 + k] = ((k & 4) ? hh : -hh) + P[2];
    }
    for (int j = 0; j < 24; ++j) corners[24 * (long long)i + j] = c[j];
    // bbox_transform.py:162-176: diag = ||p0 - p6|| in f32 (sqrt((d0^2 + d1^2) + d2^2)), deltas * diag + boxes
    const float d0 = c[0] - c[6], d1 = c[8] - c[14], d2 = c[16] - c[22];
    float ss = __fmul_rn(d0, d0);
    ss = __fadd_rn(ss, __fmul_rn(d1, d1));
    ss = __fadd_rn(ss, __fmul_rn(d2, d2));
    const float diag = sqrtf(ss);
    for (int k = 0; k < nc; ++k) {
        float pr[24];
        const float *dl = deltas + ((long long)i * nc + k) * 24;
#pragma unroll
        for (int j = 0; j < 24; ++j) pr[j] = __fadd_rn(__fmul_rn(dl[j], diag), c[j]);
        for (int j = 0; j < 24; ++j) pred_cnr_r[((long long)i * nc + k) * 24 + j] = pr[j];
        corners_to_bv_one(c, bv + ((long long)i * nc + k) * 4);                   // hstack((cnr, cnr)) (:250)
        corners_to_bv_one(pr, bv_r + ((long long)i * nc + k) * 4);
    }
}

extern "C" int mv3d_box_detect_tail(const float *rois_3d_dev, const float *bbox_pred_dev, int num_rois, int num_classes,
                                    float *corners_dev, float *pred_cnr_r_dev, float *pred_bv_dev, float *pred_bv_r_dev,
                                    void *stream)
{
    if (num_rois < 0 || num_classes <= 0) return MV3D_ERR_INVALID_ARG;
    if (num_rois == 0) return MV3D_OK;
    if (!rois_3d_dev || !bbox_pred_dev || !corners_dev || !pred_cnr_r_dev || !pred_bv_dev || !pred_bv_r_dev)
        return MV3D_ERR_INVALID_ARG;
    hipLaunchKernelGGL(box_tail_kernel, dim3((num_rois + 127) / 128), dim3(128), 0, (hipStream_t)stream, rois_3d_dev,
                       bbox_pred_dev, num_rois, num_classes, corners_dev, pred_cnr_r_dev, pred_bv_dev, pred_bv_r_dev);
    return mv3d_launch_status();
}
