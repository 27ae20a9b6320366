// SURVEY §8(f) "next" rows, the steps either side of the hot path:
//   rank 1  point_cloud_2_top  (lib/utils/read_lidar.py:10-115 == tools/read_lidar.py:10-115, ranges
//           tools/read_lidar.py:121-123): Velodyne points -> 601x601x9 BEV (8 height slices + reflectance)
//   rank 2  test-time tail of box_detect (lib/fast_rcnn/test_mv.py:240-261): lidar_3d_to_corners,
//           bbox_transform_inv_cnr (lib/fast_rcnn/bbox_transform.py:157-176), corners_to_bv
//           (lib/utils/transform.py:342-366)
#include <math.h>
#include "geometry.h"

// ---------------------------------------------------------------------------- BEV rasteriser
// The reference assigns top[y, x, i] = z + 2 and top[y, x, 8] = reflectance slice by slice with
// numpy fancy indexing: among the points that hit the same (cell, slice) the LAST one in point order
// wins, and for the reflectance channel the last slice then the last point.  Deterministic here via
// atomicMax on keys written into the output buffer itself (as u32), converted in place afterwards:
//   channels 0..7 : key = point index + 1
//   channel  8    : key = slice << 27 | (point index + 1)          (P < 2^27)
// HBM-bound: one 13 MB clear, P scattered atomics, one 13 MB read-modify-write.
#define BEV_H 601
#define BEV_W 601
#define BEV_C 9

// zero keys: four 16-byte stores per thread, a workgroup clears 16 KB contiguous (hipMemsetAsync splits the 13 MB
// clear into two 6 us fills)
__global__ __launch_bounds__(256) void bev_clear_kernel(float *top, long long n)
{
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;       // float4 index
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long q = base + u * 256, e = q * 4;
        if (e + 4 <= n) reinterpret_cast<float4 *>(top)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        else for (long long t = e; t < n; ++t) top[t] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void bev_scatter_kernel(const float *__restrict__ pts, int P, unsigned *keys)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float4 v = reinterpret_cast<const float4 *>(pts)[p];
    const double x = (double)v.x, y = (double)v.y, z = (double)v.z;
    if (!(x > 0.0 && x < 60.0)) return;                          // f_filt (read_lidar.py:58-59)
    if (!(y > -30.0 && y < 30.0)) return;                        // s_filt (:60-61)
    int xi = (int)(-v.y / 0.1f), yi = (int)(-v.x / 0.1f);       // f32 divide, astype(int32) truncates (:96-97)
    xi -= -300;                                                  // int(np.floor(-30 / 0.1)) (:102)
    yi += 600;                                                   // int(np.floor(60 / 0.1))  (:103)
    unsigned *cell = keys + ((long long)yi * BEV_W + xi) * BEV_C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                // np.arange(-2, 0.4, 0.3): 8 slices, start + i*step
        const double height = -2.0 + i * 0.3;
        if (z >= height && z < height + 0.3) {                   // (:82-83) compared in f64
            atomicMax(&cell[i], (unsigned)p + 1u);
            atomicMax(&cell[8], ((unsigned)i << 27) | ((unsigned)p + 1u));   // z_max = int(2.4 / 0.3) = 8
        }
    }
}

__global__ __launch_bounds__(256) void bev_resolve_kernel(const float *__restrict__ pts, float *top, long long n)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const unsigned key = reinterpret_cast<const unsigned *>(top)[e];
    float out = 0.0f;
    if (key) {
        const int c = (int)(e % BEV_C);
        const unsigned p = (c == 8 ? (key & 0x7FFFFFFu) : key) - 1u;
        out = (c == 8) ? pts[4 * (long long)p + 3] : (pts[4 * (long long)p + 2] + 2.0f);   // z - height_range[0] (:106)
    }
    top[e] = out;
}

extern "C" int mv3d_point_cloud_2_top(const float *points_dev, int num_points, float *top_dev, void *stream)
{
    if (num_points < 0 || num_points >= (1 << 27) || !top_dev || (num_points > 0 && !points_dev)) return MV3D_ERR_INVALID_ARG;
    if (((uintptr_t)points_dev & 15) != 0) return MV3D_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)BEV_H * BEV_W * BEV_C;
    if (((uintptr_t)top_dev & 15) != 0) return MV3D_ERR_INVALID_ARG;
    hipLaunchKernelGGL(bev_clear_kernel, dim3((unsigned)((n / 4 + 1024) / 1024)), dim3(256), 0, s, top_dev, n);
    if (num_points > 0) {
        hipLaunchKernelGGL(bev_scatter_kernel, dim3((num_points + 255) / 256), dim3(256), 0, s, points_dev, num_points,
                           reinterpret_cast<unsigned *>(top_dev));
        hipLaunchKernelGGL(bev_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, points_dev, top_dev, n);
    }
    return mv3d_launch_status();
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
