// Device-side geometry shared by proposal.hip and proposal_target.hip: the BEV <-> LIDAR <->
// corner <-> image arithmetic of lib/utils/transform.py and numpy's f32 exp / f64 floor_divide,
// restated operation for operation (provenance: oracle/mv3d_oracle.c and DESIGN.md).
#pragma once
#include <math.h>
#include "common.h"

// ---- constants of lib/utils/transform.py:3-11 and lib/rpn_msr/generate_anchors.py:37-51 ----
// Xn = Yn = int((60 - 0) // 0.1) + 1 = 600 (Python float floor-division gives 599.0).
#define BV_XN 600
#define BV_YN 600
#define BV_RES 0.1
#define TOP_Y_MIN_D (-30.0)
#define TOP_X_MIN_D 0.0
static __constant__ int c_base_anchors[16] = {-19, -8, 20, 8, -5, -2, 5, 3, -8, -19, 8, 20, -2, -5, 3, 5};

// numpy f32 exp (simd_exp_FLOAT), see oracle/mv3d_oracle.c:mv3d_ref_expf for the provenance.
__device__ __forceinline__ float np_expf(float x)
{
    if (x != x) return x;
    if (x >= 88.72283935546875f) return INFINITY;
    if (x <= -103.97208404541015625f) return 0.0f;
    const float LOG2E = 1.44269504088896341f, MAGIC = 0x1.8p+23f;
    const float C1 = -6.93145752e-1f, C2 = -1.42860677e-6f;
    const float P0 = 9.999999999980870924916e-01f, P1 = 7.257664613233124478488e-01f,
                P2 = 2.473615434895520810817e-01f, P3 = 5.114512081637298353406e-02f,
                P4 = 6.757896990527504603057e-03f, P5 = 5.082762527590693718096e-04f;
    const float Q0 = 1.0f, Q1 = -2.742335390411667452936e-01f, Q2 = 2.159509375685829852307e-02f;
    const float k = __fsub_rn(__fadd_rn(__fmul_rn(x, LOG2E), MAGIC), MAGIC);
    float r = __fmaf_rn(k, C1, x);
    r = __fmaf_rn(k, C2, r);
    float num = __fmaf_rn(P5, r, P4);
    num = __fmaf_rn(num, r, P3);
    num = __fmaf_rn(num, r, P2);
    num = __fmaf_rn(num, r, P1);
    num = __fmaf_rn(num, r, P0);
    float den = __fmaf_rn(Q2, r, Q1);
    den = __fmaf_rn(den, r, Q0);
    return ldexpf(__fdiv_rn(num, den), (int)k);
}

// numpy npy_divmod -> floor_divide for f64 (what `//` does in transform.py:17-18)
__device__ __forceinline__ double np_floor_divide(double a, double b)
{
    if (b == 0.0) return a / b;
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0; }
    }
    double fd;
    if (div != 0.0) {
        fd = floor(div);
        if (div - fd > 0.5) fd += 1.0;
    } else {
        fd = copysign(0.0, a / b);
    }
    return fd;
}

// ndarray.astype(np.int32) of an f64 on x86-64 (cvttsd2si): trunc; NaN / out of range -> INT32_MIN
__device__ __forceinline__ int32_t f64_to_i32(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

// transform.py:369-386: (P2 . R0) . Tr in f32, k ascending, fused multiply-add from a zero
// accumulator (what the build container's OpenBLAS sgemm does; pinned in tests/golden/proj_matrix.npz)
__device__ __forceinline__ void proj_matrix(const float *__restrict__ calib, float M[12])
{
    const float *P2 = calib, *R0 = calib + 24, *Tr = calib + 36;
    float m1[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __fmaf_rn(P2[i * 4 + k], R0[k * 3 + j], acc);
            m1[i * 3 + j] = acc;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc = __fmaf_rn(m1[i * 3 + k], Tr[k * 4 + j], acc);
            M[i * 4 + j] = acc;
        }
}

// transform.py:89-111 + :81-87 for one integer BEV anchor box; f64, then .astype(f32)
// (bbox_transform.py:112).  z and h are f32 constants in the reference.
__device__ __forceinline__ void anchor_to_lidar(int x1, int y1, int x2, int y2, float o[6])
{
    const double ex_len = (double)(y2 - y1) * BV_RES;
    const double ex_wid = (double)(x2 - x1) * BV_RES;
    const double cx = (double)(x1 + x2) / 2.0;
    const double cy = (double)(y1 + y2) / 2.0;
    const double y = BV_XN * BV_RES - (cx + 0.5) * BV_RES + TOP_Y_MIN_D;
    const double x = BV_YN * BV_RES - (cy + 0.5) * BV_RES + TOP_X_MIN_D;
    o[0] = (float)x; o[1] = (float)y;
    o[2] = (float)(-(1.73 - 1.56 / 2.0));     // -(LIDAR_HEIGHT - CAR_HEIGHT/2.)
    o[3] = (float)ex_len; o[4] = (float)ex_wid;
    o[5] = (float)1.56;                       // CAR_HEIGHT
}

// transform.py:483-500 for one box given its 6 decoded numbers
__device__ __forceinline__ void image_box(const float M[12], const float P[6], int32_t out[4])
{
    const float hl = P[3] / 2.0f, hw = P[4] / 2.0f, hh = P[5] / 2.0f;   // transform.py:296-313
    double xmin = 0, xmax = 0, ymin = 0, ymax = 0;
    bool nanx = false, nany = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // corner signs x:[+,+,-,-,+,+,-,-] y:[+,-,-,+,+,-,-,+] z:[-,-,-,-,+,+,+,+]
        const float sx = (k & 2) ? -hl : hl;
        const float sy = ((k + 1) & 2) ? -hw : hw;
        const float sz = (k & 4) ? hh : -hh;
        const double cx = (double)(sx + P[0]), cy = (double)(sy + P[1]), cz = (double)(sz + P[2]);
        double v[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
            acc = fma((double)M[r * 4 + 0], cx, acc);
            acc = fma((double)M[r * 4 + 1], cy, acc);
            acc = fma((double)M[r * 4 + 2], cz, acc);
            acc = fma((double)M[r * 4 + 3], 0.0, acc);   // homogeneous w = 0 (sic)
            v[r] = acc;
        }
        const double px = v[0] / v[2], py = v[1] / v[2];
        nanx |= (px != px);
        nany |= (py != py);
        if (k == 0) { xmin = xmax = px; ymin = ymax = py; }
        else {
            if (px < xmin) xmin = px;
            if (px > xmax) xmax = px;
            if (py < ymin) ymin = py;
            if (py > ymax) ymax = py;
        }
    }
    if (nanx) xmin = xmax = NAN;
    if (nany) ymin = ymax = NAN;
    out[0] = f64_to_i32(xmin); out[1] = f64_to_i32(ymin);
    out[2] = f64_to_i32(xmax); out[3] = f64_to_i32(ymax);
}


// ---- eight lanes per box (lane k of an aligned group of 8 = corner k): the per-corner work of image_box() and the
// sequential first-extreme-wins scan over the 8 results, on every lane of the group
__device__ __forceinline__ void box_corner(const float P[6], int k, float &x, float &y, float &z)
{
    const float hl = P[3] / 2.0f, hw = P[4] / 2.0f, hh = P[5] / 2.0f;   // transform.py:296-313
    x = ((k & 2) ? -hl : hl) + P[0];
    y = (((k + 1) & 2) ? -hw : hw) + P[1];
    z = ((k & 4) ? hh : -hh) + P[2];
}

__device__ __forceinline__ void image_point(const float M[12], float x, float y, float z, double &px, double &py)
{
    const double cx = (double)x, cy = (double)y, cz = (double)z;
    double v[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double acc = 0.0;
        acc = fma((double)M[r * 4 + 0], cx, acc);
        acc = fma((double)M[r * 4 + 1], cy, acc);
        acc = fma((double)M[r * 4 + 2], cz, acc);
        acc = fma((double)M[r * 4 + 3], 0.0, acc);   // homogeneous w = 0 (sic)
        v[r] = acc;
    }
    px = v[0] / v[2]; py = v[1] / v[2];
}

__device__ __forceinline__ void group8_minmax(double v, int lane_base, double &mn, double &mx, bool &any_nan)
{
    any_nan = false;
    mn = mx = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double x = __shfl(v, lane_base + k);
        any_nan |= (x != x);
        if (k == 0) { mn = mx = x; }
        else {
            if (x < mn) mn = x;
            if (x > mx) mx = x;
        }
    }
}
