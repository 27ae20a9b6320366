/*
 * mv3d_oracle.c -- CPU restatement of the MV3D hot path.  TEST INFRASTRUCTURE ONLY
 * (see mv3d_oracle.h).  Plain C99, scalar, single-threaded; compiled with
 * -ffp-contract=off so that every `*`, `+`, `/` below is one IEEE rounding, as in
 * numpy / gcc-x86-64-baseline builds of the reference.  fma() is only used where
 * it is written explicitly.
 */
#include "mv3d_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ constants
 * lib/utils/transform.py:3-11 */
#define TOP_X_MAX 60
#define TOP_X_MIN 0
#define TOP_Y_MIN (-30)
#define TOP_Y_MAX 30
static const double RES = 0.1;
static const double LIDAR_HEIGHT = 1.73;
static const double CAR_HEIGHT = 1.56;
#define NUM_ANCHORS 4

/* int((TOP_X_MAX - TOP_X_MIN) // RES) + 1 : Python int // float -> float floor-div */
static int grid_n(int lo, int hi) { return (int)mv3d_ref_floor_divide((double)(hi - lo), RES) + 1; }

/* ------------------------------------------------------------------ a1 */
void mv3d_ref_generate_anchors_bv(int64_t out[16])
{
    /* generate_anchors.py:37-51: base_size=[[3.9,1.6],[1.0,0.6]], res=0.1 */
    const double base_size[2][2] = {{3.9, 1.6}, {1.0, 0.6}};
    int64_t b[2][4];
    for (int i = 0; i < 2; ++i) {
        b[i][0] = 0; b[i][1] = 0;
        b[i][2] = (int64_t)(base_size[i][0] / 0.1);   /* int() truncates */
        b[i][3] = (int64_t)(base_size[i][1] / 0.1);
        /* in-place, column by column, in this order (:43-46) */
        b[i][0] -= b[i][2] / 2;
        b[i][1] -= b[i][3] / 2;
        b[i][2] -= b[i][2] / 2;
        b[i][3] -= b[i][3] / 2;
    }
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = b[i][j];
        /* base_anchors[:, [1,0,3,2]] (:48) */
        out[(2 + i) * 4 + 0] = b[i][1]; out[(2 + i) * 4 + 1] = b[i][0];
        out[(2 + i) * 4 + 2] = b[i][3]; out[(2 + i) * 4 + 3] = b[i][2];
    }
}

/* ------------------------------------------------------------------ a4 */
void mv3d_ref_bbox_overlaps(const double *boxes, int n, const double *query, int k,
                            double *ov)
{
    /* bbox.pyx:33-54 */
    for (long i = 0; i < (long)n * k; ++i) ov[i] = 0.0;
    for (int q = 0; q < k; ++q) {
        const double *Q = query + 4 * q;
        double qarea = (Q[2] - Q[0] + 1) * (Q[3] - Q[1] + 1);
        for (int b = 0; b < n; ++b) {
            const double *B = boxes + 4 * b;
            double iw = fmin(B[2], Q[2]) - fmax(B[0], Q[0]) + 1;
            if (iw > 0) {
                double ih = fmin(B[3], Q[3]) - fmax(B[1], Q[1]) + 1;
                if (ih > 0) {
                    double ua = (B[2] - B[0] + 1) * (B[3] - B[1] + 1) + qarea - iw * ih;
                    ov[(long)b * k + q] = iw * ih / ua;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ sorting
 * Processing order used everywhere a score sort appears: descending score,
 * ties by descending index (= stable ascending sort, reversed).  NaN sorts
 * first (numpy puts NaN last in the ascending sort). */
typedef struct { float s; int32_t i; } skey;
static int skey_cmp(const void *pa, const void *pb)
{
    const skey *a = (const skey *)pa, *b = (const skey *)pb;
    int an = a->s != a->s, bn = b->s != b->s;
    if (an != bn) return an ? -1 : 1;
    if (!an) {
        if (a->s > b->s) return -1;
        if (a->s < b->s) return 1;
    }
    return (a->i > b->i) ? -1 : (a->i < b->i);
}

/* ------------------------------------------------------------------ a15 */
/* cpu_nms.pyx:11-15 : Cython inline float32 max / min */
static inline float fmax32(float a, float b) { return a >= b ? a : b; }
static inline float fmin32(float a, float b) { return a <= b ? a : b; }

int mv3d_ref_cpu_nms(const float *dets, int n, double thresh, int presorted, int32_t *keep)
{
    if (n <= 0) return 0;
    float *areas = (float *)malloc(sizeof(float) * n);
    skey *ord = (skey *)malloc(sizeof(skey) * n);
    unsigned char *supp = (unsigned char *)calloc(n, 1);
    for (int i = 0; i < n; ++i) {
        const float *d = dets + 5 * i;
        /* :24 numpy f32: (x2 - x1 + 1) * (y2 - y1 + 1) */
        areas[i] = ((d[2] - d[0]) + 1.0f) * ((d[3] - d[1]) + 1.0f);
        ord[i].s = d[4]; ord[i].i = i;
    }
    if (!presorted) qsort(ord, n, sizeof(skey), skey_cmp);   /* :25 */
    int nk = 0, err = 0;
    for (int _i = 0; _i < n && !err; ++_i) {
        int i = ord[_i].i;
        if (supp[i]) continue;
        keep[nk++] = i;
        const float ix1 = dets[5 * i], iy1 = dets[5 * i + 1], ix2 = dets[5 * i + 2],
                    iy2 = dets[5 * i + 3], iarea = areas[i];
        for (int _j = _i + 1; _j < n; ++_j) {
            int j = ord[_j].i;
            if (supp[j]) continue;
            float xx1 = fmax32(ix1, dets[5 * j]);
            float yy1 = fmax32(iy1, dets[5 * j + 1]);
            float xx2 = fmin32(ix2, dets[5 * j + 2]);
            float yy2 = fmin32(iy2, dets[5 * j + 3]);
            /* Cython emits ((xx2 - xx1) + 1.0) with a double literal, then narrows */
            float w = fmax32(0.0f, (float)((double)(xx2 - xx1) + 1.0));
            float h = fmax32(0.0f, (float)((double)(yy2 - yy1) + 1.0));
            float inter = w * h;
            float den = (iarea + areas[j]) - inter;
            if (den == 0.0f) { err = 1; break; }   /* Cython: ZeroDivisionError */
            float ovr = inter / den;
            if ((double)ovr >= thresh) supp[j] = 1; /* :65 Python-float compare */
        }
    }
    free(areas); free(ord); free(supp);
    return err ? -1 : nk;
}

/* lib/nms/nms_kernel.cu:21-30 (devIoU) + :71 + the host reduce :117-133: boxes pre-sorted, all arithmetic f32, box j is
 * removed by an earlier kept box iff IoU > thresh (f32 compare); 0/0 = NaN does not remove and nothing is raised.
 * PARITY UNPINNED (no CUDA device to run the original): a restatement for sweeping the `_nms` entry against. */
int mv3d_ref_gpu_nms_rule(const float *dets, int n, float thresh, int32_t *keep)
{
    unsigned char *supp = (unsigned char *)calloc(n > 0 ? n : 1, 1);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (supp[i]) continue;
        keep[nk++] = i;
        const float *a = dets + 5 * i;
        const float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
        for (int j = i + 1; j < n; ++j) {
            if (supp[j]) continue;
            const float *b = dets + 5 * j;
            const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
            const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
            const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
            const float interS = width * height;
            const float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
            if (interS / (Sa + Sb - interS) > thresh) supp[j] = 1;
        }
    }
    free(supp);
    return nk;
}

/* ------------------------------------------------------------------ defined arithmetic */
/* f32 exp exactly as numpy computes it on AVX2/AVX512 hosts (third-party dependency of
 * the reference: numpy, version unpinned upstream, 2.2.6 in the build container;
 * numpy/_core/src/umath/loops_exponent_log.dispatch.c.src, simd_exp_FLOAT): Cody-Waite
 * reduction with two fused steps, a degree-5 / degree-2 rational minimax evaluated by
 * Horner with fused multiply-adds, one IEEE divide, exact scaling by 2^k.  Only f32
 * mul / fma / div are involved, so gcc (fmaf) and gfx950 (v_fma_f32, IEEE v_div) give
 * the same bits; pinned bit-for-bit against 22 004 numpy values in tests/golden/exp_log.npz. */
float mv3d_ref_expf(float x)
{
    if (x != x) return x;
    if (x >= 88.72283935546875f) return INFINITY;
    if (x <= -103.97208404541015625f) return 0.0f;
    const float LOG2E = 1.44269504088896341f;
    const float MAGIC = 0x1.8p+23f;                       /* round-to-nearest-even via add/sub */
    const float C1 = -6.93145752e-1f, C2 = -1.42860677e-6f;
    const float P0 = 9.999999999980870924916e-01f, P1 = 7.257664613233124478488e-01f,
                P2 = 2.473615434895520810817e-01f, P3 = 5.114512081637298353406e-02f,
                P4 = 6.757896990527504603057e-03f, P5 = 5.082762527590693718096e-04f;
    const float Q0 = 1.0f, Q1 = -2.742335390411667452936e-01f, Q2 = 2.159509375685829852307e-02f;
    volatile float t = x * LOG2E + MAGIC;                 /* volatile: keep the f32 rounding */
    const float k = t - MAGIC;
    float r = fmaf(k, C1, x);
    r = fmaf(k, C2, r);
    float num = fmaf(P5, r, P4);
    num = fmaf(num, r, P3);
    num = fmaf(num, r, P2);
    num = fmaf(num, r, P1);
    num = fmaf(num, r, P0);
    float den = fmaf(Q2, r, Q1);
    den = fmaf(den, r, Q0);
    return ldexpf(num / den, (int)k);
}

/* Defined natural log for f64 (targets are rounded to f32 afterwards):
 * x = m * 2^k, m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1),
 * log m = 2 s (1 + s^2/3 + s^4/5 + ... + s^38/39); |s| < 0.1716 so the
 * truncation error is < 2e-32 and rounding error a few ulp_f64. */
double mv3d_ref_log(double x)
{
    if (x != x || x < 0) return NAN;
    if (x == 0) return -INFINITY;
    if (isinf(x)) return x;
    int k;
    double m = frexp(x, &k);         /* m in [0.5,1) */
    if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 39;
    for (int d = 37; d >= 1; d -= 2) p = fma(p, z, 1.0 / d);
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double lm = 2.0 * s * p;
    return fma((double)k, LN2_HI, fma((double)k, LN2_LO, lm));
}

/* numpy/core/src/npymath/npy_math_internal.h.src npy_divmod -> floor_divide (f64);
 * this is what `//` on float arrays does in transform.py:17-18. */
double mv3d_ref_floor_divide(double a, double b)
{
    if (b == 0.0) return a / b;
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0; }
    }
    double floordiv;
    if (div != 0.0) {
        floordiv = floor(div);
        if (div - floordiv > 0.5) floordiv += 1.0;
    } else {
        floordiv = copysign(0.0, a / b);
    }
    return floordiv;
}

/* ------------------------------------------------------------------ a5 */
/* transform.py:89-111 + :81-87, one anchor.  All f64 except z,h which are f32
 * constants promoted by np.hstack. */
static void bv_anchor_to_lidar_one(int64_t x1, int64_t y1, int64_t x2, int64_t y2, double o[6])
{
    const int Xn = grid_n(TOP_X_MIN, TOP_X_MAX), Yn = grid_n(TOP_Y_MIN, TOP_Y_MAX);
    double ex_len = (double)(y2 - y1) * RES;
    double ex_wid = (double)(x2 - x1) * RES;
    double cx = (double)(x1 + x2) / 2.0;
    double cy = (double)(y1 + y2) / 2.0;
    /* _bv_to_lidar_coords(xx=cx, yy=cy) -> (x, y) */
    double y = Xn * RES - (cx + 0.5) * RES + TOP_Y_MIN;
    double x = Yn * RES - (cy + 0.5) * RES + TOP_X_MIN;
    float hz = (float)CAR_HEIGHT;                               /* np.ones(f32) * CAR_HEIGHT */
    float cz = (float)(-(LIDAR_HEIGHT - CAR_HEIGHT / 2.0));
    o[0] = x; o[1] = y; o[2] = (double)cz; o[3] = ex_len; o[4] = ex_wid; o[5] = (double)hz;
}

void mv3d_ref_bv_anchor_to_lidar(const int64_t *a, int n, double *out)
{
    for (int i = 0; i < n; ++i)
        bv_anchor_to_lidar_one(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3], out + 6 * i);
}

/* ------------------------------------------------------------------ a10 */
static void corners_one(const float p[6], float c[24])
{
    /* transform.py:296-313; l/2. etc. are f32 (python float is weak) */
    const float hl = p[3] / 2.0f, hw = p[4] / 2.0f, hh = p[5] / 2.0f;
    const float xs[8] = {hl, hl, -hl, -hl, hl, hl, -hl, -hl};
    const float ys[8] = {hw, -hw, -hw, hw, hw, -hw, -hw, hw};
    const float zs[8] = {-hh, -hh, -hh, -hh, hh, hh, hh, hh};
    for (int k = 0; k < 8; ++k) {
        c[k] = xs[k] + p[0];
        c[8 + k] = ys[k] + p[1];
        c[16 + k] = zs[k] + p[2];
    }
}
void mv3d_ref_lidar_3d_to_corners(const float *b, int n, float *corners)
{
    for (int i = 0; i < n; ++i) corners_one(b + 6 * i, corners + 24 * i);
}

/* ------------------------------------------------------------------ a11 */
/* transform.py:369-386: mat2 = (P2(3x4) . R0(4x3)) . Tr(3x4) in f32.  numpy hands
 * these to sgemm; the container's OpenBLAS accumulates k ascending with fused
 * multiply-add from a zero accumulator (verified on 200 random calibrations in
 * tests/golden/make_golden.py), which is what is restated here. */
static void proj_matrix(const float *calib, float M[12])
{
    const float *P2 = calib, *R0 = calib + 24, *Tr = calib + 36;
    float m1[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc = fmaf(P2[i * 4 + k], R0[k * 3 + j], acc);
            m1[i * 3 + j] = acc;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) acc = fmaf(m1[i * 3 + k], Tr[k * 4 + j], acc);
            M[i * 4 + j] = acc;
        }
}

/* C cast semantics of ndarray.astype(np.int32) on x86-64 (cvttsd2si): truncate
 * toward zero; NaN / out-of-range -> INT32_MIN. */
static int32_t f64_to_i32(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

static void img_box_one(const float M[12], const float c[24], int32_t out[4])
{
    /* transform.py:483-500: img_cor = mat2(f32->f64) . [corners; 0] in f64 (dgemm, k
     * ascending, fused multiply-add from zero; the w=0 row contributes +0). */
    double xmin = 0, xmax = 0, ymin = 0, ymax = 0;
    int nanx = 0, nany = 0;
    for (int k = 0; k < 8; ++k) {
        double v[3];
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
            acc = fma((double)M[r * 4 + 0], (double)c[k], acc);
            acc = fma((double)M[r * 4 + 1], (double)c[8 + k], acc);
            acc = fma((double)M[r * 4 + 2], (double)c[16 + k], acc);
            acc = fma((double)M[r * 4 + 3], 0.0, acc);
            v[r] = acc;
        }
        double px = v[0] / v[2], py = v[1] / v[2];   /* img_cor / img_cor[2] */
        if (px != px) nanx = 1;
        if (py != py) nany = 1;
        if (k == 0) { xmin = xmax = px; ymin = ymax = py; }
        else {
            if (px < xmin) xmin = px;
            if (px > xmax) xmax = px;
            if (py < ymin) ymin = py;
            if (py > ymax) ymax = py;
        }
    }
    if (nanx) xmin = xmax = NAN;     /* np.min / np.max propagate NaN */
    if (nany) ymin = ymax = NAN;
    out[0] = f64_to_i32(xmin); out[1] = f64_to_i32(ymin);
    out[2] = f64_to_i32(xmax); out[3] = f64_to_i32(ymax);
}

void mv3d_ref_lidar_cnr_to_img(const float *corners, int n, const float *calib, int32_t *img)
{
    float M[12];
    proj_matrix(calib, M);
    for (int i = 0; i < n; ++i) img_box_one(M, corners + 24 * i, img + 4 * i);
}

/* ------------------------------------------------------------------ a7 */
/* numpy minimum / maximum loops: NaN in the first operand propagates */
static inline float np_min32(float a, float b) { return (a < b || a != a) ? a : b; }
static inline float np_max32(float a, float b) { return (a >= b || a != a) ? a : b; }
int mv3d_ref_proposal_layer_3d(const float *prob, const float *pred, int H, int W,
                               const float *im_info, const float *calib,
                               const mv3d_ref_proposal_cfg *cfg,
                               float *blob_bv, float *blob_img, float *blob_3d, int *n_out,
                               double *o_anchors3d, float *o_props3d, float *o_bv_raw,
                               float *o_corners, int32_t *o_img, uint8_t *o_valid,
                               int32_t *o_order, int *n_order, int32_t *o_nms_keep)
{
    const int A = NUM_ANCHORS, N = H * W * A;
    const int Xn = grid_n(TOP_X_MIN, TOP_X_MAX), Yn = grid_n(TOP_Y_MIN, TOP_Y_MAX);
    int64_t base[16];
    mv3d_ref_generate_anchors_bv(base);
    float M[12];
    proj_matrix(calib, M);

    float *p3 = (float *)malloc(sizeof(float) * 6 * N);
    float *bv = (float *)malloc(sizeof(float) * 4 * N);
    int32_t *img = (int32_t *)malloc(sizeof(int32_t) * 4 * N);
    skey *cand = (skey *)malloc(sizeof(skey) * N);
    int nc = 0;

    /* proposal_layer_tf.py:136-147 thresholds */
    const float xmaxc = im_info[1] - 1.0f, ymaxc = im_info[0] - 1.0f;
    const float min_size = (float)cfg->min_size * im_info[2];
    const int pad = 50, w_max = 1242 + pad, h_max = 375 + pad;   /* :147 hard-coded 375x1242 */

    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w)
            for (int a = 0; a < A; ++a) {
                const int n = (h * W + w) * A + a;
                /* :79-95 anchor = base[a] + stride*(w,h,w,h) */
                const int64_t sx = (int64_t)w * cfg->feat_stride, sy = (int64_t)h * cfg->feat_stride;
                double an[6];
                bv_anchor_to_lidar_one(base[4 * a] + sx, base[4 * a + 1] + sy,
                                       base[4 * a + 2] + sx, base[4 * a + 3] + sy, an);
                if (o_anchors3d) memcpy(o_anchors3d + 6 * n, an, sizeof an);
                /* bbox_transform.py:112 boxes.astype(f32); :130-135 f32 mul, add; np.exp f32 */
                const float ax = (float)an[0], ay = (float)an[1], az = (float)an[2];
                const float al = (float)an[3], aw = (float)an[4], ah = (float)an[5];
                const float *d = pred + (size_t)(h * W + w) * 6 * A + 6 * a;   /* :105 */
                float *P = p3 + 6 * n;
                P[0] = d[0] * al + ax;
                P[1] = d[1] * aw + ay;
                P[2] = d[2] * ah + az;
                P[3] = mv3d_ref_expf(d[3]) * al;
                P[4] = mv3d_ref_expf(d[4]) * aw;
                P[5] = mv3d_ref_expf(d[5]) * ah;
                /* transform.py:131-137: f32 corner sums stored into an f64 array, then
                 * Yn - (y - TOP_Y_MIN)//RES, Xn - (x - TOP_X_MIN)//RES in f64, -> f32 */
                const double r0 = (double)(P[0] + P[3] * 0.5f), r1 = (double)(P[1] + P[4] * 0.5f);
                const double r2 = (double)(P[0] - P[3] * 0.5f), r3 = (double)(P[1] - P[4] * 0.5f);
                float *B = bv + 4 * n;
                B[0] = (float)(Yn - mv3d_ref_floor_divide(r1 - TOP_Y_MIN, RES));
                B[1] = (float)(Xn - mv3d_ref_floor_divide(r0 - TOP_X_MIN, RES));
                B[2] = (float)(Yn - mv3d_ref_floor_divide(r3 - TOP_Y_MIN, RES));
                B[3] = (float)(Xn - mv3d_ref_floor_divide(r2 - TOP_X_MIN, RES));
                if (o_bv_raw) memcpy(o_bv_raw + 4 * n, B, 4 * sizeof(float));
                float c[24];
                corners_one(P, c);
                if (o_corners) memcpy(o_corners + 24 * n, c, sizeof c);
                img_box_one(M, c, img + 4 * n);
                /* bbox_transform.py:184-190 clip: np.maximum(np.minimum(v, lim), 0) */
                B[0] = np_max32(np_min32(B[0], xmaxc), 0.0f);
                B[1] = np_max32(np_min32(B[1], ymaxc), 0.0f);
                B[2] = np_max32(np_min32(B[2], xmaxc), 0.0f);
                B[3] = np_max32(np_min32(B[3], ymaxc), 0.0f);
                /* :336-341 and :343-352 */
                const float ws = (B[2] - B[0]) + 1.0f, hs = (B[3] - B[1]) + 1.0f;
                const int32_t *I = img + 4 * n;
                int ok = (ws >= min_size) && (hs >= min_size);
                ok = ok && (-pad <= I[0]) && (I[2] <= w_max) && (-pad <= I[1]) && (I[3] <= h_max);
                if (o_valid) o_valid[n] = (uint8_t)ok;
                if (ok) {
                    cand[nc].s = prob[(size_t)(h * W + w) * 2 * A + 2 * a + 1];   /* :63 */
                    cand[nc].i = n;
                    ++nc;
                }
            }
    if (o_props3d) memcpy(o_props3d, p3, sizeof(float) * 6 * N);
    if (o_img) memcpy(o_img, img, sizeof(int32_t) * 4 * N);

    /* :161-167 */
    qsort(cand, nc, sizeof(skey), skey_cmp);
    int K = nc;
    if (cfg->pre_nms_topN > 0 && K > cfg->pre_nms_topN) K = cfg->pre_nms_topN;
    if (n_order) *n_order = K;
    if (o_order) for (int i = 0; i < K; ++i) o_order[i] = cand[i].i;

    /* :172-174 nms(np.hstack((proposals_bv, scores)), thresh)[:post_nms_topN];
     * nms_wrapper.py:16-17 empty -> [] */
    float *dets = (float *)malloc(sizeof(float) * 5 * (K > 0 ? K : 1));
    int32_t *keep = (int32_t *)malloc(sizeof(int32_t) * (K > 0 ? K : 1));
    for (int i = 0; i < K; ++i) {
        memcpy(dets + 5 * i, bv + 4 * cand[i].i, 4 * sizeof(float));
        dets[5 * i + 4] = cand[i].s;
    }
    int nk = K > 0 ? mv3d_ref_cpu_nms(dets, K, cfg->nms_thresh, 1, keep) : 0;
    int rc = 0;
    if (nk < 0) { rc = -1; nk = 0; }
    if (cfg->post_nms_topN > 0 && nk > cfg->post_nms_topN) nk = cfg->post_nms_topN;
    *n_out = nk;
    for (int r = 0; r < nk; ++r) {
        const int n = cand[keep[r]].i;
        if (o_nms_keep) o_nms_keep[r] = keep[r];
        blob_bv[5 * r] = 0.0f;
        memcpy(blob_bv + 5 * r + 1, bv + 4 * n, 4 * sizeof(float));
        blob_img[5 * r] = 0.0f;
        for (int j = 0; j < 4; ++j) blob_img[5 * r + 1 + j] = (float)img[4 * n + j];
        blob_3d[7 * r] = 0.0f;
        memcpy(blob_3d + 7 * r + 1, p3 + 6 * n, 6 * sizeof(float));
    }
    free(dets); free(keep); free(p3); free(bv); free(img); free(cand);
    return rc;
}

/* ------------------------------------------------------------------ a3 (deterministic part) */
static double iou_f64(const double B[4], const double Q[4])
{
    /* bbox.pyx:33-54 for one pair */
    double iw = fmin(B[2], Q[2]) - fmax(B[0], Q[0]) + 1;
    if (iw > 0) {
        double ih = fmin(B[3], Q[3]) - fmax(B[1], Q[1]) + 1;
        if (ih > 0) {
            double qarea = (Q[2] - Q[0] + 1) * (Q[3] - Q[1] + 1);
            double ua = (B[2] - B[0] + 1) * (B[3] - B[1] + 1) + qarea - iw * ih;
            return iw * ih / ua;
        }
    }
    return 0.0;
}

int mv3d_ref_anchor_target_stage1(int H, int W, int feat_stride, const float *im_info,
                                  const float *gt_bv, const float *gt_3d, int G,
                                  double neg_ov, double pos_ov, int clobber,
                                  int32_t *inds_inside, int32_t *argmax, double *max_ov,
                                  float *labels, float *targets)
{
    const int A = NUM_ANCHORS;
    int64_t base[16];
    mv3d_ref_generate_anchors_bv(base);
    /* anchor_target_layer_tf.py:93-98 (im_info is f32; ints compare exactly) */
    int ni = 0;
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w)
            for (int a = 0; a < A; ++a) {
                const int64_t sx = (int64_t)w * feat_stride, sy = (int64_t)h * feat_stride;
                const int64_t x1 = base[4 * a] + sx, y1 = base[4 * a + 1] + sy;
                const int64_t x2 = base[4 * a + 2] + sx, y2 = base[4 * a + 3] + sy;
                if (x1 >= 0 && y1 >= 0 && (double)x2 < (double)im_info[1] &&
                    (double)y2 < (double)im_info[0])
                    inds_inside[ni++] = (h * W + w) * A + a;
            }
    double *gtmax = (double *)malloc(sizeof(double) * (G > 0 ? G : 1));
    double *ov = (double *)malloc(sizeof(double) * (size_t)ni * (G > 0 ? G : 1));
    for (int g = 0; g < G; ++g) gtmax[g] = -1.0;
    for (int t = 0; t < ni; ++t) {
        const int n = inds_inside[t], a = n % A, cell = n / A, w = cell % W, h = cell / W;
        const int64_t sx = (int64_t)w * feat_stride, sy = (int64_t)h * feat_stride;
        const double B[4] = {(double)(base[4 * a] + sx), (double)(base[4 * a + 1] + sy),
                             (double)(base[4 * a + 2] + sx), (double)(base[4 * a + 3] + sy)};
        /* :118-119 argmax(axis=1) = first maximum */
        int am = 0; double mx = -1.0;
        for (int g = 0; g < G; ++g) {
            const double Q[4] = {gt_bv[5 * g], gt_bv[5 * g + 1], gt_bv[5 * g + 2], gt_bv[5 * g + 3]};
            const double o = iou_f64(B, Q);
            ov[(size_t)t * G + g] = o;
            if (o > mx) { mx = o; am = g; }
            if (o > gtmax[g]) gtmax[g] = o;     /* :120-122 column max */
        }
        argmax[t] = am; max_ov[t] = mx;
    }
    for (int t = 0; t < ni; ++t) {
        float lab = -1.0f;                                        /* :110-111 */
        const double mx = max_ov[t];
        if (!clobber && 0 < mx && mx < neg_ov) lab = 0.0f;       /* :125-130 */
        for (int g = 0; g < G; ++g)                               /* :123,:133 every tie */
            if (ov[(size_t)t * G + g] == gtmax[g]) { lab = 1.0f; break; }
        if (mx >= pos_ov) lab = 1.0f;                             /* :139 */
        if (clobber && mx < neg_ov) lab = 0.0f;                   /* :141-143 */
        labels[t] = lab;
        /* :165-166, bbox_transform.py:32-58 in f64 then astype(f32) */
        const int n = inds_inside[t], a = n % A, cell = n / A, w = cell % W, h = cell / W;
        const int64_t sx = (int64_t)w * feat_stride, sy = (int64_t)h * feat_stride;
        double ex[6];
        bv_anchor_to_lidar_one(base[4 * a] + sx, base[4 * a + 1] + sy, base[4 * a + 2] + sx,
                               base[4 * a + 3] + sy, ex);
        const float *gt = gt_3d + 7 * argmax[t];
        float *T = targets + 6 * t;
        T[0] = (float)(((double)gt[0] - ex[0]) / ex[4]);   /* dx / ex_widths  (sic) */
        T[1] = (float)(((double)gt[1] - ex[1]) / ex[3]);   /* dy / ex_lengths (sic) */
        T[2] = (float)(((double)gt[2] - ex[2]) / ex[5]);
        T[3] = (float)mv3d_ref_log((double)gt[3] / ex[3]);
        T[4] = (float)mv3d_ref_log((double)gt[4] / ex[4]);
        T[5] = (float)mv3d_ref_log((double)gt[5] / ex[5]);
    }
    free(gtmax); free(ov);
    return ni;
}

/* ------------------------------------------------------------------ a17 helper */
void mv3d_ref_bbox_transform_cnr(const float *ex, const float *gt, int n, float *out)
{
    /* bbox_transform.py:61-72, all f32: diag = ||gt[:,0::8] - gt[:,6::8]||_2 =
     * sqrt(sum(d*d)) with numpy's add.reduce order 0 + d0^2 + d1^2 + d2^2 */
    for (int i = 0; i < n; ++i) {
        const float *g = gt + 24 * i, *e = ex + 24 * i;
        float d0 = g[0] - g[6], d1 = g[8] - g[14], d2 = g[16] - g[22];
        float s = d0 * d0;
        s = s + d1 * d1;
        s = s + d2 * d2;
        float diag = sqrtf(s);
        for (int j = 0; j < 24; ++j) out[24 * i + j] = (g[j] - e[j]) / diag;
    }
}

/* ------------------------------------------------------------------ a18 */
int mv3d_ref_roi_pool_forward(const float *data, int B, int H, int W, int C,
                              const float *rois, int R, int PH, int PW, float scale,
                              float *top, int32_t *argmax)
{
    /* roi_pooling_op.cc:127-181 (one output element per iteration there; the same
     * arithmetic with the roi-level values hoisted here) */
    for (int n = 0; n < R; ++n) {
        const float *roi = rois + 5 * n;
        const int bi = (int)roi[0];
        if (bi < 0 || bi >= B) return -1;
        const int rsw = (int)round(roi[1] * scale), rsh = (int)round(roi[2] * scale);
        const int rew = (int)round(roi[3] * scale), reh = (int)round(roi[4] * scale);
        const int rw = (rew - rsw + 1) > 1 ? (rew - rsw + 1) : 1;   /* malformed -> 1x1 */
        const int rh = (reh - rsh + 1) > 1 ? (reh - rsh + 1) : 1;
        const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
        const float *d = data + (size_t)bi * H * W * C;
        for (int ph = 0; ph < PH; ++ph)
            for (int pw = 0; pw < PW; ++pw) {
                int hs = (int)floor(ph * bh), ws = (int)floor(pw * bw);
                int he = (int)ceil((ph + 1) * bh), we = (int)ceil((pw + 1) * bw);
                hs = hs + rsh; he = he + rsh; ws = ws + rsw; we = we + rsw;
                hs = hs < 0 ? 0 : (hs > H ? H : hs);
                he = he < 0 ? 0 : (he > H ? H : he);
                ws = ws < 0 ? 0 : (ws > W ? W : ws);
                we = we < 0 ? 0 : (we > W ? W : we);
                const int empty = (he <= hs) || (we <= ws);
                for (int c = 0; c < C; ++c) {
                    float mv = empty ? 0.0f : -FLT_MAX;
                    int mi = -1;
                    for (int hh = hs; hh < he; ++hh)
                        for (int ww = ws; ww < we; ++ww) {
                            const int idx = (hh * W + ww) * C + c;
                            if (d[idx] > mv) { mv = d[idx]; mi = idx; }
                        }
                    const size_t o = (((size_t)n * PH + ph) * PW + pw) * C + c;
                    top[o] = mv;
                    if (argmax) argmax[o] = mi;
                }
            }
    }
    return 0;
}

/* ------------------------------------------------------------------ a19 */
void mv3d_ref_roi_pool_backward(const float *top_diff, const int32_t *argmax,
                                int B, int H, int W, int C,
                                const float *rois, int R, int PH, int PW, float scale,
                                float *bottom_diff)
{
    /* roi_pooling_op.cc:373-443: per input element, ROIs ascending, ph then pw
     * ascending; f32 accumulation in that order. */
    for (int n = 0; n < B; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                float *out = bottom_diff + (((size_t)n * H + h) * W + w) * C;
                for (int c = 0; c < C; ++c) out[c] = 0.0f;
                for (int r = 0; r < R; ++r) {
                    const float *roi = rois + 5 * r;
                    if (n != (int)roi[0]) continue;
                    const int rsw = (int)round(roi[1] * scale), rsh = (int)round(roi[2] * scale);
                    const int rew = (int)round(roi[3] * scale), reh = (int)round(roi[4] * scale);
                    if (!(w >= rsw && w <= rew && h >= rsh && h <= reh)) continue;
                    const int rw = (rew - rsw + 1) > 1 ? (rew - rsw + 1) : 1;
                    const int rh = (reh - rsh + 1) > 1 ? (reh - rsh + 1) : 1;
                    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
                    int phs = (int)floor((float)(h - rsh) / bh);
                    int phe = (int)ceil((float)(h - rsh + 1) / bh);
                    int pws = (int)floor((float)(w - rsw) / bw);
                    int pwe = (int)ceil((float)(w - rsw + 1) / bw);
                    phs = phs < 0 ? 0 : (phs > PH ? PH : phs);
                    phe = phe < 0 ? 0 : (phe > PH ? PH : phe);
                    pws = pws < 0 ? 0 : (pws > PW ? PW : pws);
                    pwe = pwe < 0 ? 0 : (pwe > PW ? PW : pwe);
                    const size_t off = (size_t)r * PH * PW * C;
                    for (int ph = phs; ph < phe; ++ph)
                        for (int pw = pws; pw < pwe; ++pw) {
                            const size_t o = off + ((size_t)ph * PW + pw) * C;
                            const int base = (h * W + w) * C;
                            for (int c = 0; c < C; ++c)
                                if (argmax[o + c] == base + c) out[c] += top_diff[o + c];
                        }
                }
            }
}

/* ------------------------------------------------------------------ §8(f) next rows */
float mv3d_ref_floor_dividef(float a, float b)
{
    /* npy_divmodf, f32 throughout */
    if (b == 0.0f) return a / b;
    float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.0f) {
        if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0f; }
    }
    float floordiv;
    if (div != 0.0f) {
        floordiv = floorf(div);
        if (div - floordiv > 0.5f) floordiv += 1.0f;
    } else {
        floordiv = copysignf(0.0f, a / b);
    }
    return floordiv;
}

/* transform.py:342-357 for one box's 24 corner numbers -> (x1,y1,x2,y2) BEV, all f32:
 * np.min/np.max over the 8 x / y corners, then _lidar_to_bv_coord with f32 arrays (RES becomes f32) */
static void corners_to_bv_one(const float *c, float out[4])
{
    const int Xn = grid_n(TOP_X_MIN, TOP_X_MAX), Yn = grid_n(TOP_Y_MIN, TOP_Y_MAX);
    float xmin = c[0], xmax = c[0], ymin = c[8], ymax = c[8];
    int nx = c[0] != c[0], ny = c[8] != c[8];
    for (int k = 1; k < 8; ++k) {
        if (c[k] < xmin) xmin = c[k];
        if (c[k] > xmax) xmax = c[k];
        if (c[8 + k] < ymin) ymin = c[8 + k];
        if (c[8 + k] > ymax) ymax = c[8 + k];
        nx |= c[k] != c[k]; ny |= c[8 + k] != c[8 + k];
    }
    if (nx) xmin = xmax = NAN;
    if (ny) ymin = ymax = NAN;
    const float res = (float)RES;
    out[0] = (float)Yn - mv3d_ref_floor_dividef(ymax - (float)TOP_Y_MIN, res);
    out[1] = (float)Xn - mv3d_ref_floor_dividef(xmax - (float)TOP_X_MIN, res);
    out[2] = (float)Yn - mv3d_ref_floor_dividef(ymin - (float)TOP_Y_MIN, res);
    out[3] = (float)Xn - mv3d_ref_floor_dividef(xmin - (float)TOP_X_MIN, res);
}

void mv3d_ref_box_tail(const float *rois_3d, const float *deltas, int R, int nc, float *corners,
                       float *pred_cnr_r, float *bv, float *bv_r)
{
    for (int i = 0; i < R; ++i) {
        float *c = corners + 24 * i;
        corners_one(rois_3d + 7 * i + 1, c);                       /* test_mv.py:240-244 */
        /* bbox_transform.py:157-176: diag = ||p0 - p6|| (f32), deltas * diag + boxes */
        const float d0 = c[0] - c[6], d1 = c[8] - c[14], d2 = c[16] - c[22];
        float ss = d0 * d0;
        ss = ss + d1 * d1;
        ss = ss + d2 * d2;
        const float diag = sqrtf(ss);
        for (int k = 0; k < nc; ++k) {
            for (int j = 0; j < 24; ++j)
                pred_cnr_r[(size_t)i * 24 * nc + 24 * k + j] = deltas[(size_t)i * 24 * nc + 24 * k + j] * diag + c[j];
            corners_to_bv_one(c, bv + (size_t)i * 4 * nc + 4 * k);                 /* hstack((cnr, cnr)) */
            corners_to_bv_one(pred_cnr_r + (size_t)i * 24 * nc + 24 * k, bv_r + (size_t)i * 4 * nc + 4 * k);
        }
    }
}

/* read_lidar.py:10-115 with its own parameters.  What numpy (2.2, NEP 50) does with the dtypes, restated:
 *   x_points > fwd_range[0] ...   f32 array against a Python float: compared in f32 (the bound rounded to f32)      (:58-61)
 *   np.arange(h0, h1, zres)       length ceil((h1 - h0) / zres); values h0, h0 + zres, then h0 + i * ((h0 + zres) - h0) -- numpy fills
 *                                 from the first two elements, so the step is the ROUNDED difference, not zres                (:80)
 *   z_points >= height            f32 array against an np.float64 scalar: compared in f64; height + zres in f64               (:82-83)
 *   (-y / res).astype(int32)      f32 divide by f32(res), truncation                                                         (:96-97)
 *   z - height_range[0]           f32                                                                                        (:106)
 *   top[y, x, i] = ...            last assignment wins (point order inside a slice, slices in order); channel z_max = reflectance;
 *                                 negative indices wrap once, anything else out of range is numpy's IndexError: *err = 1      (:109-112)
 * dims[3] = (y_max + 1, x_max + 1, z_max + 1) of the output (top must hold them; call with top = NULL to get the shape only). */
void mv3d_ref_point_cloud_2_top_ranges(const float *pts, int P, double res, double zres, double side0, double side1, double fwd0, double fwd1,
                                       double h0, double h1, int *dims, float *top, int *err)
{
    const int x_max = (int)((side1 - side0) / res), y_max = (int)((fwd1 - fwd0) / res), z_max = (int)((h1 - h0) / zres);
    const int Wd = x_max + 1, Hd = y_max + 1, Cd = z_max + 1;
    if (dims) { dims[0] = Hd; dims[1] = Wd; dims[2] = Cd; }
    if (err) *err = 0;
    if (!top) return;
    memset(top, 0, sizeof(float) * (size_t)Wd * Hd * Cd);
    const int xoff = (int)floor(side0 / res), yoff = (int)floor(fwd1 / res);
    int nslice = (int)ceil((h1 - h0) / zres);
    if (nslice < 0) nslice = 0;
    const double next = h0 + zres, delta = next - h0;
    const float resf = (float)res, fwd0f = (float)fwd0, fwd1f = (float)fwd1, ylo = (float)(-side1), yhi = (float)(-side0), h0f = (float)h0;
    for (int i = 0; i < nslice; ++i) {
        const double height = i == 0 ? h0 : (i == 1 ? next : h0 + (double)i * delta);
        const double upper = height + zres;
        for (int p = 0; p < P; ++p) {
            const float x = pts[4 * p], y = pts[4 * p + 1], z = pts[4 * p + 2], r = pts[4 * p + 3];
            if (!(x > fwd0f && x < fwd1f)) continue;
            if (!(y > ylo && y < yhi)) continue;
            if (!((double)z >= height && (double)z < upper)) continue;
            int x_img = (int)(-y / resf), y_img = (int)(-x / resf);
            x_img -= xoff; y_img += yoff;
            if (x_img < 0) x_img += Wd;
            if (y_img < 0) y_img += Hd;
            if (x_img < 0 || x_img >= Wd || y_img < 0 || y_img >= Hd) { if (err) *err = 1; continue; }
            float *cell = top + ((size_t)y_img * Wd + x_img) * Cd;
            cell[i] = z - h0f;
            cell[z_max] = r;
        }
    }
}

void mv3d_ref_point_cloud_2_top(const float *pts, int P, float *top)
{
    /* the call MV3D makes (tools/read_lidar.py:121-133): (601, 601, 9) */
    mv3d_ref_point_cloud_2_top_ranges(pts, P, 0.1, 0.3, -30.0, 30.0, 0.0, 60.0, -2.0, 0.4, NULL, top, NULL);
}

/* ------------------------------------------------------------------ X1: third (front-view) ROI
 * PARITY UNPINNED: the reference has no front-view projection (lib/networks/network.py:293-315 leaves it a TODO
 * that returns None).  This restates the definition the product documents in mv3d_tf_amd/csrc/front_view.hip:
 * cylindrical projection of the MV3D paper (c = floor(atan2(y,x)/d_theta), r = floor(atan2(z, sqrt(x^2+y^2))/d_phi))
 * over a 64 x 512 map, azimuth [-45,+45] deg (column 0 = +45 deg), elevation [-24.9,+2] deg (row 0 = +2 deg), the 8
 * corners of lidar_3d_to_corners (transform.py:290-315), min/max, clipped to the map, NaN -> 0.  The arctangent is
 * a table + series built from IEEE basic operations only, so that C and HIP agree bit for bit. */
static const double ATAN16[17] = {
    0x0.0p+0, 0x1.ff55bb72cfdeap-5, 0x1.fd5ba9aac2f6ep-4, 0x1.7b97b4bce5b02p-3, 0x1.f5b75f92c80ddp-3,
    0x1.362773707ebccp-2, 0x1.6f61941e4def1p-2, 0x1.a64eec3cc23fdp-2, 0x1.dac670561bb4fp-2, 0x1.0657e94db30d0p-1,
    0x1.1e00babdefeb4p-1, 0x1.345f01cce37bbp-1, 0x1.4978fa3269ee1p-1, 0x1.5d58987169b18p-1, 0x1.700a7c5784634p-1,
    0x1.819d0b7158a4dp-1, 0x1.921fb54442d18p-1};
static const double HALF_PI = 0x1.921fb54442d18p+0, FULL_PI = 0x1.921fb54442d18p+1;

double mv3d_ref_fv_atan2(double y, double x)
{
    if (x != x || y != y) return NAN;
    if (x == 0.0) return y > 0.0 ? HALF_PI : (y < 0.0 ? -HALF_PI : 0.0);
    double a = fabs(y / x);
    const int inv = a > 1.0;
    if (inv) a = 1.0 / a;
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
    double q = ATAN16[k] + fma(t * s, p, t);
    if (inv) q = HALF_PI - q;
    const double w = x > 0.0 ? q : FULL_PI - q;
    return y < 0.0 ? -w : w;
}

static float fv_clip(double v, double hi)
{
    v = (v >= 0.0) ? v : 0.0;
    v = (v <= hi) ? v : hi;
    return (float)v;
}

void mv3d_ref_rois_3d_to_fv(const float *rois_3d, int R, float *rois_fv)
{
    const double theta_max = 0x1.921fb54442d18p-1, dtheta = 0x1.921fb54442d18p-9;
    const double phi_top = 0x1.1df46a2529d39p-5, dphi = 0x1.e0c2ec0e7b1eep-8;
    for (int i = 0; i < R; ++i) {
        float c[24];
        corners_one(rois_3d + 7 * (long)i + 1, c);
        double cmin = 0, cmax = 0, rmin = 0, rmax = 0;
        int bad = 0;
        for (int k = 0; k < 8; ++k) {
            const double x = (double)c[k], y = (double)c[8 + k], z = (double)c[16 + k];
            const double theta = mv3d_ref_fv_atan2(y, x);
            const double rho = sqrt(fma(x, x, y * y));
            const double phi = mv3d_ref_fv_atan2(z, rho);
            const double col = floor((theta_max - theta) / dtheta);
            const double row = floor((phi_top - phi) / dphi);
            if (col != col || row != row) bad = 1;
            if (k == 0) { cmin = cmax = col; rmin = rmax = row; }
            else {
                if (col < cmin) cmin = col;
                if (col > cmax) cmax = col;
                if (row < rmin) rmin = row;
                if (row > rmax) rmax = row;
            }
        }
        if (bad) cmin = cmax = rmin = rmax = NAN;
        float *o = rois_fv + 5 * (long)i;
        o[0] = rois_3d[7 * (long)i];
        o[1] = fv_clip(cmin, 511.0); o[2] = fv_clip(rmin, 63.0);
        o[3] = fv_clip(cmax, 511.0); o[4] = fv_clip(rmax, 63.0);
    }
}

/* ------------------------------------------------------------------ f3: KITTI label -> ground-truth encodings
 * lib/datasets/kitti_mv3d.py:240-272 for one frame's kept objects, after the text has been parsed:
 *   computeCorners3D (lib/utils/transform.py:441-465): camera box (f32 label row) + yaw -> 8 camera corners.  The
 *     rotation is f64 (np.cos / np.sin of the Python float, passed in), the half extents are f32 (f32 / int), the
 *     local corner matrix is f64 (vstack of f32, f64, f32 rows), np.dot = k-ascending fma from 0, then + centre (f32 -> f64).
 *   camera_to_lidar_cnr (:502-524): [inv(Tr[:, :3]) (f32, numpy/LAPACK on the host: passed in) | (-Tr[1,3], -Tr[2,3], Tr[0,3])]
 *     (f64) . [corners; 0], k-ascending fma -- the homogeneous coordinate is 0, so the translation is dropped (sic).
 *   lidar_cnr_to_3d (:172-187): centre = f32 mean of the 8 f32 corners (numpy pairwise sum of 8), sizes = (l, w, h).
 *   lidar_3d_to_bv (:113-142) + _lidar_to_bv_coord (:13-20): f32 corner sums, f64 floor-divide.
 * Outputs are the f32 arrays the roidb holds: boxes3D_cam_corners (G,24), boxes_corners (G,24), boxes_3D (G,6),
 * boxes_bv (G,4). */
void mv3d_ref_gt_encode(const float *box_cam, const double *cos_sin, int G, const float *inv_rot, const float *Tr,
                        float *cnr_cam, float *cnr_lidar, float *box_lidar, float *boxes_bv)
{
    const int Xn = grid_n(TOP_X_MIN, TOP_X_MAX), Yn = grid_n(TOP_Y_MIN, TOP_Y_MAX);           /* 600 (transform.py:8-9) */
    for (int g = 0; g < G; ++g) {
        const float *b = box_cam + 6 * g;
        const double c = cos_sin[2 * g], s = cos_sin[2 * g + 1];
        const double rot[3][3] = {{c, 0.0, s}, {0.0, 1.0, 0.0}, {-s, 0.0, c}};
        const float hl = b[3] / 2.0f, hw = b[4] / 2.0f, hgt = b[5];
        double loc[3][8];
        const int sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sz[8] = {1, -1, -1, 1, 1, -1, -1, 1};
        for (int j = 0; j < 8; ++j) {
            loc[0][j] = (double)(sx[j] > 0 ? hl : -hl);
            loc[1][j] = (j < 4) ? 0.0 : (double)(-hgt);
            loc[2][j] = (double)(sz[j] > 0 ? hw : -hw);
        }
        double cam[3][8];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 8; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 3; ++k) acc = fma(rot[i][k], loc[k][j], acc);
                cam[i][j] = acc + (double)b[i];
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 8; ++j) cnr_cam[24 * g + 8 * i + j] = (float)cam[i][j];
        const double M[3][4] = {{inv_rot[0], inv_rot[1], inv_rot[2], -(double)Tr[7]},
                                {inv_rot[3], inv_rot[4], inv_rot[5], -(double)Tr[11]},
                                {inv_rot[6], inv_rot[7], inv_rot[8], (double)Tr[3]}};
        float *cl = cnr_lidar + 24 * g;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 8; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 3; ++k) acc = fma(M[i][k], cam[k][j], acc);
                acc = fma(M[i][3], 0.0, acc);
                cl[8 * i + j] = (float)acc;
            }
        float *bl = box_lidar + 6 * g;
        for (int i = 0; i < 3; ++i) {
            const float *r = cl + 8 * i;
            const float sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            bl[i] = sum / 8.0f;
        }
        bl[3] = b[3]; bl[4] = b[4]; bl[5] = b[5];
        const float x1 = bl[0] + bl[3] * 0.5f, y1 = bl[1] + bl[4] * 0.5f;
        const float x2 = bl[0] - bl[3] * 0.5f, y2 = bl[1] - bl[4] * 0.5f;
        float *bv = boxes_bv + 4 * g;
        bv[0] = (float)((double)Yn - mv3d_ref_floor_divide((double)y1 - (double)TOP_Y_MIN, RES));
        bv[1] = (float)((double)Xn - mv3d_ref_floor_divide((double)x1 - (double)TOP_X_MIN, RES));
        bv[2] = (float)((double)Yn - mv3d_ref_floor_divide((double)y2 - (double)TOP_Y_MIN, RES));
        bv[3] = (float)((double)Xn - mv3d_ref_floor_divide((double)x2 - (double)TOP_X_MIN, RES));
    }
}
