// KITTI label rows -> ground-truth encodings on the device: one thread per labelled object.
// Replaces the per-object Python chain of lib/datasets/kitti_mv3d.py:240-272:
//   computeCorners3D      lib/utils/transform.py:441-465   camera box + yaw -> 8 camera corners
//   camera_to_lidar_cnr   lib/utils/transform.py:502-524   -> 8 LIDAR corners (inverse rotation of Tr_velo_to_cam; the
//                                                           homogeneous coordinate is 0, so the translation is dropped, sic)
//   lidar_cnr_to_3d       lib/utils/transform.py:172-187   -> LIDAR box (centre = mean of the corners)
//   lidar_3d_to_bv        lib/utils/transform.py:113-142   -> BEV pixel box (f64 floor-divide, Xn = Yn = 600)
// Rounding follows the dtypes the reference's numpy expressions have (pinned bit for bit by tests/golden/kitti_label.npz,
// which the reference's own loader produced): f32 label values and half extents, f64 rotation and matrix products
// (k-ascending fma from 0, what np.dot does on these tiny matrices), f32 stores, f32 pairwise mean of 8, f32 corner sums,
// f64 floor-divide.  Two things stay on the host because they are numpy / LAPACK library calls on a handful of scalars:
// np.cos / np.sin of the yaw (f64) and np.linalg.inv of the 3x3 f32 rotation of Tr (once per frame).
#include "geometry.h"

__global__ __launch_bounds__(64) void gt_encode_kernel(const float *__restrict__ box_cam, const double *__restrict__ cos_sin, int G,
                                                       const float *__restrict__ inv_rot, const float *__restrict__ Tr,
                                                       float *__restrict__ cnr_cam, float *__restrict__ cnr_lidar,
                                                       float *__restrict__ box_lidar, float *__restrict__ boxes_bv)
{
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= G) return;
    const float *b = box_cam + 6 * (long long)g;
    const double c = cos_sin[2 * g], s = cos_sin[2 * g + 1];
    const float hl = b[3] / 2.0f, hw = b[4] / 2.0f, hgt = b[5];
    // rows of the rotation about the camera's y axis, and of [inv(R) | (-Tr[1,3], -Tr[2,3], Tr[0,3])]
    const double rot[3][3] = {{c, 0.0, s}, {0.0, 1.0, 0.0}, {-s, 0.0, c}};
    const double M[3][4] = {{(double)inv_rot[0], (double)inv_rot[1], (double)inv_rot[2], -(double)Tr[7]},
                            {(double)inv_rot[3], (double)inv_rot[4], (double)inv_rot[5], -(double)Tr[11]},
                            {(double)inv_rot[6], (double)inv_rot[7], (double)inv_rot[8], (double)Tr[3]}};
    float lid[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // local corner j: x = +-l/2 [+,+,-,-,+,+,-,-], y = 0 (bottom) / -h (top), z = +-w/2 [+,-,-,+,+,-,-,+]
        const double lx = (double)((j & 2) ? -hl : hl);
        const double ly = (j < 4) ? 0.0 : (double)(-hgt);
        const double lz = (double)(((j + 1) & 2) ? -hw : hw);
        double cam[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double acc = 0.0;
            acc = fma(rot[i][0], lx, acc);
            acc = fma(rot[i][1], ly, acc);
            acc = fma(rot[i][2], lz, acc);
            cam[i] = acc + (double)b[i];
            cnr_cam[24 * (long long)g + 8 * i + j] = (float)cam[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double acc = 0.0;
            acc = fma(M[i][0], cam[0], acc);
            acc = fma(M[i][1], cam[1], acc);
            acc = fma(M[i][2], cam[2], acc);
            acc = fma(M[i][3], 0.0, acc);
            lid[i][j] = (float)acc;
            cnr_lidar[24 * (long long)g + 8 * i + j] = lid[i][j];
        }
    }
    float ctr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float *r = lid[i];
        // numpy's pairwise f32 sum of exactly 8 values, then / 8
        const float sum = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        ctr[i] = sum / 8.0f;
    }
    float *bl = box_lidar + 6 * (long long)g;
    bl[0] = ctr[0]; bl[1] = ctr[1]; bl[2] = ctr[2]; bl[3] = b[3]; bl[4] = b[4]; bl[5] = b[5];
    const float x1 = __fadd_rn(ctr[0], __fmul_rn(b[3], 0.5f)), y1 = __fadd_rn(ctr[1], __fmul_rn(b[4], 0.5f));
    const float x2 = __fsub_rn(ctr[0], __fmul_rn(b[3], 0.5f)), y2 = __fsub_rn(ctr[1], __fmul_rn(b[4], 0.5f));
    float *bv = boxes_bv + 4 * (long long)g;
    bv[0] = (float)(BV_YN - np_floor_divide((double)y1 - TOP_Y_MIN_D, BV_RES));
    bv[1] = (float)(BV_XN - np_floor_divide((double)x1 - TOP_X_MIN_D, BV_RES));
    bv[2] = (float)(BV_YN - np_floor_divide((double)y2 - TOP_Y_MIN_D, BV_RES));
    bv[3] = (float)(BV_XN - np_floor_divide((double)x2 - TOP_X_MIN_D, BV_RES));
}

extern "C" int mv3d_gt_encode(const float *box_cam_dev, const double *cos_sin_dev, int num_objects, const float *inv_rot_dev,
                              const float *tr_velo_to_cam_dev, float *corners_cam_dev, float *corners_lidar_dev,
                              float *boxes_3d_dev, float *boxes_bv_dev, void *stream)
{
    if (num_objects < 0) return MV3D_ERR_INVALID_ARG;
    if (num_objects == 0) return MV3D_OK;
    if (!box_cam_dev || !cos_sin_dev || !inv_rot_dev || !tr_velo_to_cam_dev || !corners_cam_dev || !corners_lidar_dev ||
        !boxes_3d_dev || !boxes_bv_dev)
        return MV3D_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gt_encode_kernel, dim3((num_objects + 63) / 64), dim3(64), 0, (hipStream_t)stream, box_cam_dev, cos_sin_dev,
                       num_objects, inv_rot_dev, tr_velo_to_cam_dev, corners_cam_dev, corners_lidar_dev, boxes_3d_dev, boxes_bv_dev);
    return mv3d_launch_status();
}
