// Rounded ROI geometry shared by the RoiPool kernels (roi_pool.hip, roi_grad_tiles.hip).
#pragma once
#include "common.h"

struct RoiGeom { int rsw, rsh, rew, reh; };

// roi_pooling_op.cc:139-143: round() (half away from zero) of the f32 product
__device__ __forceinline__ RoiGeom roi_geom(const float *roi, float scale)
{
    RoiGeom g;
    g.rsw = (int)roundf(__fmul_rn(roi[1], scale));
    g.rsh = (int)roundf(__fmul_rn(roi[2], scale));
    g.rew = (int)roundf(__fmul_rn(roi[3], scale));
    g.reh = (int)roundf(__fmul_rn(roi[4], scale));
    return g;
}
