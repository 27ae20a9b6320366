// Internal launchers shared between the translation units of libmv3d_hip.so.
#pragma once
#include "common.h"

// ---- greedy NMS over boxes already in processing order (nms.hip) -------------------
// Box p of frame f lives at boxes + f*boxes_frame_stride + q*box_stride (4 floats
// x1,y1,x2,y2) with q = idx ? idx[f*idx_frame_stride + p] : p.
struct NmsLaunch {
    const float *boxes;
    int box_stride;
    long long boxes_frame_stride;
    const int32_t *idx;
    long long idx_frame_stride;
    const int32_t *n_dev;       // optional per-frame count; n = min(n_dev[f], n_cap)
    int n_cap;                  // static upper bound on boxes per frame
    int batch;
    float thresh_f32;           // ceil_f32(thresh) for the CPU rule, thresh for the CUDA rule
    int strict_gt;              // 0: IoU >= thresh (cpu_nms.pyx:65); 1: IoU > thresh (nms_kernel.cu:71)
    int max_keep;               // <= 0: no cap
    int32_t *keep;              // (batch, keep_frame_stride) positions p, in order
    long long keep_frame_stride;
    int32_t *num_keep;          // (batch)
    int32_t *status;            // (batch) flag bits, may be NULL; must be zeroed by the caller
    void *workspace;            // mv3d_nms_ws_bytes(n_cap, batch)
};
size_t mv3d_nms_ws_bytes(int n_cap, int batch);
int mv3d_launch_nms(const NmsLaunch &L, hipStream_t stream);

// ---- rank by counting (rank.hip) -----------------------------------------------------
// keys (batch, N) u32, 0 = not a candidate.  order[f*cap + r] = i for the candidate i of
// frame f whose rank r (number of candidates that precede it: larger key, ties by larger
// index) is < cap.  Entries r >= number of candidates are left untouched.
int mv3d_launch_rank(const uint32_t *keys, int N, int batch, int32_t *order, int cap,
                     hipStream_t stream);
