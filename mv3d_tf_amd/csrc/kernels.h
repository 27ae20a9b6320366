// Internal launchers shared between the translation units of libmv3d_hip.so.
#pragma once
#include "common.h"

// ---- ROI blob emission fused into the tail of the NMS reduce kernel -----------------
// Row r < num_keep of frame f takes candidate n = order[keep[r]]; rows up to cap are zeroed.
struct EmitDev {
    int enabled;
    int N, order_cap, cap;
    const float4 *bv;           // (batch, N) clipped BEV boxes
    const int4 *img;            // (batch, N) image boxes
    const float *p3;            // (batch, N, 6) decoded 3D boxes
    const int32_t *order;       // (batch, order_cap)
    float *blob_bv, *blob_img, *blob_3d;   // (batch, cap, 5|5|7)
    int32_t *num_out;           // (batch)
};

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
    EmitDev emit;               // optional fused ROI-blob emission (proposal_layer_tf.py:188-191)
    long long *trace;           // diagnostics (mv3d_nms_device_trace), may be NULL
};
size_t mv3d_nms_ws_bytes(int n_cap, int batch);
int mv3d_launch_nms(const NmsLaunch &L, hipStream_t stream);

// ---- rank by counting (rank.hip) -----------------------------------------------------
// keys (batch, key_stride) u32, 0 = not a candidate; key_stride = mv3d_rank_key_stride(N) and the
// tail keys[N..key_stride) of every frame must be 0.  order[f*cap + r] = i for the candidate i of
// frame f whose rank r (number of candidates that precede it: larger key, ties by larger
// index) is < cap.  Entries r >= number of candidates are left untouched.
// part_counts (batch, n_parts): per-producer-workgroup candidate counts; their per-frame
// sum is written to n_valid (batch).  Both may be NULL.  workspace: mv3d_rank_ws_bytes().
// gather_src (batch, N) / gather_dst (batch, cap), optional: gather_dst[f*cap + r] = gather_src[f*N + i]
// alongside order (16-byte records in rank order for the consumer, e.g. the NMS boxes).
size_t mv3d_rank_ws_bytes(int N, int batch);
int mv3d_rank_key_stride(int N);
int mv3d_launch_rank(const uint32_t *keys, int N, int key_stride, int batch, int32_t *order, int cap,
                     const int32_t *part_counts, int n_parts, int32_t *n_valid, void *workspace,
                     hipStream_t stream, const float4 *gather_src = nullptr, float4 *gather_dst = nullptr);



// ---- RoiPoolGrad of the pair as one launch of LDS-resident map tiles (roi_grad_tiles.hip) -----------------
// views validated by the caller (roi_pair_shapes of roi_pool.hip); argmax_data holds the pair's compact codes
int mv3d_launch_roi_pair_tiles(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, hipStream_t stream);

// ---- the trunks' input layer (c_in = 16, c_out = 64, 16-bit maps) as a kernel of its own (conv_input.hip) ------------------
int mv3d_launch_conv_input_f16(int num_views, const mv3d_conv_view *views, int out_framed, int relu, hipStream_t stream);
int mv3d_launch_conv_input_bf16(int num_views, const mv3d_conv_view *views, int out_framed, int relu, hipStream_t stream);
