// Host side of the library: the training path on FRESH frames, driven from C (mv3d_train_path_* of include/mv3d_hip.h).
//
// What it stands in for: the three py_func layers lib/networks/MV3D_train.py:83-112 hangs behind the RPN heads -- proposal_layer_3d
// (lib/rpn_msr/proposal_layer_tf.py:25-202), anchor_target_layer (lib/rpn_msr/anchor_target_layer_tf.py:21-250) and
// proposal_target_layer_3d (lib/rpn_msr/proposal_target_layer_tf.py:19-94) -- as a session runs them once per training batch.  The
// device work is the existing entries (mv3d_proposal_3d, mv3d_anchor_target_stage1/2_batch, mv3d_proposal_target_stage1/2_batch_devn);
// the reference's subsampling draws between their two stages are mv3d_draw_training_subsamples (legacy_rng.hip) on numpy's own
// generator.  Through round 4 a Python class issued these calls; its ~0.3 ms of interpreter time per batch had become the limit of
// the path on new inputs (the device needs ~0.15 ms per batch), so the sequence lives here now:
//   * submit(): stage 1 + one small launch that writes the batch's reports into the pinned host buffers (tp_report_kernel; three
//     device-to-host copies when those buffers are not mapped into the device) + an event, all on the caller's stream -- one call;
//   * a helper thread per object: waits for the event, draws (slots strictly in submission order = the reference's order of draws),
//     uploads the lists, enqueues stage 2 on the same stream.  The caller's thread never touches the draws;
//   * finish(): waits for "stage 2 enqueued" and reports the row counts.
// No device memory is owned here: every buffer is the caller's (mv3d_train_path_slot).
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "common.h"
#include "../../include/mv3d_hip.h"

#define TP_MAX_BATCH 16

namespace {

enum SlotState { TP_IDLE = 0, TP_SUBMITTED = 1, TP_STAGED = 2 };

struct Slot {
    mv3d_train_path_slot buf;
    hipEvent_t ev = nullptr;
    int state = TP_IDLE, rc = MV3D_OK;
    hipStream_t stream = nullptr;
    void *mt = nullptr;
    // the batch's inputs (the caller keeps them alive until finish())
    const float *calib = nullptr;
    const float *gt_bv[TP_MAX_BATCH], *gt_3d[TP_MAX_BATCH], *gt_c[TP_MAX_BATCH];
    int G[TP_MAX_BATCH];
    int32_t sizes[5 * TP_MAX_BATCH], rows[TP_MAX_BATCH];
    std::vector<uint8_t> spill[TP_MAX_BATCH];            // a frame's foreground flags when more than travel with the report's head
    // per-frame pointer tables that never change
    float *p_bv[TP_MAX_BATCH], *p_3d[TP_MAX_BATCH];
    int32_t *p_num[TP_MAX_BATCH], *p_cnt[TP_MAX_BATCH], *a_cnt[TP_MAX_BATCH];
    uint8_t *a_fgh[TP_MAX_BATCH];
    void *a_ws[TP_MAX_BATCH], *p_ws[TP_MAX_BATCH];
    int p_cap[TP_MAX_BATCH];
    size_t p_wsz[TP_MAX_BATCH];
    // the pinned host buffers as the device sees them (hipHostGetDevicePointer; NULL: not mapped -- the reports travel as three copies)
    uint8_t *dv_report = nullptr;
    int32_t *dv_pt_counts = nullptr, *dv_num_proposals = nullptr;
};

static bool tp_force_copies()
{
#ifdef MV3D_TUNING                                                     // (experiment builds: the three copy launches for an A / B)
    return getenv("MV3D_TRAIN_PATH_COPIES") != nullptr;
#else
    return false;
#endif
}

// A batch's reports -- the head of every frame's anchor report, the proposal-target counts, the proposal counts | status words -- written
// by ONE small launch straight into the caller's pinned host buffers (16-byte stores over the host link) instead of three copy launches of
// the runtime (17 us of queue time each with eight batches in flight, profiles/r06_s_kernel_stats.txt: __amd_rocclr_copyBuffer).
__global__ __launch_bounds__(256) void tp_report_kernel(const uint8_t *__restrict__ report, size_t report_row, uint8_t *__restrict__ h_report,
                                                        size_t h_row, int B, const int32_t *__restrict__ pt_counts, int32_t *__restrict__ h_pt,
                                                        const int32_t *__restrict__ nump, int32_t *__restrict__ h_nump)
{
    const size_t v = h_row / 16;                                       // (h_row is a multiple of 16: slot_ok)
    for (size_t i = threadIdx.x; i < (size_t)B * v; i += 256) {
        const size_t b = i / v, j = i - b * v;
        reinterpret_cast<uint4 *>(h_report + b * h_row)[j] = reinterpret_cast<const uint4 *>(report + b * report_row)[j];
    }
    for (int i = threadIdx.x; i < 4 * B; i += 256) h_pt[i] = pt_counts[i];
    for (int i = threadIdx.x; i < 2 * B; i += 256) h_nump[i] = nump[i];
}

}  // namespace

struct mv3d_train_path {
    mv3d_train_path_config cfg;
    mv3d_proposal_target_params tpar[TP_MAX_BATCH];
    int depth = 0, device = 0;
    std::vector<Slot> slots;
    bool threaded = false, stop = false;
    std::thread helper;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<int> queue;
    double t_wait = 0.0, t_draw = 0.0;
};

namespace {

bool config_ok(const mv3d_train_path_config &c)
{
    return c.batch >= 1 && c.batch <= TP_MAX_BATCH && c.H > 0 && c.W > 0 && c.num_classes >= 1 && c.proposal_cap > 0 && c.anchor_cap > 0 &&
           c.roi_cap > 0 && c.max_gt > 0 && c.draw.rois_per_image <= c.roi_cap;
}

void set_params(mv3d_train_path *tp, const mv3d_train_path_config &c)
{
    tp->cfg = c;
    for (int b = 0; b < c.batch; ++b) {
        tp->tpar[b] = c.target;
        tp->tpar[b].num_classes = c.num_classes;
        tp->tpar[b].frame_index = b;
    }
}

bool slot_ok(const mv3d_train_path_config &c, const mv3d_train_path_slot &s)
{
    const size_t N = (size_t)4 * c.H * c.W, B = (size_t)c.batch;
    if (!s.blob_bv || !s.blob_img || !s.blob_3d || !s.num_proposals || !s.proposal_ws || !s.rpn_labels || !s.rpn_targets || !s.anchors ||
        !s.anchors_3d || !s.n_anchors || !s.report || !s.pt_counts || !s.anchor_ws || !s.target_ws || !s.rois_bev || !s.rois_rgb ||
        !s.rois_3d || !s.labels || !s.bbox_targets || !s.lists || !s.h_report || !s.h_pt_counts || !s.h_num_proposals || !s.h_lists ||
        !s.h_scratch)
        return false;
    if (s.report_row % MV3D_ALIGN || s.report_row < 32 + N || s.h_report_row < 32 || s.h_report_row > s.report_row) return false;
    if (s.anchor_ws_bytes % MV3D_ALIGN || s.target_ws_bytes % MV3D_ALIGN) return false;
    if (s.lists_cap < B * (3 * N + 2 * ((size_t)c.proposal_cap + c.max_gt))) return false;
    if (s.scratch_cap < (N > (size_t)c.proposal_cap + c.max_gt ? N : (size_t)c.proposal_cap + c.max_gt)) return false;
    return true;
}

void slot_tables(const mv3d_train_path_config &c, Slot &s)
{
    const mv3d_train_path_slot &u = s.buf;
    for (int b = 0; b < c.batch; ++b) {
        s.p_bv[b] = u.blob_bv + (size_t)b * c.proposal_cap * 5;
        s.p_3d[b] = u.blob_3d + (size_t)b * c.proposal_cap * 7;
        s.p_num[b] = u.num_proposals + b;
        s.p_cnt[b] = u.pt_counts + 4 * b;
        s.a_cnt[b] = (int32_t *)(u.report + (size_t)b * u.report_row);
        s.a_fgh[b] = u.report + (size_t)b * u.report_row + 32;
        s.a_ws[b] = (char *)u.anchor_ws + (size_t)b * u.anchor_ws_bytes;
        s.p_ws[b] = (char *)u.target_ws + (size_t)b * u.target_ws_bytes;
        s.p_cap[b] = c.proposal_cap;
        s.p_wsz[b] = u.target_ws_bytes;
    }
}

#define TP_HIP(expr)                                 \
    do {                                             \
        if ((expr) != hipSuccess) return MV3D_ERR_HIP; \
    } while (0)

// the host stage of a batch: wait for its reports, draw, upload the lists, enqueue stage 2
int host_stage(mv3d_train_path *tp, Slot &s)
{
    const mv3d_train_path_config &c = tp->cfg;
    const mv3d_train_path_slot &u = s.buf;
    const int B = c.batch, H = c.H, W = c.W;
    const auto t0 = std::chrono::steady_clock::now();
    TP_HIP(hipEventSynchronize(s.ev));
    const auto t1 = std::chrono::steady_clock::now();
    for (int b = 0; b < B; ++b)
        if (u.h_num_proposals[B + b] & MV3D_FLAG_ZERO_DIVISION) return MV3D_ERR_ZERO_DIVISION;      // lib/nms/cpu_nms.pyx:64
    mv3d_draw_frame fr[TP_MAX_BATCH];
    for (int b = 0; b < B; ++b) {
        const uint8_t *head = u.h_report + (size_t)b * u.h_report_row;
        const int32_t *cnt = (const int32_t *)head;                  // [n_inside, n_fg, n_bg, n_low, ...]
        fr[b].n_fg = cnt[1]; fr[b].n_bg = cnt[2]; fr[b].n_low = cnt[3];
        fr[b].pt_n_fg = u.h_pt_counts[4 * b + 1]; fr[b].pt_n_bg = u.h_pt_counts[4 * b + 2];
        fr[b].reserved0 = 0;
        if (cnt[1] < 0 || (size_t)cnt[1] > (size_t)4 * H * W) return MV3D_ERR_INVALID_ARG;
        if (32 + (size_t)cnt[1] <= u.h_report_row) {
            fr[b].fg_alive = head + 32;
        } else {                                                     // (more positives than travel with the first copy: rare)
            s.spill[b].resize((size_t)cnt[1]);
            TP_HIP(hipMemcpyAsync(s.spill[b].data(), s.a_fgh[b], (size_t)cnt[1], hipMemcpyDeviceToHost, s.stream));
            TP_HIP(hipStreamSynchronize(s.stream));
            fr[b].fg_alive = s.spill[b].data();
        }
    }
    int rc = mv3d_draw_training_subsamples(s.mt, B, fr, &c.draw, u.h_lists, u.lists_cap, s.sizes, u.h_scratch, u.scratch_cap);
    const auto t2 = std::chrono::steady_clock::now();
    {
        std::lock_guard<std::mutex> g(tp->m);
        tp->t_wait += std::chrono::duration<double>(t1 - t0).count();
        tp->t_draw += std::chrono::duration<double>(t2 - t1).count();
    }
    if (rc != MV3D_OK) return rc;
    size_t off[5 * TP_MAX_BATCH + 1];
    off[0] = 0;
    for (int k = 0; k < 5 * B; ++k) off[k + 1] = off[k] + (size_t)s.sizes[k];
    // (stage 2 reading the lists in place from the pinned buffer instead of this upload was measured: no gain, profiles/r06_t_path_copies.txt)
    const int32_t *const lists = u.lists;
    if (off[5 * B]) TP_HIP(hipMemcpyAsync(u.lists, u.h_lists, off[5 * B] * sizeof(int32_t), hipMemcpyHostToDevice, s.stream));
    const int32_t *lst[5][TP_MAX_BATCH];
    int cnt[5][TP_MAX_BATCH];
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 5; ++k) {
            cnt[k][b] = s.sizes[5 * b + k];
            lst[k][b] = cnt[k][b] ? lists + off[5 * b + k] : nullptr;
        }
    rc = mv3d_anchor_target_stage2_batch(B, H, W, &c.anchor, lst[0], cnt[0], lst[1], cnt[1], lst[2], cnt[2], u.rpn_labels, u.anchors,
                                         u.anchors_3d, u.n_anchors, c.anchor_cap, s.a_ws, u.anchor_ws_bytes, s.stream);
    if (rc != MV3D_OK) return rc;
    float *o_bev[TP_MAX_BATCH], *o_rgb[TP_MAX_BATCH], *o_fv[TP_MAX_BATCH], *o_3d[TP_MAX_BATCH], *o_tg[TP_MAX_BATCH];
    int32_t *o_lab[TP_MAX_BATCH];
    const float *cal[TP_MAX_BATCH];
    size_t row = 0;
    for (int b = 0; b < B; ++b) {
        const int S = cnt[3][b] + cnt[4][b];
        s.rows[b] = S;
        if (row + (size_t)S > (size_t)B * c.roi_cap) return MV3D_ERR_WORKSPACE;
        o_bev[b] = S ? u.rois_bev + row * 5 : nullptr;
        o_rgb[b] = S ? u.rois_rgb + row * 5 : nullptr;
        o_fv[b] = (S && u.rois_fv) ? u.rois_fv + row * 5 : nullptr;
        o_3d[b] = S ? u.rois_3d + row * 7 : nullptr;
        o_lab[b] = S ? u.labels + row : nullptr;
        o_tg[b] = S ? u.bbox_targets + row * 24 * c.num_classes : nullptr;
        cal[b] = s.calib + (size_t)b * 48;
        row += (size_t)S;
    }
    return mv3d_proposal_target_stage2_batch_devn(B, s.p_bv, s.p_3d, s.p_cap, s.p_num, s.gt_bv, s.gt_3d, s.gt_c, s.G, cal, tp->tpar, lst[3],
                                                  cnt[3], lst[4], cnt[4], o_bev, o_rgb, o_lab, o_tg, o_3d, u.rois_fv ? o_fv : nullptr, s.p_ws,
                                                  s.p_wsz, s.stream);
}

// (the C-ABI promises "no exceptions": an allocation failure of the spill buffer becomes a status)
int host_stage_noexcept(mv3d_train_path *tp, Slot &s)
{
    try {
        return host_stage(tp, s);
    } catch (...) {
        return MV3D_ERR_WORKSPACE;
    }
}

void helper_main(mv3d_train_path *tp)
{
    (void)hipSetDevice(tp->device);
    for (;;) {
        int k;
        {
            std::unique_lock<std::mutex> g(tp->m);
            tp->cv_work.wait(g, [&] { return tp->stop || !tp->queue.empty(); });
            if (tp->queue.empty()) return;                            // (stop: drained)
            k = tp->queue.front();
            tp->queue.pop_front();
        }
        Slot &s = tp->slots[k];
        const int rc = host_stage_noexcept(tp, s);
        {
            std::lock_guard<std::mutex> g(tp->m);
            s.rc = rc;
            s.state = TP_STAGED;
        }
        tp->cv_done.notify_all();
    }
}

}  // namespace

extern "C" int mv3d_train_path_create(const mv3d_train_path_config *config, int depth, const mv3d_train_path_slot *slots, int helper_thread,
                                      mv3d_train_path **out)
{
    if (!config || !slots || !out || depth < 1 || depth > 64 || !config_ok(*config)) return MV3D_ERR_INVALID_ARG;
    for (int k = 0; k < depth; ++k)
        if (!slot_ok(*config, slots[k])) return MV3D_ERR_INVALID_ARG;
    mv3d_train_path *tp = new (std::nothrow) mv3d_train_path();
    if (!tp) return MV3D_ERR_HIP;
    set_params(tp, *config);
    tp->depth = depth;
    if (hipGetDevice(&tp->device) != hipSuccess) { delete tp; return MV3D_ERR_HIP; }
    try {
        tp->slots.resize((size_t)depth);
    } catch (...) {
        delete tp;
        return MV3D_ERR_HIP;
    }
    for (int k = 0; k < depth; ++k) {
        Slot &s = tp->slots[k];
        s.buf = slots[k];
        slot_tables(tp->cfg, s);
        // the pinned report buffers as the device addresses them (one launch writes the reports); anything unmapped or unaligned: copies
        {
            void *d0 = nullptr, *d1 = nullptr, *d2 = nullptr;
            const bool mapped = hipHostGetDevicePointer(&d0, s.buf.h_report, 0) == hipSuccess && hipHostGetDevicePointer(&d1, s.buf.h_pt_counts, 0) == hipSuccess &&
                                hipHostGetDevicePointer(&d2, s.buf.h_num_proposals, 0) == hipSuccess && d0 && d1 && d2;
            (void)hipGetLastError();                                   // (a buffer that is not pinned leaves an error behind: not ours to report)
            if (mapped && s.buf.h_report_row % 16 == 0 && (uintptr_t)d0 % 16 == 0 && (uintptr_t)s.buf.report % 16 == 0 && !tp_force_copies()) {
                s.dv_report = (uint8_t *)d0; s.dv_pt_counts = (int32_t *)d1; s.dv_num_proposals = (int32_t *)d2;
            }
        }
        // (blocking sync: the helper sleeps while it waits for a batch's reports instead of spinning on a core -- one process per GPU
        // on a node shares the host's cores with seven others; with several batches in flight the wake-up latency is hidden)
        if (hipEventCreateWithFlags(&s.ev, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) {
            for (int j = 0; j < k; ++j) (void)hipEventDestroy(tp->slots[j].ev);
            delete tp;
            return MV3D_ERR_HIP;
        }
    }
    tp->threaded = helper_thread != 0;
    if (tp->threaded) {
        try {
            tp->helper = std::thread(helper_main, tp);
        } catch (...) {                                               // (no thread to be had: refuse rather than run without the helper)
            for (Slot &s : tp->slots) (void)hipEventDestroy(s.ev);
            delete tp;
            return MV3D_ERR_HIP;
        }
    }
    *out = tp;
    return MV3D_OK;
}

extern "C" int mv3d_train_path_configure(mv3d_train_path *tp, const mv3d_train_path_config *c)
{
    if (!tp || !c || !config_ok(*c)) return MV3D_ERR_INVALID_ARG;
    const mv3d_train_path_config &o = tp->cfg;
    if (c->batch != o.batch || c->H != o.H || c->W != o.W || c->num_classes != o.num_classes || c->proposal_cap != o.proposal_cap ||
        c->anchor_cap != o.anchor_cap || c->roi_cap != o.roi_cap || c->max_gt != o.max_gt)
        return MV3D_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(tp->m);
    for (const Slot &s : tp->slots)
        if (s.state != TP_IDLE) return MV3D_ERR_INVALID_ARG;
    set_params(tp, *c);
    return MV3D_OK;
}

extern "C" int mv3d_train_path_submit(mv3d_train_path *tp, int slot, const float *prob, const float *pred, const float *im_info,
                                      const float *calib, const float *const *gt_bv, const float *const *gt_3d,
                                      const float *const *gt_corners, const int *G, void *mt19937_state, void *stream)
{
    if (!tp || slot < 0 || slot >= tp->depth || !prob || !pred || !im_info || !calib || !gt_bv || !gt_3d || !gt_corners || !G ||
        !mt19937_state)
        return MV3D_ERR_INVALID_ARG;
    const mv3d_train_path_config &c = tp->cfg;
    const int B = c.batch;
    for (int b = 0; b < B; ++b)
        if (G[b] < 1 || G[b] > c.max_gt || !gt_bv[b] || !gt_3d[b] || !gt_corners[b]) return MV3D_ERR_INVALID_ARG;
    Slot &s = tp->slots[slot];
    {
        std::lock_guard<std::mutex> g(tp->m);
        if (s.state != TP_IDLE) return MV3D_ERR_INVALID_ARG;
    }
    const mv3d_train_path_slot &u = s.buf;
    s.stream = (hipStream_t)stream;
    s.mt = mt19937_state;
    s.calib = calib;
    for (int b = 0; b < B; ++b) { s.gt_bv[b] = gt_bv[b]; s.gt_3d[b] = gt_3d[b]; s.gt_c[b] = gt_corners[b]; s.G[b] = G[b]; }
    int rc = mv3d_proposal_3d(prob, pred, B, c.H, c.W, im_info, calib, &c.proposal, u.blob_bv, u.blob_img, u.blob_3d, u.num_proposals,
                              u.num_proposals + B, u.proposal_ws, u.proposal_ws_bytes, stream);
    if (rc != MV3D_OK) return rc;
    rc = mv3d_anchor_target_stage1_batch(B, c.H, c.W, im_info, s.gt_bv, s.gt_3d, s.G, &c.anchor, u.rpn_labels, u.rpn_targets, s.a_cnt,
                                         s.a_fgh, s.a_ws, u.anchor_ws_bytes, stream);
    if (rc != MV3D_OK) return rc;
    rc = mv3d_proposal_target_stage1_batch_devn(B, s.p_bv, s.p_3d, s.p_cap, s.p_num, s.gt_bv, s.gt_3d, s.G, tp->tpar, s.p_cnt, s.p_ws,
                                                s.p_wsz, stream);
    if (rc != MV3D_OK) return rc;
    // the reports: one launch that writes them into the pinned host buffers (three device-to-host copies when those are not mapped)
    if (s.dv_report) {
        hipLaunchKernelGGL(tp_report_kernel, dim3(1), dim3(256), 0, s.stream, (const uint8_t *)u.report, (size_t)u.report_row, s.dv_report,
                           (size_t)u.h_report_row, B, (const int32_t *)u.pt_counts, s.dv_pt_counts, (const int32_t *)u.num_proposals,
                           s.dv_num_proposals);
        if (mv3d_launch_status() != MV3D_OK) return MV3D_ERR_HIP;
    } else {
        TP_HIP(hipMemcpy2DAsync(u.h_report, u.h_report_row, u.report, u.report_row, u.h_report_row, (size_t)B, hipMemcpyDeviceToHost, s.stream));
        TP_HIP(hipMemcpyAsync(u.h_pt_counts, u.pt_counts, (size_t)B * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
        TP_HIP(hipMemcpyAsync(u.h_num_proposals, u.num_proposals, (size_t)B * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
    }
    TP_HIP(hipEventRecord(s.ev, s.stream));
    {
        std::lock_guard<std::mutex> g(tp->m);
        s.state = TP_SUBMITTED;
        s.rc = MV3D_OK;
        if (tp->threaded) {
            try {
                tp->queue.push_back(slot);
            } catch (...) {
                s.state = TP_IDLE;
                return MV3D_ERR_HIP;
            }
        }
    }
    if (tp->threaded) tp->cv_work.notify_one();
    return MV3D_OK;
}

extern "C" int mv3d_train_path_finish(mv3d_train_path *tp, int slot, int32_t *rows_out, int32_t *num_proposals_out, int32_t *sizes_out)
{
    if (!tp || slot < 0 || slot >= tp->depth || !rows_out || !num_proposals_out) return MV3D_ERR_INVALID_ARG;
    Slot &s = tp->slots[slot];
    int rc;
    if (tp->threaded) {
        std::unique_lock<std::mutex> g(tp->m);
        if (s.state == TP_IDLE) return MV3D_ERR_INVALID_ARG;
        tp->cv_done.wait(g, [&] { return s.state == TP_STAGED; });
        rc = s.rc;
    } else {
        {
            std::lock_guard<std::mutex> g(tp->m);
            if (s.state != TP_SUBMITTED) return MV3D_ERR_INVALID_ARG;
        }
        rc = host_stage_noexcept(tp, s);
    }
    const int B = tp->cfg.batch;
    if (rc == MV3D_OK) {
        for (int b = 0; b < B; ++b) { rows_out[b] = s.rows[b]; num_proposals_out[b] = s.buf.h_num_proposals[b]; }
        if (sizes_out)
            for (int k = 0; k < 5 * B; ++k) sizes_out[k] = s.sizes[k];
    }
    std::lock_guard<std::mutex> g(tp->m);
    s.state = TP_IDLE;
    return rc;
}

extern "C" int mv3d_train_path_host_seconds(mv3d_train_path *tp, double *wait_s, double *draw_s)
{
    if (!tp) return MV3D_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(tp->m);
    if (wait_s) *wait_s = tp->t_wait;
    if (draw_s) *draw_s = tp->t_draw;
    tp->t_wait = tp->t_draw = 0.0;
    return MV3D_OK;
}

extern "C" void mv3d_train_path_destroy(mv3d_train_path *tp)
{
    if (!tp) return;
    if (tp->threaded) {
        {
            std::lock_guard<std::mutex> g(tp->m);
            tp->stop = true;                                          // (the helper drains its queue first)
        }
        tp->cv_work.notify_all();
        tp->helper.join();
    }
    for (Slot &s : tp->slots)
        if (s.ev) (void)hipEventDestroy(s.ev);
    delete tp;
}
