// RoiPool / RoiPoolGrad for gfx950, NHWC f32.
// Replaces lib/roi_pooling_layer/roi_pooling_op_gpu.cu.cc:20-110 (forward) / :113-215
// (backward) and the CPU kernels roi_pooling_op.cc:74-190 / :319-452; same arithmetic.
//
// Both kernels are HBM-bound streaming kernels:
//  forward   one work item = (roi, ph, pw, 4 consecutive channels); 16-byte loads/stores,
//            channel-fastest so a wave touches 1 KiB contiguous per pixel of the bin; the
//            bin bounds are uniform per wave (C/4 is a multiple of 64 for C = 512) so the
//            bin loops do not diverge.  Output (top + argmax, R*PH*PW*C*8 B) dominates.
//  backward  the reference's deterministic gather, kept bit-identical (ROIs ascending, then
//            ph, pw ascending, f32 adds in that order), but instead of every input element
//            scanning all R ROIs (O(H*W*C*R)), the rounded ROI geometry is tabled once per
//            workgroup in LDS and every wave owns whole input pixels: 64 ROIs per ballot,
//            then only the containing ROIs' candidate bins, 16-byte loads over the channels.
//            No atomics, no dependence on scheduling.
#include <float.h>
#include <stdlib.h>
#include <math.h>
#include "common.h"
#include "kernels.h"
#include "roi_geom.h"
#include "roi_grad_plan.h"

template <int VEC>
struct VecT;
template <> struct VecT<4> { typedef float4 F; typedef int4 I; };
template <> struct VecT<1> { typedef float F; typedef int I; };

__device__ __forceinline__ void upd(float v, int idx, float &mv, int &mi)
{
    if (v > mv) { mv = v; mi = idx; }          // strict >: first maximum wins, NaN never wins
}

__device__ __forceinline__ void upd4(const float4 x, int idx, float4 &mv, int4 &mi)
{
    upd(x.x, idx + 0, mv.x, mi.x);
    upd(x.y, idx + 1, mv.y, mi.y);
    upd(x.z, idx + 2, mv.z, mi.z);
    upd(x.w, idx + 3, mv.w, mi.w);
}

// items = R*PH*PW*(C/VEC), grid-stride
template <int VEC>
__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(const float *__restrict__ data, float scale, int B, int R,
                                                           int H, int W, int C, int PH, int PW,
                                                           const float *__restrict__ rois, float *__restrict__ top,
                                                           int *__restrict__ argmax)
{
    const int CV = C / VEC;
    const long long items = (long long)R * PH * PW * CV;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(it % CV);
        long long t = it / CV;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH);
        const int n = (int)(t / PH);
        const float *roi = rois + 5 * n;
        const int bi = (int)roi[0];
        const RoiGeom g = roi_geom(roi, scale);
        const int rw = max(g.rew - g.rsw + 1, 1), rh = max(g.reh - g.rsh + 1, 1);   // :146-147
        const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;        // :148-151
        int hs = (int)floorf(__fmul_rn((float)ph, bh)), ws = (int)floorf(__fmul_rn((float)pw, bw));
        int he = (int)ceilf(__fmul_rn((float)(ph + 1), bh)), we = (int)ceilf(__fmul_rn((float)(pw + 1), bw));
        hs = min(max(hs + g.rsh, 0), H); he = min(max(he + g.rsh, 0), H);          // :159-162
        ws = min(max(ws + g.rsw, 0), W); we = min(max(we + g.rsw, 0), W);
        const bool bad_batch = (bi < 0 || bi >= B);
        const bool empty = (he <= hs) || (we <= ws) || bad_batch;
        const int c0 = cv * VEC;
        float mv[VEC];
        int mi[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { mv[v] = empty ? 0.0f : -FLT_MAX; mi[v] = -1; }
        if (!empty) {
            const float *d = data + (long long)bi * H * W * C;
            for (int h = hs; h < he; ++h)
                for (int w = ws; w < we; ++w) {
                    const int idx = (h * W + w) * C + c0;
                    if (VEC == 4) {
                        const float4 x = *reinterpret_cast<const float4 *>(d + idx);
                        upd(x.x, idx + 0, mv[0], mi[0]);
                        upd(x.y, idx + 1, mv[1 % VEC], mi[1 % VEC]);
                        upd(x.z, idx + 2, mv[2 % VEC], mi[2 % VEC]);
                        upd(x.w, idx + 3, mv[3 % VEC], mi[3 % VEC]);
                    } else {
                        upd(d[idx], idx, mv[0], mi[0]);
                    }
                }
        }
        const long long o = (((long long)n * PH + ph) * PW + pw) * C + c0;
        if (VEC == 4) {
            *reinterpret_cast<float4 *>(top + o) = make_float4(mv[0], mv[1 % VEC], mv[2 % VEC], mv[3 % VEC]);
            if (argmax) *reinterpret_cast<int4 *>(argmax + o) = make_int4(mi[0], mi[1 % VEC], mi[2 % VEC], mi[3 % VEC]);
        } else {
            top[o] = mv[0];
            if (argmax) argmax[o] = mi[0];
        }
    }
}

// Fast forward for C/4 in {64,128,256} (C = 256, 512, 1024), XCD-aware.
//
// The 8 XCDs have private 4 MiB L2s and workgroup b is placed on XCD b % 8 (observed; only speed
// depends on it).  A feature map (BEV 11.8 MB, RGB 14.6 MB) does not fit one L2, and with ROIs
// spread over all XCDs every L2 would stream the whole map.  So the CHANNELS are split instead:
// workgroup b handles channel slice b % 8 (C/8 channels = 256 B per pixel for C = 512) of bin
// group b / 8 -- each XCD then only ever touches its own 1/8 of the map (<= 1.8 MB, L2 resident)
// while the overlapping bins of neighbouring ROIs re-read it.  The bin geometry (f32 divides,
// round / floor / ceil) is computed once per bin by the first lanes and broadcast through LDS.
#define LDS_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local")
struct BinGeom { int hs, he, ws, we; int base; int pad0, pad1, pad2; };   // base < 0: empty / bad batch index

__device__ __forceinline__ void upd4pos(const float4 x, int pos, float4 &mv, int4 &mi)
{
    upd(x.x, pos, mv.x, mi.x);
    upd(x.y, pos, mv.y, mi.y);
    upd(x.z, pos, mv.z, mi.z);
    upd(x.w, pos, mv.w, mi.w);
}

// NOARG (inference: argmax_data = NULL in every view): the maximum alone, same strict > (first maximum, NaN never wins): half the
// vector-ALU work of the scan and no index registers
__device__ __forceinline__ void upd4max(const float4 x, float4 &mv)
{
    mv.x = x.x > mv.x ? x.x : mv.x; mv.y = x.y > mv.y ? x.y : mv.y; mv.z = x.z > mv.z ? x.z : mv.z; mv.w = x.w > mv.w ? x.w : mv.w;
}

// COMPACT (the pair, mv3d_roi_pool_forward_views_pair): the argmax plane is PRIVATE to the pair -- only its own RoiPoolGrad reads
// it -- and holds, instead of the reference's int32 flat indices, the position of the first maximum in the bin's scan order,
// (h - hstart) * (wend - wstart) + (w - wstart):
//   bins of <= 255 pixels (every bin of a ROI smaller than ~100 x 100 feature pixels): ONE BYTE per pooled value, 0xFF = "none"
//       (the reference's -1), in the byte plane at the front of the caller's argmax buffer, [0, N), N = R * PH * PW * C;
//   larger bins (ROIs far larger than the map): 16-bit codes, 0xFFFF = none, in the escape plane behind it, bytes [N, 3 N).
// Which plane a bin uses follows from its geometry alone, so the gradient's index knows it (RPC_BIG in a list entry).  3 / 8 of the
// pair's record bytes (8 -> 5 B per pooled value) are neither written here nor read by the gradient; the index carries the code each
// candidate pixel has inside each of its bins, so the test `argmax == this pixel` (roi_pooling_op.cc:433) is one compare.
// mv3d_roi_pool_argmax_decode gives the reference's plane back (tests, bench).
#define RPC_BIG 0x10000              // list entry: the bin has more than 255 pixels -- its codes are in the 16-bit escape plane
// the rectangle of bin (ph, pw) of the ROI row rr[5] as roi_pooling_op.cc:139-162 computes it
__device__ __forceinline__ BinGeom fwd_bin_rect(const float rr[5], const int ph, const int pw, const float scale, const int B, const int H, const int W,
                                                const int PH, const int PW)
{
    BinGeom g;
    g.pad0 = g.pad1 = g.pad2 = 0;
    const int bi = (int)rr[0];
    const RoiGeom q = roi_geom(rr, scale);
    const int rw = max(q.rew - q.rsw + 1, 1), rh = max(q.reh - q.rsh + 1, 1);   // roi_pooling_op.cc:146-147
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;        // :148-151
    int hs = (int)floorf(__fmul_rn((float)ph, bh)), ws = (int)floorf(__fmul_rn((float)pw, bw));
    int he = (int)ceilf(__fmul_rn((float)(ph + 1), bh)), we = (int)ceilf(__fmul_rn((float)(pw + 1), bw));
    g.hs = min(max(hs + q.rsh, 0), H); g.he = min(max(he + q.rsh, 0), H);      // :159-162
    g.ws = min(max(ws + q.rsw, 0), W); g.we = min(max(we + q.rsw, 0), W);
    const bool empty = (g.he <= g.hs) || (g.we <= g.ws) || bi < 0 || bi >= B;
    g.base = empty ? -1 : bi;
    return g;
}
// the rectangle of pooled bin `bin` (= (roi n, ph, pw) flattened)
__device__ __forceinline__ BinGeom fwd_bin_geom(const long long bin, const long long nbins, const float *__restrict__ rois, const float scale,
                                                const int B, const int H, const int W, const int PH, const int PW)
{
    BinGeom g;
    g.hs = g.he = g.ws = g.we = 0; g.base = -1; g.pad0 = g.pad1 = g.pad2 = 0;
    if (bin < nbins) {
        // 32-bit divisions (the launcher guarantees R*PH*PW < 2^31): a 64-bit divide is ~100 instructions,
        // three of them in front of the barrier were the long pole of every workgroup
        const unsigned ub = (unsigned)bin, upw = (unsigned)PW, uph = (unsigned)PH;
        const unsigned t = ub / upw;
        const int pw = (int)(ub - t * upw);
        const unsigned n = t / uph;
        const int ph = (int)(t - n * uph);
        const float *roi = rois + 5 * (long long)n;
        float rr[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) rr[u] = roi[u];
        g = fwd_bin_rect(rr, ph, pw, scale, B, H, W, PH, PW);
    }
    return g;
}

template <int FWD_PASSES, bool COMPACT, bool NOARG, int TOPT>
__device__ __forceinline__ void roi_pool_fwd_xcd_pool(const BinGeom *s_g, const unsigned block, const float *__restrict__ data,
                                                       int B, int R, int H, int W, int C, int PH, int PW, float *__restrict__ top,
                                                       int *__restrict__ argmax, int tpb_shift);

// (Round 5, measured and dropped: the rectangles computed per thread in registers -- no LDS, no barrier, the 8 - 32 lanes of a bin
// redundantly: 42 - 46 us instead of 35, profiles/r05_fwd_local_ab.txt; a persistent grid that requests the next group's ROI rows
// while it pools the current one: 39 - 41 us, profiles/r05_v_fwd_persist_ab.txt, tools/experiments/roi_pool_fwd_persistent_r05.hip.txt.)
// TOPT (with NOARG, the serving graph in 16-bit mode): the pooled maximum stored as f16 (1) / bf16 (2) -- the values a cast of the f32
// output would give, without the f32 output and the cast launch
template <int FWD_PASSES, bool COMPACT = false, bool NOARG = false, int TOPT = 0>
__device__ __forceinline__ void roi_pool_fwd_xcd_block(BinGeom *s_g /* LDS, FWD_PASSES * 32 entries */, const unsigned block,
                                                        const float *__restrict__ data, float scale,
                                                        int B, int R, int H, int W, int C, int PH, int PW,
                                                        const float *__restrict__ rois, float *__restrict__ top,
                                                        int *__restrict__ argmax, int tpb_shift)
{
    const int bpp = 256 >> tpb_shift;                // bins per pass (32, 16 or 8): C/32 = 8, 16 or 32 float4 lanes per bin
    const long long nbins = (long long)R * PH * PW;
    const long long bin0 = (long long)(block >> 3) * (FWD_PASSES * bpp);
    if (threadIdx.x < FWD_PASSES * bpp) s_g[threadIdx.x] = fwd_bin_geom(bin0 + threadIdx.x, nbins, rois, scale, B, H, W, PH, PW);
    __syncthreads();
    roi_pool_fwd_xcd_pool<FWD_PASSES, COMPACT, NOARG, TOPT>(s_g, block, data, B, R, H, W, C, PH, PW, top, argmax, tpb_shift);
}

// the pooling phase of roi_pool_fwd_xcd_block: the workgroup's bins from their rectangles in LDS
template <int FWD_PASSES, bool COMPACT, bool NOARG, int TOPT>
__device__ __forceinline__ void roi_pool_fwd_xcd_pool(const BinGeom *s_g, const unsigned block, const float *__restrict__ data,
                                                       int B, int R, int H, int W, int C, int PH, int PW, float *__restrict__ top,
                                                       int *__restrict__ argmax, int tpb_shift)
{
    const int tpb = 1 << tpb_shift, bpp = 256 >> tpb_shift, slice = block & 7;
    const long long nbins = (long long)R * PH * PW;
    const long long bin0 = (long long)(block >> 3) * (FWD_PASSES * bpp);
    const int sub = threadIdx.x >> tpb_shift;        // bin inside the pass
    const int c0 = (slice * tpb + (threadIdx.x & (tpb - 1))) * 4;
#pragma unroll
    for (int pass = 0; pass < FWD_PASSES; ++pass) {
        const int lb = pass * bpp + sub;
        const long long bin = bin0 + lb;
        if (bin < nbins) {
            const BinGeom g = s_g[lb];
            float4 mv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int4 mi = make_int4(-1, -1, -1, -1);
            if (g.base >= 0) {
                mv = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
                const float *d = data + (long long)g.base * H * W * C + c0;
                int pos = 0;                         // (COMPACT) position in the bin's scan order
                // (Measured and dropped in round 5, profiles/r05_u_fwd_clamped_rows_probe.txt: row steps of four with the offsets clamped
                // to the row's last pixel -- no one-pixel tail loop, one round trip per row of a 1 - 4 pixel wide bin.  36.3 vs 35.4 us
                // here, 708 vs 658 us on the TEST-cfg batch: at eight waves per SIMD the tail's round trips are hidden, the compares of
                // the re-read pixels are not -- the kernel is bound by vector-ALU issue and stores, not by load latency.)
                for (int h = g.hs; h < g.he; ++h) {
                    int w = g.ws;
                    // four independent 16-byte loads in flight, consumed in scan order (first max wins)
                    for (; w + 4 <= g.we; w += 4) {
                        const int idx = (h * W + w) * C;
                        const float4 x0 = *reinterpret_cast<const float4 *>(d + idx);
                        const float4 x1 = *reinterpret_cast<const float4 *>(d + idx + C);
                        const float4 x2 = *reinterpret_cast<const float4 *>(d + idx + 2 * C);
                        const float4 x3 = *reinterpret_cast<const float4 *>(d + idx + 3 * C);
                        if (NOARG) {
                            upd4max(x0, mv); upd4max(x1, mv); upd4max(x2, mv); upd4max(x3, mv);
                        } else if (COMPACT) {
                            upd4pos(x0, pos, mv, mi); upd4pos(x1, pos + 1, mv, mi); upd4pos(x2, pos + 2, mv, mi); upd4pos(x3, pos + 3, mv, mi);
                            pos += 4;
                        } else {
                            upd4(x0, idx + c0, mv, mi);
                            upd4(x1, idx + C + c0, mv, mi);
                            upd4(x2, idx + 2 * C + c0, mv, mi);
                            upd4(x3, idx + 3 * C + c0, mv, mi);
                        }
                    }
                    for (; w < g.we; ++w) {
                        const int idx = (h * W + w) * C;
                        if (NOARG) upd4max(*reinterpret_cast<const float4 *>(d + idx), mv);
                        else if (COMPACT) upd4pos(*reinterpret_cast<const float4 *>(d + idx), pos++, mv, mi);
                        else upd4(*reinterpret_cast<const float4 *>(d + idx), idx + c0, mv, mi);
                    }
                }
            }
            const long long o = bin * C + c0;
            // streaming stores: the outputs are written once and never re-read here, they must not push the
            // XCD's slice of the feature map out of its L2
            typedef float f4v __attribute__((ext_vector_type(4)));
            typedef int i4v __attribute__((ext_vector_type(4)));
            const f4v mvv = {mv.x, mv.y, mv.z, mv.w};
            if (TOPT == 1) {
                typedef _Float16 h4v __attribute__((ext_vector_type(4)));
                const h4v hv = {(_Float16)mv.x, (_Float16)mv.y, (_Float16)mv.z, (_Float16)mv.w};
                __builtin_nontemporal_store(hv, reinterpret_cast<h4v *>(reinterpret_cast<_Float16 *>(top) + o));
            } else if (TOPT == 2) {
                typedef __bf16 b4v __attribute__((ext_vector_type(4)));
                const b4v hv = {(__bf16)mv.x, (__bf16)mv.y, (__bf16)mv.z, (__bf16)mv.w};
                __builtin_nontemporal_store(hv, reinterpret_cast<b4v *>(reinterpret_cast<__bf16 *>(top) + o));
            } else
                __builtin_nontemporal_store(mvv, reinterpret_cast<f4v *>(top + o));
            if (NOARG) {
            } else if (COMPACT) {
                unsigned char *const plane8 = reinterpret_cast<unsigned char *>(argmax);
                if (g.base < 0 || (g.he - g.hs) * (g.we - g.ws) <= 255) {      // four one-byte codes (-1 -> 0xFF), 4 bytes per lane
                    const unsigned cv = ((unsigned)mi.x & 0xffu) | (((unsigned)mi.y & 0xffu) << 8) | (((unsigned)mi.z & 0xffu) << 16) | ((unsigned)mi.w << 24);
                    __builtin_nontemporal_store(cv, reinterpret_cast<unsigned *>(plane8 + o));
                } else {                             // a bin of more than 255 pixels: four 16-bit codes in the escape plane
                    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
                    const u2v cv = {((unsigned)mi.x & 0xffffu) | ((unsigned)mi.y << 16), ((unsigned)mi.z & 0xffffu) | ((unsigned)mi.w << 16)};
                    __builtin_nontemporal_store(cv, reinterpret_cast<u2v *>(reinterpret_cast<unsigned short *>(plane8 + nbins * C) + o));
                }
            } else {
                const i4v miv = {mi.x, mi.y, mi.z, mi.w};
                if (argmax) __builtin_nontemporal_store(miv, reinterpret_cast<i4v *>(argmax + o));
            }
        }
    }
}

template <int FWD_PASSES>
__global__ __launch_bounds__(256) void roi_pool_fwd_xcd_kernel(const float *__restrict__ data, float scale, int B, int R,
                                                               int H, int W, int C, int PH, int PW,
                                                               const float *__restrict__ rois, float *__restrict__ top,
                                                               int *__restrict__ argmax, int tpb_shift)
{
    __shared__ BinGeom s_g[FWD_PASSES * 32];
    roi_pool_fwd_xcd_block<FWD_PASSES>(s_g, blockIdx.x, data, scale, B, R, H, W, C, PH, PW, rois, top, argmax, tpb_shift);
}

// Several views (the BEV and RGB maps of one step) in ONE launch: the second view's workgroups fill the
// machine while the first view drains, and one kernel boundary disappears.
struct RoiViewDev {
    const float *data, *rois;
    float *top;
    int *argmax;
    float scale;
    int B, R, H, W, C, tpb_shift;
    unsigned first_block;        // first workgroup of this view
};
struct RoiViewPack { RoiViewDev v[MV3D_MAX_ROI_VIEWS]; int n, PH, PW; };

template <int FWD_PASSES, bool NOARG = false, int TOPT = 0>
__global__ __launch_bounds__(256) void roi_pool_fwd_xcd_multi_kernel(RoiViewPack p)
{
    __shared__ BinGeom s_g[FWD_PASSES * 32];
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && blockIdx.x >= p.v[j].first_block) k = j;
    const RoiViewDev &v = p.v[k];
    roi_pool_fwd_xcd_block<FWD_PASSES, false, NOARG, TOPT>(s_g, blockIdx.x - v.first_block, v.data, v.scale, v.B, v.R, v.H, v.W, v.C, p.PH, p.PW, v.rois,
                                                     v.top, v.argmax, v.tpb_shift);
}

// Forward on maps that are NOT cache-resident (mv3d_roi_pool_forward_views_cold).  The forward reads a map in 256-B
// pieces (one XCD slice of a pixel), every workgroup behind a dependent round trip; when those pieces come from HBM the
// round trip is 2-3 us and the kernel runs at half its speed (58 vs 31 us on the training batch).  The cold variant puts
// prefetch workgroups at the FRONT of the same grid (dispatched first): each owns 16 pixels of a map row, marks the pixels
// some ROI's rectangle covers (one ROI per thread, LDS mask) and streams the marked pixels -- whole pixels, C floats
// contiguous -- throwing the values away, so that the ~20 % of a map the pooling workgroups are going to touch sit in
// the memory-side cache (which every XCD reads) when those get there: 40 us for both together.  Same results.
#define PF_PIX 16
struct RoiPrefetchPack { unsigned first_block[MV3D_MAX_ROI_VIEWS]; unsigned blocks; };
__device__ __forceinline__ void roi_prefetch_block(unsigned *s_mask_p /* LDS, one word */, const RoiViewPack &p, const RoiPrefetchPack &pf,
                                                   const unsigned block, int *sink)
{
    unsigned &s_mask = *s_mask_p;
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && block >= pf.first_block[j]) k = j;
    const RoiViewDev &v = p.v[k];
    const unsigned g = block - pf.first_block[k];
    const unsigned gpr = (unsigned)(v.W + PF_PIX - 1) / PF_PIX;
    const int w0 = (int)(g % gpr) * PF_PIX;
    const unsigned gh = g / gpr;
    const int h = (int)(gh % (unsigned)v.H), n = (int)(gh / (unsigned)v.H);
    if (threadIdx.x == 0) s_mask = 0u;
    __syncthreads();
    unsigned m = 0u;
    for (int r = threadIdx.x; r < v.R; r += 256) {
        const float *roi = v.rois + 5 * (long long)r;
        if ((int)roi[0] != n) continue;
        const RoiGeom q = roi_geom(roi, v.scale);
        // the bins of a ROI reach one pixel past its rounded end (ceil of the last bin): be generous by one
        if (h < q.rsh || h > q.reh + 1) continue;
        const int a = max(q.rsw, w0), b = min(q.rew + 1, w0 + PF_PIX - 1);
        if (a <= b) m |= ((2u << (b - w0)) - 1u) & ~((1u << (a - w0)) - 1u);
    }
    if (m) atomicOr(&s_mask, m);
    __syncthreads();
    const unsigned mask = s_mask;
    if (!mask) return;
    const int c4 = v.C / 4;
    const float4 *row = reinterpret_cast<const float4 *>(v.data + (((long long)n * v.H + h) * v.W + w0) * v.C);
    const int tot = min(PF_PIX, v.W - w0) * c4;
    float acc = 0.0f;
    for (int t = threadIdx.x; t < tot; t += 256)
        if ((mask >> (t / c4)) & 1u) acc += row[t].x;
    if (acc == 1.2345678e-30f && sink) sink[0] = 1;               // keeps the loads alive
}

// forward with the prefetch workgroups at the front of the same grid (pf.blocks is a multiple of 8: the forward's
// workgroup -> XCD slice mapping is kept)
template <int FWD_PASSES, bool NOARG = false, int TOPT = 0>
__global__ __launch_bounds__(256) void roi_pool_fwd_xcd_multi_cold_kernel(RoiViewPack p, RoiPrefetchPack pf, int *sink)
{
    __shared__ BinGeom s_g[FWD_PASSES * 32];
    __shared__ unsigned s_mask;
    if (blockIdx.x < pf.blocks) { roi_prefetch_block(&s_mask, p, pf, blockIdx.x, sink); return; }
    const unsigned blk = blockIdx.x - pf.blocks;
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && blk >= p.v[j].first_block) k = j;
    const RoiViewDev &v = p.v[k];
    roi_pool_fwd_xcd_block<FWD_PASSES, false, NOARG, TOPT>(s_g, blk - v.first_block, v.data, v.scale, v.B, v.R, v.H, v.W, v.C, p.PH, p.PW, v.rois, v.top,
                                                     v.argmax, v.tpb_shift);
}

#define BWD_CHUNK 1024      // ROIs whose geometry is staged in LDS at a time
#define BWD_PIX 4           // input pixels per workgroup (1 per wave)
#define BWD_CAND 128        // candidate (roi, bin) records a wave collects before it drains them

// grid = ceil(B*H*W / BWD_PIX) workgroups of 256 threads.  The rounded ROI geometry (the part with
// the f32 multiplies / round()) is computed once per workgroup into LDS; every WAVE then owns one
// input pixel: it scans the table 64 ROIs per ballot (ascending) and, for every containing ROI, appends
// the offsets of the candidate bins (ph, pw ascending) to a per-wave list in LDS.  The list is drained
// four candidates at a time -- across ROI boundaries, so that every round of 16-byte loads is full and the
// kernel's critical path (the pixel under the most ROIs) is a quarter as many memory round trips as it
// has candidates -- and the adds happen in list order = the reference's summation order roi -> ph -> pw:
// the f32 sums are bit-identical.  No atomics, no barriers after the table is built, nothing depends on
// scheduling.
template <int VEC, int NACC>
__device__ __forceinline__ void bwd_drain(const long long *cand, const int ncand, const int want0, const int lane, const int CV,
                                          const float *__restrict__ top_diff, const int *__restrict__ argmax,
                                          float (&acc)[NACC][VEC])
{
    for (int t0 = 0; t0 < ncand; t0 += 4) {
        long long o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = cand[min(t0 + u, ncand - 1)];       // wave-uniform (LDS broadcast)
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            const int cv = lane + 64 * k;
            if (cv < CV) {
                const int c0 = cv * VEC;
                const int want = want0 + c0;
                if (VEC == 4) {
                    int4 am[4];
                    float4 td[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        am[u] = *reinterpret_cast<const int4 *>(argmax + o[u] + c0);
                        td[u] = *reinterpret_cast<const float4 *>(top_diff + o[u] + c0);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (t0 + u < ncand) {
                            if (am[u].x == want + 0) acc[k][0] += td[u].x;
                            if (am[u].y == want + 1) acc[k][1 % VEC] += td[u].y;
                            if (am[u].z == want + 2) acc[k][2 % VEC] += td[u].z;
                            if (am[u].w == want + 3) acc[k][3 % VEC] += td[u].w;
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (t0 + u < ncand && argmax[o[u] + c0] == want) acc[k][0] += top_diff[o[u] + c0];
                }
            }
        }
    }
}

template <int VEC, int NACC>
__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(const float *__restrict__ top_diff, float scale, int B, int R,
                                                           int H, int W, int C, int PH, int PW,
                                                           const float *__restrict__ rois, float *__restrict__ bottom_diff,
                                                           const int *__restrict__ argmax)
{
    __shared__ int4 s_geom[BWD_CHUNK];
    __shared__ int s_bi[BWD_CHUNK];
    __shared__ long long s_cand[BWD_PIX][BWD_CAND];
    const int CV = C / VEC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long npix = (long long)B * H * W;
    const long long pix = (long long)blockIdx.x * BWD_PIX + wave;
    const bool live = pix < npix;                             // wave-uniform
    const int w = (int)(pix % W), h = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
    const int want0 = (h * W + w) * C;
    long long *cand = s_cand[wave];
    float acc[NACC][VEC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[k][v] = 0.0f;

    for (int base = 0; base < R; base += BWD_CHUNK) {
        const int cnt = min(R - base, BWD_CHUNK);
        __syncthreads();                              // previous chunk fully consumed
        for (int r = threadIdx.x; r < cnt; r += blockDim.x) {
            const float *roi = rois + 5 * (base + r);
            const RoiGeom g = roi_geom(roi, scale);
            s_geom[r] = make_int4(g.rsw, g.rsh, g.rew, g.reh);
            s_bi[r] = (int)roi[0];
        }
        __syncthreads();
        if (!live) continue;
        int ncand = 0;
        for (int r0 = 0; r0 < cnt; r0 += 64) {
            const int r = r0 + lane;
            bool in = false;
            if (r < cnt) {
                const int4 g = s_geom[r];
                // roi_pooling_op.cc:392-403: batch match, containment on the unclamped rounded ROI
                in = (n == s_bi[r]) && (w >= g.x && w <= g.z && h >= g.y && h <= g.w);
            }
            unsigned long long bal = __ballot(in);
            while (bal) {                             // ascending ROI order
                const int rr = r0 + __builtin_ctzll(bal);
                bal &= bal - 1;
                const int4 g = s_geom[rr];
                const int rw = max(g.z - g.x + 1, 1), rh = max(g.w - g.y + 1, 1);
                const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
                // :423-426 (identical to the CUDA form roi_pooling_op_gpu.cu.cc:169-172)
                int phs = (int)floorf((float)(h - g.y) / bh), phe = (int)ceilf((float)(h - g.y + 1) / bh);
                int pws = (int)floorf((float)(w - g.x) / bw), pwe = (int)ceilf((float)(w - g.x + 1) / bw);
                phs = min(max(phs, 0), PH); phe = min(max(phe, 0), PH);
                pws = min(max(pws, 0), PW); pwe = min(max(pwe, 0), PW);
                const long long off = (long long)(base + rr) * PH * PW * C;
                const int nw = pwe - pws, nbins = (phe - phs) * nw;    // <= PH * PW candidates, (ph, pw) order
                for (int b0 = 0; b0 < nbins; b0 += 64) {
                    const int take = min(nbins - b0, 64);
                    if (ncand + take > BWD_CAND) {            // list full: drain it, in order
                        LDS_FENCE();
                        bwd_drain<VEC, NACC>(cand, ncand, want0, lane, CV, top_diff, argmax, acc);
                        LDS_FENCE();
                        ncand = 0;
                    }
                    const int t = b0 + lane;
                    if (lane < take) cand[ncand + lane] = off + ((long long)(phs + t / nw) * PW + (pws + t % nw)) * C;
                    ncand += take;
                }
            }
        }
        LDS_FENCE();
        bwd_drain<VEC, NACC>(cand, ncand, want0, lane, CV, top_diff, argmax, acc);
        LDS_FENCE();
    }
    if (!live) return;
    float *out = bottom_diff + pix * C;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const int cv = lane + 64 * k;
        if (cv < CV) {
            if (VEC == 4)
                *reinterpret_cast<float4 *>(out + cv * 4) = make_float4(acc[k][0], acc[k][1 % VEC], acc[k][2 % VEC], acc[k][3 % VEC]);
            else
                out[cv] = acc[k][0];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Fast RoiPoolGrad for C % 64 == 0 (C = 512: the MV3D maps), all views of a step in ONE launch.
//
// Same gather and the same f32 summation order as above (ROIs ascending, ph, pw ascending), re-shaped around what
// bounded the kernel above on the training workload (3 views x 2 frames x 128 sampled ROIs: 228 us, 0.11 of the HBM
// roofline):
//   * XCD / L2: a wave there owned all C channels of a pixel and neighbouring pixels ran on different XCDs, so every
//     (roi, bin) record was pulled over the fabric by up to 8 private L2s.  Here the CHANNELS are sliced like in the
//     forward: workgroup b works on channels [64 (b % nsl), +64) -- with nsl = 8 slice s only ever lives on XCD s, its
//     L2 sees 1/8 of every record, and the re-reads of a record (each is a candidate of 1.4 - 3.4 pixels on the
//     training workload) hit that L2.
//   * geometry: ~80 % of the pixels lie in no ROI at all and the rest in a handful.  A workgroup = one row segment of
//     16 (or 4) pixels x one slice first filters the ROIs by frame, row and column span (256 ROIs per pass, the f32
//     divides only for survivors) into an ordered LDS list, typically 0 - 30 entries; the pixels then only walk that.
//   * critical path: a wave handles ONE pixel at a time with one channel per lane -- per candidate two dword loads,
//     a compare, a select and an add, four candidates in flight -- so the pixel under the most bins (328 on the 8 x 64
//     front-view map) costs ~2.5 k instructions instead of ~8 k.
// Adding +0.0f for a non-matching channel is bit-neutral: the running sum starts at +0.0f and a sum that started at
// +0.0f is never -0.0f.
#define BW_THREADS 256
#define BW_CHUNK 256                 // ROIs filtered per pass
#define BW_CAND 512                  // candidate records a wave lists before it drains them
#define BW_MAXPXG 64                 // pixels of a row segment per workgroup, at most
struct RoiGradViewDev {
    const float *top_diff, *rois;
    const int *argmax;
    float *bottom_diff;
    float scale;
    int B, R, H, W, C;
    int pxg;                         // pixels per workgroup (a row segment), <= BW_MAXPXG
    int gpr;                         // segments per row = ceil(W / pxg)
    int nsl;                         // channel slices = C / 64
    unsigned first_block;
};
struct RoiGradPack { RoiGradViewDev v[MV3D_MAX_ROI_VIEWS]; int n, PH, PW; };

template <int U>
__device__ __forceinline__ float bwd_drain_records(const int *cand, const int t0, const int ncand, const int *__restrict__ am,
                                                   const float *__restrict__ td, const int C, const int want, float a)
{
    int am_v[U];
    float td_v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long o = (long long)__builtin_amdgcn_readfirstlane(cand[min(t0 + u, ncand - 1)]) * C;
        am_v[u] = am[o];
        td_v[u] = td[o];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (t0 + u < ncand) a += (am_v[u] == want) ? td_v[u] : 0.0f;
    return a;
}

__global__ __launch_bounds__(BW_THREADS) void roi_pool_bwd_sliced_kernel(RoiGradPack p)
{
    __shared__ int s_roi[BW_CHUNK], s_rsw[BW_CHUNK], s_rew[BW_CHUNK], s_prow[BW_CHUNK];
    __shared__ float s_bw[BW_CHUNK];
    __shared__ int s_wcnt[4];
    __shared__ int s_cand[4][BW_CAND];
    extern __shared__ float s_carry[];                   // [pxg][64] partial sums between ROI passes (only when R > BW_CHUNK)
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && blockIdx.x >= p.v[j].first_block) k = j;
    const RoiGradViewDev &v = p.v[k];
    const int PH = p.PH, PW = p.PW, H = v.H, W = v.W, C = v.C, R = v.R;
    const unsigned b = blockIdx.x - v.first_block;
    const int slice = (int)(b % (unsigned)v.nsl);
    const unsigned g = b / (unsigned)v.nsl;
    const int w0 = (int)(g % (unsigned)v.gpr) * v.pxg;
    const int npx = min(v.pxg, W - w0);                  // pixels of this segment
    const unsigned gh = g / (unsigned)v.gpr;
    const int h = (int)(gh % (unsigned)H), n = (int)(gh / (unsigned)H);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = slice * 64 + lane;
    const int *am = v.argmax + c;
    const float *td = v.top_diff + c;
    float *out = v.bottom_diff + (((long long)n * H + h) * W + w0) * C + c;
    int *cand = s_cand[wave];
    const int esub = max(1, min(64, BW_CAND / (PH * PW)));      // entries whose bins always fit the candidate list
    const int npass = max(1, (R + BW_CHUNK - 1) / BW_CHUNK);

    for (int pass = 0; pass < npass; ++pass) {
        // ---- pass part 1: ROIs of this frame that contain row h and touch the columns of the segment, in ROI order
        if (pass) __syncthreads();                         // the previous list is fully consumed
        const int r = pass * BW_CHUNK + (int)threadIdx.x;
        bool ok = false;
        int rsw = 0, rew = 0, prow = 0;
        float bw = 1.0f;
        if (r < R) {
            const float *roi = v.rois + 5 * (long long)r;
            const RoiGeom q = roi_geom(roi, v.scale);
            // roi_pooling_op.cc:392-403: batch match, containment on the unclamped rounded ROI
            ok = ((int)roi[0] == n) && h >= q.rsh && h <= q.reh && q.rew >= w0 && q.rsw < w0 + npx;
            if (ok) {
                const int rh = max(q.reh - q.rsh + 1, 1), rw = max(q.rew - q.rsw + 1, 1);
                const float bh = (float)rh / (float)PH;
                // :423-426 (identical to the CUDA form roi_pooling_op_gpu.cu.cc:169-172)
                int phs = (int)floorf((float)(h - q.rsh) / bh), phe = (int)ceilf((float)(h - q.rsh + 1) / bh);
                phs = min(max(phs, 0), PH); phe = min(max(phe, 0), PH);
                ok = phe > phs;
                rsw = q.rsw; rew = q.rew; prow = phs | (phe << 8);
                bw = (float)rw / (float)PW;
            }
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int pos = __popcll(bal & ((1ull << lane) - 1ull));
        int nlist = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int t = s_wcnt[q]; if (q < wave) pos += t; nlist += t; }
        if (ok) { s_roi[pos] = r; s_rsw[pos] = rsw; s_rew[pos] = rew; s_prow[pos] = prow; s_bw[pos] = bw; }
        __syncthreads();
        const bool last = (pass == npass - 1);
        // ---- part 2: pixel j of the segment belongs to wave j % 4 (neighbouring, similarly loaded pixels spread over the
        // waves); one pixel at a time, one channel per lane
        for (int j = wave; j < npx; j += 4) {
            const int w = w0 + j;
            const int want = (h * W + w) * C + c;
            float a = pass ? s_carry[j * 64 + lane] : 0.0f;
            for (int e0 = 0; e0 < nlist; e0 += 64) {
                // lane i: entry e0 + i -> its candidate bins for pixel (h, w)
                const int e = e0 + lane;
                int nb = 0, phs = 0, pws = 0, nw = 1, rec0 = 0;
                if (e < nlist) {
                    const int xs = s_rsw[e], xe = s_rew[e];
                    if (w >= xs && w <= xe) {
                        const float fbw = s_bw[e];
                        int x0 = (int)floorf((float)(w - xs) / fbw), x1 = (int)ceilf((float)(w - xs + 1) / fbw);
                        x0 = min(max(x0, 0), PW); x1 = min(max(x1, 0), PW);
                        if (x1 > x0) {
                            const int pr = s_prow[e];
                            phs = pr & 255; pws = x0; nw = x1 - x0;
                            nb = ((pr >> 8) - phs) * nw;
                            rec0 = s_roi[e] * PH * PW;
                        }
                    }
                }
                if (!__any(nb > 0)) continue;
                int tot = nb;                                     // inclusive prefix over the lanes
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { const int t = __shfl_up(tot, m); if (lane >= m) tot += t; }
                const int all = __builtin_amdgcn_readfirstlane(__shfl(tot, 63));
                // entries are taken in pieces whose candidates fit the list: all 64 at once when they do (the normal
                // case), `esub` at a time otherwise
                const int step = (all <= BW_CAND) ? 64 : esub;
                for (int s0 = 0; s0 < 64; s0 += step) {
                    const int before = (s0 == 0) ? 0 : __shfl(tot, s0 - 1);
                    const int upto = __shfl(tot, min(s0 + step, 64) - 1);
                    const int ncand = __builtin_amdgcn_readfirstlane(upto - before);
                    if (ncand == 0) continue;
                    if (lane >= s0 && lane < s0 + step && nb > 0) {
                        int o = tot - nb - before;                // exclusive prefix inside the piece
                        for (int ph = phs; ph < phs + nb / nw; ++ph)
                            for (int pw = pws; pw < pws + nw; ++pw) cand[o++] = rec0 + ph * PW + pw;
                    }
                    LDS_FENCE();
                    // drain in list order = reference order roi -> ph -> pw; 16 (then 4) records in flight: the walk of a
                    // pixel is a chain of dependent memory round trips, and the hottest pixel sets the kernel's time
                    int t0 = 0;
                    for (; t0 + 16 <= ncand; t0 += 16) a = bwd_drain_records<16>(cand, t0, ncand, am, td, C, want, a);
                    for (; t0 < ncand; t0 += 4) a = bwd_drain_records<4>(cand, t0, ncand, am, td, C, want, a);
                    LDS_FENCE();                                  // the list is reused by the next piece
                }
            }
            if (last) __builtin_nontemporal_store(a, out + (long long)j * C);
            else s_carry[j * 64 + lane] = a;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Indexed RoiPoolGrad (used when the caller passes a workspace): the same gather, split into
//   roi_bwd_index_kernel<false>, <true>
//                          ONCE per launch and per pixel (not per channel slice): which (roi, bin) records are candidates
//                          of pixel (n, h, w), in reference order roi -> ph -> pw.  A workgroup = 16 pixels of a row: it
//                          filters the ROIs (frame, row, column span) and evaluates the (pixel, roi) pairs in parallel
//                          (the f32 divides of roi_pooling_op.cc:423-426).  <false> zero-fills the pixels of bottom_diff
//                          (streaming stores, issued first) and writes the segment's sizes; <true> takes its slab offsets
//                          as plain sums over the preceding segments' sizes (the sizing launch is complete: no atomics,
//                          no look-back protocol, nothing that has to be zero on entry), writes the candidate lists (byte
//                          offsets of the records) and one item (pixel, offset, count, view) per pixel that has candidates
//                          -- ~80 % of the pixels have none and are finished by the zero fill.
//   roi_bwd_gather_kernel  a fixed grid; wave = (item, 64-channel slice): walks the item's list (record offset = scalar
//                          offset of a buffer load, 32 records in flight, the next item's header and offsets requested
//                          while this item's records fly), adds in list order and overwrites the pixel's slice.  Slice s
//                          of every record is only ever read by XCD s.  No LDS, no barrier.
// Why this shape (each point measured): the workgroup dispatcher hands out ~2.5 workgroups / ns (a kernel with
// one 4-wave workgroup per 4 pixels and slice, 54 k workgroups, takes 20 us to do NOTHING) -> a fixed gather grid over the
// compact item list; a same-address global atomic costs ~20 ns however many workgroups issue it (1 k slab allocations =
// 20 us) -> sizes and offsets in two launches instead; a wave that walks pixels one after the other behind LDS round trips
// pays every pixel's memory round trips in sequence (the sliced kernel above: 102 us on the training batch).
// Workspace: [0, 256) header {-, item count}, per-segment sizes (2 ints per 16-pixel segment), the item list (int4 per
// pixel of all views), the candidate pool.  Pool bound per ROI: the sum over the rows of (phe - phs) is <= PH + 2 rows + 3
// (floor / ceil slack), likewise over the columns.  Every word is written before it is read: no memset, no zero contract.
#define BWI_PIX 16
#define BWG_GROUPS 256                // gather grid = BWG_GROUPS x nsl workgroups of 4 waves (~ what the chip holds at once)
#define BWI_SPIN_LIMIT (1 << 16)      // polls of one published word before a look-back gives up (and flags the index invalid)
struct RoiGradIdxPack {
    long long *trace; int4 *items; int *pool; int *header;
    int *seg_tot, *seg_mask;                         // per 16-pixel segment: candidates, 16-bit mask of the pixels that have any
    unsigned first_block[MV3D_MAX_ROI_VIEWS]; int gpr[MV3D_MAX_ROI_VIEWS];
};

// LDS of one index workgroup
struct RoiIdxShared {
    int red[8];
    int roi[BW_CHUNK], rsw[BW_CHUNK], rew[BW_CHUNK], prow[BW_CHUNK];
    float bw[BW_CHUNK];
    unsigned char nb[BW_CHUNK][BWI_PIX], xr[BW_CHUNK][BWI_PIX];
    int wcnt[4], cnt[BWI_PIX], base[BWI_PIX], run[BWI_PIX];
    int part[16][BWI_PIX];
};

// One index workgroup = one 16-pixel segment of a map row.
// FILL = false: zero-fill + sizes (candidates of the segment, mask of its non-empty pixels); FILL = true: slab offsets from the
// sizes of the preceding segments (a plain sum: the sizing launch is complete), items and candidate lists.  No atomics on global
// memory, no state that has to be zero on entry.
template <bool FILL>
__device__ __forceinline__ void roi_bwd_index_block(RoiIdxShared &S, const RoiGradPack &p, const RoiGradIdxPack &ix, const unsigned block,
                                                    const unsigned nblocks)
{
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && block >= ix.first_block[j]) k = j;
    const RoiGradViewDev &v = p.v[k];
    const int PH = p.PH, PW = p.PW, H = v.H, W = v.W, R = v.R, C = v.C;
    const unsigned g = block - ix.first_block[k];
    const int gpr = ix.gpr[k];
    const int w0 = (int)(g % (unsigned)gpr) * BWI_PIX;
    const int npx = min(BWI_PIX, W - w0);
    const unsigned gh = g / (unsigned)gpr;
    const int h = (int)(gh % (unsigned)H), n = (int)(gh / (unsigned)H);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = threadIdx.x & (BWI_PIX - 1), q = threadIdx.x / BWI_PIX;      // pixel of the segment, entry stripe
    const int npass = (R + BW_CHUNK - 1) / BW_CHUNK;
    const long long pix0 = ((long long)n * H + h) * W + w0;
    if (FILL == ((h & 1) != 0)) {   // every pixel of the segment starts as zeros (the gather kernel overwrites the ones that have
        // candidates); even rows by the sizing launch, odd rows by the list launch: each launch is a chain of barriers and LDS
        // round trips with the memory system idle, so half of the 55 MB of streaming stores hides under each.  The
        // segment's npx * C floats are contiguous
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v z = {0.0f, 0.0f, 0.0f, 0.0f};
        f4v *dst = reinterpret_cast<f4v *>(v.bottom_diff + pix0 * C);
        const int n4 = npx * (C / 4);
        for (int t = threadIdx.x; t < n4; t += 256) __builtin_nontemporal_store(z, dst + t);
    }
    if (threadIdx.x < BWI_PIX) { S.cnt[threadIdx.x] = 0; S.run[threadIdx.x] = 0; }
    int nlist = 0;

    // filter one pass of ROIs into the ordered LDS list, then evaluate every (entry, pixel) pair: number of candidate
    // bins and the column range
    auto build = [&](const int pass) -> int {
        __syncthreads();
        const int r = pass * BW_CHUNK + (int)threadIdx.x;
        bool ok = false;
        int rsw = 0, rew = 0, prow = 0;
        float bw = 1.0f;
        if (r < R) {
            const float *roi = v.rois + 5 * (long long)r;
            const RoiGeom t = roi_geom(roi, v.scale);
            ok = ((int)roi[0] == n) && h >= t.rsh && h <= t.reh && t.rew >= w0 && t.rsw < w0 + npx;   // roi_pooling_op.cc:392-403
            if (ok) {
                const int rh = max(t.reh - t.rsh + 1, 1), rw = max(t.rew - t.rsw + 1, 1);
                const float bh = (float)rh / (float)PH;
                int phs = (int)floorf((float)(h - t.rsh) / bh), phe = (int)ceilf((float)(h - t.rsh + 1) / bh);   // :423-426
                phs = min(max(phs, 0), PH); phe = min(max(phe, 0), PH);
                ok = phe > phs;
                rsw = t.rsw; rew = t.rew; prow = phs | (phe << 8);
                bw = (float)rw / (float)PW;
            }
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) S.wcnt[wave] = __popcll(bal);
        __syncthreads();
        int pos = __popcll(bal & ((1ull << lane) - 1ull));
        int nl = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int cw = S.wcnt[t]; if (t < wave) pos += cw; nl += cw; }
        if (ok) { S.roi[pos] = r; S.rsw[pos] = rsw; S.rew[pos] = rew; S.prow[pos] = prow; S.bw[pos] = bw; }
        __syncthreads();
        int local = 0;
        const int w = w0 + j;
        for (int e = q; e < nl; e += 256 / BWI_PIX) {
            int nb = 0, xr = 0;
            const int xs = S.rsw[e], xe = S.rew[e];
            if (j < npx && w >= xs && w <= xe) {
                const float fbw = S.bw[e];
                int x0 = (int)floorf((float)(w - xs) / fbw), x1 = (int)ceilf((float)(w - xs + 1) / fbw);
                x0 = min(max(x0, 0), PW); x1 = min(max(x1, 0), PW);
                if (x1 > x0) {
                    const int pr = S.prow[e];
                    nb = ((pr >> 8) - (pr & 255)) * (x1 - x0);
                    xr = x0 | (x1 << 4);
                }
            }
            S.nb[e][j] = (unsigned char)nb;
            S.xr[e][j] = (unsigned char)xr;
            local += nb;
        }
        if (local) atomicAdd(&S.cnt[j], local);
        return nl;
    };

    for (int pass = 0; pass < npass; ++pass) nlist = build(pass);
    __syncthreads();
    if (!FILL) {
        if (threadIdx.x < 64) {
            const int c = (lane < BWI_PIX) ? S.cnt[lane] : 0;
            int tot = c;
#pragma unroll
            for (int m = 1; m < BWI_PIX; m <<= 1) tot += __shfl_xor(tot, m);
            const unsigned mask = (unsigned)(__ballot(c > 0) & 0xffffull);
            if (lane == 0) { ix.seg_tot[block] = (tot << 5) | __popc(mask); ix.seg_mask[block] = (int)mask; }
        }
        return;
    }
    {   // slab offsets = sums over the preceding segments' sizes (candidates << 5 | pixels with any; eight words per thread requested
        // together: one word at a time the loop was a chain of dependent round trips at the head of every workgroup)
        int a = 0, b = 0;
        for (int t0 = 0; t0 < (int)block; t0 += 256 * 8) {
            int pk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 256 + (int)threadIdx.x;
                pk[u] = t < (int)block ? ix.seg_tot[t] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += pk[u] >> 5; b += pk[u] & 31; }
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
        if (lane == 0) { S.red[wave] = a; S.red[4 + wave] = b; }
        __syncthreads();
    }
    if (threadIdx.x < 64) {                      // sizes -> offsets inside the segment's slabs; items
        const int base0 = S.red[0] + S.red[1] + S.red[2] + S.red[3], ibase = S.red[4] + S.red[5] + S.red[6] + S.red[7];
        const int c = (lane < BWI_PIX) ? S.cnt[lane] : 0;
        int inc = c;
#pragma unroll
        for (int m = 1; m < BWI_PIX; m <<= 1) { const int t = __shfl_up(inc, m); if (lane >= m) inc += t; }
        const unsigned long long ne = __ballot(c > 0);
        if (lane < BWI_PIX) S.base[lane] = base0 + inc - c;
        if (c > 0) ix.items[ibase + __popcll(ne & ((1ull << lane) - 1ull))] = make_int4((int)(pix0 + lane), base0 + inc - c, c, k);
        if (lane == 0 && block == nblocks - 1) ix.header[1] = ibase + __popcll(ne);        // number of items of the launch
    }
    __syncthreads();
    for (int pass = 0; pass < npass; ++pass) {
        if (npass > 1) { nlist = build(pass); }
        __syncthreads();
        // per pixel: where each entry's bins go, in entry order = an exclusive prefix of nb over the entries.  16 stripes of
        // CONSECUTIVE entries per pixel (thread = (stripe, pixel)): stripe sums, a prefix over the stripes, then the thread walks
        // its own entries with the running offset in a register and writes their records' byte offsets (a serial walk by one
        // thread per pixel was the slowest part of the launch on segments with a few hundred entries)
        const int sp = threadIdx.x & (BWI_PIX - 1), st = threadIdx.x / BWI_PIX;       // pixel, stripe
        const int L = (nlist + 15) / 16, e0 = st * L, e1 = min(nlist, e0 + L);
        int sum = 0;
        for (int e = e0; e < e1; ++e) sum += S.nb[e][sp];
        S.part[st][sp] = sum;
        __syncthreads();
        int run = S.run[sp];
        for (int t = 0; t < st; ++t) run += S.part[t][sp];
        int *dst = ix.pool + S.base[sp] + run;
        for (int e = e0; e < e1; ++e) {
            const int nb = S.nb[e][sp];
            if (nb == 0) continue;
            const int xr = S.xr[e][sp], x0 = xr & 15, x1 = xr >> 4, pr = S.prow[e];
            const int rec0 = S.roi[e] * PH * PW;
            for (int ph = pr & 255; ph < (pr >> 8); ++ph)
                for (int pw = x0; pw < x1; ++pw) *dst++ = (rec0 + ph * PW + pw) * C * 4;     // the record's byte offset
        }
        __syncthreads();
        if (st == 15) S.run[sp] = run + sum;                         // (the last stripe ends at the pass's total)
    }
}

template <bool FILL>
__global__ __launch_bounds__(256) void roi_bwd_index_kernel(RoiGradPack p, RoiGradIdxPack ix)
{
    __shared__ RoiIdxShared S;
    roi_bwd_index_block<FILL>(S, p, ix, blockIdx.x, gridDim.x);
}

// records u0 .. u0 + W - 1 of the 64 whose byte offsets sit in the lanes of `cur`: per record one v_readlane (-> SGPR) and two
// buffer loads (CPL dwords per lane) whose scalar offset is that SGPR, then CPL x (compare / select / add); W x 2 loads per
// lane in flight
template <int CPL> struct BwdVec;
template <> struct BwdVec<1> { typedef unsigned int T; static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t r, int v, int s) { return __builtin_amdgcn_raw_buffer_load_b32(r, v, s, 0); } };
template <> struct BwdVec<2> { typedef unsigned int T __attribute__((ext_vector_type(2))); static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t r, int v, int s) { return __builtin_amdgcn_raw_buffer_load_b64(r, v, s, 0); } };
template <> struct BwdVec<4> { typedef unsigned int T __attribute__((ext_vector_type(4))); static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t r, int v, int s) { return __builtin_amdgcn_raw_buffer_load_b128(r, v, s, 0); } };
template <int CPL> __device__ __forceinline__ unsigned bwd_elem(const typename BwdVec<CPL>::T &x, int j) { return x[j]; }
template <> __device__ __forceinline__ unsigned bwd_elem<1>(const unsigned int &x, int) { return x; }

template <int CPL, int W, bool MASKED>
__device__ __forceinline__ void bwd_drain_lanes(const int cur, const int u0, const int m, const __amdgpu_buffer_rsrc_t ra,
                                                const __amdgpu_buffer_rsrc_t rt, const int voff, const int want, float (&a)[CPL])
{
    typename BwdVec<CPL>::T am_v[W], td_v[W];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const int so = __builtin_amdgcn_readlane(cur, MASKED ? min(u0 + u, 63) : u0 + u);
        am_v[u] = BwdVec<CPL>::ld(ra, voff, so);
        td_v[u] = BwdVec<CPL>::ld(rt, voff, so);
    }
#pragma unroll
    for (int u = 0; u < W; ++u)
        if (!MASKED || u0 + u < m) {
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                a[j] += ((int)bwd_elem<CPL>(am_v[u], j) == want + j) ? __builtin_bit_cast(float, bwd_elem<CPL>(td_v[u], j)) : 0.0f;
        }
}

// A persistent grid; wave = (item, slice of 64 CPL channels).  XCD x (= blockIdx % 8) works on slice x % nsl of the items of
// part x / nsl (the item list is in pixel order: a part is a band of rows / frames), so that every record slice is pulled
// into ONE private L2.  CPL > 1 makes the pieces a slice reads from a record larger (64 CPL x 4 B: 256 B at CPL = 1, 1 KB at
// CPL = 4): cold HBM reads of 256-B pieces scattered at 2 KB stride run at ~3 TB/s (DRAM row misses), the same bytes in
// larger pieces do not.  A wave walks its items i, i + stride, ... software-pipelined: while item n's records are in flight,
// the offsets of item n + 1 and the header of item n + 2 are already requested.
template <int CPL>
__device__ __forceinline__ void roi_bwd_gather_block(const RoiGradPack &p, const RoiGradIdxPack &ix, const int nsl, const unsigned vblock,
                                                     const unsigned vgrid)
{
    const int lane = threadIdx.x & 63;
    const int xcd = (int)(vblock & 7);
    const int slice = xcd % nsl, part = xcd / nsl, nparts = 8 / nsl;
    const int n_all = __builtin_amdgcn_readfirstlane(ix.header[1]);
    const int per = (n_all + nparts - 1) / nparts;
    const int i_end = min(n_all, (part + 1) * per);                     // this part's items: [part * per, i_end)
    const int stride = (int)(vgrid >> 3) * 4;
    int i = part * per + (int)(vblock >> 3) * 4 + (int)(threadIdx.x >> 6);
    // diagnostics (tools/roi_bwd_trace.py): 8 words per wave {start, items loaded, offsets loaded, first item done, end, items,
    // candidates, -}
    long long *tr = ix.trace ? ix.trace + 8 * ((long long)vblock * 4 + (threadIdx.x >> 6)) : nullptr;
    long long t_first = 0;
    int n_done = 0, n_cand = 0;
    if (tr && lane == 0) tr[0] = (long long)__builtin_readcyclecounter();
    if (i >= i_end) return;
    const int4 zero4 = make_int4(0, 0, 0, 0);
    int4 it = ix.items[i];
    int4 it1 = (i + stride < i_end) ? ix.items[i + stride] : zero4;
    if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) tr[1] = (long long)__builtin_readcyclecounter(); }
    int idx = ix.pool[it.y + min(lane, it.z - 1)];
    if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) tr[2] = (long long)__builtin_readcyclecounter(); }
    const int voff = lane * 4 * CPL;
    constexpr int W = CPL == 1 ? 32 : (CPL == 2 ? 16 : 8);
    for (; i < i_end; i += stride) {
        const int pix = __builtin_amdgcn_readfirstlane(it.x), off = __builtin_amdgcn_readfirstlane(it.y);
        const int cnt = __builtin_amdgcn_readfirstlane(it.z), k = __builtin_amdgcn_readfirstlane(it.w);
        const bool more = i + stride < i_end;                              // wave-uniform
        const int4 nxt = it1;
        if (i + 2 * stride < i_end) it1 = ix.items[i + 2 * stride];
        int idx1 = 0;
        if (more) idx1 = ix.pool[__builtin_amdgcn_readfirstlane(nxt.y) + min(lane, __builtin_amdgcn_readfirstlane(nxt.z) - 1)];
        const RoiGradViewDev &v = p.v[k];
        const int C = v.C;
        const int c = slice * 64 * CPL + lane * CPL;
        const int want = (pix % (v.H * v.W)) * C + c;
        const int *cand = ix.pool + off;
        // the slice lives in the (wave-uniform) base address, the lane in the vector offset, the record's byte offset is
        // the scalar offset of the load
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(v.argmax + slice * 64 * CPL), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)(v.top_diff + slice * 64 * CPL), 0, 0x7fffffff, 0x00020000);
        float a[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) a[j] = 0.0f;
        // 64 record offsets per vector load (lane l holds candidate t0 + l); the next 64 are requested before the current
        // ones are consumed
        for (int t0 = 0; t0 < cnt; t0 += 64) {
            const int cur = idx;
            if (t0 + 64 < cnt) idx = cand[min(t0 + 64 + lane, cnt - 1)];
            const int m = min(64, cnt - t0);
            int u0 = 0;
            for (; u0 + W <= m; u0 += W) bwd_drain_lanes<CPL, W, false>(cur, u0, m, ra, rt, voff, want, a);
            if (W > 8 && u0 + 8 <= m) { bwd_drain_lanes<CPL, 8, false>(cur, u0, m, ra, rt, voff, want, a); u0 += 8; }
            if (W > 16 && u0 + 8 <= m) { bwd_drain_lanes<CPL, 8, false>(cur, u0, m, ra, rt, voff, want, a); u0 += 8; }
            if (W > 16 && u0 + 8 <= m) { bwd_drain_lanes<CPL, 8, false>(cur, u0, m, ra, rt, voff, want, a); u0 += 8; }
            if (u0 + 4 <= m) { bwd_drain_lanes<CPL, 4, false>(cur, u0, m, ra, rt, voff, want, a); u0 += 4; }
            if (u0 < m) bwd_drain_lanes<CPL, 4, true>(cur, u0, m, ra, rt, voff, want, a);
        }
        float *out = v.bottom_diff + (long long)pix * C + c;
        if (CPL == 1) __builtin_nontemporal_store(a[0], out);
        else if (CPL == 2) { typedef float f2v __attribute__((ext_vector_type(2))); const f2v av = {a[0], a[1 % CPL]}; __builtin_nontemporal_store(av, reinterpret_cast<f2v *>(out)); }
        else { typedef float f4v __attribute__((ext_vector_type(4))); const f4v av = {a[0], a[1 % CPL], a[2 % CPL], a[3 % CPL]}; __builtin_nontemporal_store(av, reinterpret_cast<f4v *>(out)); }
        it = nxt; idx = idx1;
        if (tr) { if (n_done == 0) t_first = (long long)__builtin_readcyclecounter(); ++n_done; n_cand += cnt; }
    }
    if (tr && lane == 0) { tr[3] = t_first; tr[4] = (long long)__builtin_readcyclecounter(); tr[5] = n_done; tr[6] = n_cand; tr[7] = 0; }
}

template <int CPL>
__global__ __launch_bounds__(256) void roi_bwd_gather_kernel(RoiGradPack p, RoiGradIdxPack ix, int nsl)
{
    roi_bwd_gather_block<CPL>(p, ix, nsl, blockIdx.x, gridDim.x);
}

// ===============================================================================================================================
// The PAIR (mv3d_roi_pool_forward_views_pair / mv3d_roi_pool_backward_views_pair): RoiPool and its gradient with a private
// argmax plane of one-byte codes between them (16-bit codes for bins of more than 255 pixels: COMPACT above).
//
//   forward          the XCD-sliced pooling kernels above with COMPACT codes: 5 instead of 8 bytes per pooled value are written.
//   backward, launches 1 + 2  one workgroup per 16-pixel segment of a map row, as the plain indexed RoiPoolGrad above: SIZES
//                    (filter the ROIs by frame, row, column span; an upper bound of every pixel's list: all bins the reference's
//                    test lets through), then LISTS (slab offsets = plain sums over the preceding segments' sizes -- the sizing
//                    launch is complete -- items, and the candidate lists themselves); the LISTS launch zero-fills the pixels
//                    WITHOUT an item under its latency chain (the ROI and size loads are issued BEFORE the fill's stores: a load's wait would
//                    otherwise wait for every older store as well).  No atomics, no state that has to be zero on entry.
//   candidate lists  entries {record byte offset into top_diff, code of THIS pixel inside THAT bin} in the reference's order
//                    roi -> ph -> pw; only bins whose forward rectangle [hstart, hend) x [wstart, wend) (roi_pooling_op.cc:153-162)
//                    contains the pixel are listed: the forward scans nothing else, so a bin outside of which the pixel lies can
//                    never name it (roi_pooling_op.cc:433 is false for it whatever top_diff holds) -- and a bin that covers the
//                    pixel without passing the reference's candidate test (:401-431: the test uses the UNCLAMPED rounded ROI, a
//                    last bin's ceil() may reach one pixel past it) stays out, as the reference leaves it out.
//   backward, launch 3  the gather sums each candidate pixel's records in list order (= the reference's f32 summation order): per
//                    record one code byte + one f32 per lane, `code == this pixel's code ? top_diff : +0`, and writes the pixel.
// Bit-identical to the plain entries (tests/test_roi_pair.py).  Measured and dropped in round 5 (tools/experiments/
// roi_pair_index_in_forward_r05.hip.txt, profiles/r05_c_*, r05_d_*): the index built by workgroups INSIDE the forward launch
// (forward 39 -> 56 us: latency-bound workgroups under a write-saturated memory system), and a single-pass index with a look-back
// over agent-scope words (38 - 49 us for the launch: 1744 workgroups x ~870 predecessor words through the memory side).
// header words of the workspace (ints): [1] number of items
struct RoiPairIdx {
    int4 *items; int2 *pool; int *header; int *seg_tot, *seg_mask;
    unsigned first_block[MV3D_MAX_ROI_VIEWS]; int gpr[MV3D_MAX_ROI_VIEWS];
    unsigned nseg;
    int nslots;                                      // item slots (one per pixel of every view)
    int dbg;                                         // experiment builds (MV3D_TUNING): parts of the index launches switched off, 0 otherwise
    long long *trace;                                // experiment builds: 8 cycle stamps per workgroup of the lists launch (tools/roi_idx_trace.py)
};

struct RoiPairShared {
    int red[8];
    int roi[BW_CHUNK], rsw[BW_CHUNK], rew[BW_CHUNK], rsh[BW_CHUNK], prow[BW_CHUNK];
    float bw[BW_CHUNK], bh[BW_CHUNK];
    unsigned char nb[BW_CHUNK][BWI_PIX], xr[BW_CHUNK][BWI_PIX];
    int wcnt[4], cnt[BWI_PIX], base[BWI_PIX], run[BWI_PIX];
    int part[16][BWI_PIX];
    int last;
};

// The forward's rectangle of a bin is a product of a row range and a column range (roi_pooling_op.cc:153-162), and so is the
// reference's candidate set of a pixel (:423-431), so the listed bins of pixel (h, w) under one ROI are [pa, pb) x [qa, qb): the
// bins ph of the reference's range whose rows hstart(ph) <= h < hend(ph) contain the pixel's row -- a contiguous run, the bins being
// ordered -- times the same for the columns.  The row run is found once per (ROI, segment) by the thread that filters the ROI, the
// column run once per (ROI, pixel); a bin's code is then (h - hstart(ph)) * (wend(pw) - wstart(pw)) + (w - wstart(pw)).
__device__ __forceinline__ int roi_pair_lo(const int p, const float bin, const int start, const int limit)
{
    return min(max((int)floorf(__fmul_rn((float)p, bin)) + start, 0), limit);        // hstart / wstart as roi_pool_fwd_xcd_block computes them
}
__device__ __forceinline__ int roi_pair_hi(const int p, const float bin, const int start, const int limit)
{
    return min(max((int)ceilf(__fmul_rn((float)(p + 1), bin)) + start, 0), limit);    // hend / wend
}
// the run [a, b) of bins of [p0, p1) whose extent contains coordinate x; packed a | b << 8 (a == b: none)
__device__ __forceinline__ int roi_pair_run(const int p0, const int p1, const float bin, const int start, const int limit, const int x)
{
    int a = p1, b = p1;
    for (int p = p0; p < p1; ++p) {
        const bool in = x >= roi_pair_lo(p, bin, start, limit) && x < roi_pair_hi(p, bin, start, limit);
        if (in && a == p1) a = p;
        if (!in && a != p1) { b = p; break; }
    }
    return a | (b << 8);
}

// FILL = false: sizes -- the UNPRUNED number of candidate bins of every pixel (an upper bound of its list, cheap: no per-bin
// arithmetic) and the mask of pixels that have any; FILL = true: offsets, items, the pruned lists with their codes.
template <bool FILL>
__device__ __forceinline__ void roi_pair_index_block(RoiPairShared &S, const RoiGradPack &p, const RoiPairIdx &ix, const unsigned block)
{
    const unsigned nblocks = ix.nseg;
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && block >= ix.first_block[j]) k = j;
    const RoiGradViewDev &v = p.v[k];
    const int PH = p.PH, PW = p.PW, H = v.H, W = v.W, R = v.R, C = v.C;
    const unsigned g = block - ix.first_block[k];
    const int gpr = ix.gpr[k];
    const int w0 = (int)(g % (unsigned)gpr) * BWI_PIX;
    const int npx = min(BWI_PIX, W - w0);
    const unsigned gh = g / (unsigned)gpr;
    const int h = (int)(gh % (unsigned)H), n = (int)(gh / (unsigned)H);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = threadIdx.x & (BWI_PIX - 1), q = threadIdx.x / BWI_PIX;      // pixel of the segment, entry stripe
    const int npass = (R + BW_CHUNK - 1) / BW_CHUNK;
    const long long pix0 = ((long long)n * H + h) * W + w0;
    long long *tr = (FILL && ix.trace) ? ix.trace + 8 * (long long)block : nullptr;
#define RPI_STAMP(K) do { if (tr && threadIdx.x == 0) tr[K] = (long long)__builtin_readcyclecounter(); } while (0)
    RPI_STAMP(0);
    // (lists) the pixels the sizing launch gave an item (upper bound > 0): the FIRST request of the workgroup -- the fill below skips
    // those pixels and should not have to wait for the younger requests as well
    unsigned my_mask = 0;
    if (FILL) my_mask = (unsigned)ix.seg_mask[block];
    // the first pass's ROI of this thread: requested BEFORE the fill's stores (a wait for a load also waits for every older store)
    float r0[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if ((int)threadIdx.x < R) {
        const float *roi = v.rois + 5 * (long long)threadIdx.x;
#pragma unroll
        for (int u = 0; u < 5; ++u) r0[u] = roi[u];
    }
    // (lists) the segment's slab and item offsets = sums over the preceding segments' sizes: requested before the fill as well
    int a = 0, b = 0;
    if (FILL) {
        // (seg_tot holds candidates << 5 | pixels with any; eight words per thread requested together: summed one word at a time the
        // loop was a chain of dependent round trips -- ~2 us each -- at the head of every workgroup)
        for (int t0 = 0; t0 < ((ix.dbg & 2) ? 0 : (int)block); t0 += 256 * 8) {
            int pk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 256 + (int)threadIdx.x;
                pk[u] = t < (int)block ? ix.seg_tot[t] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += pk[u] >> 5; b += pk[u] & 31; }
        }
    }
    if (FILL && !(ix.dbg & 1)) {    // the pixels of the segment WITHOUT an item become zeros here (every item's pixel is written by the
        // gather, an empty pruned list as +0): the list launch is a ~15 us chain of barriers, LDS round trips and little memory traffic,
        // the ~40 MB of streaming stores ride under it (the sizing launch is short and stays short without them).  npx * C floats
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v z = {0.0f, 0.0f, 0.0f, 0.0f};
        f4v *dst = reinterpret_cast<f4v *>(v.bottom_diff + pix0 * C);
        const int n4 = npx * (C / 4), psh = C == 512 ? 7 : 6;          // (the pair takes C = 256 | 512: roi_pair_shapes)
        const unsigned skip = (ix.dbg & 16) ? 0u : my_mask;            // (dbg 16, experiment builds: zero every pixel as before)
        for (int t = threadIdx.x; t < n4; t += 256)
            if (!((skip >> (t >> psh)) & 1u)) __builtin_nontemporal_store(z, dst + t);
    }
    if (threadIdx.x < BWI_PIX) { S.cnt[threadIdx.x] = 0; S.run[threadIdx.x] = 0; }
    int nlist = 0;

    // filter one pass of ROIs into the ordered LDS list (roi_pooling_op.cc:392-403, :423-426 for the rows), then count every
    // (entry, pixel) pair's bins: all the reference's test lets through (sizes), or the ones the lists will hold (lists)
    auto build = [&](const int pass) -> int {
        __syncthreads();
        const int r = pass * BW_CHUNK + (int)threadIdx.x;
        bool ok = false;
        int rsw = 0, rew = 0, rsh = 0, prow = 0;
        float bw = 1.0f, bh = 1.0f;
        if (r < R) {
            float rr[5];
            if (pass == 0) {
#pragma unroll
                for (int u = 0; u < 5; ++u) rr[u] = r0[u];
            } else {
                const float *roi = v.rois + 5 * (long long)r;
#pragma unroll
                for (int u = 0; u < 5; ++u) rr[u] = roi[u];
            }
            const RoiGeom t = roi_geom(rr, v.scale);
            ok = ((int)rr[0] == n) && h >= t.rsh && h <= t.reh && t.rew >= w0 && t.rsw < w0 + npx;
            if (ok) {
                const int rh = max(t.reh - t.rsh + 1, 1), rw = max(t.rew - t.rsw + 1, 1);
                bh = (float)rh / (float)PH;
                int phs = (int)floorf((float)(h - t.rsh) / bh), phe = (int)ceilf((float)(h - t.rsh + 1) / bh);
                phs = min(max(phs, 0), PH); phe = min(max(phe, 0), PH);
                ok = phe > phs;
                rsw = t.rsw; rew = t.rew; rsh = t.rsh; prow = phs | (phe << 8);
                bw = (float)rw / (float)PW;
                if (FILL && ok && !(ix.dbg & 8)) { // the lists hold only the bins whose forward rows contain h
                    prow = roi_pair_run(phs, phe, bh, rsh, H, h);
                    ok = (prow >> 8) > (prow & 255);
                }
            }
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) S.wcnt[wave] = __popcll(bal);
        __syncthreads();
        int pos = __popcll(bal & ((1ull << lane) - 1ull));
        int nl = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int cw = S.wcnt[t]; if (t < wave) pos += cw; nl += cw; }
        if (ok) { S.roi[pos] = r; S.rsw[pos] = rsw; S.rew[pos] = rew; S.rsh[pos] = rsh; S.prow[pos] = prow; S.bw[pos] = bw; S.bh[pos] = bh; }
        __syncthreads();
        int local = 0;
        const int w = w0 + j;
        for (int e = q; e < nl; e += 256 / BWI_PIX) {
            int nb = 0, xr = 0;
            const int xs = S.rsw[e], xe = S.rew[e];
            if (j < npx && w >= xs && w <= xe) {
                const float fbw = S.bw[e];
                int x0 = (int)floorf((float)(w - xs) / fbw), x1 = (int)ceilf((float)(w - xs + 1) / fbw);      // :425-426
                x0 = min(max(x0, 0), PW); x1 = min(max(x1, 0), PW);
                if (x1 > x0) {
                    const int pr = S.prow[e];
                    if (FILL && !(ix.dbg & 8)) {    // ... and whose forward columns contain w
                        const int run = roi_pair_run(x0, x1, fbw, xs, W, w);
                        x0 = run & 255; x1 = run >> 8;
                    }
                    nb = ((pr >> 8) - (pr & 255)) * (x1 - x0);
                    xr = x0 | (x1 << 4);
                }
            }
            if (FILL) { S.nb[e][j] = (unsigned char)nb; S.xr[e][j] = (unsigned char)xr; }
            local += nb;
        }
        if (local) atomicAdd(&S.cnt[j], local);
        return nl;
    };

    if (!FILL) {
        for (int pass = 0; pass < npass; ++pass) nlist = build(pass);
        __syncthreads();
        if (threadIdx.x < 64) {
            const int c = (lane < BWI_PIX) ? S.cnt[lane] : 0;
            int tot = c;
#pragma unroll
            for (int m = 1; m < BWI_PIX; m <<= 1) tot += __shfl_xor(tot, m);
            const unsigned mask = (unsigned)(__ballot(c > 0) & 0xffffull);
            if (lane == 0) { ix.seg_tot[block] = (tot << 5) | __popc(mask); ix.seg_mask[block] = (int)mask; }
        }
        return;
    }
    // ---- lists
    RPI_STAMP(1);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
    RPI_STAMP(2);                                     // (the prefix words have arrived)
    if (lane == 0) { S.red[wave] = a; S.red[4 + wave] = b; }
    // the pruned counts of every pass: a pixel's list is written pass after pass, its length is known after the last one
    if (npass == 1) nlist = build(0);
    else {
        // (R > 256: the pixel counts of all passes first, then the passes again for the lists -- rare on this path)
        for (int pass = 0; pass < npass; ++pass) nlist = build(pass);
    }
    __syncthreads();
    if (npass == 1 && nlist == 0 && my_mask == 0u) {  // no ROI touches the segment (half of them): no item, nothing to list
        if (threadIdx.x == 0 && block == nblocks - 1) {
            int bt = 0;
            for (int u = 0; u < 4; ++u) bt += S.red[4 + u];
            ix.header[1] = bt;                        // (the sizing launch gave this segment no item either)
        }
        return;
    }
    RPI_STAMP(3);                                     // (filter + counts done)
    if (tr && threadIdx.x == 0) tr[7] = nlist;
    if (threadIdx.x < 64) {                      // upper-bound slab offsets of the segment's pixels come from the SIZES launch's rule
        const int base0 = S.red[0] + S.red[1] + S.red[2] + S.red[3], ibase = S.red[4] + S.red[5] + S.red[6] + S.red[7];
        // this pixel's slab: the unpruned counts are not kept, so the pixels share the segment's slab in pixel order by their PRUNED
        // counts (pruned <= unpruned, pixel by pixel: the slab is never overrun)
        const int c = (lane < BWI_PIX) ? S.cnt[lane] : 0;
        int inc = c;
#pragma unroll
        for (int m = 1; m < BWI_PIX; m <<= 1) { const int t = __shfl_up(inc, m); if (lane >= m) inc += t; }
        const bool has = lane < BWI_PIX && ((my_mask >> lane) & 1u);
        const unsigned long long ne = __ballot(has);
        if (lane < BWI_PIX) S.base[lane] = base0 + inc - c;
        // an item per pixel the sizing launch counted (its list may turn out empty: count 0, the gather then writes the zero again)
        // (dbg 32 / 64, experiment builds: lists cut to 32 / 8 entries -- wrong sums, the gather's time without its long items)
        const int cc = (ix.dbg & 32) ? min(c, 32) : ((ix.dbg & 64) ? min(c, 8) : c);
        if (has) ix.items[ibase + __popcll(ne & ((1ull << lane) - 1ull))] = make_int4((int)(pix0 + lane), base0 + inc - c, cc, k);
        if (lane == 0 && block == nblocks - 1) ix.header[1] = ibase + __popcll(ne);        // number of items of the launch
    }
    __syncthreads();
    for (int pass = 0; pass < npass; ++pass) {
        if (npass > 1) { nlist = build(pass); }
        __syncthreads();
        // per pixel: where each entry's bins go = an exclusive prefix of nb over the entries: 16 stripes of consecutive entries per
        // pixel (thread = (stripe, pixel)), stripe sums, a prefix over the stripes, then the thread walks its own entries
        const int sp = threadIdx.x & (BWI_PIX - 1), st = threadIdx.x / BWI_PIX;       // pixel, stripe
        const int L = (nlist + 15) / 16, e0 = st * L, e1 = min(nlist, e0 + L);
        int sum = 0;
        for (int e = e0; e < e1; ++e) sum += S.nb[e][sp];
        S.part[st][sp] = sum;
        __syncthreads();
        int run = S.run[sp];
        for (int t = 0; t < st; ++t) run += S.part[t][sp];
        int2 *dst = ix.pool + S.base[sp] + run;
        for (int e = e0; e < ((ix.dbg & 4) ? e0 : e1); ++e) {
            if (S.nb[e][sp] == 0) continue;
            const int xr = S.xr[e][sp], pr = S.prow[e], ys = S.rsh[e], xs = S.rsw[e];
            const float fbh = S.bh[e], fbw = S.bw[e];
            const int rec0 = S.roi[e] * PH * PW, w = w0 + sp;
            const int qa = xr & 15, ncols = (xr >> 4) - qa;
            // (a tiny ROI -- far objects on the 8 x 64 front-view map -- puts all PH x PW bins on one pixel: up to 49 entries per (ROI,
            // pixel), and such segments are the launch's long pole.  The columns' start / width are the same for every row of bins:
            // up to eight of them are kept in registers, so that an entry costs a multiply-add and a store)
            if (ncols <= 8) {
                int cw[8], cd[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int pw = min(qa + u, PW - 1);
                    const int ws = roi_pair_lo(pw, fbw, xs, W);
                    cw[u] = roi_pair_hi(pw, fbw, xs, W) - ws;
                    cd[u] = w - ws;
                }
                for (int ph = pr & 255; ph < (pr >> 8); ++ph) {
                    const int hs = roi_pair_lo(ph, fbh, ys, H), bhgt = roi_pair_hi(ph, fbh, ys, H) - hs;
                    const int dh = h - hs, rb = (rec0 + ph * PW + qa) * C * 4;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < ncols) dst[u] = make_int2(rb + u * C * 4, (dh * cw[u] + cd[u]) | (bhgt * cw[u] > 255 ? RPC_BIG : 0));
                    dst += ncols;
                }
            } else {
                for (int ph = pr & 255; ph < (pr >> 8); ++ph) {
                    const int hs = roi_pair_lo(ph, fbh, ys, H), bhgt = roi_pair_hi(ph, fbh, ys, H) - hs, dh = h - hs;
                    for (int pw = qa; pw < qa + ncols; ++pw) {
                        const int ws = roi_pair_lo(pw, fbw, xs, W), we = roi_pair_hi(pw, fbw, xs, W);
                        *dst++ = make_int2((rec0 + ph * PW + pw) * C * 4, (dh * (we - ws) + (w - ws)) | (bhgt * (we - ws) > 255 ? RPC_BIG : 0));
                    }
                }
            }
        }
        __syncthreads();
        if (st == 15) S.run[sp] = run + sum;                         // (the last stripe ends at the pass's total)
    }
    RPI_STAMP(4);
#undef RPI_STAMP
}

// the gather of the pair: wave = (item, 64-channel slice) as roi_bwd_gather_block; per record one code byte and one f32 per lane
template <int W, bool MASKED>
__device__ __forceinline__ void roi_pair_drain(const int cur_o, const int cur_k, const int u0, const int m, const __amdgpu_buffer_rsrc_t rc,
                                               const __amdgpu_buffer_rsrc_t rt, const int lane, float &a)
{
    unsigned char cd[W];
    float td[W];
    int sk[W];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const int l = MASKED ? min(u0 + u, 63) : u0 + u;
        const int so = __builtin_amdgcn_readlane(cur_o, l);
        sk[u] = __builtin_amdgcn_readlane(cur_k, l);
        cd[u] = __builtin_amdgcn_raw_buffer_load_b8(rc, lane, so >> 2, 0);
        td[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, lane * 4, so, 0));
    }
#pragma unroll
    for (int u = 0; u < W; ++u)
        if (!MASKED || u0 + u < m) a += ((int)cd[u] == sk[u]) ? td[u] : 0.0f;
}
// a batch of entries with bins of more than 255 pixels among them (ROIs far larger than the map: rare): entry by entry, each from
// its own plane, in the same order
__device__ __forceinline__ void roi_pair_drain_mixed(const int cur_o, const int cur_k, const int m, const __amdgpu_buffer_rsrc_t rc,
                                                     const __amdgpu_buffer_rsrc_t rc16, const __amdgpu_buffer_rsrc_t rt, const int lane, float &a)
{
    for (int u = 0; u < m; ++u) {
        const int so = __builtin_amdgcn_readlane(cur_o, u), k = __builtin_amdgcn_readlane(cur_k, u);
        const int code = (k & RPC_BIG) ? (int)__builtin_amdgcn_raw_buffer_load_b16(rc16, lane * 2, so >> 1, 0)
                                       : (int)__builtin_amdgcn_raw_buffer_load_b8(rc, lane, so >> 2, 0);
        const float td = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, lane * 4, so, 0));
        a += (code == (k & 0xffff)) ? td : 0.0f;
    }
}

// (Round 5, measured and dropped -- profiles/r05_z_gather_whole_ab.txt: the eight waves of a 512-thread workgroup taking ONE item and a
// slice each, any workgroup any item -- a record's 2 KB requested by one CU at one time instead of as 256-byte pieces by eight XCDs:
// RoiPoolGrad 79 - 103 us for grids of 4096 - 512 workgroups against 68 us.  The slices stay on their XCDs.)
__device__ __forceinline__ void roi_pair_gather_block(const RoiGradPack &p, const RoiPairIdx &ix, const int nsl, const unsigned vblock,
                                                      const unsigned vgrid)
{
    const int lane = threadIdx.x & 63;
    const int xcd = (int)(vblock & 7);
    const int slice = xcd % nsl, part = xcd / nsl, nparts = 8 / nsl;
    const int stride = (int)(vgrid >> 3) * 4;
    const int i0 = (int)(vblock >> 3) * 4 + (int)(threadIdx.x >> 6);
    // (eight slices: this wave's first item does not depend on the number of items -- its header is requested together with that
    // number, one dependent round trip less at the head of every wave; the item array has a slot per pixel, so the read is in bounds)
    int4 it = make_int4(0, 0, 0, 0);
    if (nparts == 1) it = ix.items[i0];
    const int n_all = __builtin_amdgcn_readfirstlane(ix.header[1]);
    const int per = (n_all + nparts - 1) / nparts;
    const int i_end = min(n_all, (part + 1) * per);                     // this part's items: [part * per, i_end)
    int i = part * per + i0;
    if (i >= i_end) return;
    const int4 zero4 = make_int4(0, 0, 0, 0);
    if (nparts != 1) it = ix.items[i];
    int4 it1 = (i + stride < i_end) ? ix.items[i + stride] : zero4;
    int2 idx = ix.pool[it.y + max(min(lane, it.z - 1), 0)];       // (a list may be empty: every bin of the pixel pruned)
    constexpr int W = 32;
    for (; i < i_end; i += stride) {
        const int pix = __builtin_amdgcn_readfirstlane(it.x), off = __builtin_amdgcn_readfirstlane(it.y);
        const int cnt = __builtin_amdgcn_readfirstlane(it.z), k = __builtin_amdgcn_readfirstlane(it.w);
        const bool more = i + stride < i_end;                              // wave-uniform
        const int4 nxt = it1;
        if (i + 2 * stride < i_end) it1 = ix.items[i + 2 * stride];
        int2 idx1 = make_int2(0, 0);
        if (more) idx1 = ix.pool[__builtin_amdgcn_readfirstlane(nxt.y) + max(min(lane, __builtin_amdgcn_readfirstlane(nxt.z) - 1), 0)];
        const RoiGradViewDev &v = p.v[k];
        const int C = v.C;
        const int c = slice * 64 + lane;
        const int2 *cand = ix.pool + off;
        // the slice lives in the (wave-uniform) base address, the lane in the vector offset, the record's byte offset is the scalar
        // offset of the load (a quarter of it for the byte plane, half for the escape plane: both in the caller's argmax buffer)
        const unsigned char *const plane8 = (const unsigned char *)v.argmax;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(plane8 + slice * 64), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rc16 = __builtin_amdgcn_make_buffer_rsrc(
            (void *)((const unsigned short *)(plane8 + (long long)v.R * p.PH * p.PW * C) + slice * 64), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)(v.top_diff + slice * 64), 0, 0x7fffffff, 0x00020000);
        float a = 0.0f;
        for (int t0 = 0; t0 < cnt; t0 += 64) {
            const int2 cur = idx;
            if (t0 + 64 < cnt) idx = cand[min(t0 + 64 + lane, cnt - 1)];
            const int m = min(64, cnt - t0);
            if (__ballot((cur.y & RPC_BIG) != 0 && lane < m) != 0ull) {      // (wave-uniform; never taken on the proposal path's ROIs)
                roi_pair_drain_mixed(cur.x, cur.y, m, rc, rc16, rt, lane, a);
                continue;
            }
            int u0 = 0;
            for (; u0 + W <= m; u0 += W) roi_pair_drain<W, false>(cur.x, cur.y, u0, m, rc, rt, lane, a);
            if (u0 + 16 <= m) { roi_pair_drain<16, false>(cur.x, cur.y, u0, m, rc, rt, lane, a); u0 += 16; }
            if (u0 + 8 <= m) { roi_pair_drain<8, false>(cur.x, cur.y, u0, m, rc, rt, lane, a); u0 += 8; }
            if (u0 + 4 <= m) { roi_pair_drain<4, false>(cur.x, cur.y, u0, m, rc, rt, lane, a); u0 += 4; }
            if (u0 < m) roi_pair_drain<4, true>(cur.x, cur.y, u0, m, rc, rt, lane, a);
        }
        __builtin_nontemporal_store(a, v.bottom_diff + (long long)pix * C + c);
        it = nxt; idx = idx1;
    }
}

template <bool FILL>
__global__ __launch_bounds__(256) void roi_pair_index_kernel(RoiGradPack p, RoiPairIdx ix)
{
    __shared__ RoiPairShared S;
    roi_pair_index_block<FILL>(S, p, ix, blockIdx.x);
}

__global__ __launch_bounds__(256) void roi_pair_gather_kernel(RoiGradPack p, RoiPairIdx ix, int nsl)
{
    roi_pair_gather_block(p, ix, nsl, blockIdx.x, gridDim.x);
}


// the pair's forward: the multi-view pooling kernels with COMPACT argmax codes.  The first eight workgroups of the grid belong to the
// PLAN of the pair's backward launch (roi_grad_plan.h; eight so that the workgroup -> XCD slice mapping of the pooling workgroups is kept):
// workgroup 0 estimates every map tile's entry stream from the ROIs and writes the backward's work list -- hot tiles cut into sub-tiles,
// first -- into the unused quarter of view 0's argmax buffer, under the pooling workgroups' shadow; the other seven return at once.
struct RoiPlanArgs { RgtPack lay; int on, hot_entries, hot_max; };
#define PAIR_PLAN_BLOCKS 8u
union PairFwdLds {
    BinGeom g[4 * 32];
    struct { int heat[RGT_PLAN_TILES]; int scan[RGT_PLAN_SCAN(RGT_PLAN_THREADS)]; } plan;
};
template <int FWD_PASSES>
__global__ __launch_bounds__(256) void roi_pool_fwd_pair_kernel(RoiViewPack p, RoiPlanArgs pl)
{
    __shared__ PairFwdLds lds;
    unsigned blk = blockIdx.x;
    if (pl.on) {
        if (blk == 0) { rgt_plan_block<RGT_PLAN_THREADS>(pl.lay, const_cast<int4 *>(pl.lay.work), const_cast<int *>(pl.lay.n_work), pl.hot_entries, pl.hot_max, lds.plan.heat, lds.plan.scan); return; }
        if (blk < PAIR_PLAN_BLOCKS) return;
        blk -= PAIR_PLAN_BLOCKS;
    }
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && blk >= p.v[j].first_block) k = j;
    const RoiViewDev &v = p.v[k];
    roi_pool_fwd_xcd_block<FWD_PASSES, true>(lds.g, blk - v.first_block, v.data, v.scale, v.B, v.R, v.H, v.W, v.C, p.PH, p.PW, v.rois, v.top,
                                             v.argmax, v.tpb_shift);
}

template <int FWD_PASSES>
__global__ __launch_bounds__(256) void roi_pool_fwd_pair_cold_kernel(RoiViewPack p, RoiPrefetchPack pf, int *sink, RoiPlanArgs pl)
{
    __shared__ PairFwdLds lds;
    __shared__ unsigned s_mask;
    unsigned blk = blockIdx.x;
    if (pl.on) {
        if (blk == 0) { rgt_plan_block<RGT_PLAN_THREADS>(pl.lay, const_cast<int4 *>(pl.lay.work), const_cast<int *>(pl.lay.n_work), pl.hot_entries, pl.hot_max, lds.plan.heat, lds.plan.scan); return; }
        if (blk < PAIR_PLAN_BLOCKS) return;
        blk -= PAIR_PLAN_BLOCKS;
    }
    if (blk < pf.blocks) { roi_prefetch_block(&s_mask, p, pf, blk, sink); return; }
    blk -= pf.blocks;
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && blk >= p.v[j].first_block) k = j;
    const RoiViewDev &v = p.v[k];
    roi_pool_fwd_xcd_block<FWD_PASSES, true>(lds.g, blk - v.first_block, v.data, v.scale, v.B, v.R, v.H, v.W, v.C, p.PH, p.PW, v.rois, v.top,
                                             v.argmax, v.tpb_shift);
}

// the pair's private argmax planes (byte codes; 16-bit codes for bins of more than 255 pixels) -> the reference's int32 plane (flat
// index inside the frame, -1 for none); tests and bench.py's verification only.  One thread per pooled value; the bin geometry as
// the forward computes it.
__global__ __launch_bounds__(256) void roi_argmax_decode_kernel(const unsigned char *__restrict__ plane8, const float *__restrict__ rois,
                                                                float scale, int R, int H, int W, int C, int PH, int PW, int *__restrict__ out)
{
    const long long total = (long long)R * PH * PW * C;
    const unsigned short *const plane16 = reinterpret_cast<const unsigned short *>(plane8 + total);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH);
        const int n = (int)(t / PH);
        const float *roi = rois + 5 * (long long)n;
        const RoiGeom q = roi_geom(roi, scale);
        const int rw = max(q.rew - q.rsw + 1, 1), rh = max(q.reh - q.rsh + 1, 1);
        const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
        const int hs = min(max((int)floorf(__fmul_rn((float)ph, bh)) + q.rsh, 0), H);
        const int he = min(max((int)ceilf(__fmul_rn((float)(ph + 1), bh)) + q.rsh, 0), H);
        const int ws = min(max((int)floorf(__fmul_rn((float)pw, bw)) + q.rsw, 0), W);
        const int we = min(max((int)ceilf(__fmul_rn((float)(pw + 1), bw)) + q.rsw, 0), W);
        const bool big = he > hs && we > ws && (he - hs) * (we - ws) > 255;
        const unsigned code = big ? (unsigned)plane16[i] : (unsigned)plane8[i];
        int res = -1;
        if (code != (big ? 0xffffu : 0xffu)) {
            const int bwid = max(we - ws, 1);
            res = ((hs + (int)code / bwid) * W + ws + (int)code % bwid) * C + c;
        }
        out[i] = res;
    }
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }
static bool roi_prefetch_plan(int num_views, const mv3d_roi_view *views, RoiPrefetchPack &pf);


// Bins per workgroup (passes x bins per pass).  A small job (one frame: 2 x 14 700 bins) wants many
// workgroups in flight -- 2 passes (4 passes cost 5 % at batch 4); a batch of 16 frames runs ~6 % faster
// with 4 (interleaved A/B runs; the run-to-run spread at that size is larger than the effect).
static int fwd_passes(long long total_bins) { return total_bins > 8 * 29400 ? 4 : 2; }

extern "C" int mv3d_roi_pool_forward(const float *bottom_data, float spatial_scale, int batch_size, int num_rois,
                                     int height, int width, int channels, int pooled_height, int pooled_width,
                                     const float *bottom_rois, float *top_data, int32_t *argmax_data, void *stream)
{
    if (batch_size <= 0 || num_rois < 0 || height <= 0 || width <= 0 || channels <= 0 || pooled_height <= 0 ||
        pooled_width <= 0 || !bottom_data || (num_rois > 0 && (!bottom_rois || !top_data)))
        return MV3D_ERR_INVALID_ARG;                         // no ROIs: empty outputs, their pointers may be NULL
    if ((long long)height * width * channels > 0x7fffffffLL) return MV3D_ERR_INVALID_ARG;   // argmax is i32
    if ((long long)num_rois * pooled_height * pooled_width > 0x7fffffffLL) return MV3D_ERR_INVALID_ARG;
    if (num_rois == 0) return MV3D_OK;
    const bool v4 = (channels % 4 == 0) && aligned16(bottom_data) && aligned16(top_data) &&
                    (!argmax_data || aligned16(argmax_data));
    const long long items = (long long)num_rois * pooled_height * pooled_width * (channels / (v4 ? 4 : 1));
    long long blocks = (items + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;            // grid-stride the rest
    hipStream_t s = (hipStream_t)stream;
    const int cv4 = channels / 4;
    if (v4 && (cv4 == 64 || cv4 == 128 || cv4 == 256)) {
        const int tpb_shift = cv4 == 64 ? 3 : (cv4 == 128 ? 4 : 5);      // threads per bin = cv4 / 8
        const long long nbins = (long long)num_rois * pooled_height * pooled_width;
        const int passes = fwd_passes(nbins);
        const long long per_block = (long long)passes * (256 >> tpb_shift);
        const long long groups = (nbins + per_block - 1) / per_block;
        if (passes == 4)
            hipLaunchKernelGGL(roi_pool_fwd_xcd_kernel<4>, dim3((unsigned)(groups * 8)), dim3(256), 0, s, bottom_data,
                               spatial_scale, batch_size, num_rois, height, width, channels, pooled_height, pooled_width,
                               bottom_rois, top_data, argmax_data, tpb_shift);
        else
            hipLaunchKernelGGL(roi_pool_fwd_xcd_kernel<2>, dim3((unsigned)(groups * 8)), dim3(256), 0, s, bottom_data,
                               spatial_scale, batch_size, num_rois, height, width, channels, pooled_height, pooled_width,
                               bottom_rois, top_data, argmax_data, tpb_shift);
        return mv3d_launch_status();
    }
    if (v4)
        hipLaunchKernelGGL(roi_pool_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, bottom_data, spatial_scale,
                           batch_size, num_rois, height, width, channels, pooled_height, pooled_width, bottom_rois,
                           top_data, argmax_data);
    else
        hipLaunchKernelGGL(roi_pool_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, bottom_data, spatial_scale,
                           batch_size, num_rois, height, width, channels, pooled_height, pooled_width, bottom_rois,
                           top_data, argmax_data);
    return mv3d_launch_status();
}

static int roi_pool_backward_generic(const float *top_diff, float spatial_scale, int batch_size, int num_rois,
                                     int height, int width, int channels, int pooled_height, int pooled_width,
                                     const float *bottom_rois, float *bottom_diff, const int32_t *argmax_data,
                                     void *stream)
{
    if (batch_size <= 0 || num_rois < 0 || height <= 0 || width <= 0 || channels <= 0 || pooled_height <= 0 ||
        pooled_width <= 0 || !bottom_diff || (num_rois > 0 && (!bottom_rois || !top_diff || !argmax_data)))
        return MV3D_ERR_INVALID_ARG;
    if ((long long)height * width * channels > 0x7fffffffLL) return MV3D_ERR_INVALID_ARG;
    const long long pixels = (long long)batch_size * height * width;
    if (pixels > 0x7fffffffLL) return MV3D_ERR_INVALID_ARG;
    const bool v4 = (channels % 4 == 0) && aligned16(top_diff) && aligned16(bottom_diff) && aligned16(argmax_data);
    const int cv = channels / (v4 ? 4 : 1);
    if (cv > 64 * 16) return MV3D_ERR_INVALID_ARG;       // C <= 4096 (vectorised) / 1024 (scalar)
    const int nacc = cv <= 64 ? 1 : (cv <= 128 ? 2 : (cv <= 256 ? 4 : (cv <= 512 ? 8 : 16)));
    const unsigned blocks = (unsigned)((pixels + BWD_PIX - 1) / BWD_PIX);
    hipStream_t s = (hipStream_t)stream;
#define MV3D_BWD(V, N)                                                                                             \
    hipLaunchKernelGGL((roi_pool_bwd_kernel<V, N>), dim3(blocks), dim3(256), 0, s, top_diff, spatial_scale, batch_size, \
                       num_rois, height, width, channels, pooled_height, pooled_width, bottom_rois, bottom_diff,   \
                       argmax_data)
    if (v4) {
        switch (nacc) { case 1: MV3D_BWD(4, 1); break; case 2: MV3D_BWD(4, 2); break; case 4: MV3D_BWD(4, 4); break;
                        case 8: MV3D_BWD(4, 8); break; default: MV3D_BWD(4, 16); break; }
    } else {
        switch (nacc) { case 1: MV3D_BWD(1, 1); break; case 2: MV3D_BWD(1, 2); break; case 4: MV3D_BWD(1, 4); break;
                        case 8: MV3D_BWD(1, 8); break; default: MV3D_BWD(1, 16); break; }
    }
#undef MV3D_BWD
    return mv3d_launch_status();
}

static int roi_pool_forward_views_impl(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                       bool cold, void *stream, int top_type = 0)
{
    if (num_views <= 0 || num_views > MV3D_MAX_ROI_VIEWS || !views || pooled_height <= 0 || pooled_width <= 0)
        return MV3D_ERR_INVALID_ARG;
    bool fast = true;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_view &w = views[k];
        if (w.batch_size <= 0 || w.num_rois < 0 || w.height <= 0 || w.width <= 0 || w.channels <= 0 || !w.bottom_data ||
            (w.num_rois > 0 && (!w.bottom_rois || !w.top_data)) ||
            (long long)w.num_rois * pooled_height * pooled_width > 0x7fffffffLL)
            return MV3D_ERR_INVALID_ARG;
        const int cv4 = w.channels / 4;
        fast = fast && (w.channels % 4 == 0) && (cv4 == 64 || cv4 == 128 || cv4 == 256) && aligned16(w.bottom_data) &&
               aligned16(w.top_data) && (!w.argmax_data || aligned16(w.argmax_data));
    }
    if (top_type != 0) {                                  // 16-bit tops: the XCD-sliced maximum-only kernels or nothing
        for (int k = 0; k < num_views; ++k)
            if (views[k].argmax_data) return MV3D_ERR_INVALID_ARG;
        if (!fast || (top_type != 1 && top_type != 2)) return MV3D_ERR_INVALID_ARG;
    }
    if (!fast) {                                          // generic shapes: one launch per view
        for (int k = 0; k < num_views; ++k) {
            const mv3d_roi_view &w = views[k];
            const int rc = mv3d_roi_pool_forward(w.bottom_data, w.spatial_scale, w.batch_size, w.num_rois, w.height, w.width,
                                                 w.channels, pooled_height, pooled_width, w.bottom_rois, w.top_data,
                                                 w.argmax_data, stream);
            if (rc != MV3D_OK) return rc;
        }
        return MV3D_OK;
    }
    RoiViewPack p;
    p.n = num_views; p.PH = pooled_height; p.PW = pooled_width;
    unsigned blocks = 0;
    long long total_bins = 0;
    for (int k = 0; k < num_views; ++k) total_bins += (long long)views[k].num_rois * pooled_height * pooled_width;
    const int passes = fwd_passes(total_bins);
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_view &w = views[k];
        const int cv4 = w.channels / 4;
        RoiViewDev &v = p.v[k];
        v.data = w.bottom_data; v.rois = w.bottom_rois; v.top = w.top_data; v.argmax = w.argmax_data; v.scale = w.spatial_scale;
        v.B = w.batch_size; v.R = w.num_rois; v.H = w.height; v.W = w.width; v.C = w.channels;
        v.tpb_shift = cv4 == 64 ? 3 : (cv4 == 128 ? 4 : 5);
        v.first_block = blocks;
        const long long nbins = (long long)w.num_rois * pooled_height * pooled_width;
        const long long per_block = (long long)passes * (256 >> v.tpb_shift);
        blocks += (unsigned)(((nbins + per_block - 1) / per_block) * 8);
    }
    for (int k = num_views; k < MV3D_MAX_ROI_VIEWS; ++k) p.v[k] = p.v[0];
    if (blocks == 0) return MV3D_OK;
    bool noarg = true;                                    // inference: no view wants the argmax plane -> the maximum-only scan
    for (int k = 0; k < num_views; ++k) noarg = noarg && !views[k].argmax_data;
    hipStream_t s = (hipStream_t)stream;
    RoiPrefetchPack pf;
    if (cold && roi_prefetch_plan(num_views, views, pf)) {
        pf.blocks = (pf.blocks + 7u) & ~7u;
        const dim3 grid(blocks + pf.blocks);
        if (top_type == 1) {
            if (passes == 4) hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_cold_kernel<4, true, 1>), grid, dim3(256), 0, s, p, pf, (int *)nullptr);
            else hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_cold_kernel<2, true, 1>), grid, dim3(256), 0, s, p, pf, (int *)nullptr);
        } else if (top_type == 2) {
            if (passes == 4) hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_cold_kernel<4, true, 2>), grid, dim3(256), 0, s, p, pf, (int *)nullptr);
            else hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_cold_kernel<2, true, 2>), grid, dim3(256), 0, s, p, pf, (int *)nullptr);
        } else if (noarg) {
            if (passes == 4) hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_cold_kernel<4, true>), grid, dim3(256), 0, s, p, pf, (int *)nullptr);
            else hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_cold_kernel<2, true>), grid, dim3(256), 0, s, p, pf, (int *)nullptr);
        } else {
            if (passes == 4) hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_cold_kernel<4>), grid, dim3(256), 0, s, p, pf, (int *)nullptr);
            else hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_cold_kernel<2>), grid, dim3(256), 0, s, p, pf, (int *)nullptr);
        }
        return mv3d_launch_status();
    }
    if (top_type == 1) {
        if (passes == 4) hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_kernel<4, true, 1>), dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_kernel<2, true, 1>), dim3(blocks), dim3(256), 0, s, p);
    } else if (top_type == 2) {
        if (passes == 4) hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_kernel<4, true, 2>), dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_kernel<2, true, 2>), dim3(blocks), dim3(256), 0, s, p);
    } else if (noarg) {
        if (passes == 4) hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_kernel<4, true>), dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_kernel<2, true>), dim3(blocks), dim3(256), 0, s, p);
    } else {
        if (passes == 4) hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_kernel<4>), dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((roi_pool_fwd_xcd_multi_kernel<2>), dim3(blocks), dim3(256), 0, s, p);
    }
    return mv3d_launch_status();
}

extern "C" int mv3d_roi_pool_forward_views(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                           void *stream)
{
    return roi_pool_forward_views_impl(num_views, views, pooled_height, pooled_width, false, stream);
}

extern "C" int mv3d_roi_pool_forward_views_cold(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                                void *stream)
{
    return roi_pool_forward_views_impl(num_views, views, pooled_height, pooled_width, true, stream);
}

extern "C" int mv3d_roi_pool_forward_views_half(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                                int top_type, int cold_maps, void *stream)
{
    if (top_type != 1 && top_type != 2) return MV3D_ERR_INVALID_ARG;
    return roi_pool_forward_views_impl(num_views, views, pooled_height, pooled_width, cold_maps != 0, stream, top_type);
}

static bool roi_prefetch_plan(int num_views, const mv3d_roi_view *views, RoiPrefetchPack &pf)
{
    unsigned blocks = 0;
    for (int k = 0; k < MV3D_MAX_ROI_VIEWS; ++k) pf.first_block[k] = 0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_view &w = views[k];
        if (w.channels % 4 != 0 || !aligned16(w.bottom_data)) return false;             // a hint: nothing to do for odd layouts
        pf.first_block[k] = blocks;
        if (w.num_rois > 0) blocks += (unsigned)((long long)w.batch_size * w.height * ((w.width + PF_PIX - 1) / PF_PIX));
    }
    pf.blocks = blocks;
    return blocks > 0;
}

static bool bwd_fast_ok(int channels, int pooled_height, int pooled_width, int height, int width, int batch_size)
{
    return channels % 64 == 0 && channels / 64 <= 64 && (long long)pooled_height * pooled_width <= BW_CAND &&
           pooled_height < 256 && pooled_width < 256 && (long long)batch_size * height * width < (1ll << 26);
}

static size_t bwd_pool_entries(const mv3d_roi_grad_view &w, int PH, int PW)
{
    return (size_t)w.num_rois * (size_t)(PH + 2 * w.height + 3) * (size_t)(PW + 2 * w.width + 3);
}

static unsigned bwd_segments(const mv3d_roi_grad_view &w) { return (unsigned)((long long)w.batch_size * w.height * ((w.width + BWI_PIX - 1) / BWI_PIX)); }

// The candidate index of RoiPoolGrad (lists of (roi, bin) records per pixel) and who may use it: same C for all views, C in
// {64, 128, 256, 512}, pooled sizes <= 15, 16-byte aligned buffers where given, everything inside 31-bit offsets.
static bool bwd_index_eligible(int num_views, const mv3d_roi_grad_view *views, int PH, int PW)
{
    long long all_pix = 0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[k];
        if (!bwd_fast_ok(w.channels, PH, PW, w.height, w.width, w.batch_size)) return false;
        // (64-channel slices, one per XCD or XCD group: C = 64, 128, 256 or 512; other widths take the sliced kernel)
        if (w.channels != views[0].channels || w.channels % 64 != 0 || w.channels > 512 || 512 % w.channels != 0) return false;
        if ((w.bottom_diff && !aligned16(w.bottom_diff)) || (w.top_diff && !aligned16(w.top_diff)) || (w.argmax_data && !aligned16(w.argmax_data)))
            return false;
        // record byte offsets are 31-bit scalars in the gather kernel
        if ((long long)w.num_rois * PH * PW * w.channels * 4 >= 0x7fffffffLL) return false;
        all_pix += (long long)w.batch_size * w.height * w.width;
    }
    return all_pix < 0x7fffffffLL && PH <= 15 && PW <= 15 && PH * PW <= 255;
}

extern "C" size_t mv3d_roi_pool_backward_workspace_bytes(int num_views, const mv3d_roi_grad_view *views, int pooled_height,
                                                         int pooled_width)
{
    if (num_views <= 0 || num_views > MV3D_MAX_ROI_VIEWS || !views || pooled_height <= 0 || pooled_width <= 0) return 0;
    size_t total = MV3D_ALIGN, nseg = 0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[k];
        if (w.batch_size <= 0 || w.height <= 0 || w.width <= 0 || w.num_rois < 0) return 0;
        total += mv3d_align_up((size_t)w.batch_size * w.height * w.width * sizeof(int4));
        total += mv3d_align_up(bwd_pool_entries(w, pooled_height, pooled_width) * sizeof(int));
        nseg += bwd_segments(w);
    }
    // header | candidates per segment | mask per segment | items | pool
    return total + 2 * mv3d_align_up(nseg * sizeof(int)) + MV3D_ALIGN;
}

// The views in index order (the densest view first -- candidates per pixel ~ R PH PW / pixels: the 8 x 64 front-view map carries
// ~10 x the lists of the others -- so that its pixels head the item list and the gather's waves start with the long lists instead
// of ending with them: the kernel's time is its slowest wave) and the segment numbering
struct BwdOrder { RoiGradPack p; unsigned first_block[MV3D_MAX_ROI_VIEWS]; int gpr[MV3D_MAX_ROI_VIEWS]; unsigned iblocks; size_t n_items, pool_entries; };
static void bwd_order(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, BwdOrder &o)
{
    RoiGradPack &p = o.p;
    p.n = num_views; p.PH = PH; p.PW = PW;
    int ord[MV3D_MAX_ROI_VIEWS];
    for (int k = 0; k < MV3D_MAX_ROI_VIEWS; ++k) ord[k] = k;
    for (int a = 0; a < num_views; ++a)
        for (int b = a + 1; b < num_views; ++b) {
            const mv3d_roi_grad_view &x = views[ord[a]], &y = views[ord[b]];
            const double dx = (double)x.num_rois / ((double)x.batch_size * x.height * x.width);
            const double dy = (double)y.num_rois / ((double)y.batch_size * y.height * y.width);
            if (dy > dx) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
        }
    o.iblocks = 0; o.n_items = 0; o.pool_entries = 0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[ord[k]];
        RoiGradViewDev &v = p.v[k];
        v.top_diff = w.top_diff; v.rois = w.bottom_rois; v.argmax = w.argmax_data; v.bottom_diff = w.bottom_diff;
        v.scale = w.spatial_scale; v.B = w.batch_size; v.R = w.num_rois; v.H = w.height; v.W = w.width; v.C = w.channels;
        v.nsl = w.channels / 64;
        v.first_block = 0; v.pxg = 4; v.gpr = 0;
        o.n_items += (size_t)w.batch_size * w.height * w.width;
        o.first_block[k] = o.iblocks;
        o.gpr[k] = (w.width + BWI_PIX - 1) / BWI_PIX;
        o.iblocks += bwd_segments(w);
        o.pool_entries += bwd_pool_entries(w, PH, PW);
    }
    for (int k = num_views; k < MV3D_MAX_ROI_VIEWS; ++k) { p.v[k] = p.v[0]; o.first_block[k] = 0; o.gpr[k] = 1; }
}

struct BwdIndexPlan { RoiGradPack p; RoiGradIdxPack ix; unsigned iblocks; };
static int bwd_index_plan(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, void *workspace, BwdIndexPlan &pl)
{
    BwdOrder o;
    bwd_order(num_views, views, PH, PW, o);
    if (o.pool_entries > 0x7fffffffull) return MV3D_ERR_INVALID_ARG;
    pl.p = o.p;
    RoiGradIdxPack &ix = pl.ix;
    ix = RoiGradIdxPack{};
    for (int k = 0; k < MV3D_MAX_ROI_VIEWS; ++k) { ix.first_block[k] = o.first_block[k]; ix.gpr[k] = o.gpr[k]; }
    char *ws = (char *)workspace;
    size_t off = MV3D_ALIGN;
    ix.header = (int *)ws;
    ix.seg_tot = (int *)(ws + off); off += mv3d_align_up((size_t)o.iblocks * sizeof(int));
    ix.seg_mask = (int *)(ws + off); off += mv3d_align_up((size_t)o.iblocks * sizeof(int));
    ix.items = (int4 *)(ws + off); off += mv3d_align_up(o.n_items * sizeof(int4));
    ix.pool = (int *)(ws + off);
    ix.trace = nullptr;
#ifdef MV3D_TUNING                                                     // diagnostics (tools/roi_bwd_trace.py), experiment builds only
    ix.trace = getenv("MV3D_BWD_TRACE") ? (long long *)strtoull(getenv("MV3D_BWD_TRACE"), nullptr, 10) : nullptr;
#endif
    pl.iblocks = o.iblocks;
    return MV3D_OK;
}

// the pair's workspace: header | upper-bound sizes per segment | masks per segment | items | pool of {record offset, code} entries
static size_t roi_pair_workspace_bytes(int num_views, const mv3d_roi_grad_view *views, int PH, int PW)
{
    size_t total = MV3D_ALIGN, nseg = 0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[k];
        total += mv3d_align_up((size_t)w.batch_size * w.height * w.width * sizeof(int4));
        total += mv3d_align_up(bwd_pool_entries(w, PH, PW) * sizeof(int2));
        nseg += bwd_segments(w);
    }
    return total + 2 * mv3d_align_up(nseg * sizeof(int)) + MV3D_ALIGN;
}

struct RoiPairPlan { RoiGradPack p; RoiPairIdx ix; };
static int roi_pair_plan(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, void *workspace, RoiPairPlan &pl)
{
    BwdOrder o;
    bwd_order(num_views, views, PH, PW, o);
    if (o.pool_entries > 0x7fffffffull) return MV3D_ERR_INVALID_ARG;
    pl.p = o.p;
    RoiPairIdx &ix = pl.ix;
    ix = RoiPairIdx{};
    for (int k = 0; k < MV3D_MAX_ROI_VIEWS; ++k) { ix.first_block[k] = o.first_block[k]; ix.gpr[k] = o.gpr[k]; }
    char *ws = (char *)workspace;
    size_t off = MV3D_ALIGN;
    ix.header = (int *)ws;
    ix.seg_tot = (int *)(ws + off); off += mv3d_align_up((size_t)o.iblocks * sizeof(int));
    ix.seg_mask = (int *)(ws + off); off += mv3d_align_up((size_t)o.iblocks * sizeof(int));
    ix.items = (int4 *)(ws + off); off += mv3d_align_up(o.n_items * sizeof(int4));
    ix.pool = (int2 *)(ws + off);
    ix.nseg = o.iblocks;
    ix.nslots = (int)o.n_items;
    ix.dbg = 0;
    ix.trace = nullptr;
#ifdef MV3D_TUNING
    ix.dbg = getenv("MV3D_IDX_DBG") ? atoi(getenv("MV3D_IDX_DBG")) : 0;
    ix.trace = getenv("MV3D_IDX_TRACE") ? (long long *)strtoull(getenv("MV3D_IDX_TRACE"), nullptr, 10) : nullptr;
#endif
    return MV3D_OK;
}

// channels per lane of the gather: 1 (64-channel slices, 256-B pieces of a record per wave) measured best on the training batch:
// 75 us for the three launches vs 82 (2 channels, 512-B pieces) and 105 (4 channels, 1-KB pieces, 8 records in flight): the
// walk is bound by its dependent round trips, and a lane with more channels holds fewer records in flight
static int bwd_gather_cpl(int channels)
{
#ifdef MV3D_TUNING                                                     // tuning hooks, experiment builds only
    static const int cpl_env = getenv("MV3D_BWG_CPL") ? atoi(getenv("MV3D_BWG_CPL")) : 0;
#else
    const int cpl_env = 0;
#endif
    int cpl = 1;
    if (cpl_env == 2 && channels % 128 == 0) cpl = 2;
    if (cpl_env == 4 && channels % 256 == 0) cpl = 4;
    if (8 % (channels / (64 * cpl)) != 0 || channels / (64 * cpl) > 8) cpl = 0;     // (C = 64 k, k not a divisor of 8)
    return cpl;
}

static int bwd_gather_groups()
{
#ifdef MV3D_TUNING
    static const int groups = getenv("MV3D_BWG_GROUPS") ? atoi(getenv("MV3D_BWG_GROUPS")) : BWG_GROUPS;
    return groups;
#else
    return BWG_GROUPS;
#endif
}

static int grad_views_check(int num_views, const mv3d_roi_grad_view *views, int pooled_height, int pooled_width)
{
    if (num_views <= 0 || num_views > MV3D_MAX_ROI_VIEWS || !views || pooled_height <= 0 || pooled_width <= 0)
        return MV3D_ERR_INVALID_ARG;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[k];
        if (w.batch_size <= 0 || w.num_rois < 0 || w.height <= 0 || w.width <= 0 || w.channels <= 0 || !w.bottom_diff ||
            (w.num_rois > 0 && (!w.bottom_rois || !w.top_diff || !w.argmax_data)) ||
            (long long)w.height * w.width * w.channels > 0x7fffffffLL ||
            (long long)w.num_rois * pooled_height * pooled_width > 0x7fffffffLL)
            return MV3D_ERR_INVALID_ARG;
    }
    return MV3D_OK;
}

extern "C" int mv3d_roi_pool_backward_views(int num_views, const mv3d_roi_grad_view *views, int pooled_height, int pooled_width,
                                            void *workspace, size_t workspace_bytes, void *stream)
{
    const int rc0 = grad_views_check(num_views, views, pooled_height, pooled_width);
    if (rc0 != MV3D_OK) return rc0;
    bool fast = true;
    for (int k = 0; k < num_views; ++k)
        fast = fast && bwd_fast_ok(views[k].channels, pooled_height, pooled_width, views[k].height, views[k].width, views[k].batch_size);
    if (workspace && ((uintptr_t)workspace % MV3D_ALIGN)) return MV3D_ERR_WORKSPACE;
    if (!fast) {                                          // generic shapes: one launch of the generic kernel per view
        for (int k = 0; k < num_views; ++k) {
            const mv3d_roi_grad_view &w = views[k];
            const int rc = roi_pool_backward_generic(w.top_diff, w.spatial_scale, w.batch_size, w.num_rois, w.height, w.width,
                                                     w.channels, pooled_height, pooled_width, w.bottom_rois, w.bottom_diff,
                                                     w.argmax_data, stream);
            if (rc != MV3D_OK) return rc;
        }
        return MV3D_OK;
    }
    const bool indexed = workspace && bwd_index_eligible(num_views, views, pooled_height, pooled_width) &&
                         workspace_bytes >= mv3d_roi_pool_backward_workspace_bytes(num_views, views, pooled_height, pooled_width);
    if (indexed) {
        BwdIndexPlan pl;
        const int rc = bwd_index_plan(num_views, views, pooled_height, pooled_width, workspace, pl);
        if (rc != MV3D_OK) return rc;
        hipLaunchKernelGGL(roi_bwd_index_kernel<false>, dim3(pl.iblocks), dim3(256), 0, (hipStream_t)stream, pl.p, pl.ix);
        hipLaunchKernelGGL(roi_bwd_index_kernel<true>, dim3(pl.iblocks), dim3(256), 0, (hipStream_t)stream, pl.p, pl.ix);
        const int cpl = bwd_gather_cpl(views[0].channels);
        const dim3 gg((unsigned)(bwd_gather_groups() * 8));
        if (cpl == 4) hipLaunchKernelGGL(roi_bwd_gather_kernel<4>, gg, dim3(256), 0, (hipStream_t)stream, pl.p, pl.ix, views[0].channels / 256);
        else if (cpl == 2) hipLaunchKernelGGL(roi_bwd_gather_kernel<2>, gg, dim3(256), 0, (hipStream_t)stream, pl.p, pl.ix, views[0].channels / 128);
        else if (cpl == 1) hipLaunchKernelGGL(roi_bwd_gather_kernel<1>, gg, dim3(256), 0, (hipStream_t)stream, pl.p, pl.ix, views[0].channels / 64);
        else return MV3D_ERR_INVALID_ARG;
        return mv3d_launch_status();
    }
    RoiGradPack p;
    p.n = num_views; p.PH = pooled_height; p.PW = pooled_width;
    unsigned blocks = 0;
    size_t carry = 0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[k];
        RoiGradViewDev &v = p.v[k];
        v.top_diff = w.top_diff; v.rois = w.bottom_rois; v.argmax = w.argmax_data; v.bottom_diff = w.bottom_diff;
        v.scale = w.spatial_scale; v.B = w.batch_size; v.R = w.num_rois; v.H = w.height; v.W = w.width; v.C = w.channels;
        v.nsl = w.channels / 64;
        v.first_block = blocks;
        // sliced kernel: a workgroup = one row segment x one slice.  Long segments amortise the ROI filter and the launch
        // of a workgroup over many (mostly empty) pixels; small maps get short segments so that the launch still fills the chip
        const long long rows = (long long)w.batch_size * w.height * v.nsl;
        long long gpr = (w.width + BW_MAXPXG - 1) / BW_MAXPXG;
        if (rows * gpr < 1024) gpr = (1024 + rows - 1) / rows;
        if (gpr > w.width) gpr = w.width;
        v.pxg = (int)((w.width + gpr - 1) / gpr);
        v.gpr = (w.width + v.pxg - 1) / v.pxg;
        blocks += (unsigned)(rows * v.gpr);
        if (w.num_rois > BW_CHUNK && (size_t)v.pxg * 64 * sizeof(float) > carry) carry = (size_t)v.pxg * 64 * sizeof(float);
    }
    for (int k = num_views; k < MV3D_MAX_ROI_VIEWS; ++k) p.v[k] = p.v[0];
    hipLaunchKernelGGL(roi_pool_bwd_sliced_kernel, dim3(blocks), dim3(BW_THREADS), carry, (hipStream_t)stream, p);
    return mv3d_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// The pair: RoiPool with a private compact argmax plane, RoiPoolGrad = index + zero fill (one launch) and gather (one launch).
static void grad_views_of(int num_views, const mv3d_roi_view *views, mv3d_roi_grad_view *g)
{
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_view &w = views[k];
        g[k].bottom_diff = nullptr; g[k].bottom_rois = w.bottom_rois; g[k].top_diff = nullptr; g[k].argmax_data = nullptr;
        g[k].spatial_scale = w.spatial_scale; g[k].batch_size = w.batch_size; g[k].num_rois = w.num_rois; g[k].height = w.height;
        g[k].width = w.width; g[k].channels = w.channels;
    }
}

// Do these views go through the pair's own kernels (compact argmax codes)?  Decided from the SHAPES alone (channel widths the
// XCD-sliced forward handles, at least one ROI per view, the index's own conditions, bins whose scan positions fit 16 bits), so that
// the forward, the backward and the decode entries take the same decision independently.
static bool roi_pair_shapes(int num_views, const mv3d_roi_grad_view *g, int PH, int PW)
{
    if (!bwd_index_eligible(num_views, g, PH, PW)) return false;
    for (int k = 0; k < num_views; ++k) {
        const int cv4 = g[k].channels / 4;
        if (g[k].channels % 4 != 0 || !(cv4 == 64 || cv4 == 128 || cv4 == 256) || g[k].num_rois <= 0) return false;
        if ((long long)g[k].height * g[k].width > 0xfffeLL) return false;           // a bin never has more scan positions than the map has pixels
    }
    return true;
}

extern "C" size_t mv3d_roi_pool_pair_workspace_bytes(int num_views, const mv3d_roi_grad_view *views, int pooled_height, int pooled_width)
{
    const size_t plain = mv3d_roi_pool_backward_workspace_bytes(num_views, views, pooled_height, pooled_width);      // (validates)
    if (plain == 0) return 0;
    const size_t pair = roi_pair_workspace_bytes(num_views, views, pooled_height, pooled_width);
    return pair > plain ? pair : plain;                   // (shapes outside the pair's kernels use the plain layout in the same buffer)
}

extern "C" int mv3d_roi_pool_forward_views_pair(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                                int cold_maps, void *stream)
{
    if (num_views <= 0 || num_views > MV3D_MAX_ROI_VIEWS || !views || pooled_height <= 0 || pooled_width <= 0) return MV3D_ERR_INVALID_ARG;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_view &w = views[k];
        if (w.batch_size <= 0 || w.num_rois < 0 || w.height <= 0 || w.width <= 0 || w.channels <= 0 || !w.bottom_data ||
            (w.num_rois > 0 && (!w.bottom_rois || !w.top_data || !w.argmax_data)) ||
            (long long)w.num_rois * pooled_height * pooled_width > 0x7fffffffLL)
            return MV3D_ERR_INVALID_ARG;
    }
    mv3d_roi_grad_view g[MV3D_MAX_ROI_VIEWS];
    grad_views_of(num_views, views, g);
    if (!roi_pair_shapes(num_views, g, pooled_height, pooled_width))
        // shapes outside the pair's kernels: the plain forward (int32 argmax plane); mv3d_roi_pool_backward_views_pair takes the same
        // decision from the same shapes
        return roi_pool_forward_views_impl(num_views, views, pooled_height, pooled_width, cold_maps != 0, stream);
    for (int k = 0; k < num_views; ++k)                    // (pointer conditions: a caller that misses them gets an error, not a silent other path)
        if (!aligned16(views[k].bottom_data) || !aligned16(views[k].top_data) || !aligned16(views[k].argmax_data)) return MV3D_ERR_INVALID_ARG;
    RoiViewPack p;
    p.n = num_views; p.PH = pooled_height; p.PW = pooled_width;
    unsigned blocks = 0;
    long long total_bins = 0;
    for (int k = 0; k < num_views; ++k) total_bins += (long long)views[k].num_rois * pooled_height * pooled_width;
    const int passes = fwd_passes(total_bins);
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_view &w = views[k];
        const int cv4 = w.channels / 4;
        RoiViewDev &v = p.v[k];
        v.data = w.bottom_data; v.rois = w.bottom_rois; v.top = w.top_data; v.argmax = w.argmax_data; v.scale = w.spatial_scale;
        v.B = w.batch_size; v.R = w.num_rois; v.H = w.height; v.W = w.width; v.C = w.channels;
        v.tpb_shift = cv4 == 64 ? 3 : (cv4 == 128 ? 4 : 5);
        v.first_block = blocks;
        const long long nbins = (long long)w.num_rois * pooled_height * pooled_width;
        const long long per_block = (long long)passes * (256 >> v.tpb_shift);
        blocks += (unsigned)(((nbins + per_block - 1) / per_block) * 8);
    }
    for (int k = num_views; k < MV3D_MAX_ROI_VIEWS; ++k) p.v[k] = p.v[0];
    hipStream_t s = (hipStream_t)stream;
    // the plan of the pair's backward launch (the decision mv3d_launch_roi_pair_tiles takes from the same shapes)
    RoiPlanArgs pl;
    for (int k = 0; k < num_views; ++k) g[k].argmax_data = views[k].argmax_data;
    unsigned tile_blocks = 0, plan_blocks = 0;
    bool planned = false;
    if (!mv3d_rgt_layout(num_views, g, pooled_height, pooled_width, pl.lay, &tile_blocks, &planned, &plan_blocks, &pl.hot_entries, &pl.hot_max)) planned = false;
    pl.on = planned ? 1 : 0;
    const unsigned front = planned ? PAIR_PLAN_BLOCKS : 0u;
    RoiPrefetchPack pf;
    if (cold_maps && roi_prefetch_plan(num_views, views, pf)) {
        pf.blocks = (pf.blocks + 7u) & ~7u;
        if (passes == 4) hipLaunchKernelGGL(roi_pool_fwd_pair_cold_kernel<4>, dim3(front + blocks + pf.blocks), dim3(256), 0, s, p, pf, (int *)nullptr, pl);
        else hipLaunchKernelGGL(roi_pool_fwd_pair_cold_kernel<2>, dim3(front + blocks + pf.blocks), dim3(256), 0, s, p, pf, (int *)nullptr, pl);
        return mv3d_launch_status();
    }
    if (passes == 4) hipLaunchKernelGGL(roi_pool_fwd_pair_kernel<4>, dim3(front + blocks), dim3(256), 0, s, p, pl);
    else hipLaunchKernelGGL(roi_pool_fwd_pair_kernel<2>, dim3(front + blocks), dim3(256), 0, s, p, pl);
    return mv3d_launch_status();
}

extern "C" int mv3d_roi_pool_backward_views_pair(int num_views, const mv3d_roi_grad_view *views, int pooled_height, int pooled_width,
                                                 void *workspace, size_t workspace_bytes, void *stream)
{
    const int rc0 = grad_views_check(num_views, views, pooled_height, pooled_width);
    if (rc0 != MV3D_OK) return rc0;
    if (workspace && ((uintptr_t)workspace % MV3D_ALIGN)) return MV3D_ERR_WORKSPACE;
    // the decision mv3d_roi_pool_forward_views_pair took from the same shapes: compact codes in argmax_data?
    if (!roi_pair_shapes(num_views, views, pooled_height, pooled_width))
        return mv3d_roi_pool_backward_views(num_views, views, pooled_height, pooled_width, workspace, workspace_bytes, stream);
    for (int k = 0; k < num_views; ++k)
        if (!aligned16(views[k].bottom_diff) || !aligned16(views[k].top_diff) || !aligned16(views[k].argmax_data)) return MV3D_ERR_INVALID_ARG;
#ifdef MV3D_TUNING
    static const int tiles_env = getenv("MV3D_PAIR_TILES") ? atoi(getenv("MV3D_PAIR_TILES")) : -1;
#else
    const int tiles_env = -1;
#endif
    // ONE launch of map tiles, no index, no fill, no scratch memory (roi_grad_tiles.hip) -- with or without a workspace: the round-5
    // structure behind a workspace (index + zero fill, gather: three launches) is level alone and 3 - 7 % slower in the path, and a
    // planning launch that cuts the hot tiles costs what it saves (profiles/r06_g).  (Experiment builds, MV3D_PAIR_TILES=0: index + gather.)
    if (tiles_env != 0)
        return mv3d_launch_roi_pair_tiles(num_views, views, pooled_height, pooled_width, (hipStream_t)stream);
    if (!workspace) return MV3D_ERR_WORKSPACE;
    if (workspace_bytes < roi_pair_workspace_bytes(num_views, views, pooled_height, pooled_width)) return MV3D_ERR_WORKSPACE;
    RoiPairPlan pl;
    const int rc = roi_pair_plan(num_views, views, pooled_height, pooled_width, workspace, pl);
    if (rc != MV3D_OK) return rc;
    hipLaunchKernelGGL(roi_pair_index_kernel<false>, dim3(pl.ix.nseg), dim3(256), 0, (hipStream_t)stream, pl.p, pl.ix);
    hipLaunchKernelGGL(roi_pair_index_kernel<true>, dim3(pl.ix.nseg), dim3(256), 0, (hipStream_t)stream, pl.p, pl.ix);
    hipLaunchKernelGGL(roi_pair_gather_kernel, dim3((unsigned)(bwd_gather_groups() * 8)), dim3(256), 0, (hipStream_t)stream, pl.p, pl.ix,
                       views[0].channels / 64);
    return mv3d_launch_status();
}

// The argmax plane a forward of the pair left in `argmax_data` -> the reference's int32 plane (num_rois, PH, PW, C) in `argmax_out`
// (a buffer of its own).  For views outside the pair's kernels the plane already IS the reference's: copied.  Tests / verification:
// the pair itself never needs it.
extern "C" int mv3d_roi_pool_argmax_decode(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                           int32_t *const *argmax_out, void *stream)
{
    if (num_views <= 0 || num_views > MV3D_MAX_ROI_VIEWS || !views || pooled_height <= 0 || pooled_width <= 0 || !argmax_out)
        return MV3D_ERR_INVALID_ARG;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_view &w = views[k];
        if (w.batch_size <= 0 || w.num_rois < 0 || w.height <= 0 || w.width <= 0 || w.channels <= 0 ||
            (w.num_rois > 0 && (!w.bottom_rois || !w.argmax_data || !argmax_out[k])))
            return MV3D_ERR_INVALID_ARG;
    }
    mv3d_roi_grad_view g[MV3D_MAX_ROI_VIEWS];
    grad_views_of(num_views, views, g);
    const bool pair = roi_pair_shapes(num_views, g, pooled_height, pooled_width);
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_view &w = views[k];
        const long long total = (long long)w.num_rois * pooled_height * pooled_width * w.channels;
        if (total == 0) continue;
        if (!pair) {
            if (argmax_out[k] != w.argmax_data &&
                hipMemcpyAsync(argmax_out[k], w.argmax_data, (size_t)total * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                return MV3D_ERR_HIP;
            continue;
        }
        if ((void *)argmax_out[k] == (void *)w.argmax_data) return MV3D_ERR_INVALID_ARG;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        hipLaunchKernelGGL(roi_argmax_decode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char *)w.argmax_data,
                           w.bottom_rois, w.spatial_scale, w.num_rois, w.height, w.width, w.channels, pooled_height, pooled_width, argmax_out[k]);
    }
    return mv3d_launch_status();
}

extern "C" int mv3d_roi_pool_backward(const float *top_diff, float spatial_scale, int batch_size, int num_rois,
                                      int height, int width, int channels, int pooled_height, int pooled_width,
                                      const float *bottom_rois, float *bottom_diff, const int32_t *argmax_data,
                                      void *stream)
{
    mv3d_roi_grad_view w;
    w.bottom_diff = bottom_diff; w.bottom_rois = bottom_rois; w.top_diff = top_diff; w.argmax_data = argmax_data;
    w.spatial_scale = spatial_scale; w.batch_size = batch_size; w.num_rois = num_rois; w.height = height; w.width = width;
    w.channels = channels;
    return mv3d_roi_pool_backward_views(1, &w, pooled_height, pooled_width, nullptr, 0, stream);
}

// roi_pooling_op_gpu.h:18-27, argument for argument: 1 = launched, 0 = refused / launch error
extern "C" int mv3d_ROIPoolForwardLaucher(const float *bottom_data, float spatial_scale, int num_rois, int height, int width,
                                          int channels, int pooled_height, int pooled_width, const float *bottom_rois,
                                          float *top_data, int32_t *argmax_data, void *stream)
{
    return mv3d_roi_pool_forward(bottom_data, spatial_scale, 0x7fffffff, num_rois, height, width, channels, pooled_height, pooled_width,
                                 bottom_rois, top_data, argmax_data, stream) == MV3D_OK;
}

extern "C" int mv3d_ROIPoolBackwardLaucher(const float *top_diff, float spatial_scale, int batch_size, int num_rois, int height,
                                           int width, int channels, int pooled_height, int pooled_width, const float *bottom_rois,
                                           float *bottom_diff, const int32_t *argmax_data, void *stream)
{
    return mv3d_roi_pool_backward(top_diff, spatial_scale, batch_size, num_rois, height, width, channels, pooled_height, pooled_width,
                                  bottom_rois, bottom_diff, argmax_data, stream) == MV3D_OK;
}
