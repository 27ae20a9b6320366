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
#include <math.h>
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

template <int FWD_PASSES>
__device__ __forceinline__ void roi_pool_fwd_xcd_block(const unsigned block, const float *__restrict__ data, float scale,
                                                        int B, int R, int H, int W, int C, int PH, int PW,
                                                        const float *__restrict__ rois, float *__restrict__ top,
                                                        int *__restrict__ argmax, int tpb_shift)
{
    __shared__ BinGeom s_g[FWD_PASSES * 32];
    const int tpb = 1 << tpb_shift;                  // threads per bin = C/32 (8, 16 or 32 float4 lanes)
    const int bpp = 256 >> tpb_shift;                // bins per pass (32, 16 or 8)
    const int slice = block & 7;
    const long long nbins = (long long)R * PH * PW;
    const long long bin0 = (long long)(block >> 3) * (FWD_PASSES * bpp);
    if (threadIdx.x < FWD_PASSES * bpp) {
        BinGeom g;
        g.hs = g.he = g.ws = g.we = 0; g.base = -1; g.pad0 = g.pad1 = g.pad2 = 0;
        const long long bin = bin0 + threadIdx.x;
        if (bin < nbins) {
            // 32-bit divisions (the launcher guarantees R*PH*PW < 2^31): a 64-bit divide is ~100 instructions,
            // three of them in front of the barrier were the long pole of every workgroup
            const unsigned ub = (unsigned)bin, upw = (unsigned)PW, uph = (unsigned)PH;
            const unsigned t = ub / upw;
            const int pw = (int)(ub - t * upw);
            const unsigned n = t / uph;
            const int ph = (int)(t - n * uph);
            const float *roi = rois + 5 * (long long)n;
            const int bi = (int)roi[0];
            const RoiGeom q = roi_geom(roi, scale);
            const int rw = max(q.rew - q.rsw + 1, 1), rh = max(q.reh - q.rsh + 1, 1);   // roi_pooling_op.cc:146-147
            const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;        // :148-151
            int hs = (int)floorf(__fmul_rn((float)ph, bh)), ws = (int)floorf(__fmul_rn((float)pw, bw));
            int he = (int)ceilf(__fmul_rn((float)(ph + 1), bh)), we = (int)ceilf(__fmul_rn((float)(pw + 1), bw));
            g.hs = min(max(hs + q.rsh, 0), H); g.he = min(max(he + q.rsh, 0), H);      // :159-162
            g.ws = min(max(ws + q.rsw, 0), W); g.we = min(max(we + q.rsw, 0), W);
            const bool empty = (g.he <= g.hs) || (g.we <= g.ws) || bi < 0 || bi >= B;
            g.base = empty ? -1 : bi;
        }
        s_g[threadIdx.x] = g;
    }
    __syncthreads();
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
                for (int h = g.hs; h < g.he; ++h) {
                    int w = g.ws;
                    // four independent 16-byte loads in flight, consumed in scan order (first max wins)
                    for (; w + 4 <= g.we; w += 4) {
                        const int idx = (h * W + w) * C;
                        const float4 x0 = *reinterpret_cast<const float4 *>(d + idx);
                        const float4 x1 = *reinterpret_cast<const float4 *>(d + idx + C);
                        const float4 x2 = *reinterpret_cast<const float4 *>(d + idx + 2 * C);
                        const float4 x3 = *reinterpret_cast<const float4 *>(d + idx + 3 * C);
                        upd4(x0, idx + c0, mv, mi);
                        upd4(x1, idx + C + c0, mv, mi);
                        upd4(x2, idx + 2 * C + c0, mv, mi);
                        upd4(x3, idx + 3 * C + c0, mv, mi);
                    }
                    for (; w < g.we; ++w) {
                        const int idx = (h * W + w) * C;
                        upd4(*reinterpret_cast<const float4 *>(d + idx), idx + c0, mv, mi);
                    }
                }
            }
            const long long o = bin * C + c0;
            // streaming stores: the outputs are written once and never re-read here, they must not push the
            // XCD's slice of the feature map out of its L2
            typedef float f4v __attribute__((ext_vector_type(4)));
            typedef int i4v __attribute__((ext_vector_type(4)));
            const f4v mvv = {mv.x, mv.y, mv.z, mv.w};
            const i4v miv = {mi.x, mi.y, mi.z, mi.w};
            __builtin_nontemporal_store(mvv, reinterpret_cast<f4v *>(top + o));
            if (argmax) __builtin_nontemporal_store(miv, reinterpret_cast<i4v *>(argmax + o));
        }
    }
}

template <int FWD_PASSES>
__global__ __launch_bounds__(256) void roi_pool_fwd_xcd_kernel(const float *__restrict__ data, float scale, int B, int R,
                                                               int H, int W, int C, int PH, int PW,
                                                               const float *__restrict__ rois, float *__restrict__ top,
                                                               int *__restrict__ argmax, int tpb_shift)
{
    roi_pool_fwd_xcd_block<FWD_PASSES>(blockIdx.x, data, scale, B, R, H, W, C, PH, PW, rois, top, argmax, tpb_shift);
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

template <int FWD_PASSES>
__global__ __launch_bounds__(256) void roi_pool_fwd_xcd_multi_kernel(RoiViewPack p)
{
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && blockIdx.x >= p.v[j].first_block) k = j;
    const RoiViewDev &v = p.v[k];
    roi_pool_fwd_xcd_block<FWD_PASSES>(blockIdx.x - v.first_block, v.data, v.scale, v.B, v.R, v.H, v.W, v.C, p.PH, p.PW, v.rois, v.top,
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

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

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

extern "C" int mv3d_roi_pool_backward(const float *top_diff, float spatial_scale, int batch_size, int num_rois,
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

extern "C" int mv3d_roi_pool_forward_views(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                           void *stream)
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
    if (passes == 4) hipLaunchKernelGGL(roi_pool_fwd_xcd_multi_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(roi_pool_fwd_xcd_multi_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return mv3d_launch_status();
}

extern "C" int mv3d_roi_pool_backward_views(int num_views, const mv3d_roi_grad_view *views, int pooled_height, int pooled_width,
                                            void *stream)
{
    if (num_views <= 0 || num_views > MV3D_MAX_ROI_VIEWS || !views) return MV3D_ERR_INVALID_ARG;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[k];
        const int rc = mv3d_roi_pool_backward(w.top_diff, w.spatial_scale, w.batch_size, w.num_rois, w.height, w.width, w.channels,
                                              pooled_height, pooled_width, w.bottom_rois, w.bottom_diff, w.argmax_data, stream);
        if (rc != MV3D_OK) return rc;
    }
    return MV3D_OK;
}
