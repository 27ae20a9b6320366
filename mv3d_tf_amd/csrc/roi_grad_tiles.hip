// RoiPoolGrad of the PAIR as ONE launch: an ordered scatter into LDS-resident map tiles (roi_pooling_op.cc:373-443).
//
// The pair's argmax plane holds, per pooled value, the scan position of the first maximum inside its bin's forward rectangle
// (roi_pool.hip, COMPACT): the code NAMES the pixel the value's gradient goes to.  So the gradient needs no per-pixel candidate
// lists at all -- a (roi, bin) record is relevant to a map tile iff the bin's forward rectangle meets the tile, and where a
// record's 64 channels go inside the tile follows from their codes.
//
//   workgroup = 4 waves = a strip of 4 adjacent tiles (tile = 2^ths x 2^tws pixels, <= 16) x one 64-channel slice
//               (workgroup b -> slice b % nsl: slice s of every record is only ever read on XCD s, like the forward);
//   filter      thread t takes ROI t of the view (256 per pass): frame, bounding box against the strip -> ordered hit list in LDS
//               with the ROI's rounded corner and bin sizes (the f32 divides happen once per ROI and workgroup);
//   expansion   a wave walks the hits that meet ITS tile; lane = bin (ph, pw) computes the bin's rectangle exactly as the forward
//               does, bins that meet the tile are appended (ballot order = the reference's ph, pw order) to a 128-entry LDS ring:
//               {record byte offset, bin width + its reciprocal, rectangle origin relative to the tile};
//   drain       64 entries at a time: W records' code bytes + top_diff slices in flight per wave (record offset = scalar offset
//               of the buffer loads), then per record  code -> (h, w)  by one multiply-shift, target = the tile's f32 accumulator
//               of that pixel in LDS (lane = channel: lanes never collide) or a junk slot when the code names a pixel of another
//               tile / no pixel; read-add-write in record order, four records per LDS round trip with the sums forwarded between
//               records that hit the same accumulator.  ROIs ascending, then ph, pw: the reference's f32 summation order, so
//               the sums are bit-identical to the per-pixel gather (a sum starts at +0 and only ever adds what the reference adds);
//   write-out   the tile's accumulators, zeros included: every pixel of bottom_diff is written exactly once, no fill launch,
//               no workspace, no index.
#include "common.h"
#include "kernels.h"
#include "roi_geom.h"
#include <limits.h>
#include <stdlib.h>

#define RGT_MAXPX 16
#define RGT_RING 128
#define RGT_BIG 0x80000000u

struct RgtView {
    const float *top_diff, *rois;
    const unsigned char *plane8;
    float *bottom_diff;
    float scale;
    int B, R, H, W, C;
    int ths, tws;                    // log2 of the tile's rows / columns
    int tiles_x, tiles_y, strips_x;  // tiles per row, tile rows, strips (4 tiles) per tile row
    int nsl;                         // channel slices = C / 64
    unsigned first_block;
};
struct RgtPack {
    RgtView v[MV3D_MAX_ROI_VIEWS];
    int n, PH, PW, inv_pw;
    int dbg;                         // experiment builds (MV3D_TUNING): phases switched off, 0 otherwise
    long long *trace;                // experiment builds: 8 wall-clock stamps per wave (tools/roi_tiles_trace.py), NULL otherwise
};
#ifdef MV3D_TUNING
#define RGT_DBG(d, x) ((d) & (x))
#else
#define RGT_DBG(d, x) 0
#endif
#define RGT_STAMP(i) do { if (RGT_DBG(1, 1) && p.trace && lane == 0) p.trace[((long long)blockIdx.x * 4 + wave) * 8 + (i)] = (long long)wall_clock64(); } while (0)

struct RgtShared {
    float acc[4][(RGT_MAXPX + 1) * 64];          // per wave: the tile's accumulators [pixel][channel] + one junk row
    int4 ring[4][RGT_RING];
    int4 g0[256];                                // hits of the pass: {roi, rsh, rsw, bits(bin height)}
    int4 g1[256];                                //                   {bits(bin width), first column, last column + 1 (conservative), -}
    int wcnt[4];
};

// four records, in order: acc[a[j]] += v[j]; one LDS round trip, the running sums forwarded between records on the same accumulator
__device__ __forceinline__ void rgt_add4(float *acc, const int a[4], const float v[4])
{
    const float r0 = acc[a[0]], r1 = acc[a[1]], r2 = acc[a[2]], r3 = acc[a[3]];
    const float s0 = r0 + v[0];
    const float s1 = (a[1] == a[0] ? s0 : r1) + v[1];
    const float s2 = (a[2] == a[1] ? s1 : (a[2] == a[0] ? s0 : r2)) + v[2];
    const float s3 = (a[3] == a[2] ? s2 : (a[3] == a[1] ? s1 : (a[3] == a[0] ? s0 : r3))) + v[3];
    acc[a[0]] = s0; acc[a[1]] = s1; acc[a[2]] = s2; acc[a[3]] = s3;
}

// the accumulator (float index into acc) of one record's lane: entry words y (bin width | reciprocal << 8), z (origin + 256, packed)
__device__ __forceinline__ int rgt_target(const int c, const int y, const int z, const int TH, const int TW, const int tws, const int lane,
                                          const bool live)
{
    const int bw = y & 0xff, mm = (y >> 8) & 0x1ffff;
    const int q = (int)(((unsigned)c * (unsigned)mm) >> 16);           // c / bw, exact for c < 256, bw < 256
    const int h = q + (z & 0xffff) - 256, w = c - q * bw + (z >> 16) - 256;
    const bool ok = live && c != 255 && (unsigned)h < (unsigned)TH && (unsigned)w < (unsigned)TW;
    return ((ok ? (h << tws) + w : RGT_MAXPX) << 6) + lane;
}

template <int W, bool MASKED>
__device__ __forceinline__ void rgt_drain(const int4 e, const int u0, const int m, const __amdgpu_buffer_rsrc_t rc,
                                          const __amdgpu_buffer_rsrc_t rt, const int lane, float *acc, const int TH, const int TW, const int tws, const int dbg)
{
    unsigned char cd[W];
    float td[W];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const int l = MASKED ? min(u0 + u, m - 1) : u0 + u;
        const int so = __builtin_amdgcn_readlane(e.x, l);
        cd[u] = __builtin_amdgcn_raw_buffer_load_b8(rc, lane, so >> 2, 0);
        td[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, lane * 4, so, 0));
    }
#pragma unroll
    for (int g = 0; g < W; g += 4) {
        if (MASKED && u0 + g >= m) break;
        int a[4];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = g + j;
            const int l = MASKED ? min(u0 + u, m - 1) : u0 + u;
            a[j] = rgt_target((int)cd[u], __builtin_amdgcn_readlane(e.y, l), __builtin_amdgcn_readlane(e.z, l), TH, TW, tws, lane,
                              !MASKED || u0 + u < m);
            v[j] = td[u];
        }
        if (RGT_DBG(dbg, 4)) { acc[(RGT_MAXPX << 6) + lane] += (float)(a[0] + a[1] + a[2] + a[3]) + v[0] + v[1] + v[2] + v[3]; continue; }
        rgt_add4(acc, a, v);
    }
}

// entries with bins of more than 255 pixels among them (ROIs far larger than the map: rare): entry by entry, each from its own
// plane, in the same order
__device__ __forceinline__ void rgt_drain_mixed(const int4 e, const int m, const __amdgpu_buffer_rsrc_t rc, const __amdgpu_buffer_rsrc_t rc16,
                                                const __amdgpu_buffer_rsrc_t rt, const int lane, float *acc, const int TH, const int TW,
                                                const int tws)
{
    for (int u = 0; u < m; ++u) {
        const int so = __builtin_amdgcn_readlane(e.x, u), y = __builtin_amdgcn_readlane(e.y, u);
        const int z = __builtin_amdgcn_readlane(e.z, u), w4 = __builtin_amdgcn_readlane(e.w, u);
        const float td = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, lane * 4, so, 0));
        int a;
        if ((unsigned)y & RGT_BIG) {
            const int c = (int)__builtin_amdgcn_raw_buffer_load_b16(rc16, lane * 2, so >> 1, 0);
            const int bw = y & 0xffff;
            const int q = c / bw;
            const int h = q + z, w = c - q * bw + w4;
            const bool ok = c != 0xffff && (unsigned)h < (unsigned)TH && (unsigned)w < (unsigned)TW;
            a = ((ok ? (h << tws) + w : RGT_MAXPX) << 6) + lane;
        } else {
            a = rgt_target((int)__builtin_amdgcn_raw_buffer_load_b8(rc, lane, so >> 2, 0), y, z, TH, TW, tws, lane, true);
        }
        acc[a] = acc[a] + td;
    }
}

template <int W>
__global__ __launch_bounds__(256) void roi_pair_tiles_kernel(RgtPack p)
{
    __shared__ RgtShared S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && blockIdx.x >= p.v[j].first_block) k = j;
    const RgtView &v = p.v[k];
    const unsigned lb = blockIdx.x - v.first_block;
    const int nsl = v.nsl;
    const int slice = (int)(lb % (unsigned)nsl);
    unsigned t = lb / (unsigned)nsl;
    const int sx = (int)(t % (unsigned)v.strips_x); t /= (unsigned)v.strips_x;
    const int ty = (int)(t % (unsigned)v.tiles_y);
    const int b = (int)(t / (unsigned)v.tiles_y);
    const int H = v.H, Wd = v.W, C = v.C, R = v.R, PH = p.PH, PW = p.PW, PHW = PH * PW;
    const int ths = v.ths, tws = v.tws, TH = 1 << ths, TW = 1 << tws;
    const int tx = sx * 4 + wave;
    const bool tile_ok = tx < v.tiles_x;                               // (wave-uniform)
    const int th0 = ty << ths, th1 = min(th0 + TH, H);
    const int tw0 = tx << tws, tw1 = min(tw0 + TW, Wd);
    const int sw0 = (sx * 4) << tws, sw1 = min(sw0 + 4 * TW, Wd);      // the strip's columns
    float *const acc = S.acc[wave];
    int4 *const ring = S.ring[wave];
    RGT_STAMP(0);
    // this wave's accumulators (and the junk row) start at +0
    for (int i = lane; i < (RGT_MAXPX + 1) * 16; i += 64) reinterpret_cast<float4 *>(acc)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const unsigned char *const plane8 = v.plane8;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(plane8 + slice * 64), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc16 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const unsigned short *)(plane8 + (long long)R * PHW * C) + slice * 64), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)(v.top_diff + slice * 64), 0, 0x7fffffff, 0x00020000);
    int head = 0, tail = 0;                                            // ring positions (wave-uniform)
    for (int base = 0; base < R; base += 256) {
        // ---- filter: ROI base + t against the strip
        const int roi = base + (int)threadIdx.x;
        bool hit = false;
        int4 q0 = make_int4(0, 0, 0, 0), q1 = make_int4(0, 0, 0, 0);
        if (roi < R) {
            float rr[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) rr[u] = v.rois[5 * (long long)roi + u];
            const RoiGeom q = roi_geom(rr, v.scale);
            const int rw = max(q.rew - q.rsw + 1, 1), rh = max(q.reh - q.rsh + 1, 1);   // roi_pooling_op.cc:146-147
            const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;        // :148-151
            // conservative bounding box of the ROI's bins (every bin's rows lie in [rsh, rsh + rh + 1]); coordinates outside the
            // range where that arithmetic is exact (NaN / inf / absurd boxes) are left to the exact per-bin test of the expansion
            const bool sane = abs(q.rsw) < (1 << 24) && abs(q.rsh) < (1 << 24) && abs(q.rew) < (1 << 24) && abs(q.reh) < (1 << 24);
            int c0 = INT_MIN, c1 = INT_MAX;
            bool rows = true;
            if (sane) { c0 = q.rsw; c1 = q.rsw + rw + 2; rows = q.rsh < th1 && q.rsh + rh + 2 > th0; }
            hit = (int)rr[0] == b && rows && c0 < sw1 && c1 > sw0;
            q0 = make_int4(roi, q.rsh, q.rsw, __builtin_bit_cast(int, bh));
            q1 = make_int4(__builtin_bit_cast(int, bw), c0, c1, 0);
        }
        if (RGT_DBG(p.dbg, 8)) hit = false;
        const unsigned long long bal = __ballot(hit);
        if (base == 0) RGT_STAMP(1);
        if (lane == 0) S.wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, nh = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = S.wcnt[w]; if (w < wave) off += c; nh += c; }
        if (hit) {
            const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
            S.g0[pos] = q0; S.g1[pos] = q1;
        }
        __syncthreads();
        if (base == 0) RGT_STAMP(2);
        // ---- expansion + drain: this wave's tile
        if (tile_ok && !RGT_DBG(p.dbg, 1)) {
            for (int j = 0; j < nh; ++j) {
                const int4 h1v = S.g1[j];
                const int c0 = __builtin_amdgcn_readfirstlane(h1v.y), c1 = __builtin_amdgcn_readfirstlane(h1v.z);
                if (c0 >= tw1 || c1 <= tw0) continue;                  // the ROI does not reach this tile's columns
                const int4 h0v = S.g0[j];
                const int r = __builtin_amdgcn_readfirstlane(h0v.x), rsh = __builtin_amdgcn_readfirstlane(h0v.y);
                const int rsw = __builtin_amdgcn_readfirstlane(h0v.z);
                const float bh = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(h0v.w));
                const float bw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(h1v.x));
                for (int bin0 = 0; bin0 < PHW; bin0 += 64) {
                    const int bin = bin0 + lane;
                    const int ph = (int)(((unsigned)bin * (unsigned)p.inv_pw) >> 16), pw = bin - ph * PW;
                    // the bin's rectangle as the forward computes it (roi_pool.hip fwd_bin_rect, roi_pooling_op.cc:153-162)
                    const int hs0 = (int)floorf(__fmul_rn((float)ph, bh)), ws0 = (int)floorf(__fmul_rn((float)pw, bw));
                    const int he0 = (int)ceilf(__fmul_rn((float)(ph + 1), bh)), we0 = (int)ceilf(__fmul_rn((float)(pw + 1), bw));
                    const int hs = min(max(hs0 + rsh, 0), H), he = min(max(he0 + rsh, 0), H);
                    const int ws = min(max(ws0 + rsw, 0), Wd), we = min(max(we0 + rsw, 0), Wd);
                    const bool meets = bin < PHW && he > hs && we > ws && hs < th1 && he > th0 && ws < tw1 && we > tw0;
                    const unsigned long long mb = __ballot(meets);
                    if (mb == 0ull) continue;
                    if (meets) {
                        const int bwid = we - ws;
                        const bool big = (he - hs) * bwid > 255;
                        int4 e;
                        e.x = ((r * PHW + bin) * C) * 4;
                        e.y = big ? (int)((unsigned)bwid | RGT_BIG) : (bwid | (int)((65535u / (unsigned)bwid + 1u) << 8));
                        e.z = big ? hs - th0 : ((hs - th0 + 256) | ((ws - tw0 + 256) << 16));
                        e.w = ws - tw0;
                        ring[(tail + __popcll(mb & ((1ull << lane) - 1ull))) & (RGT_RING - 1)] = e;
                    }
                    tail += __popcll(mb);
                    if (RGT_DBG(p.dbg, 2)) { head = tail; continue; }
                    if (tail - head >= 64) {
                        const int4 e = ring[(head + lane) & (RGT_RING - 1)];
                        if (__ballot(((unsigned)e.y & RGT_BIG) != 0u) != 0ull) rgt_drain_mixed(e, 64, rc, rc16, rt, lane, acc, TH, TW, tws);
                        else
#pragma unroll 1
                            for (int u0 = 0; u0 < 64; u0 += W) rgt_drain<W, false>(e, u0, 64, rc, rt, lane, acc, TH, TW, tws, p.dbg);
                        head += 64;
                    }
                }
            }
        }
        if (base + 256 < R) __syncthreads();                           // the hit list is rewritten by the next pass
    }
    RGT_STAMP(3);
    if (RGT_DBG(1, 1) && p.trace && lane == 0) p.trace[((long long)blockIdx.x * 4 + wave) * 8 + 6] = tail;
    if (!tile_ok) return;
    if (tail > head) {
        const int m = tail - head;
        const int4 e = ring[(head + min(lane, m - 1)) & (RGT_RING - 1)];
        if (__ballot(((unsigned)e.y & RGT_BIG) != 0u) != 0ull) rgt_drain_mixed(e, m, rc, rc16, rt, lane, acc, TH, TW, tws);
        else {
            int u0 = 0;
#pragma unroll 1
            for (; u0 + W <= m; u0 += W) rgt_drain<W, false>(e, u0, m, rc, rt, lane, acc, TH, TW, tws, p.dbg);
            if (u0 < m) rgt_drain<W, true>(e, u0, m, rc, rt, lane, acc, TH, TW, tws, p.dbg);
        }
    }
    RGT_STAMP(4);
    // ---- write-out: every pixel of the tile, zeros included (four pixels x 256 B per store instruction)
    float *const out = v.bottom_diff + (long long)b * H * Wd * C + slice * 64 + (lane & 15) * 4;
    const int npx = TH * TW;
    typedef float f4v __attribute__((ext_vector_type(4)));
    for (int i = 0; i < npx; i += 4) {
        const int px = i + (lane >> 4);
        const int h = th0 + (px >> tws), w = tw0 + (px & (TW - 1));
        if (px < npx && h < H && w < Wd) {
            const float4 x = *reinterpret_cast<const float4 *>(acc + px * 64 + (lane & 15) * 4);
            const f4v xv = {x.x, x.y, x.z, x.w};
            __builtin_nontemporal_store(xv, reinterpret_cast<f4v *>(out + ((long long)h * Wd + w) * C));
        }
    }
    RGT_STAMP(5);
}

static int rgt_env(const char *name, int dflt)
{
#ifdef MV3D_TUNING                                                     // tuning hooks, experiment builds only
    const char *s = getenv(name);
    return s ? atoi(s) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// The caller (mv3d_roi_pool_backward_views_pair) has validated the views: roi_pair_shapes (same C in {256, 512}, pooled sizes <= 15,
// PH * PW <= 255, maps of <= 65534 pixels, 31-bit record offsets), 16-byte aligned buffers.
int mv3d_launch_roi_pair_tiles(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, hipStream_t stream)
{
    RgtPack p;
    p.n = num_views; p.PH = PH; p.PW = PW; p.inv_pw = 65536 / PW + 1;
    // densest view first (records per pixel): its tiles carry the longest record streams and must not start last
    int order[MV3D_MAX_ROI_VIEWS];
    double dens[MV3D_MAX_ROI_VIEWS];
    for (int k = 0; k < num_views; ++k) {
        order[k] = k;
        dens[k] = (double)views[k].num_rois * PH * PW / ((double)views[k].batch_size * views[k].height * views[k].width);
    }
    for (int i = 1; i < num_views; ++i)
        for (int j = i; j > 0 && dens[order[j]] > dens[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    if (rgt_env("MV3D_RGT_ORDER", 0) == 1 && num_views == 3) { const int t = order[1]; order[1] = order[2]; order[2] = t; }
    const int force_px = rgt_env("MV3D_RGT_PX", 0);
    unsigned blocks = 0;
    for (int i = 0; i < num_views; ++i) {
        const mv3d_roi_grad_view &w = views[order[i]];
        RgtView &v = p.v[i];
        v.top_diff = w.top_diff; v.rois = w.bottom_rois; v.plane8 = (const unsigned char *)w.argmax_data; v.bottom_diff = w.bottom_diff;
        v.scale = w.spatial_scale; v.B = w.batch_size; v.R = w.num_rois; v.H = w.height; v.W = w.width; v.C = w.channels;
        // tile size from the record density: a tile's record stream is one wave's serial work
        const double d = dens[order[i]];
        int px = d >= 6.0 ? 2 : (d >= 3.0 ? 4 : (d >= 1.5 ? 8 : 16));
        if (force_px) px = i == 0 ? (force_px & 0xff) : (i == 1 ? (force_px >> 8) & 0xff : (force_px >> 16) & 0xff);
        if (px != 1 && px != 2 && px != 4 && px != 8 && px != 16) px = 16;
        int lg = 0;
        while ((1 << lg) < px) ++lg;
        v.ths = lg / 2; v.tws = lg - v.ths;                            // 1x1, 1x2, 2x2, 2x4, 4x4
        v.tiles_x = (w.width + (1 << v.tws) - 1) >> v.tws;
        v.tiles_y = (w.height + (1 << v.ths) - 1) >> v.ths;
        v.strips_x = (v.tiles_x + 3) / 4;
        v.nsl = w.channels / 64;
        v.first_block = blocks;
        const long long nb = (long long)w.batch_size * v.tiles_y * v.strips_x * v.nsl;
        if (nb + blocks > 0x7fffffffLL) return MV3D_ERR_INVALID_ARG;
        blocks += (unsigned)nb;
    }
    for (int k = num_views; k < MV3D_MAX_ROI_VIEWS; ++k) p.v[k] = p.v[0];
    p.dbg = rgt_env("MV3D_RGT_DBG", 0);
    p.trace = nullptr;
#ifdef MV3D_TUNING
    if (getenv("MV3D_RGT_TRACE")) p.trace = (long long *)strtoull(getenv("MV3D_RGT_TRACE"), nullptr, 0);
#endif
    const int W = rgt_env("MV3D_RGT_W", 16);
    if (W == 64) hipLaunchKernelGGL(roi_pair_tiles_kernel<64>, dim3(blocks), dim3(256), 0, stream, p);
    else if (W == 32) hipLaunchKernelGGL(roi_pair_tiles_kernel<32>, dim3(blocks), dim3(256), 0, stream, p);
    else if (W == 8) hipLaunchKernelGGL(roi_pair_tiles_kernel<8>, dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(roi_pair_tiles_kernel<16>, dim3(blocks), dim3(256), 0, stream, p);
    return mv3d_launch_status();
}
