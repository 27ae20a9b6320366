// RoiPoolGrad of the PAIR as ONE launch: an ordered, code-directed scatter into LDS-resident map tiles (roi_pooling_op.cc:373-443).
//
// The pair's argmax plane holds, per pooled value, the scan position of the first maximum inside its bin's forward rectangle
// (roi_pool.hip, COMPACT): the code NAMES the pixel the value's gradient goes to.  So the gradient needs no per-pixel candidate
// lists: a (roi, bin) record is relevant to a map tile iff the bin's forward rectangle meets the tile, and which of the tile's
// pixels a channel's value goes to follows from its code.
//
//   workgroup = ONE wave = one tile (2^ths x 2^tws pixels, <= 16) x one 64-channel slice (workgroup b -> slice b % nsl: slice s of
//               every record is only ever read on XCD s, like the forward).  No barrier anywhere: a wave's slot is free the moment
//               its tile is done;
//   filter      lane = ROI, 256 ROI rows requested at once: frame and bounding box against the tile (the f32 divides of the bin
//               sizes once per ROI and wave).  A ROI whose rounded end lies before its start passes NO pixel through the reference's
//               containment test (roi_pooling_op.cc:401-404) although the forward pools a forced 1 x 1 region for it: dropped here;
//   expansion   per surviving ROI lane = bin (ph, pw) computes the bin's rectangle exactly as the forward does, cut to the rounded ROI
//               (f32: 7 * (57 / 7) > 57, so the last bin of a 57-wide ROI reaches one column past the ROI's end; the forward pools
//               that column, the reference's backward drops what lands there -- same test, :401-404); per bin (lane order = the
//               reference's ph, pw order) one entry per ROW of the tile its rectangle meets is appended (a pixel only ever sees one row
//               segment of a bin, so every pixel meets its bins in the reference's order; positions by one prefix sum over the lanes' row
//               counts) to a 256-entry LDS ring of 8-byte entries
//               {record byte offset | first pixel of the row segment | its length - 1, code of that pixel};
//   drain       W entries at a time, all their code bytes + top_diff slices requested at once (record offset = scalar offset of
//               the buffer loads); per entry and lane  code -> pixel of the segment  (a subtract and a compare, branch-free) = the
//               f32 accumulator [pixel][channel] in LDS, or a junk slot when the code names a pixel outside the segment / no
//               pixel (lane = channel: lanes never collide); read-add-write in entry
//               order, four entries per LDS round trip with the sums forwarded between entries that hit the same accumulator.
//               ROIs ascending, then ph, pw: the reference's f32 summation order, bit-identical to the per-pixel gather (a sum
//               starts at +0 and only ever adds what the reference adds);
//   write-out   the tile's accumulators, zeros included: every pixel of bottom_diff is written exactly once -- no fill launch,
//               no workspace, no index.
// The reference's backward additionally asks ph in [phstart(h), phend(h)) (:423-431, f32 divides).  For every pooled size <= 15 and
// every ROI extent <= RGT_EXACT_MAX that range contains all bins whose forward rectangle holds the pixel
// (tests/test_roi_geometry.py, exhaustive), so inside the rounded ROI the code alone decides; larger ROIs evaluate the reference's
// expressions per pixel (rgt_exact_span).
// (Other structures, measured and dropped, sources under tools/experiments/: a strip of four tiles per 256-thread workgroup with a
// shared filter, 88 us; accumulators in registers indexed by the wave-uniform pixel, 119 us (profiles/r05_ad, r05_af); round 6: every
// load pipelined through accumulation-register sets, three groups in flight across the expansion code -- the loads were never the
// limit, a record costs ~140 ns of ONE wave's instruction stream with or without them (profiles/r06_a, r06_b); four waves sharing
// the draining of four tiles through tickets and an in-order add token -- the token's critical section is as slow as the adds it
// orders, 121 - 172 us (profiles/r06_c, r06_d).)
#include "common.h"
#include "kernels.h"
#include "roi_geom.h"
#include "roi_grad_plan.h"
#include <limits.h>
#include <stdlib.h>

#define RGT_MAXPX 16
#define RGT_RING 256                 // ring entries: a chunk of bins appends <= 240 (60 bins x 4 rows | 64 x 2), < 16 are left over from the last drain
#define RGT_BIGBIT 0x100             // entry word 0: the bin has more than 255 pixels (16-bit codes in the escape plane)
#define RGT_EXACT_MAX 2048           // ROI extents (map pixels) up to which "forward rectangle inside backward range" is proven exhaustively

#ifdef MV3D_TUNING
#define RGT_DBG(d, x) ((d) & (x))
#else
#define RGT_DBG(d, x) 0
#endif
#define RGT_STAMP(i) do { if (RGT_DBG(1, 1) && p.trace && lane == 0) p.trace[(long long)blockIdx.x * 8 + (i)] = (long long)wall_clock64(); } while (0)

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

// The tile pixel (slot; RGT_MAXPX = none of this tile) a value with code c goes to under one entry = one row segment of (rectangle x
// tile).  Entry words: x = record byte offset | slot of the segment's first pixel [3:0] | pixels - 1 [5:4] | big [8], y = code of that
// pixel.  A code c names pixel d of the segment iff c - code0 = d <= pixels - 1.
__device__ __forceinline__ int rgt_target(const int x, const int y, const int c, const bool live)
{
    const int d = c - y;
    return live && (unsigned)d <= (unsigned)((x >> 4) & 3) ? (x & 15) + d : RGT_MAXPX;
}

// CPL channels per lane (a record's slice = 64 CPL channels, one load of CPL code bytes and one of CPL floats per record and lane).
// Shipped: CPL = 1.  Two / four channels per lane halve / quarter the waves, filters and expansions but lengthen every wave's serial
// record stream by as much: 98 / 155 us against 68 (profiles/r05_aj_tiles_cpl.txt; tools/experiments/roi_grad_tiles_onewave_cpl_r05.hip.txt).
template <int CPL> struct RgtVec;
template <> struct RgtVec<1> {
    typedef unsigned int C; typedef float T;
    static __device__ __forceinline__ C ldc(__amdgpu_buffer_rsrc_t r, int lane, int s) { return __builtin_amdgcn_raw_buffer_load_b8(r, lane, s, 0); }
    static __device__ __forceinline__ T ldt(__amdgpu_buffer_rsrc_t r, int lane, int s) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, s, 0)); }
    static __device__ __forceinline__ float el(const T &t, int) { return t; }
};

// m <= W entries of the ring (entry u in lane u of e), none of them a big bin: every load first (rgt_issue), then the ordered adds
// (rgt_adds).  Split in two so that the pipelined kernel can request a group's bytes one group ahead of its adds.
template <int W, int CPL>
__device__ __forceinline__ void rgt_issue(typename RgtVec<CPL>::C (&cd)[W], typename RgtVec<CPL>::T (&td)[W], const int2 e, const int m,
                                          const __amdgpu_buffer_rsrc_t rc, const __amdgpu_buffer_rsrc_t rt, const int lane)
{
    typedef RgtVec<CPL> V;
#pragma unroll
    for (int g = 0; g < W; g += 4) {
        if (g < m) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int so = __builtin_amdgcn_readlane(e.x, min(g + j, m - 1)) & ~1023;
                cd[g + j] = V::ldc(rc, lane, so >> 2);
                td[g + j] = V::ldt(rt, lane, so);
            }
        }
    }
}

template <int W, int CPL>
__device__ __forceinline__ void rgt_adds(float *acc, const typename RgtVec<CPL>::C (&cd)[W], const typename RgtVec<CPL>::T (&td)[W], const int2 e,
                                         const int m, const int lane, const int dbg)
{
    typedef RgtVec<CPL> V;
#pragma unroll
    for (int g = 0; g < W; g += 4) {
        if (g < m) {
            int xs[4], ys[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int l = min(g + j, m - 1);
                xs[j] = __builtin_amdgcn_readlane(e.x, l); ys[j] = __builtin_amdgcn_readlane(e.y, l);
            }
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                int a[4];
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = CPL == 1 ? (int)cd[g + j] : (int)((cd[g + j] >> (8 * k)) & 0xffu);
                    a[j] = (rgt_target(xs[j], ys[j], c, g + j < m) * 64 + lane) * CPL + k;
                    v[j] = V::el(td[g + j], k);
                }
                if (RGT_DBG(dbg, 4)) { acc[(RGT_MAXPX * 64 + lane) * CPL] += (float)(a[0] + a[1] + a[2] + a[3]) + v[0] + v[1] + v[2] + v[3]; continue; }
                rgt_add4(acc, a, v);
            }
        }
    }
}

template <int W, int CPL>
__device__ __forceinline__ void rgt_drain(float *acc, const int2 e, const int m, const __amdgpu_buffer_rsrc_t rc,
                                          const __amdgpu_buffer_rsrc_t rt, const int lane, const int tws, const int dbg)
{
    typename RgtVec<CPL>::C cd[W];
    typename RgtVec<CPL>::T td[W];
    rgt_issue<W, CPL>(cd, td, e, m, rc, rt, lane);
    rgt_adds<W, CPL>(acc, cd, td, e, m, lane, dbg);
}

// entries with bins of more than 255 pixels among them (ROIs far larger than the map: rare): entry by entry, each from its own
// plane, in the same order
template <int CPL>
__device__ __forceinline__ void rgt_drain_mixed(float *acc, const int2 e, const int m, const __amdgpu_buffer_rsrc_t rc,
                                                const __amdgpu_buffer_rsrc_t rc16, const __amdgpu_buffer_rsrc_t rt, const int lane, const int tws)
{
    for (int u = 0; u < m; ++u) {
        const int x = __builtin_amdgcn_readlane(e.x, u), y = __builtin_amdgcn_readlane(e.y, u);
        const int so = x & ~1023;
        const bool big = (x & RGT_BIGBIT) != 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int ch = lane * CPL + k;
            const float td = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, ch * 4, so, 0));
            const int c = big ? (int)__builtin_amdgcn_raw_buffer_load_b16(rc16, ch * 2, so >> 1, 0)
                              : (int)__builtin_amdgcn_raw_buffer_load_b8(rc, ch, so >> 2, 0);
            const int a = (rgt_target(x, y, c, true) * 64 + lane) * CPL + k;
            acc[a] = acc[a] + td;
        }
    }
}

// A ONE-PIXEL unit (a sub-tile of a cut 2 x 2 tile: the streams of the launch's hottest pixels) keeps its sum in a register, lane =
// channel: every entry names that pixel with code e.y, so a value is added iff its code equals it -- `s += code == y ? v : +0`, which is
// the reference's sum bit for bit: a sum that starts at +0 and adds in the same order never becomes -0, so the +0 of a value that goes
// elsewhere changes nothing.  Three vector instructions per entry instead of a routed LDS read-add-write.
template <int W>
__device__ __forceinline__ void rgt_drain1(float &s, const int2 e, const int m, const __amdgpu_buffer_rsrc_t rc, const __amdgpu_buffer_rsrc_t rt,
                                           const int lane)
{
    unsigned cd[W];
    float td[W];
    rgt_issue<W, 1>(cd, td, e, m, rc, rt, lane);
#pragma unroll
    for (int j = 0; j < W; ++j) {
        if (j < m) {
            const int y = __builtin_amdgcn_readlane(e.y, j);
            s = s + ((int)cd[j] == y ? td[j] : 0.0f);
        }
    }
}

template <int W, int CPL>
__device__ __forceinline__ void rgt_drain_any(float *acc, float &s1, const bool onepx, const int2 e, const int m, const __amdgpu_buffer_rsrc_t rc,
                                              const __amdgpu_buffer_rsrc_t rc16, const __amdgpu_buffer_rsrc_t rt, const int lane, const int tws,
                                              const int dbg)
{
    const bool mixed = __ballot((e.x & RGT_BIGBIT) != 0 && lane < m) != 0ull;
    if (CPL == 1 && onepx && !mixed) { rgt_drain1<W>(s1, e, m, rc, rt, lane); return; }
    if (CPL == 1 && onepx) acc[lane] = s1;                             // (rare: 16-bit codes in a one-pixel unit -- through the LDS slot and back)
    if (mixed) rgt_drain_mixed<CPL>(acc, e, m, rc, rc16, rt, lane, tws);
    else rgt_drain<W, CPL>(acc, e, m, rc, rt, lane, tws, dbg);
    if (CPL == 1 && onepx) s1 = acc[lane];
}

// The pixels x of [lo, hi] that list pooled index p under the reference's backward (roi_pooling_op.cc:423-431): floor((x - start) / bin) <= p <
// ceil((x - start + 1) / bin), both clamped to [0, P]; an interval (the bounds grow with x).  Only for ROIs beyond RGT_EXACT_MAX.
__device__ __forceinline__ int2 rgt_exact_span(const int lo, const int hi, const int start, const float bin, const int P, const int p)
{
    int a = INT_MAX, b = INT_MIN;                                      // (an empty [lo, hi] stays empty)
    for (int x = lo; x <= hi; ++x) {
        const int s = min(max((int)floorf((float)(x - start) / bin), 0), P), e = min(max((int)ceilf((float)(x - start + 1) / bin), 0), P);
        if (s <= p && p < e) { a = min(a, x); b = max(b, x); }
    }
    return make_int2(a, b);
}

#ifdef RGT_WAVES6                                                       // (experiment builds: six waves per SIMD at 80 VGPRs, three spilled)
#define RGT_OCC __attribute__((amdgpu_waves_per_eu(6, 6)))
#else
#define RGT_OCC
#endif
template <int W, int CPL>
__global__ __launch_bounds__(64) RGT_OCC void roi_pair_tiles_kernel(RgtPack p)
{
    __shared__ float acc[(RGT_MAXPX + 1) * 64 * CPL];                        // the tile's accumulators [pixel][channel] + one junk row
    __shared__ int2 ring[RGT_RING];
    const int lane = threadIdx.x;
    int k = 0, slice, b, ths, tws, th0, tw0;
    bool no_roi = false;                                               // (planned mode) no ROI touches the tile: only the zeros are written
    if (p.work) {
        const int nsl = p.v[0].nsl;                                    // (one C for all views of the pair)
        slice = (int)(blockIdx.x % (unsigned)nsl);
        const unsigned unit = blockIdx.x / (unsigned)nsl;
        if (unit >= (unsigned)*p.n_work) return;
        const int4 d = p.work[unit];
        if (d.x & (1 << 24)) return;                                   // (a sub-tile outside the map)
        no_roi = (d.x & RGT_UNIT_EMPTY) != 0;
        k = d.x & 15; b = (d.x >> 4) & 0xfff; ths = (d.x >> 16) & 15; tws = (d.x >> 20) & 15; th0 = d.y; tw0 = d.z;
        // (a list this library's forward did not write -- the caller broke the pair's contract -- must not send the wave outside the buffers)
#ifndef RGT_NO_GUARD
        if (k >= p.n || ths > 2 || tws > 2 || b >= p.v[k].B || (unsigned)th0 >= (unsigned)p.v[k].H || (unsigned)tw0 >= (unsigned)p.v[k].W) return;
#endif
    } else {
#pragma unroll
        for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
            if (j < p.n && blockIdx.x >= p.v[j].first_block) k = j;
        const RgtView &u = p.v[k];
        const unsigned lb = blockIdx.x - u.first_block;
        slice = (int)(lb % (unsigned)u.nsl);
        unsigned t = lb / (unsigned)u.nsl;
        const int tx = (int)(t % (unsigned)u.tiles_x); t /= (unsigned)u.tiles_x;
        const int ty = (int)(t % (unsigned)u.tiles_y);
        b = (int)(t / (unsigned)u.tiles_y);
        ths = u.ths; tws = u.tws; th0 = ty << ths; tw0 = tx << tws;
    }
    const RgtView &v = p.v[k];
    const int H = v.H, Wd = v.W, C = v.C, R = v.R, PH = p.PH, PW = p.PW, PHW = PH * PW;
    const int TH = 1 << ths, TW = 1 << tws;
    const int th1 = min(th0 + TH, H);
    const int tw1 = min(tw0 + TW, Wd);
    RGT_STAMP(0);
    const unsigned char *const plane8 = v.plane8;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(plane8 + slice * 64 * CPL), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc16 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const unsigned short *)(plane8 + (long long)R * PHW * C) + slice * 64 * CPL), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)(v.top_diff + slice * 64 * CPL), 0, 0x7fffffff, 0x00020000);
    int head = 0, tail = 0;                                            // ring positions (wave-uniform)
    const bool onepx = (ths | tws) == 0;                               // a one-pixel unit: the sum in a register (rgt_drain1)
    float s1 = 0.0f;
    const int CH = TH > 2 ? 60 : 64;                                   // bins per expansion chunk: CH x TH + (W - 1) entries fit the ring
    static_assert(W <= 16 && 60 * 4 + W <= RGT_RING, "ring too small for a chunk");
    const int bin1 = lane < CH ? lane : PHW;                           // the first expansion chunk's bin of this lane, (ph, pw), and their float forms
    const int ph1 = (int)(((unsigned)bin1 * (unsigned)p.inv_pw) >> 16), pw1 = bin1 - ph1 * PW;
    const float fph0_1 = (float)ph1, fph1_1 = (float)(ph1 + 1), fpw0_1 = (float)pw1, fpw1_1 = (float)(pw1 + 1);
    for (int base = 0; base < R; base += 256) {
        if (no_roi || RGT_DBG(p.dbg, 16)) {                            // the write-out alone (experiment builds: for every tile)
            for (int i = lane; i < (RGT_MAXPX + 1) * 16 * CPL; i += 64) reinterpret_cast<float4 *>(acc)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            break;
        }
        // ---- the rows of 256 ROIs, requested at once (lane = ROI, four passes)
        float rr[4][5];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int roi = min(base + 64 * q + lane, R - 1);
#pragma unroll
            for (int u = 0; u < 5; ++u) rr[q][u] = v.rois[5 * (long long)roi + u];
        }
        if (base == 0) {                                               // (under the ROI rows' latency) the accumulators start at +0
            for (int i = lane; i < (RGT_MAXPX + 1) * 16 * CPL; i += 64) reinterpret_cast<float4 *>(acc)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const int roi0 = base + 64 * q;
            if (roi0 >= R) break;
            // ---- filter: ROI roi0 + lane against the tile
            float r5[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) r5[u] = q == 0 ? rr[0][u] : (q == 1 ? rr[1][u] : (q == 2 ? rr[2][u] : rr[3][u]));
            const RoiGeom g = roi_geom(r5, v.scale);
            // every entry lies inside the ROI's rounded rectangle (the expansion cuts a bin's rectangle to it, :401-404), so a ROI whose
            // rectangle misses the tile has nothing for it; coordinates outside the range where that arithmetic is exact (NaN / inf / absurd
            // boxes) are left to the exact per-bin test of the expansion
            const bool sane = abs(g.rsw) < (1 << 24) && abs(g.rsh) < (1 << 24) && abs(g.rew) < (1 << 24) && abs(g.reh) < (1 << 24);
            // (:401-404: h in [rsh, reh], w in [rsw, rew] -- an end before its start lets nothing through)
            bool hit = roi0 + lane < R && (int)r5[0] == b && g.reh >= g.rsh && g.rew >= g.rsw;
            if (sane) hit = hit && g.rsh < th1 && g.reh >= th0 && g.rsw < tw1 && g.rew >= tw0;
            if (RGT_DBG(p.dbg, 8)) hit = false;
            unsigned long long todo = __ballot(hit);
            if (base == 0 && q == 0) RGT_STAMP(1);
            if (RGT_DBG(p.dbg, 1)) todo = 0ull;
            // the bin sizes of this pass's ROIs, lane = ROI (roi_pooling_op.cc:148-151): one pair of f32 divides per pass with a hit instead of
            // one per hit ROI (a wave that got past the work list's empty flag usually has many)
            float bhl = 0.0f, bwl = 0.0f;
            if (todo != 0ull) {
                bhl = (float)(g.reh - g.rsh + 1) / (float)PH;
                bwl = (float)(g.rew - g.rsw + 1) / (float)PW;
            }
            // ---- expansion + drain
            while (todo != 0ull) {
                const int j = (int)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int r = roi0 + j, rsh = __builtin_amdgcn_readlane(g.rsh, j), rsw = __builtin_amdgcn_readlane(g.rsw, j);
                const int reh = __builtin_amdgcn_readlane(g.reh, j), rew = __builtin_amdgcn_readlane(g.rew, j);
                const int rhj = reh - rsh + 1, rwj = rew - rsw + 1;    // (>= 1: the filter drops the others)
                const float bh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bhl), j));
                const float bw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bwl), j));
                const bool exact = rhj > RGT_EXACT_MAX || rwj > RGT_EXACT_MAX;
                for (int bin0 = 0; bin0 < PHW; bin0 += CH) {
                    // (the first chunk's lane -> bin map and its float forms are the wave's: bin1 .. fpw1 above; 7 x 7 bins are one chunk)
                    int bin = bin1, ph = ph1, pw = pw1;
                    float fph0 = fph0_1, fph1 = fph1_1, fpw0 = fpw0_1, fpw1 = fpw1_1;
                    if (bin0 != 0) {
                        bin = lane < CH ? bin0 + lane : PHW;
                        ph = (int)(((unsigned)bin * (unsigned)p.inv_pw) >> 16); pw = bin - ph * PW;
                        fph0 = (float)ph; fph1 = (float)(ph + 1); fpw0 = (float)pw; fpw1 = (float)(pw + 1);
                    }
                    // the bin's rectangle as the forward computes it (roi_pool.hip fwd_bin_rect, roi_pooling_op.cc:153-162)
                    const int hs0 = (int)floorf(__fmul_rn(fph0, bh)), ws0 = (int)floorf(__fmul_rn(fpw0, bw));
                    const int he0 = (int)ceilf(__fmul_rn(fph1, bh)), we0 = (int)ceilf(__fmul_rn(fpw1, bw));
                    const int hs = min(max(hs0 + rsh, 0), H), he = min(max(he0 + rsh, 0), H);
                    const int ws = min(max(ws0 + rsw, 0), Wd), we = min(max(we0 + rsw, 0), Wd);
                    // (rectangle x tile), cut to the rounded ROI: hs >= rsh and ws >= rsw hold by construction
                    int ih0 = max(hs, th0), ih1 = min(min(he, th1) - 1, reh), iw0 = max(ws, tw0), iw1 = min(min(we, tw1) - 1, rew);
                    if (exact && bin < PHW && he > hs && we > ws) {
                        const int2 sh = rgt_exact_span(ih0, ih1, rsh, bh, PH, ph), sw = rgt_exact_span(iw0, iw1, rsw, bw, PW, pw);
                        ih0 = sh.x; ih1 = sh.y; iw0 = sw.x; iw1 = sw.y;
                    }
                    const bool meets = bin < PHW && he > hs && we > ws && ih1 >= ih0 && iw1 >= iw0;
                    if (__ballot(meets) == 0ull) continue;
                    const int bwid = we - ws;
                    const int ex = (((r * PHW + bin) * C) * 4) | (iw0 - tw0) | ((iw1 - iw0) << 4) | ((he - hs) * bwid > 255 ? RGT_BIGBIT : 0);
                    const int ey = (iw0 - ws) - hs * bwid;
                    // ---- one entry per row of the tile a bin's rectangle meets, bins outermost (ph, pw order), a bin's rows in turn: a pixel
                    // meets at most one row segment of a bin, so every pixel still sees its bins in the reference's order.  Positions by ONE
                    // prefix sum over the lanes' row counts (0 .. 4: three ballots of their bits) instead of a ballot and a branch per tile row
                    const int nrows = meets ? ih1 - ih0 + 1 : 0;       // (ih0 .. ih1 lies inside the tile's rows)
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const unsigned long long b0 = __ballot((nrows & 1) != 0), b1 = __ballot((nrows & 2) != 0);
                    int pre = __popcll(b0 & below) + 2 * __popcll(b1 & below), tot = __popcll(b0) + 2 * __popcll(b1);
                    if (TH > 2) {
                        const unsigned long long b2 = __ballot((nrows & 4) != 0);
                        pre += 4 * __popcll(b2 & below); tot += 4 * __popcll(b2);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < TH && i < nrows) {
                            const int h = ih0 + i;
                            ring[(tail + pre + i) & (RGT_RING - 1)] = make_int2(ex + ((h - th0) << tws), ey + h * bwid);
                        }
                    }
                    tail += tot;
                    if (RGT_DBG(p.dbg, 2)) { head = tail; continue; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // (ring entries written by other lanes of this wave)
                    __builtin_amdgcn_wave_barrier();
                    while (tail - head >= W) {
                        const int2 e = ring[(head + min(lane, W - 1)) & (RGT_RING - 1)];
                        rgt_drain_any<W, CPL>(acc, s1, onepx, e, W, rc, rc16, rt, lane, tws, p.dbg);
                        head += W;
                    }
                }
            }
        }
    }
    RGT_STAMP(3);
    if (RGT_DBG(1, 1) && p.trace && lane == 0) p.trace[(long long)blockIdx.x * 8 + 6] = tail;
    if (tail > head) {
        const int m = tail - head;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int2 e = ring[(head + min(lane, m - 1)) & (RGT_RING - 1)];
        rgt_drain_any<W, CPL>(acc, s1, onepx, e, m, rc, rc16, rt, lane, tws, p.dbg);
    }
    RGT_STAMP(4);
    if (CPL == 1 && onepx && !no_roi) acc[lane] = s1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");             // (the accumulators: written per channel lane, read 16 B per lane)
    __builtin_amdgcn_wave_barrier();
    // ---- write-out: every pixel of the tile, zeros included (16 B per lane: 4 / CPL pixels x 256 CPL bytes per store instruction)
    constexpr int LPP = 16 * CPL;                                      // lanes per pixel
    float *const out = v.bottom_diff + (long long)b * H * Wd * C + slice * 64 * CPL + (lane & (LPP - 1)) * 4;
    const int npx = TH * TW;
    typedef float f4v __attribute__((ext_vector_type(4)));
    for (int i = 0; i < npx; i += 64 / LPP) {
        const int px = i + lane / LPP;
        const int h = th0 + (px >> tws), w = tw0 + (px & (TW - 1));
        if (px < npx && h < H && w < Wd) {
            const float4 x = *reinterpret_cast<const float4 *>(acc + px * 64 * CPL + (lane & (LPP - 1)) * 4);
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

// The launch layout (see roi_grad_plan.h).  The caller has validated the views: roi_pair_shapes (same C in {256, 512}, pooled sizes <= 15,
// PH * PW <= 255, maps of <= 65534 pixels, 31-bit record offsets), 16-byte aligned buffers.
bool mv3d_rgt_layout(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, RgtPack &p, unsigned *blocks_out, bool *planned,
                     unsigned *plan_blocks, int *hot_entries, int *hot_max)
{
    p.n = num_views; p.PH = PH; p.PW = PW; p.inv_pw = 65536 / PW + 1;
    p.work = nullptr; p.n_work = nullptr; p.dbg = 0; p.trace = nullptr;
    *planned = false; *plan_blocks = 0;
    // launch order: views with dense record streams first by density (their tiles carry the longest serial streams and must not
    // start last), then the sparse ones, the larger map first (its ROIs cover more pixels: longer streams per tile)
    int order[MV3D_MAX_ROI_VIEWS];
    double dens[MV3D_MAX_ROI_VIEWS], key[MV3D_MAX_ROI_VIEWS];
    for (int k = 0; k < num_views; ++k) {
        order[k] = k;
        const double px = (double)views[k].batch_size * views[k].height * views[k].width;
        dens[k] = (double)views[k].num_rois * PH * PW / px;
        key[k] = dens[k] >= 1.5 ? 1e12 * dens[k] : px;
    }
    for (int i = 1; i < num_views; ++i)
        for (int j = i; j > 0 && key[order[j]] > key[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    if (rgt_env("MV3D_RGT_ORDER", 0) == 1 && num_views == 3) { const int t = order[1]; order[1] = order[2]; order[2] = t; }
    const int cpl = 1;
    const int force_px = rgt_env("MV3D_RGT_PX", 0);
    unsigned blocks = 0;
    for (int i = 0; i < num_views; ++i) {
        const mv3d_roi_grad_view &w = views[order[i]];
        RgtView &v = p.v[i];
        v.top_diff = w.top_diff; v.rois = w.bottom_rois; v.plane8 = (const unsigned char *)w.argmax_data; v.bottom_diff = w.bottom_diff;
        v.scale = w.spatial_scale; v.B = w.batch_size; v.R = w.num_rois; v.H = w.height; v.W = w.width; v.C = w.channels;
        // tile size from the record density: a tile's record stream is one wave's serial work
        const double d = dens[order[i]];
        // (measured with eight batches in flight, profiles/r05_at_tiles_in_path.txt: 2 x 2 on a dense view, 4 x 4 on the sparse ones;
        // alone the first sparse view likes 2 x 4 by ~2 us, in the path 4 x 4 is 4 % faster)
        int px = d >= 1.5 ? 4 : 16;
        if (force_px) px = i == 0 ? (force_px & 0xff) : (i == 1 ? (force_px >> 8) & 0xff : (force_px >> 16) & 0xff);
        if (px != 1 && px != 2 && px != 4 && px != 8 && px != 16) px = 16;
        int lg = 0;
        while ((1 << lg) < px) ++lg;
        v.ths = lg / 2; v.tws = lg - v.ths;                            // 1x1, 1x2, 2x2, 2x4, 4x4
        v.tiles_x = (w.width + (1 << v.tws) - 1) >> v.tws;
        v.tiles_y = (w.height + (1 << v.ths) - 1) >> v.ths;
        v.nsl = w.channels / (64 * cpl);
        v.first_block = blocks;
        const long long nb = (long long)w.batch_size * v.tiles_y * v.tiles_x * v.nsl;
        if (nb + blocks > 0x7fffffffLL) return false;
        blocks += (unsigned)nb;
    }
    for (int k = num_views; k < MV3D_MAX_ROI_VIEWS; ++k) p.v[k] = p.v[0];
    *blocks_out = blocks;
    // ---- planned mode: the work list in the last quarter of view 0's argmax buffer (the caller's view 0; bytes neither launch touches)
    long long ntiles = 0;
    bool one_c = true, frames_ok = true;
    for (int i = 0; i < num_views; ++i) {
        ntiles += (long long)p.v[i].B * p.v[i].tiles_y * p.v[i].tiles_x;
        one_c = one_c && p.v[i].nsl == p.v[0].nsl;
        frames_ok = frames_ok && p.v[i].B <= 0xfff;
    }
    *hot_entries = rgt_env("MV3D_RGT_HOT", RGT_HOT_ENTRIES);
    *hot_max = rgt_env("MV3D_RGT_HOT_MAX", RGT_HOT_MAX);
    if (*hot_max > 1024) *hot_max = 1024;
    const long long n0 = (long long)views[0].num_rois * PH * PW * views[0].channels;
    if (rgt_env("MV3D_RGT_PLAN", RGT_PLAN_DEFAULT) != 0 && *hot_entries > 0 && *hot_max > 0 && ntiles <= RGT_PLAN_TILES && one_c && frames_ok && views[0].argmax_data &&
        (long long)MV3D_ALIGN + (ntiles + 3LL * *hot_max) * 16 <= n0) {
        char *const q = (char *)views[0].argmax_data + 3 * n0;
        p.n_work = (const int *)q;
        p.work = (const int4 *)(q + MV3D_ALIGN);
        *planned = true;
        *plan_blocks = (unsigned)(ntiles + 3LL * *hot_max) * (unsigned)p.v[0].nsl;
    }
    return true;
}

int mv3d_launch_roi_pair_tiles(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, hipStream_t stream)
{
    RgtPack p;
    unsigned blocks = 0, plan_blocks = 0;
    bool planned = false;
    int hot = 0, hot_max = 0;
    if (!mv3d_rgt_layout(num_views, views, PH, PW, p, &blocks, &planned, &plan_blocks, &hot, &hot_max)) return MV3D_ERR_INVALID_ARG;
    if (planned) blocks = plan_blocks;                                 // (the list the pair's forward launch wrote)
    p.dbg = rgt_env("MV3D_RGT_DBG", 0);
    p.trace = nullptr;
#ifdef MV3D_TUNING
    if (getenv("MV3D_RGT_TRACE")) p.trace = (long long *)strtoull(getenv("MV3D_RGT_TRACE"), nullptr, 0);
#endif
    // 16 records in flight per wave (6 waves per SIMD); 32 (4 waves per SIMD) is 3 % faster alone and 4 % slower with eight batches
    // in flight (profiles/r05_am_tiles_w32_bench_ab.txt)
#ifdef MV3D_TUNING
    if (rgt_env("MV3D_RGT_W", 16) == 8) { hipLaunchKernelGGL((roi_pair_tiles_kernel<8, 1>), dim3(blocks), dim3(64), 0, stream, p); return mv3d_launch_status(); }
    if (rgt_env("MV3D_RGT_W", 16) == 12) { hipLaunchKernelGGL((roi_pair_tiles_kernel<12, 1>), dim3(blocks), dim3(64), 0, stream, p); return mv3d_launch_status(); }
#endif
    hipLaunchKernelGGL((roi_pair_tiles_kernel<16, 1>), dim3(blocks), dim3(64), 0, stream, p);
    return mv3d_launch_status();
}
