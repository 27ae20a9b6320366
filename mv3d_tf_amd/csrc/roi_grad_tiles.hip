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
//               that column, the reference's backward drops what lands there -- same test, :401-404); per ROW of the tile the bins whose
//               rectangle meets it are appended (ballot order = the reference's ph, pw order; a pixel only ever sees the bins of its own
//               row, so row-major emission keeps every pixel's order) to a 128-entry LDS ring of 8-byte entries
//               {record byte offset | first pixel of the row segment | its length - 1, code of that pixel};
//   drain       a software pipeline over GROUPS of eight entries: a group's 8 code bytes + 8 top_diff slices are requested into one of
//               FOUR register sets by inline-asm buffer loads (record offset = scalar offset; the compiler never sees a load it would
//               wait for), three groups stay in flight across the expansion code while the fourth is added: `s_waitcnt vmcnt(32)`
//               retires exactly the oldest group.  Per entry and lane  code -> pixel of the segment  (a subtract and a compare,
//               branch-free) = the f32 accumulator [pixel][channel] in LDS, or a junk slot when the code names a pixel outside
//               the segment / no pixel (lane = channel: lanes never collide); the stream is padded with null entries to whole
//               groups; read-add-write in record
//               order, four records per LDS round trip with the sums forwarded between records that hit the same accumulator.
//               ROIs ascending, then ph, pw: the reference's f32 summation order, bit-identical to the per-pixel gather (a sum
//               starts at +0 and only ever adds what the reference adds);
//   write-out   the tile's accumulators, zeros included: every pixel of bottom_diff is written exactly once -- no fill launch,
//               no workspace, no index.
// The reference's backward additionally asks ph in [phstart(h), phend(h)) (:423-431, f32 divides).  For every pooled size <= 15 and
// every ROI extent <= RGT_EXACT_MAX that range contains all bins whose forward rectangle holds the pixel
// (tests/test_roi_pair.py::test_forward_rectangles_inside_backward_ranges, exhaustive), so inside the rounded ROI the code alone decides;
// larger ROIs evaluate the reference's expressions per pixel (rgt_exact_span).
// (Earlier versions, git history / profiles/EXPERIMENTS.md R5.10: a strip of four tiles per 256-thread workgroup with a shared filter, 88 us;
// accumulators in registers indexed by the wave-uniform pixel, 119 us; every group's loads waited for before its adds (16 at a time),
// 66 us: a tile under 430 records was a chain of 27 exposed round trips.)
#include "common.h"
#include "kernels.h"
#include "roi_geom.h"
#include <limits.h>
#include <stdlib.h>

#define RGT_MAXPX 16
#define RGT_RING 128
#define RGT_BIGBIT 0x100             // entry word 0: the bin has more than 255 pixels (16-bit codes in the escape plane)
#define RGT_G 8                      // records per group = per register set
#define RGT_EXACT_MAX 2048           // ROI extents (map pixels) up to which "forward rectangle inside backward range" is proven exhaustively

struct RgtView {
    const float *top_diff, *rois;
    const unsigned char *plane8;
    float *bottom_diff;
    float scale;
    int B, R, H, W, C;
    int ths, tws;                    // log2 of the tile's rows / columns
    int tiles_x, tiles_y;
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
#define RGT_STAMP(i) do { if (RGT_DBG(1, 1) && p.trace && lane == 0) p.trace[(long long)blockIdx.x * 8 + (i)] = (long long)wall_clock64(); } while (0)

typedef int rgt_v4i __attribute__((ext_vector_type(4)));
// raw buffer resource over the whole address space above `p` (what __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000) builds),
// as four plain words an inline-asm "s" operand takes
__device__ __forceinline__ rgt_v4i rgt_rsrc(const void *p)
{
    const unsigned long long a = (unsigned long long)p;
    const rgt_v4i r = {(int)(unsigned)a, (int)(unsigned)(a >> 32) & 0xffff, 0x7fffffff, 0x00020000};
    return r;
}

// ---- groups in flight live in ACCUMULATION registers.  A group = eight records = 8 top_diff words + 8 code bytes per lane (two codes per
// register: d16 / d16_hi loads); set i owns a[12 i .. 12 i + 7] (top_diff) and a[12 i + 8 .. 12 i + 11] (codes).  The compiler allocates
// no AGPR in this kernel and is told about these by the clobber lists (they count towards the wave's register budget), so a value in
// flight can never be copied, spilled or renamed between its request and its use -- with VGPR operands the register allocator did
// exactly that (v_mov of a register whose load had not landed).  rgt_land = s_waitcnt + v_accvgpr_read into ordinary values.
#define RGT_SET0 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
#define RGT_SET1 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31
#define RGT_SET2 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47
#define RGT_SET3 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63
// record J of the group `en` (entry J in lane J): one code byte and one top_diff word per lane, record byte offset = scalar offset
// (no d16 packing of two codes per register: with SRAM ECC a d16 load clears the other half)
#define RGT_LD2(CD, TD, J)                                                                                                          \
    {                                                                                                                               \
        const int so_ = __builtin_amdgcn_readlane(en.x, J) & ~1023;                                                                 \
        asm volatile("buffer_load_ubyte a" #CD ", %0, %1, %2 offen\n\tbuffer_load_dword a" #TD ", %3, %4, %5 offen"                 \
                     :: "v"(lane), "s"(qc), "s"(so_ >> 2), "v"(lane4), "s"(qt), "s"(so_) : "a" #CD, "a" #TD);                      \
    }
#define RGT_ISSUE_(T0, T1, T2, T3, T4, T5, T6, T7, C0, C1, C2, C3, C4, C5, C6, C7)                                                  \
    RGT_LD2(C0, T0, 0) RGT_LD2(C1, T1, 1) RGT_LD2(C2, T2, 2) RGT_LD2(C3, T3, 3)                                                    \
    RGT_LD2(C4, T4, 4) RGT_LD2(C5, T5, 5) RGT_LD2(C6, T6, 6) RGT_LD2(C7, T7, 7)
#define RGT_ISSUE(SET) RGT_ISSUE_(SET)
// at most N vector-memory requests outstanding: loads return in order, so with N = 16 x (groups requested after this one) the set has landed
#define RGT_LAND_(N, T0, T1, T2, T3, T4, T5, T6, T7, C0, C1, C2, C3, C4, C5, C6, C7)                                                \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_accvgpr_read_b32 %0, a" #T0 "\n\tv_accvgpr_read_b32 %1, a" #T1                       \
                 "\n\tv_accvgpr_read_b32 %2, a" #T2 "\n\tv_accvgpr_read_b32 %3, a" #T3 "\n\tv_accvgpr_read_b32 %4, a" #T4           \
                 "\n\tv_accvgpr_read_b32 %5, a" #T5 "\n\tv_accvgpr_read_b32 %6, a" #T6 "\n\tv_accvgpr_read_b32 %7, a" #T7           \
                 "\n\tv_accvgpr_read_b32 %8, a" #C0 "\n\tv_accvgpr_read_b32 %9, a" #C1 "\n\tv_accvgpr_read_b32 %10, a" #C2          \
                 "\n\tv_accvgpr_read_b32 %11, a" #C3 "\n\tv_accvgpr_read_b32 %12, a" #C4 "\n\tv_accvgpr_read_b32 %13, a" #C5        \
                 "\n\tv_accvgpr_read_b32 %14, a" #C6 "\n\tv_accvgpr_read_b32 %15, a" #C7                                            \
                 : "=v"(td[0]), "=v"(td[1]), "=v"(td[2]), "=v"(td[3]), "=v"(td[4]), "=v"(td[5]), "=v"(td[6]), "=v"(td[7]),          \
                   "=v"(cd[0]), "=v"(cd[1]), "=v"(cd[2]), "=v"(cd[3]), "=v"(cd[4]), "=v"(cd[5]), "=v"(cd[6]), "=v"(cd[7]))
#define RGT_LAND(N, SET) RGT_LAND_(N, SET)

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
// pixel.  A code c names pixel d of the segment iff c - code0 = d <= pixels - 1.  The null entry (x = 0, y = RGT_NULLCODE) names none.
#define RGT_NULLCODE 0x1ffff
__device__ __forceinline__ int rgt_target(const int x, const int y, const int c)
{
    const int d = c - y;
    return (unsigned)d <= (unsigned)((x >> 4) & 3) ? (x & 15) + d : RGT_MAXPX;
}

// entries with bins of more than 255 pixels among them (ROIs far larger than the map: rare): entry by entry, each from its own
// plane, in the same order (compiler-tracked loads: their wait also retires every group in flight)
__device__ __forceinline__ void rgt_drain_mixed(float *acc, const int2 e, const __amdgpu_buffer_rsrc_t rc, const __amdgpu_buffer_rsrc_t rc16,
                                                const __amdgpu_buffer_rsrc_t rt, const int lane)
{
    for (int u = 0; u < RGT_G; ++u) {
        const int x = __builtin_amdgcn_readlane(e.x, u), y = __builtin_amdgcn_readlane(e.y, u);
        const int so = x & ~1023;
        const float td = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, lane * 4, so, 0));
        const int c = (x & RGT_BIGBIT) ? (int)__builtin_amdgcn_raw_buffer_load_b16(rc16, lane * 2, so >> 1, 0)
                                       : (int)__builtin_amdgcn_raw_buffer_load_b8(rc, lane, so >> 2, 0);
        const int a = rgt_target(x, y, c) * 64 + lane;
        acc[a] = acc[a] + td;
    }
}

// the eight entries of a landed group (entry u in lane u of e), in order
__device__ __forceinline__ void rgt_consume(float *acc, const float td[RGT_G], const unsigned cd[RGT_G], const int2 e,
                                            const __amdgpu_buffer_rsrc_t rc, const __amdgpu_buffer_rsrc_t rc16, const __amdgpu_buffer_rsrc_t rt,
                                            const int lane)
{
    if (__ballot((e.x & RGT_BIGBIT) != 0) != 0ull) { rgt_drain_mixed(acc, e, rc, rc16, rt, lane); return; }
#pragma unroll
    for (int g = 0; g < RGT_G; g += 4) {
        int a[4];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = g + j;
            a[j] = rgt_target(__builtin_amdgcn_readlane(e.x, u), __builtin_amdgcn_readlane(e.y, u), (int)cd[u]) * 64 + lane;
            v[j] = td[u];
        }
        rgt_add4(acc, a, v);
    }
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

__global__ __launch_bounds__(64) void roi_pair_tiles_kernel(RgtPack p)
{
    __shared__ float acc[(RGT_MAXPX + 1) * 64];                              // the tile's accumulators [pixel][channel] + one junk row
    __shared__ int2 ring[RGT_RING];
    __shared__ int4 geo[256];                                          // rounded geometry of the ROIs that reach the tile (of 256 rows)
    __shared__ unsigned char ridx[256];                                // ... and their row numbers
    const int lane = threadIdx.x, lane4 = lane * 4;
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && blockIdx.x >= p.v[j].first_block) k = j;
    const RgtView &v = p.v[k];
    const unsigned lb = blockIdx.x - v.first_block;
    const int nsl = v.nsl;
    const int slice = (int)(lb % (unsigned)nsl);
    unsigned t = lb / (unsigned)nsl;
    const int tx = (int)(t % (unsigned)v.tiles_x); t /= (unsigned)v.tiles_x;
    const int ty = (int)(t % (unsigned)v.tiles_y);
    const int b = (int)(t / (unsigned)v.tiles_y);
    const int H = v.H, Wd = v.W, C = v.C, R = v.R, PH = p.PH, PW = p.PW, PHW = PH * PW;
    const int ths = v.ths, tws = v.tws, TH = 1 << ths, TW = 1 << tws;
    const int th0 = ty << ths, th1 = min(th0 + TH, H);
    const int tw0 = tx << tws, tw1 = min(tw0 + TW, Wd);
    RGT_STAMP(0);
    const unsigned char *const plane8 = v.plane8;
    const void *const base_c = plane8 + slice * 64, *const base_t = v.top_diff + slice * 64;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)base_c, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc16 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const unsigned short *)(plane8 + (long long)R * PHW * C) + slice * 64), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)base_t, 0, 0x7fffffff, 0x00020000);
    const rgt_v4i qc = rgt_rsrc(base_c), qt = rgt_rsrc(base_t);
    // ---- the drain pipeline: register sets 0..3 as a queue of groups in flight (oldest = set qh, nq of them).  Ring positions
    // (wave-uniform): done <= head <= tail -- entries before `done` are added, before `head` requested (a group's entries stay in the ring
    // until it is added: <= 24 in flight + < 8 waiting + <= 64 appended at once <= RGT_RING)
    int done = 0, head = 0, tail = 0;
    int nq = 0, qh = 0;
    long long t_wait = 0, t_cons = 0, t_issue = 0;                     // experiment builds: ticks inside the pump's phases
    // request the next eight entries of the ring as a group; with three groups in flight the oldest one is retired: waited for, the
    // new group requested, then added -- its adds run under the flight of the three younger groups
    auto pump = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // (ring entries written by other lanes of this wave)
        __builtin_amdgcn_wave_barrier();
        const int2 en = ring[(head + (lane & (RGT_G - 1))) & (RGT_RING - 1)];
        head += RGT_G;
        if (nq == 3) {
            const int2 eo = ring[(done + (lane & (RGT_G - 1))) & (RGT_RING - 1)];
            done += RGT_G;
            float td[RGT_G];
            unsigned cd[RGT_G];
            long long c0 = 0, c1 = 0;
            if (RGT_DBG(1, 1) && p.trace) { c0 = (long long)wall_clock64(); asm volatile("s_waitcnt vmcnt(32)"); c1 = (long long)wall_clock64(); t_wait += c1 - c0; }
            switch (qh) {
            case 0: RGT_LAND(32, RGT_SET0); RGT_ISSUE(RGT_SET3) break;
            case 1: RGT_LAND(32, RGT_SET1); RGT_ISSUE(RGT_SET0) break;
            case 2: RGT_LAND(32, RGT_SET2); RGT_ISSUE(RGT_SET1) break;
            default: RGT_LAND(32, RGT_SET3); RGT_ISSUE(RGT_SET2) break;
            }
            if (RGT_DBG(1, 1) && p.trace) c0 = (long long)wall_clock64();
            rgt_consume(acc, td, cd, eo, rc, rc16, rt, lane);
            if (RGT_DBG(1, 1) && p.trace) { __builtin_amdgcn_s_waitcnt(0xc07f); t_cons += (long long)wall_clock64() - c0; t_issue += c0 - c1; }
            qh = (qh + 1) & 3;
        } else {                                                       // filling the queue: qh == 0, the new group goes to set nq
            switch (nq) {
            case 0: RGT_ISSUE(RGT_SET0) break;
            case 1: RGT_ISSUE(RGT_SET1) break;
            default: RGT_ISSUE(RGT_SET2) break;
            }
            ++nq;
        }
    };
    for (int base = 0; base < R; base += 256) {
        // ---- the rows of 256 ROIs, requested at once (lane = ROI, four passes)
        float rr[4][5];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int roi = min(base + 64 * q + lane, R - 1);
#pragma unroll
            for (int u = 0; u < 5; ++u) rr[q][u] = v.rois[5 * (long long)roi + u];
        }
        if (base == 0) {                                               // (under the ROI rows' latency) the accumulators start at +0
            for (int i = lane; i < (RGT_MAXPX + 1) * 16; i += 64) reinterpret_cast<float4 *>(acc)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        // ---- filter: ROI against the tile; the survivors' rounded geometry compacted (ascending) into LDS -- the expansion below keeps no
        // per-ROI register
        int nhit = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int roi0 = base + 64 * q;
            const RoiGeom g = roi_geom(rr[q], v.scale);
            const int rw = max(g.rew - g.rsw + 1, 1), rh = max(g.reh - g.rsh + 1, 1);   // roi_pooling_op.cc:146-147
            // conservative bounding box of the ROI's bins (every bin's rows lie in [rsh, rsh + rh + 1]); coordinates outside the
            // range where that arithmetic is exact (NaN / inf / absurd boxes) are left to the exact per-bin test of the expansion
            const bool sane = abs(g.rsw) < (1 << 24) && abs(g.rsh) < (1 << 24) && abs(g.rew) < (1 << 24) && abs(g.reh) < (1 << 24);
            // (:401-404: h in [rsh, reh], w in [rsw, rew] -- an end before its start lets nothing through)
            bool hit = roi0 + lane < R && (int)rr[q][0] == b && g.reh >= g.rsh && g.rew >= g.rsw;
            if (sane) hit = hit && g.rsh < th1 && g.rsh + rh + 2 > th0 && g.rsw < tw1 && g.rsw + rw + 2 > tw0;
            if (RGT_DBG(p.dbg, 8 | 1)) hit = false;
            const unsigned long long mh = __ballot(hit);
            if (hit) {
                const int pos = nhit + __popcll(mh & ((1ull << lane) - 1ull));
                geo[pos] = make_int4(g.rsh, g.rsw, g.reh, g.rew);
                ridx[pos] = (unsigned char)(64 * q + lane);
            }
            nhit += __popcll(mh);
        }
        if (base == 0) RGT_STAMP(1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- expansion + drain
        {
            for (int i = 0; i < nhit; ++i) {
                const int4 gg = geo[i];
                const int r = base + (int)ridx[i];
                const int rsh = __builtin_amdgcn_readfirstlane(gg.x), rsw = __builtin_amdgcn_readfirstlane(gg.y);
                const int reh = __builtin_amdgcn_readfirstlane(gg.z), rew = __builtin_amdgcn_readfirstlane(gg.w);
                const int rhj = reh - rsh + 1, rwj = rew - rsw + 1;    // (>= 1: the filter drops the others)
                const float bh = (float)rhj / (float)PH;               // roi_pooling_op.cc:148-151 (only for hits)
                const float bw = (float)rwj / (float)PW;
                const bool exact = rhj > RGT_EXACT_MAX || rwj > RGT_EXACT_MAX;
                for (int bin0 = 0; bin0 < PHW; bin0 += 64) {
                    const int bin = bin0 + lane;
                    const int ph = (int)(((unsigned)bin * (unsigned)p.inv_pw) >> 16), pw = bin - ph * PW;
                    // the bin's rectangle as the forward computes it (roi_pool.hip fwd_bin_rect, roi_pooling_op.cc:153-162)
                    const int hs0 = (int)floorf(__fmul_rn((float)ph, bh)), ws0 = (int)floorf(__fmul_rn((float)pw, bw));
                    const int he0 = (int)ceilf(__fmul_rn((float)(ph + 1), bh)), we0 = (int)ceilf(__fmul_rn((float)(pw + 1), bw));
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
                    // ---- one entry per row of the tile a bin's rectangle meets, rows outermost
                    for (int h = th0; h < th1; ++h) {
                        const bool on = meets && ih0 <= h && h <= ih1;
                        const unsigned long long mr = __ballot(on);
                        if (mr == 0ull) continue;
                        if (on) ring[(tail + __popcll(mr & ((1ull << lane) - 1ull))) & (RGT_RING - 1)] = make_int2(ex + ((h - th0) << tws), ey + h * bwid);
                        tail += __popcll(mr);
                        if (RGT_DBG(p.dbg, 2)) { done = head = tail; continue; }
                        while (tail - head >= RGT_G) pump();
                    }
                }
            }
        }
    }
    RGT_STAMP(3);
    if (RGT_DBG(1, 1) && p.trace && lane == 0) {
        p.trace[(long long)blockIdx.x * 8 + 6] = tail | (t_issue << 32);
        p.trace[(long long)blockIdx.x * 8 + 2] = t_wait; p.trace[(long long)blockIdx.x * 8 + 7] = t_cons;
    }
    // ---- the stream padded to whole groups with null entries (no pixel, a cached record), the last group requested
    if (tail > head) {
        if (lane < RGT_G) ring[(tail + lane) & (RGT_RING - 1)] = make_int2(0, RGT_NULLCODE);     // (lanes past the pad write slots nobody reads)
        tail = (tail + RGT_G - 1) & ~(RGT_G - 1);
        // (positions: head and done only ever move by whole groups, so head is a multiple of eight here)
        pump();
    }
    // ---- the groups still in flight, oldest first (nothing is requested any more: one full wait)
    for (; nq > 0; --nq, qh = (qh + 1) & 3) {
        const int2 eo = ring[(done + (lane & (RGT_G - 1))) & (RGT_RING - 1)];
        done += RGT_G;
        float td[RGT_G];
        unsigned cd[RGT_G];
        switch (qh) {
        case 0: RGT_LAND(0, RGT_SET0); break;
        case 1: RGT_LAND(0, RGT_SET1); break;
        case 2: RGT_LAND(0, RGT_SET2); break;
        default: RGT_LAND(0, RGT_SET3); break;
        }
        rgt_consume(acc, td, cd, eo, rc, rc16, rt, lane);
    }
    RGT_STAMP(4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");             // (the accumulators: written per channel lane, read 16 B per lane)
    __builtin_amdgcn_wave_barrier();
    // ---- write-out: every pixel of the tile, zeros included (16 B per lane: 4 pixels x 256 bytes per store instruction)
    constexpr int LPP = 16;                                            // lanes per pixel
    float *const out = v.bottom_diff + (long long)b * H * Wd * C + slice * 64 + (lane & (LPP - 1)) * 4;
    const int npx = TH * TW;
    typedef float f4v __attribute__((ext_vector_type(4)));
    for (int i = 0; i < npx; i += 64 / LPP) {
        const int px = i + lane / LPP;
        const int h = th0 + (px >> tws), w = tw0 + (px & (TW - 1));
        if (px < npx && h < H && w < Wd) {
            const float4 x = *reinterpret_cast<const float4 *>(acc + px * 64 + (lane & (LPP - 1)) * 4);
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
        v.nsl = w.channels / 64;
        v.first_block = blocks;
        const long long nb = (long long)w.batch_size * v.tiles_y * v.tiles_x * v.nsl;
        if (nb + blocks > 0x7fffffffLL) return MV3D_ERR_INVALID_ARG;
        blocks += (unsigned)nb;
    }
    for (int k = num_views; k < MV3D_MAX_ROI_VIEWS; ++k) p.v[k] = p.v[0];
    p.dbg = rgt_env("MV3D_RGT_DBG", 0);
    p.trace = nullptr;
#ifdef MV3D_TUNING
    if (getenv("MV3D_RGT_TRACE")) p.trace = (long long *)strtoull(getenv("MV3D_RGT_TRACE"), nullptr, 0);
#endif
    hipLaunchKernelGGL(roi_pair_tiles_kernel, dim3(blocks), dim3(64), 0, stream, p);
    return mv3d_launch_status();
}
