// RoiPoolGrad for gfx950 as an ORDERED per-tile record stream, all views of a step in one launch, no workspace.
// Replaces lib/roi_pooling_layer/roi_pooling_op.cc:319-452 (CPU) / roi_pooling_op_gpu.cu.cc:113-215 (CUDA); same sums,
// bit for bit: per (pixel, channel) the f32 adds happen ROIs ascending, then ph, pw ascending (roi_pooling_op.cc:385-443).
//
// The reference is a gather: every input element scans the ROIs that contain its pixel and, for the candidate bins
// [phstart, phend) x [pwstart, pwend) of that pixel (:423-426), adds top_diff where argmax names the element.  A per-pixel
// gather re-fetches a (roi, bin) record (argmax + top_diff, 8 B per channel) once per candidate PIXEL -- 1.4 to 3.3 times
// on the training workload -- and needs a per-pixel index built by extra launches.  Here:
//   * the unit of work is a TILE of TH x TW pixels (1 x 1 ... 2 x 4) of one frame's bottom_diff; a workgroup has one wave
//     per 64-channel slice, the accumulators of a wave are one VGPR per tile pixel.  A tile that no ROI reaches just writes
//     zeros: the zero fill of bottom_diff is the write-out of this kernel;
//   * geometry, ONCE per tile (all waves of the workgroup): the ROIs of the frame whose rounded rectangle meets the tile
//     (the rounded ROIs of the view are cached in LDS); for each, per tile row the reference's [phstart, phend) and per
//     tile column [pwstart, pwend) (the same f32 divides), transposed into per-ph row masks and per-pw column masks.  A
//     record (roi, ph, pw) is LIVE for the tile when both masks are non-empty; its candidate pixels inside the tile are
//     rowmask x colmask -- exactly the pixels whose reference loop visits that bin.  Live records are listed in LDS in
//     reference order: position = the ROI's offset (prefix sum over live-bin counts) + the bin's rank among the ROI's
//     live bins;
//   * stream: every wave walks the list once, its slice of a record = two 256-B loads (argmax, top_diff) whose record
//     offset is a scalar; two groups of RT_W records in flight.  Per record and tile pixel: a compare against the pixel's
//     index (replaced by an impossible value when the pixel is not a candidate: two scalar instructions), a select and an
//     add, in list order -- ~8 instructions per record at 1 x 1, ~45 at 2 x 4: a single wave issues an instruction every
//     4 - 5 clocks, so the instruction count per record is what a long list costs;
//   * a record is fetched once per TILE it can reach, by ONE workgroup in whole 2-KB pieces per array;
//   * PERSISTENT and software-pipelined: the workgroups of a view walk its tiles b, b + n, ...; the list of tile j + 1 is
//     built (LDS and VALU work between barriers) while the first record groups of tile j are in flight, so neither the
//     geometry nor the first memory round trip of a tile is paid in sequence.  Workgroups are split over the views in
//     proportion to their estimated work.
// Adding +0.0f for a non-matching channel is bit-neutral (a sum that started at +0.0f is never -0.0f).
// Measured on the way: tile accumulators in LDS updated with ds_add_f32 -- the LDS float atomic runs at ~0.5
// lane / clock / CU (72 us of the launch); decoding argmax to a pixel per lane -- more VALU than comparing against the few
// candidate pixels; one workgroup per tile without the pipeline -- every phase is a latency and they add up; 4 x 4 tiles
// with SGPR-indexed accumulators -- ~80 instructions per record, a 450-record workgroup took 180 us.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "roi_geom.h"

#define RT_MAXT 512                  // threads per workgroup, at most (8 slices)
#define RT_PASS 256                  // ROIs filtered per pass
#define RT_CAP 1024                  // live records listed per batch
#define RT_KB 64                     // overlapping ROIs evaluated per batch (one wave scans their counts)
#define RT_W 16                      // records per load group; two groups in flight
#define RT_OCC_WORDS 2048             // tile occupancy bitmap (64 K tiles per view; larger views go without)

struct RoiTileViewDev {
    const float *top_diff, *rois;
    const int *argmax;
    float *bottom_diff;
    float scale;
    int B, R, H, W, C;
    int shape;                       // 0: 1 x 1, 1: 1 x 2, 2: 2 x 2, 3: 2 x 4 pixels per tile
    int tiles_x, tiles_y;
    unsigned tiles, first_block, blocks;   // tiles of the view; its workgroups are [first_block, first_block + blocks)
    unsigned magic_tpf, magic_tx;    // ceil(2^32 / (tiles_x tiles_y)), ceil(2^32 / tiles_x)
    unsigned rec_bytes;              // R * PH * PW * C * 4 < 2^31
};
struct RoiTilePack {
    RoiTileViewDev v[MV3D_MAX_ROI_VIEWS];
    int n, PH, PW;
    unsigned magic_phw, magic_pw;
    int dbg;                         // tuning builds: 1 = no stream, 4 = no geometry
    long long *trace;                // tuning builds: 8 words per workgroup
};

__device__ __forceinline__ unsigned rt_div(unsigned t, unsigned magic, int d) { return d == 1 ? t : __umulhi(t, magic); }

struct RtTile {                      // wave-uniform description of a tile and of its first batch
    int n, ty0, tx0, ty1, tx1;       // frame, pixel rectangle (clipped to the map)
    int K, kcut, nrec;               // pass 0: overlapping ROIs, ROIs / records in the first batch
};

template <int NPX>
struct RtStream {
    __amdgpu_buffer_rsrc_t ra, rt;   // argmax / top_diff of the view
    const int2 *list;                // {record byte offset, pixel mask}
    int n, voff, chan, dbg;
    int wq[NPX];                     // (pixel index inside the frame) * C per tile pixel
};

// group g = records [g * RT_W, +RT_W) of the list: lane u < RT_W reads entry u, the record's byte offset goes to an SGPR
template <int NPX>
__device__ __forceinline__ void rt_issue(const RtStream<NPX> &s, const int g, int2 &ent, int (&am)[RT_W], float (&td)[RT_W])
{
    const int lane = threadIdx.x & 63;
    ent = s.list[min(g * RT_W + (lane & (RT_W - 1)), s.n - 1)];
#pragma unroll
    for (int u = 0; u < RT_W; ++u) {
        const int off = __builtin_amdgcn_readlane(ent.x, u);
        am[u] = (int)__builtin_amdgcn_raw_buffer_load_b32(s.ra, s.voff, off, 0);
        td[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s.rt, s.voff, off, 0));
    }
}

template <int NPX>
__device__ __forceinline__ void rt_consume(const RtStream<NPX> &s, const int g, const int2 &ent, const int (&am)[RT_W],
                                           const float (&td)[RT_W], float (&acc)[NPX])
{
#pragma unroll
    for (int u = 0; u < RT_W; ++u) {
        if (g * RT_W + u >= s.n) break;                                // wave-uniform
        const int rel = am[u] - s.chan;
        if (NPX == 1) {
            acc[0] += (rel == s.wq[0]) ? td[u] : 0.0f;                 // (a listed record has the tile's pixel as candidate)
        } else {
            // only the record's candidate pixels (1 - 3 of the tile's, typically) cost vector instructions: a scalar branch
            // per pixel of the tile (the empty asm keeps the compiler from turning the branches into selects)
            const unsigned m = (unsigned)__builtin_amdgcn_readlane(ent.y, u);
#pragma unroll
            for (int q = 0; q < NPX; ++q)
                if (m & (1u << q)) {
                    asm volatile("");
                    acc[q] += (rel == s.wq[q]) ? td[u] : 0.0f;
                }
        }
    }
}

// groups [gi, G) of the list in order, group gi already issued into A (and gi + 1 into B when it exists)
template <int NPX>
__device__ __forceinline__ void rt_drain(const RtStream<NPX> &s, int gi, const int G, int2 &eA, int (&amA)[RT_W], float (&tdA)[RT_W],
                                         int2 &eB, int (&amB)[RT_W], float (&tdB)[RT_W], float (&acc)[NPX])
{
    while (gi + 3 < G) {
        rt_consume<NPX>(s, gi, eA, amA, tdA, acc);
        rt_issue<NPX>(s, gi + 2, eA, amA, tdA);
        rt_consume<NPX>(s, gi + 1, eB, amB, tdB, acc);
        rt_issue<NPX>(s, gi + 3, eB, amB, tdB);
        gi += 2;
    }
    const int rem = G - gi;                                            // 1, 2 or 3 groups left, the first two are in flight
    rt_consume<NPX>(s, gi, eA, amA, tdA, acc);
    if (rem == 3) rt_issue<NPX>(s, gi + 2, eA, amA, tdA);
    if (rem >= 2) rt_consume<NPX>(s, gi + 1, eB, amB, tdB, acc);
    if (rem == 3) rt_consume<NPX>(s, gi + 2, eA, amA, tdA, acc);
}

struct RtShared {
    int2 list[2][RT_CAP];                                             // {record byte offset, pixel mask}, double-buffered
    int4 geo[RT_PASS];                                                // rounded ROIs of the cached pass: (rsw, rsh, rew, reh)
    int frame[RT_PASS];
    int ov_roi[RT_PASS];
    int4 ov_g[RT_PASS];                                               // the ROIs that meet the tile, ascending
    unsigned short se[RT_KB][8];                                      // per tile row (0..3) / column (4..7): start | end << 8
    unsigned char pm[RT_KB][32];                                      // per ph (0..15) row mask, per pw (16..31) column mask
    int live[RT_KB], off[RT_KB];
    unsigned occ[RT_OCC_WORDS];                                       // bit per tile of the view: some ROI's rectangle meets it
    unsigned long long bal[2][RT_PASS / 64];                          // alternating between calls: one barrier per call
    int misc[2];
};

template <int TH, int TWL>
__device__ __forceinline__ void rt_run(const RoiTilePack &p, const RoiTileViewDev &v, const unsigned b, RtShared &sh, long long *tr)
{
    constexpr int TW = 1 << TWL, NPX = TH * TW;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int PH = p.PH, PW = p.PW, PHW = PH * PW, H = v.H, W = v.W, C = v.C, R = v.R;
    const int ps = min(nt, RT_PASS);                                  // ROIs filtered per pass
    const int npass = (R + ps - 1) / ps;
    int cpass = -1;                                                   // the pass sh.geo / sh.frame hold
    int fcall = 0;
#ifdef MV3D_TUNING
    long long t_geo = 0, t_str = 0, t_mark = 0;
    int n_listed = 0, n_tiles = 0;
#define RT_MARK() do { if (tr) t_mark = (long long)__builtin_readcyclecounter(); } while (0)
#define RT_ACC(x) do { if (tr) (x) += (long long)__builtin_readcyclecounter() - t_mark; } while (0)
#else
#define RT_MARK()
#define RT_ACC(x)
#endif

    auto decode = [&](const unsigned tile, RtTile &t) {
        const int tpf = v.tiles_x * v.tiles_y;
        const unsigned n = rt_div(tile, v.magic_tpf, tpf), rem = tile - n * (unsigned)tpf;
        const unsigned tyi = rt_div(rem, v.magic_tx, v.tiles_x), txi = rem - tyi * (unsigned)v.tiles_x;
        t.n = (int)n; t.ty0 = (int)tyi * TH; t.tx0 = (int)txi << TWL;
        t.ty1 = min(t.ty0 + TH, H) - 1; t.tx1 = min(t.tx0 + TW, W) - 1;
    };

    // ---- G0: the ROIs of pass `pass` of frame t.n whose rounded rectangle meets the tile (roi_pooling_op.cc:392-403);
    // returns their number; when there are any, they are listed ascending in sh.ov_* (and the call ends with a barrier)
    auto filter = [&](const RtTile &t, const int pass) -> int {
        if (cpass != pass) {
            __syncthreads();                                           // (nobody still reads the cached ROIs)
            const int r = pass * ps + tid;
            if (tid < ps && r < R) {
                const float *roi = v.rois + 5 * (long long)r;
                const RoiGeom g = roi_geom(roi, v.scale);
                sh.geo[tid] = make_int4(g.rsw, g.rsh, g.rew, g.reh);
                sh.frame[tid] = (int)roi[0];
            }
            cpass = pass;
            __syncthreads();
        }
        bool ok = false;
        int4 g = make_int4(0, 0, 0, 0);
        if (tid < ps && pass * ps + tid < R) {
            g = sh.geo[tid];
            ok = sh.frame[tid] == t.n && g.w >= t.ty0 && g.y <= t.ty1 && g.z >= t.tx0 && g.x <= t.tx1;
        }
        const unsigned long long bal = __ballot(ok);
        fcall ^= 1;
        if (lane == 0 && wave < RT_PASS / 64) sh.bal[fcall][wave] = bal;
        __syncthreads();
        int pos = __popcll(bal & ((1ull << lane) - 1ull)), K = 0;
#pragma unroll
        for (int w = 0; w < RT_PASS / 64; ++w) { const int cw = w * 64 < ps ? __popcll(sh.bal[fcall][w]) : 0; if (w < wave) pos += cw; K += cw; }
#ifdef MV3D_TUNING
        if (p.dbg & 4) K = 0;
#endif
        if (K > 0) {
            if (ok) { sh.ov_roi[pos] = pass * ps + tid; sh.ov_g[pos] = g; }
            __syncthreads();
        }
        return K;
    };

    // ---- G1 / G2: ROIs [k0, k0 + kb) of sh.ov_* -> the live records of the tile in reference order in `list`; returns the
    // number of ROIs taken (those whose records fit the list) and the number of records.  Ends with a barrier.
    auto build = [&](const RtTile &t, const int k0, const int K, int2 *list, int &kcut, int &nrec) {
        const int kb = min(K - k0, RT_KB);
        // G1a: (roi, tile row | tile column) -> the reference's [start, end) of pooled indices (:423-426)
        for (int i0 = tid; i0 < kb * 8; i0 += nt) {
            const int kk = i0 >> 3, i = i0 & 7, ii = i & 3;
            const bool isrow = i < 4;
            const int4 q = sh.ov_g[k0 + kk];
            const int ty0 = t.ty0, tx0 = t.tx0, ph_ = PH, pw_ = PW, h_ = H, w_ = W;
            const int x = (isrow ? ty0 : tx0) + ii, lo = isrow ? q.y : q.x, hi = isrow ? q.w : q.z;
            const int P = isrow ? ph_ : pw_;
            const bool valid = isrow ? (ii < TH && x < h_) : (ii < TW && x < w_);
            unsigned se = 0;
            if (valid && x >= lo && x <= hi) {
                const float bsz = (float)max(hi - lo + 1, 1) / (float)P;
                int st = (int)floorf((float)(x - lo) / bsz), en = (int)ceilf((float)(x - lo + 1) / bsz);
                st = min(max(st, 0), P); en = min(max(en, 0), P);
                se = (unsigned)st | ((unsigned)en << 8);
            }
            sh.se[kk][i] = (unsigned short)se;
        }
        __syncthreads();
        // G1b: transposed: (roi, ph) -> mask of tile rows whose range holds ph; (roi, pw) -> mask of columns
        for (int i0 = tid; i0 < kb * 32; i0 += nt) {
            const int kk = i0 >> 5, pi = i0 & 31, pp = pi & 15;
            const int ph_ = PH, pw_ = PW;
            const int base = pi < 16 ? 0 : 4, P = pi < 16 ? ph_ : pw_;
            unsigned m = 0;
            if (pp < P) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned se = sh.se[kk][base + i];
                    if (pp >= (int)(se & 255u) && pp < (int)(se >> 8)) m |= 1u << i;
                }
            }
            sh.pm[kk][pi] = (unsigned char)m;
        }
        __syncthreads();
        // G1c: live bins per ROI = (#ph with rows) x (#pw with columns); offsets; how many ROIs fit the list
        if (tid < 64) {
            unsigned rl = 0, cl = 0;
            if (tid < kb) {
                for (int ph = 0; ph < PH; ++ph) if (sh.pm[tid][ph]) rl |= 1u << ph;
                for (int pw = 0; pw < PW; ++pw) if (sh.pm[tid][16 + pw]) cl |= 1u << pw;
            }
            const int cnt = __popc(rl) * __popc(cl);
            int inc = cnt;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { const int x = __shfl_up(inc, m); if (lane >= m) inc += x; }
            const bool fit = tid < kb && inc <= RT_CAP;
            const int kc = __popcll(__ballot(fit));                   // (inc is monotone: the fitting ROIs are a prefix)
            if (tid < kb) { sh.live[tid] = (int)(rl | (cl << 16)); sh.off[tid] = inc - cnt; }
            if (tid == kc - 1) sh.misc[1] = inc;
            if (tid == 0) sh.misc[0] = kc;
        }
        __syncthreads();
        kcut = sh.misc[0]; nrec = sh.misc[1];
        // G2: position of a live record = ROI offset + rank among the ROI's live bins
        for (unsigned i0 = tid; i0 < (unsigned)(kcut * PHW); i0 += nt) {
            const unsigned kk = rt_div(i0, p.magic_phw, PHW), bin = i0 - kk * (unsigned)PHW;
            const unsigned ph = rt_div(bin, p.magic_pw, PW), pw = bin - ph * (unsigned)PW;
            const unsigned rm = sh.pm[kk][ph], cm = sh.pm[kk][16 + pw];
            if (rm && cm) {
                const unsigned live = (unsigned)sh.live[kk], rl = live & 0xffffu, cl = live >> 16;
                const int rank = __popc(rl & ((1u << ph) - 1u)) * __popc(cl) + __popc(cl & ((1u << pw) - 1u));
                unsigned pm = 0;
#pragma unroll
                for (int i = 0; i < TH; ++i) if (rm & (1u << i)) pm |= cm << (i * TW);
                list[sh.off[kk] + rank] = make_int2((sh.ov_roi[k0 + kk] * PHW + (int)bin) * C * 4, (int)pm);
            }
        }
        __syncthreads();
    };

    // ---- which tiles does any ROI reach at all (typically a fifth of them): one pass over the ROIs per workgroup, a bit per
    // tile.  A tile whose bit is clear is finished by its zero write-out, without a barrier.
    const bool use_occ = v.tiles <= RT_OCC_WORDS * 32u;
    if (use_occ) {
        for (int i = tid; i < (int)((v.tiles + 31u) >> 5); i += nt) sh.occ[i] = 0u;
        __syncthreads();
        for (int r = tid; r < R; r += nt) {
            const float *roi = v.rois + 5 * (long long)r;
            const RoiGeom g = roi_geom(roi, v.scale);
            const int fr = (int)roi[0];
            const int y0 = max(g.rsh, 0), y1 = min(g.reh, H - 1), x0 = max(g.rsw, 0), x1 = min(g.rew, W - 1);
            if (fr >= 0 && fr < v.B && y0 <= y1 && x0 <= x1) {
                for (int ty = y0 / TH; ty <= y1 / TH; ++ty) {
                    const unsigned base = (unsigned)(fr * v.tiles_y + ty) * (unsigned)v.tiles_x;
                    const unsigned i0 = base + (unsigned)(x0 >> TWL), i1 = base + (unsigned)(x1 >> TWL);
                    for (unsigned wi = i0 >> 5; wi <= (i1 >> 5); ++wi) {
                        const unsigned lo = max(i0, wi << 5) & 31u, hi = min(i1, (wi << 5) + 31u) & 31u;
                        atomicOr(&sh.occ[wi], (0xffffffffu >> (31u - hi)) & (0xffffffffu << lo));
                    }
                }
            }
        }
        __syncthreads();
    }

    // first batch of a tile (pass 0, ROIs from 0) into `list`
    auto prepare = [&](const unsigned tile, RtTile &t, int2 *list) {
        RT_MARK();
        decode(tile, t);
        t.K = 0; t.kcut = 0; t.nrec = 0;
        const bool reached = !use_occ || ((__builtin_amdgcn_readfirstlane(sh.occ[tile >> 5]) >> (tile & 31u)) & 1u);
        if (reached) {
            t.K = filter(t, 0);
            if (t.K > 0) build(t, 0, t.K, list, t.kcut, t.nrec);
        }
        RT_ACC(t_geo);
    };

    const unsigned stride = v.blocks;
    unsigned tile = b;
    if (tile >= v.tiles) return;
    RtStream<NPX> s;
    s.ra = __builtin_amdgcn_make_buffer_rsrc((void *)v.argmax, 0, (int)v.rec_bytes, 0x00020000);
    s.rt = __builtin_amdgcn_make_buffer_rsrc((void *)v.top_diff, 0, (int)v.rec_bytes, 0x00020000);
    s.chan = wave * 64 + lane; s.voff = s.chan * 4; s.dbg = p.dbg;
    RtTile cur, nxt;
    prepare(tile, cur, sh.list[0]);
    int2 eA, eB;
    int amA[RT_W], amB[RT_W];
    float tdA[RT_W], tdB[RT_W];
    for (int j = 0;; ++j) {
        int2 *list = sh.list[j & 1];
        s.list = list; s.n = cur.nrec;
#pragma unroll
        for (int q = 0; q < NPX; ++q) s.wq[q] = ((cur.ty0 + (q >> TWL)) * W + cur.tx0 + (q & (TW - 1))) * C;
        float acc[NPX];
#pragma unroll
        for (int q = 0; q < NPX; ++q) acc[q] = 0.0f;
        int G = (cur.nrec + RT_W - 1) / RT_W;
#ifdef MV3D_TUNING
        n_listed += cur.nrec; ++n_tiles;
        if (p.dbg & 1) G = 0;
#endif
        // the first record groups of this tile go out, then the next tile's list is built while they fly
        if (G >= 1) rt_issue<NPX>(s, 0, eA, amA, tdA);
        if (G >= 2) rt_issue<NPX>(s, 1, eB, amB, tdB);
        const unsigned tnext = tile + stride;
        const bool more = tnext < v.tiles;
        if (more) prepare(tnext, nxt, sh.list[(j & 1) ^ 1]);
        RT_MARK();
        if (G >= 1) rt_drain<NPX>(s, 0, G, eA, amA, tdA, eB, amB, tdB, acc);
        // the rest of the tile, not pipelined (rare): further batches of pass 0 (more than RT_KB overlapping ROIs or more
        // than RT_CAP live records) and further passes (R > RT_PASS)
        if (cur.kcut < cur.K || (npass > 1 && (!use_occ || ((__builtin_amdgcn_readfirstlane(sh.occ[tile >> 5]) >> (tile & 31u)) & 1u)))) {
            for (int pass = 0; pass < npass; ++pass) {
                __syncthreads();
                const int K = filter(cur, pass);                       // (the scratch of pass 0 was reused by prepare())
                for (int k0 = pass == 0 ? cur.kcut : 0; k0 < K;) {
                    int kc = 0, nr = 0;
                    build(cur, k0, K, list, kc, nr);
                    s.n = nr;
                    const int G2 = (nr + RT_W - 1) / RT_W;
                    if (G2 >= 1) {
                        rt_issue<NPX>(s, 0, eA, amA, tdA);
                        if (G2 >= 2) rt_issue<NPX>(s, 1, eB, amB, tdB);
                        rt_drain<NPX>(s, 0, G2, eA, amA, tdA, eB, amB, tdB, acc);
                    }
                    k0 += kc;
                    __syncthreads();                                   // the list is rebuilt
                }
            }
        }
        RT_ACC(t_str);
        // write-out: every pixel of the tile, this wave's 256 B per pixel (streaming stores)
        {
            float *out = v.bottom_diff + (long long)cur.n * H * W * C + s.chan;
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                const int h = cur.ty0 + (q >> TWL), w = cur.tx0 + (q & (TW - 1));
                if (h <= cur.ty1 && w <= cur.tx1) __builtin_nontemporal_store(acc[q], out + ((long long)h * W + w) * C);
            }
        }
        if (!more) break;
        cur = nxt; tile = tnext;
    }
#ifdef MV3D_TUNING
    if (tr && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[7] = (long long)__builtin_readcyclecounter(); tr[3] = t_geo; tr[4] = t_str; tr[5] = n_listed; tr[6] = n_tiles; }
#endif
}

__global__ __launch_bounds__(RT_MAXT) void roi_grad_tile_kernel(RoiTilePack p)
{
    __shared__ RtShared sh;
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && blockIdx.x >= p.v[j].first_block) k = j;
    const RoiTileViewDev v = p.v[k];                                  // (one view per workgroup: a copy in SGPRs)
    const unsigned b = blockIdx.x - v.first_block;
    long long *tr = nullptr;
#ifdef MV3D_TUNING
    tr = p.trace ? p.trace + 8 * (long long)blockIdx.x : nullptr;
    if (tr && threadIdx.x == 0) { tr[0] = (long long)__builtin_readcyclecounter(); tr[1] = k; }
#endif
    switch (v.shape) {
    case 0: rt_run<1, 0>(p, v, b, sh, tr); break;
    case 1: rt_run<1, 1>(p, v, b, sh, tr); break;
    case 2: rt_run<2, 1>(p, v, b, sh, tr); break;
    default: rt_run<2, 2>(p, v, b, sh, tr); break;
    }
}

static unsigned rt_magic(int d) { return d <= 1 ? 0u : (unsigned)(0xFFFFFFFFull / (unsigned)d) + 1u; }

static const int rt_shapes[4][2] = {{1, 0}, {1, 1}, {2, 1}, {2, 2}};         // rows, log2(columns)

// Tile shape of a view: records per pixel = R PH PW / pixels (if every ROI lay on the map).  Dense maps (the 8 x 64 front
// view under 128 ROIs per frame: 12 records per pixel) get single pixels -- the per-record work of the stream grows with the
// tile and their lists are long --, sparse maps larger tiles (a record meets fewer of them: less traffic).
static int rt_pick_shape(const mv3d_roi_grad_view &w, int PH, int PW)
{
    const double dens = (double)w.num_rois * PH * PW / ((double)w.batch_size * w.height * w.width);
    int best = 0;
    for (int i = 0; i < 4; ++i)
        if ((rt_shapes[i][0] << rt_shapes[i][1]) * dens <= 10.0) best = i;
    return best;
}

bool mv3d_roi_grad_tiles_ok(int num_views, const mv3d_roi_grad_view *views, int PH, int PW)
{
    if (PH > 15 || PW > 15 || PH * PW > 255) return false;
    const int C = views[0].channels;
    if (C < 64 || (C & (C - 1)) || C > 512) return false;               // 64-channel slices, <= 8 waves per workgroup
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[k];
        if (w.channels != C) return false;
        if (((uintptr_t)w.bottom_diff | (uintptr_t)w.top_diff | (uintptr_t)w.argmax_data) & 15) return false;
        if ((long long)w.num_rois * PH * PW * C * 4 >= 0x7fffffffLL) return false;         // record byte offsets are i32
        if ((long long)w.height * w.width * C >= 0x7fffffffLL) return false;
        if ((long long)w.batch_size * w.height * w.width >= (1ll << 24)) return false;     // (tile index) x (tiles per frame) < 2^32
    }
    return true;
}

int mv3d_launch_roi_grad_tiles(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, const int *tile_override,
                               hipStream_t stream)
{
    RoiTilePack p;
    p.n = num_views; p.PH = PH; p.PW = PW;
    const int nsl = views[0].channels / 64;
    p.magic_phw = rt_magic(PH * PW); p.magic_pw = rt_magic(PW);
    p.dbg = 0; p.trace = nullptr;
    int groups = 512;                                                  // persistent workgroups: 2 per CU
#ifdef MV3D_TUNING
    if (getenv("MV3D_RT_DBG")) p.dbg = atoi(getenv("MV3D_RT_DBG"));
    if (getenv("MV3D_RT_TRACE")) p.trace = (long long *)strtoull(getenv("MV3D_RT_TRACE"), nullptr, 10);
    if (getenv("MV3D_RT_GROUPS")) groups = atoi(getenv("MV3D_RT_GROUPS"));
#endif
    // workgroups per view in proportion to the estimated work: a fixed cost per tile (geometry + write-out) and the records
    // a tile streams (every record of the view once per tile it reaches: ~1 + (PH + PW) / (rows + columns of the ROI in tiles))
    double weight[MV3D_MAX_ROI_VIEWS], wsum = 0.0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[k];
        RoiTileViewDev &v = p.v[k];
        v.top_diff = w.top_diff; v.rois = w.bottom_rois; v.argmax = w.argmax_data; v.bottom_diff = w.bottom_diff;
        v.scale = w.spatial_scale;
        v.B = w.batch_size; v.R = w.num_rois; v.H = w.height; v.W = w.width; v.C = w.channels;
        v.shape = rt_pick_shape(w, PH, PW);
        if (tile_override && tile_override[2 * k] > 0) {
            v.shape = -1;
            for (int i = 0; i < 4; ++i) if (rt_shapes[i][0] == tile_override[2 * k] && rt_shapes[i][1] == tile_override[2 * k + 1]) v.shape = i;
            if (v.shape < 0) return MV3D_ERR_INVALID_ARG;
        }
        const int th = rt_shapes[v.shape][0], twl = rt_shapes[v.shape][1];
        v.tiles_x = (w.width + (1 << twl) - 1) >> twl;
        v.tiles_y = (w.height + th - 1) / th;
        v.tiles = (unsigned)((long long)w.batch_size * v.tiles_x * v.tiles_y);
        v.magic_tpf = rt_magic(v.tiles_x * v.tiles_y); v.magic_tx = rt_magic(v.tiles_x);
        v.rec_bytes = (unsigned)((long long)w.num_rois * PH * PW * w.channels * 4);
        const double npx = (double)(th << twl);
        weight[k] = v.tiles * (1.0 + 0.1 * npx) + (double)w.num_rois * PH * PW * (0.3 + 0.1 * npx) * (npx >= 8 ? 1.4 : (npx >= 4 ? 1.6 : (npx >= 2 ? 2.0 : 2.4)));
        wsum += weight[k];
    }
#ifdef MV3D_TUNING
    if (const char *e = getenv("MV3D_RT_WEIGHTS")) {                   // "w0,w1,..." per view of the call
        wsum = 0.0;
        int i = 0;
        for (const char *q = e; *q && i < num_views; ++i) { weight[i] = atof(q); while (*q && *q != ',') ++q; if (*q) ++q; }
        for (int k = 0; k < num_views; ++k) wsum += weight[k];
    }
#endif
    unsigned blocks = 0;
    for (int k = 0; k < num_views; ++k) {
        RoiTileViewDev &v = p.v[k];
        unsigned nb = (unsigned)(groups * weight[k] / wsum + 0.5);
        if (nb < 1) nb = 1;
        if (nb > v.tiles) nb = v.tiles;
        v.first_block = blocks; v.blocks = nb;
        blocks += nb;
    }
    for (int k = num_views; k < MV3D_MAX_ROI_VIEWS; ++k) { p.v[k] = p.v[0]; p.v[k].first_block = 0xffffffffu; }
    if (blocks == 0) return MV3D_OK;
    hipLaunchKernelGGL(roi_grad_tile_kernel, dim3(blocks), dim3(64 * nsl), 0, stream, p);
    return mv3d_launch_status();
}
