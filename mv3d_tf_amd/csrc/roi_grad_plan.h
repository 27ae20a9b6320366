// The RoiPool pair's tile layout and work list, shared by the pair's two launches: the FORWARD launch (roi_pool.hip) carries one planning
// workgroup that writes the work list of the BACKWARD launch (roi_grad_tiles.hip) into bytes of the pair's private argmax buffer that
// neither launch otherwise touches (the last quarter of view 0's buffer: one-byte codes in the first quarter, 16-bit escape codes in the
// two behind it), so the tiles under long entry streams are cut into sub-tiles -- first in the list -- without a launch of their own.
#pragma once
#include "common.h"
#include "kernels.h"
#include "roi_geom.h"

#define RGT_PLAN_TILES 3072          // tiles of all views the planner's LDS heat map holds (12 KB; more tiles: the static grid)
#define RGT_HOT_ENTRIES 250          // estimated entries from which a tile is cut
#define RGT_HOT_MAX 128              // at most this many tiles are cut (the grid is sized for 3 extra units each)
#define RGT_UNIT_EMPTY (1 << 25)     // work-list unit: no ROI touches the tile
#define RGT_PLAN_CLASSES 5
#define RGT_PLAN_SCAN(T) (RGT_PLAN_CLASSES * ((T) / 64) + MV3D_MAX_ROI_VIEWS)
#define RGT_PLAN_THREADS 256         // (the forward's workgroup size)
#ifndef RGT_PLAN_DEFAULT
#define RGT_PLAN_DEFAULT 1           // (A / B builds: 0 = the static tile grid)
#endif

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
    // planned mode: the launch's units come from the work list the pair's FORWARD launch wrote (rgt_plan_block) -- the tiles under long
    // entry streams cut into four sub-tiles, first in the list; NULL = every view's tile grid as it lies
    const int4 *work;                // unit: x = view | frame << 4 | ths << 16 | tws << 20 | skip << 24, y = first row, z = first column
    const int *n_work;               // units in the list
    int dbg;                         // experiment builds (MV3D_TUNING): phases switched off, 0 otherwise
    long long *trace;                // experiment builds: 8 wall-clock stamps per wave (tools/roi_tiles_trace.py), NULL otherwise
};

// The launch layout of the views (launch order, tile shapes, grid offsets) from the SHAPES alone, and whether the pair runs planned: both
// launches take the same decision independently.  Returns the static grid's workgroups in *blocks (0: too many), planned mode in *planned
// with p.work / p.n_work set and the planned grid in *plan_blocks.
bool mv3d_rgt_layout(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, RgtPack &p, unsigned *blocks, bool *planned,
                     unsigned *plan_blocks, int *hot_entries, int *hot_max);

// ---- planned mode: which tiles sit under long entry streams?  ONE workgroup estimates every tile's stream from the ROIs' rounded
// geometry (thread = ROI: rows covered x bins per row x bins across, added into an LDS heat map), cuts the tiles above RGT_HOT_ENTRIES into
// four sub-tiles (rows first: a 4 x 4 tile becomes four 1 x 4 rows, a 2 x 2 tile four pixels -- the sub-tiles are ordinary units, every
// one filtered, expanded and drained by a wave of its own, no communication) and writes the launch's work list LONGEST ESTIMATE FIRST: the
// sub-tiles, the hot tiles beyond the cap, then three classes of shorter streams, the tiles no ROI touches last and flagged (their waves
// write zeros without a ROI filter).  What it buys is the makespan: one wave's instruction stream is the limit of a stream (~100 - 140 ns per
// entry, profiles/r06_a), a 4 x 4 tile under 465 entries is 65 us however fast the other 17 k waves finish, and a grid dispatched in order
// should end with its shortest units (profiles/r06_p: 66 -> 44 us).  T threads; heat: RGT_PLAN_TILES ints of LDS, scan: RGT_PLAN_SCAN(T) ints.
template <int T>
__device__ __forceinline__ void rgt_plan_block(const RgtPack &p, int4 *work, int *n_work, const int hot_entries, const int hot_max, int *heat,
                                               int *scan)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ntiles = 0;
    for (int k = 0; k < p.n; ++k) ntiles += p.v[k].B * p.v[k].tiles_y * p.v[k].tiles_x;
    for (int i = tid; i < ntiles; i += T) heat[i] = 0;
    if (tid < MV3D_MAX_ROI_VIEWS) scan[RGT_PLAN_CLASSES * (T / 64) + tid] = 0;  // a view with a ROI the estimate does not follow: none of its tiles counts as empty
    __syncthreads();
    int tile0 = 0;
    for (int k = 0; k < p.n; ++k) {
        const RgtView &v = p.v[k];
        const int TH = 1 << v.ths, TW = 1 << v.tws;
        for (int r = tid; r < v.R; r += T) {
            float r5[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) r5[u] = v.rois[5 * (long long)r + u];
            const RoiGeom g = roi_geom(r5, v.scale);
            const int b = (int)r5[0];
            if (b < 0 || b >= v.B || g.reh < g.rsh || g.rew < g.rsw) continue;          // (no pixel passes roi_pooling_op.cc:401-404: no entry anywhere)
            const int y0 = max(g.rsh, 0), y1 = min(g.reh, v.H - 1), x0 = max(g.rsw, 0), x1 = min(g.rew, v.W - 1);
            if (y1 < y0 || x1 < x0) continue;
            const int ty0 = y0 >> v.ths, ty1 = y1 >> v.ths, tx0 = x0 >> v.tws, tx1 = x1 >> v.tws;
            // (a ROI over that many tiles leaves a few bins in each; coordinates outside exact integer range are left to the kernel's own tests)
            const bool sane = abs(g.rsw) < (1 << 24) && abs(g.rsh) < (1 << 24) && abs(g.rew) < (1 << 24) && abs(g.reh) < (1 << 24);
            if (!sane || (long long)(ty1 - ty0 + 1) * (tx1 - tx0 + 1) > 128) { scan[RGT_PLAN_CLASSES * (T / 64) + k] = 1; continue; }
            // (an estimate: bins per map row / column from the reciprocal bin sizes, no divide inside the loops -- this workgroup's
            // serial work must stay well inside the pooling workgroups' shadow)
            const float inv_bh = (float)p.PH / (float)(g.reh - g.rsh + 1), inv_bw = (float)p.PW / (float)(g.rew - g.rsw + 1);
            const float per_row = fminf((float)p.PH, 1.0f + inv_bh);               // bins whose rows hold one map row
            int *const hrow = heat + tile0 + b * v.tiles_y * v.tiles_x;
            for (int ty = ty0; ty <= ty1; ++ty) {
                const int rows = min(y1, (ty << v.ths) + TH - 1) - max(y0, ty << v.ths) + 1;
                const float down = (float)rows * per_row;
                int *const hr = hrow + ty * v.tiles_x;
                // the first and the last tile of the row may be covered in part, the ones between in full: one value for all of those
                const int cols0 = min(x1, (tx0 << v.tws) + TW - 1) - x0 + 1;
                atomicAdd(&hr[tx0], max((int)(down * fminf((float)p.PW, 1.0f + (float)cols0 * inv_bw) + 0.5f), 1));
                if (tx1 > tx0) {
                    const int cols1 = x1 - (tx1 << v.tws) + 1;
                    atomicAdd(&hr[tx1], max((int)(down * fminf((float)p.PW, 1.0f + (float)cols1 * inv_bw) + 0.5f), 1));
                    const int mid = max((int)(down * fminf((float)p.PW, 1.0f + (float)TW * inv_bw) + 0.5f), 1);
                    for (int tx = tx0 + 1; tx < tx1; ++tx) atomicAdd(&hr[tx], mid);
                }
            }
        }
        tile0 += v.B * v.tiles_y * v.tiles_x;
    }
    __syncthreads();
    // ---- classify by estimate -- 0: hot (cut while the cap lasts), 1 .. 3: long, medium, short streams, 4: no ROI near -- and order the list
    // longest first (a wave slot freed late should pick up short work: the launch ends with its last wave): positions by a block scan of the
    // class counts over the tiles in grid order (thread t: tiles [t * per, t * per + per)), grid order kept inside a class
    constexpr int NW = T / 64;
    int *const flags = scan + RGT_PLAN_CLASSES * NW;
    const int per = (ntiles + T - 1) / T;
    const int first = tid * per, last = min(first + per, ntiles);
    const int lim1 = max(hot_entries / 2, 2), lim2 = max(hot_entries / 8, 1);
    int cnt[RGT_PLAN_CLASSES], inc[RGT_PLAN_CLASSES], pos[RGT_PLAN_CLASSES], tot[RGT_PLAN_CLASSES];
#pragma unroll
    for (int c = 0; c < RGT_PLAN_CLASSES; ++c) cnt[c] = 0;
    for (int i = first; i < last; ++i) {
        const int e = heat[i], c = e >= hot_entries ? 0 : (e >= lim1 ? 1 : (e >= lim2 ? 2 : (e > 0 ? 3 : 4)));
#pragma unroll
        for (int u = 0; u < RGT_PLAN_CLASSES; ++u) cnt[u] += c == u;
    }
#pragma unroll
    for (int c = 0; c < RGT_PLAN_CLASSES; ++c) {                       // inclusive scans inside the wave ...
        int x = cnt[c];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(x, o);
            if (lane >= o) x += a;
        }
        inc[c] = x;
        if (lane == 63) scan[c * NW + wave] = x;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < RGT_PLAN_CLASSES; ++c) {                       // ... and over the waves: exclusive position inside the class, class total
        int before = 0, all = 0;
        for (int w = 0; w < NW; ++w) { if (w < wave) before += scan[c * NW + w]; all += scan[c * NW + w]; }
        pos[c] = before + inc[c] - cnt[c];
        tot[c] = all;
    }
    const int cut = min(tot[0], hot_max);                              // hot tiles beyond the cap stay whole (first among the whole ones)
    int off[RGT_PLAN_CLASSES];                                         // the list: 4 x cut sub-tile units, then the classes in turn
    off[0] = 4 * cut - cut;                                            // (a whole hot tile at hot position q >= cut sits at 4 cut + q - cut)
    off[1] = 4 * cut + (tot[0] - cut);
#pragma unroll
    for (int c = 2; c < RGT_PLAN_CLASSES; ++c) off[c] = off[c - 1] + tot[c - 1];
    // the view, frame and tile coordinates of this thread's first tile (grid position `first`), advanced tile by tile
    int k = 0, tx = 0, ty = 0, b = 0;
    {
        int base = 0;
        for (int j = 0; j < p.n; ++j) {
            const int n = p.v[j].B * p.v[j].tiles_y * p.v[j].tiles_x;
            if (first < base + n || j == p.n - 1) { k = j; break; }
            base += n;
        }
        int t = max(first - base, 0);
        tx = t % p.v[k].tiles_x; t /= p.v[k].tiles_x;
        ty = t % p.v[k].tiles_y; b = t / p.v[k].tiles_y;
    }
    for (int i = first; i < last; ++i) {
        const RgtView &v = p.v[k];
        const int e = heat[i], c = e >= hot_entries ? 0 : (e >= lim1 ? 1 : (e >= lim2 ? 2 : (e > 0 ? 3 : 4)));
        int q = 0;
#pragma unroll
        for (int u = 0; u < RGT_PLAN_CLASSES; ++u) { if (c == u) { q = pos[u]; pos[u] += 1; } }
        if (c == 0 && q < cut) {
            const int sy = min(v.ths, 2), sx = min(v.tws, 2 - sy);     // split bits: rows first
            const int ths2 = v.ths - sy, tws2 = v.tws - sx;
            for (int j = 0; j < 4; ++j) {
                const int jy = j >> sx, jx = j & ((1 << sx) - 1);
                const int y = (ty << v.ths) + (jy << ths2), x = (tx << v.tws) + (jx << tws2);
                const bool skip = j >= (1 << (sy + sx)) || y >= v.H || x >= v.W;
                work[4 * q + j] = make_int4(k | (b << 4) | (ths2 << 16) | (tws2 << 20) | (skip ? 1 << 24 : 0), y, x, e);
            }
        } else {
            // a tile no ROI's rounded rectangle touches has no entry (every entry lies inside its ROI's rounded rectangle cut to the map:
            // the expansion's ih0 .. ih1 x iw0 .. iw1): its unit is flagged EMPTY and the wave only writes the zeros
            const bool empty = e == 0 && flags[k] == 0;
            int o = 0;
#pragma unroll
            for (int u = 0; u < RGT_PLAN_CLASSES; ++u) { if (c == u) o = off[u]; }
            work[o + q] = make_int4(k | (b << 4) | (v.ths << 16) | (v.tws << 20) | (empty ? RGT_UNIT_EMPTY : 0), ty << v.ths, tx << v.tws, e);
        }
        if (++tx == v.tiles_x) { tx = 0; if (++ty == v.tiles_y) { ty = 0; if (++b == v.B) { b = 0; ++k; } } }
    }
    if (tid == 0) *n_work = 3 * cut + ntiles;
}


