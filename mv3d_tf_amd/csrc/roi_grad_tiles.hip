// RoiPoolGrad for gfx950 as an ORDERED per-tile record stream, all views of a step in one launch, no workspace.
// Replaces lib/roi_pooling_layer/roi_pooling_op.cc:319-452 (CPU) / roi_pooling_op_gpu.cu.cc:113-215 (CUDA); same sums,
// bit for bit: per (pixel, channel) the f32 adds happen ROIs ascending, then ph, pw ascending (roi_pooling_op.cc:385-443).
//
// The reference is a gather: every input element scans the ROIs that contain its pixel and, for the candidate bins
// [phstart, phend) x [pwstart, pwend) of that pixel (:423-426), adds top_diff where argmax names the element.  A per-pixel
// gather re-fetches a (roi, bin) record (argmax + top_diff, 8 B per channel) once per candidate PIXEL -- 1.4 to 3.3 times
// on the training workload -- and needs a per-pixel index built by extra launches.  Here:
//   * a workgroup owns a TILE of th x tw pixels (<= 16) of one frame's bottom_diff, wave s of it the 64-channel slice s;
//     the accumulators of a wave are 16 VGPRs per lane (one per tile pixel).  A tile that no ROI reaches just writes zeros:
//     the zero fill of bottom_diff is the write-out of this kernel, not a launch of its own;
//   * geometry, ONCE per tile (all waves of the workgroup): the ROIs of the frame whose rounded rectangle meets the tile;
//     for each, per tile row the reference's [phstart, phend) and per tile column [pwstart, pwend) (the same f32 divides),
//     transposed into per-ph row masks and per-pw column masks.  A record (roi, ph, pw) is LIVE for the tile when both masks
//     are non-empty; its candidate pixels inside the tile are rowmask x colmask -- exactly the pixels whose reference loop
//     visits that bin.  Live records are listed in LDS in reference order: position = the ROI's offset (prefix sum over
//     live-bin counts) + the bin's rank among the ROI's live bins; no compaction pass;
//   * stream: every wave walks the list once: a record's slice is fetched by ONE 8-byte load per lane -- lanes 0..31 read
//     two channels of argmax, lanes 32..63 the same two channels of top_diff -- and one v_permlane32_swap leaves (argmax,
//     top_diff) of one channel in every lane; two groups of 16 records in flight.  For each candidate pixel of the record
//     (scalar loop over the mask bits) a compare, a select and an add into that pixel's VGPR, in list order;
//   * a record is fetched once per TILE it can reach (1.1 - 1.4 times with 4 x 4 tiles) in whole 2-KB pieces per array.
// Adding +0.0f for a non-matching channel is bit-neutral (a sum that started at +0.0f is never -0.0f).
// Tried first and measured: the same tiles as f32 accumulators in LDS updated with ds_add_f32 -- the LDS
// float atomic runs at ~0.5 lane / clock / CU on gfx950 (72 us of the launch), and decoding argmax to a pixel per lane
// costs more VALU than comparing against the few candidate pixels.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "roi_geom.h"

#define RT_MAXT 512                  // threads per workgroup, at most (8 slices)
#define RT_PASS 256                  // ROIs filtered per pass
#define RT_CAP 1024                  // live records listed per batch
#define RT_KB 64                     // overlapping ROIs evaluated per batch (one wave scans their counts)
#define RT_W 8                       // records per load group; two groups in flight
#define RT_NPIX 16                   // pixels per tile, at most

struct RoiTileViewDev {
    const float *top_diff, *rois;
    const int *argmax;
    float *bottom_diff;
    float scale;
    int B, R, H, W, C;
    int th, twl;                     // tile = th rows x (1 << twl) columns, th <= 4, twl <= 2, th << twl <= RT_NPIX
    int tiles_x, tiles_y;
    unsigned first_tile;
};
struct RoiTilePack {
    RoiTileViewDev v[MV3D_MAX_ROI_VIEWS];
    int n, PH, PW;
    int wpg, gpt;                    // waves (= slices) per workgroup, workgroups per tile: wpg * gpt = C / 64
    unsigned total_tiles, magic_phw, magic_pw;
    int dbg;                         // tuning builds: 1 = no stream, 2 = loads without adds, 4 = no geometry
    long long *trace;                // tuning builds: 8 cycle stamps per workgroup
};

__device__ __forceinline__ unsigned rt_div(unsigned t, unsigned magic, int d) { return d == 1 ? t : __umulhi(t, magic); }

typedef float rt_acc_t __attribute__((ext_vector_type(RT_NPIX)));
typedef unsigned rt_u2 __attribute__((ext_vector_type(2)));

struct RtStream {
    const char *lane_base;           // lanes 0..31: argmax + channel pair, lanes 32..63: top_diff + the same pair
    const int2 *list;                // {record byte offset, pixel mask}
    int n, chan, dbg;
    int twl, tmask, W, C, ty0, tx0;  // tile pixel q = (row q >> twl, column q & tmask)
};

// group g = records [g * RT_W, +RT_W) of the list: lane u < RT_W reads entry u, the record's byte offset goes to an SGPR
__device__ __forceinline__ void rt_issue(const RtStream &s, const int g, int2 &ent, rt_u2 (&d)[RT_W])
{
    const int lane = threadIdx.x & 63;
    ent = s.list[min(g * RT_W + (lane & (RT_W - 1)), s.n - 1)];
#pragma unroll
    for (int u = 0; u < RT_W; ++u) {
        const int off = __builtin_amdgcn_readlane(ent.x, u);
        d[u] = *reinterpret_cast<const rt_u2 *>(s.lane_base + (unsigned)off);
    }
}

__device__ __forceinline__ void rt_consume(const RtStream &s, const int g, const int2 &ent, const rt_u2 (&d)[RT_W], rt_acc_t &acc)
{
#pragma unroll
    for (int u = 0; u < RT_W; ++u) {
        if (g * RT_W + u >= s.n) break;                                // wave-uniform
        unsigned m = (unsigned)__builtin_amdgcn_readlane(ent.y, u);
        // lanes 32..63 of x (top_diff, even channel) <-> lanes 0..31 of y (argmax, odd channel): afterwards x = argmax and
        // y = top_diff of ONE channel in every lane (lane l < 32: channel 2 l, lane 32 + l: channel 2 l + 1)
        const auto sw = __builtin_amdgcn_permlane32_swap(d[u].x, d[u].y, false, false);
        const int rel = (int)sw[0] - s.chan;
        const float td = __builtin_bit_cast(float, sw[1]);
#ifdef MV3D_TUNING
        if (s.dbg & 2) { if (rel == 0x7ffffff0 && td == 1e30f) acc[0] += td; continue; }
#endif
        while (m) {                                                    // the record's candidate pixels (scalar loop)
            const int q = __builtin_ctz(m);
            m &= m - 1;
            const int wq = ((s.ty0 + (q >> s.twl)) * s.W + s.tx0 + (q & s.tmask)) * s.C;
            acc[q] += (rel == wq) ? td : 0.0f;                         // VGPR indexed by an SGPR
        }
    }
}

__global__ __launch_bounds__(RT_MAXT) void roi_grad_tile_kernel(RoiTilePack p)
{
    __shared__ int2 s_list[RT_CAP];                                   // {record byte offset, pixel mask}
    __shared__ int s_ov_roi[RT_PASS];
    __shared__ int4 s_ov_g[RT_PASS];                                  // (rsw, rsh, rew, reh) of the overlapping ROIs
    __shared__ unsigned short s_se[RT_KB][8];                         // per tile row (0..3) / column (4..7): start | end << 8
    __shared__ unsigned char s_pm[RT_KB][32];                         // per ph (0..15) row mask, per pw (16..31) column mask
    __shared__ int s_live[RT_KB], s_off[RT_KB];
    __shared__ int s_wcnt[RT_PASS / 64], s_misc[2];

    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MV3D_TUNING
    long long *tr = p.trace ? p.trace + 8 * (long long)blockIdx.x : nullptr;
    long long t_geo = 0, t_str = 0, t_mark;
    int n_listed = 0, n_batches = 0;
    if (tr && tid == 0) tr[0] = (long long)__builtin_readcyclecounter();
#define RT_MARK() (t_mark = (long long)__builtin_readcyclecounter())
#define RT_ACC(x) ((x) += (long long)__builtin_readcyclecounter() - t_mark)
#else
#define RT_MARK()
#define RT_ACC(x)
#endif
    const unsigned tile = blockIdx.x / (unsigned)p.gpt;
    const int slice = (int)(blockIdx.x % (unsigned)p.gpt) * p.wpg + wave;
    int k = 0;
#pragma unroll
    for (int j = 1; j < MV3D_MAX_ROI_VIEWS; ++j)
        if (j < p.n && tile >= p.v[j].first_tile) k = j;
    const RoiTileViewDev &v = p.v[k];
    const int PH = p.PH, PW = p.PW, PHW = PH * PW, H = v.H, W = v.W, C = v.C, R = v.R;
    const int th = v.th, twl = v.twl, tw = 1 << twl;
    const unsigned tv = tile - v.first_tile;
    const int txi = (int)(tv % (unsigned)v.tiles_x);
    const unsigned trest = tv / (unsigned)v.tiles_x;
    const int tyi = (int)(trest % (unsigned)v.tiles_y), n = (int)(trest / (unsigned)v.tiles_y);
    const int ty0 = tyi * th, tx0 = txi << twl;
    const int ty1 = min(ty0 + th, H) - 1, tx1 = min(tx0 + tw, W) - 1;

    RtStream s;
    {
        const int half = lane >> 5, l32 = lane & 31;
        s.lane_base = (half ? (const char *)v.top_diff : (const char *)v.argmax) + (size_t)(slice * 64 + l32 * 2) * 4;
        s.chan = slice * 64 + l32 * 2 + half;
        s.list = s_list; s.n = 0; s.dbg = p.dbg;
        s.twl = twl; s.tmask = tw - 1; s.W = W; s.C = C; s.ty0 = ty0; s.tx0 = tx0;
    }
    rt_acc_t acc = 0.0f;

    const int ps = min(nt, RT_PASS);                                   // ROIs filtered per pass
    const int npass = (R + ps - 1) / ps;
    for (int pass = 0; pass < npass; ++pass) {
        // ---- G0: the ROIs of frame n whose rounded rectangle meets the tile, ascending (roi_pooling_op.cc:392-403)
        const int r = pass * ps + tid;
        bool ok = false;
        RoiGeom g = {0, 0, 0, 0};
        if (tid < ps && r < R) {
            const float *roi = v.rois + 5 * (long long)r;
            g = roi_geom(roi, v.scale);
            ok = ((int)roi[0] == n) && g.reh >= ty0 && g.rsh <= ty1 && g.rew >= tx0 && g.rsw <= tx1;
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0 && wave < RT_PASS / 64) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int pos = __popcll(bal & ((1ull << lane) - 1ull)), K = 0;
#pragma unroll
        for (int t = 0; t < RT_PASS / 64; ++t) { const int cw = t * 64 < ps ? s_wcnt[t] : 0; if (t < wave) pos += cw; K += cw; }
        if (ok) { s_ov_roi[pos] = r; s_ov_g[pos] = make_int4(g.rsw, g.rsh, g.rew, g.reh); }
        __syncthreads();
#ifdef MV3D_TUNING
        if (p.dbg & 4) K = 0;
        if (tr && tid == 0 && pass == 0) tr[1] = (long long)__builtin_readcyclecounter();
#endif

        for (int k0 = 0; k0 < K;) {
            const int kb = min(K - k0, RT_KB);
            RT_MARK();
            // ---- G1a: (roi, tile row | tile column) -> the reference's [start, end) of pooled indices (:423-426)
            for (int t = tid; t < kb * 8; t += nt) {
                const int kk = t >> 3, i = t & 7, ii = i & 3;
                const bool isrow = i < 4;
                const int4 q = s_ov_g[k0 + kk];
                const int x = (isrow ? ty0 : tx0) + ii, lo = isrow ? q.y : q.x, hi = isrow ? q.w : q.z;
                const int P = isrow ? PH : PW;
                const bool valid = isrow ? (ii < th && x < H) : (ii < tw && x < W);
                unsigned se = 0;
                if (valid && x >= lo && x <= hi) {
                    const float b = (float)max(hi - lo + 1, 1) / (float)P;
                    int st = (int)floorf((float)(x - lo) / b), en = (int)ceilf((float)(x - lo + 1) / b);
                    st = min(max(st, 0), P); en = min(max(en, 0), P);
                    se = (unsigned)st | ((unsigned)en << 8);
                }
                s_se[kk][i] = (unsigned short)se;
            }
            __syncthreads();
            // ---- G1b: transposed: (roi, ph) -> mask of tile rows whose range holds ph; (roi, pw) -> mask of columns
            for (int t = tid; t < kb * 32; t += nt) {
                const int kk = t >> 5, pi = t & 31, pp = pi & 15;
                const int base = pi < 16 ? 0 : 4, P = pi < 16 ? PH : PW;
                unsigned m = 0;
                if (pp < P) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned se = s_se[kk][base + i];
                        if (pp >= (int)(se & 255u) && pp < (int)(se >> 8)) m |= 1u << i;
                    }
                }
                s_pm[kk][pi] = (unsigned char)m;
            }
            __syncthreads();
            // ---- G1c: live bins per ROI = (#ph with rows) x (#pw with columns); offsets; how many ROIs fit the list
            if (tid < 64) {
                unsigned rl = 0, cl = 0;
                if (tid < kb) {
                    for (int ph = 0; ph < PH; ++ph) if (s_pm[tid][ph]) rl |= 1u << ph;
                    for (int pw = 0; pw < PW; ++pw) if (s_pm[tid][16 + pw]) cl |= 1u << pw;
                }
                const int cnt = __popc(rl) * __popc(cl);
                int inc = cnt;
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { const int t = __shfl_up(inc, m); if (lane >= m) inc += t; }
                const bool fit = tid < kb && inc <= RT_CAP;
                const unsigned long long fb = __ballot(fit);
                const int kcut = __popcll(fb);                       // (inc is monotone: the fitting ROIs are a prefix)
                if (tid < kb) { s_live[tid] = (int)(rl | (cl << 16)); s_off[tid] = inc - cnt; }
                if (tid == kcut - 1) s_misc[1] = inc;
                if (tid == 0) s_misc[0] = kcut;
            }
            __syncthreads();
            const int kcut = s_misc[0], nrec = s_misc[1];
            // ---- G2: live records in reference order: position = ROI offset + rank among the ROI's live bins
            for (unsigned t = tid; t < (unsigned)(kcut * PHW); t += nt) {
                const unsigned kk = rt_div(t, p.magic_phw, PHW), bin = t - kk * (unsigned)PHW;
                const unsigned ph = rt_div(bin, p.magic_pw, PW), pw = bin - ph * (unsigned)PW;
                const unsigned rm = s_pm[kk][ph], cm = s_pm[kk][16 + pw];
                if (rm && cm) {
                    const unsigned live = (unsigned)s_live[kk], rl = live & 0xffffu, cl = live >> 16;
                    const int rank = __popc(rl & ((1u << ph) - 1u)) * __popc(cl) + __popc(cl & ((1u << pw) - 1u));
                    unsigned pm = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (rm & (1u << i)) pm |= cm << (i << twl);
                    s_list[s_off[kk] + rank] = make_int2((s_ov_roi[k0 + kk] * PHW + (int)bin) * C * 4, (int)pm);
                }
            }
            __syncthreads();
            RT_ACC(t_geo);
            RT_MARK();
#ifdef MV3D_TUNING
            n_listed += nrec; ++n_batches;
            if (p.dbg & 1) { k0 += kcut; if (k0 < K) __syncthreads(); continue; }
#endif
            // ---- stream the listed records in order, two groups of RT_W in flight
            if (nrec > 0) {
                s.n = nrec;
                const int G = (nrec + RT_W - 1) / RT_W;
                rt_u2 dA[RT_W], dB[RT_W];
                int2 eA, eB;
                rt_issue(s, 0, eA, dA);
                int gi = 0;
                while (gi + 2 <= G - 1) {
                    rt_issue(s, gi + 1, eB, dB);
                    rt_consume(s, gi, eA, dA, acc);
                    rt_issue(s, gi + 2, eA, dA);
                    rt_consume(s, gi + 1, eB, dB, acc);
                    gi += 2;
                }
                if (gi + 1 <= G - 1) {
                    rt_issue(s, gi + 1, eB, dB);
                    rt_consume(s, gi, eA, dA, acc);
                    rt_consume(s, gi + 1, eB, dB, acc);
                } else {
                    rt_consume(s, gi, eA, dA, acc);
                }
            }
            RT_ACC(t_str);
            k0 += kcut;
            if (k0 < K) __syncthreads();                               // the list and the masks are rebuilt
        }
        if (pass + 1 < npass) __syncthreads();
    }
#ifdef MV3D_TUNING
    if (tr && tid == 0) { tr[2] = (long long)__builtin_readcyclecounter(); tr[3] = t_geo; tr[4] = t_str; tr[5] = n_listed; tr[6] = n_batches; }
#endif
    // write-out: every pixel of the tile, this wave's 256 B per pixel (streaming stores)
    float *out = v.bottom_diff + (long long)n * H * W * C + s.chan;
#pragma unroll
    for (int q = 0; q < RT_NPIX; ++q) {
        const int h = ty0 + (q >> twl), w = tx0 + (q & (tw - 1));
        if ((q >> twl) < th && h < H && w < W) __builtin_nontemporal_store(acc[q], out + ((long long)h * W + w) * C);
    }
#ifdef MV3D_TUNING
    if (tr && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[7] = (long long)__builtin_readcyclecounter(); }
#endif
}

static unsigned rt_magic(int d) { return d <= 1 ? 0u : (unsigned)(0xFFFFFFFFull / (unsigned)d) + 1u; }

// Tile shape of a view: aim at ~`target` live records per tile on average (records per pixel = R PH PW / pixels, if every ROI
// lay on the map); dense maps (the 8 x 64 front view under 128 ROIs per frame) get small tiles so that no tile's list is
// long, sparse maps get large ones (fewer workgroups, a record meets fewer tiles).
static void rt_pick_tile(const mv3d_roi_grad_view &w, int PH, int PW, int &th, int &twl)
{
    static const int shapes[][2] = {{1, 0}, {1, 1}, {2, 1}, {2, 2}, {4, 2}};
    const double dens = (double)w.num_rois * PH * PW / ((double)w.batch_size * w.height * w.width);
    const double target = 16.0;
    int best = 0;
    for (int i = 0; i < 5; ++i) {
        const int a = shapes[i][0] << shapes[i][1];
        if (a * dens <= target * 1.25 || i == 0) best = i;
    }
    th = shapes[best][0]; twl = shapes[best][1];
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
        if ((long long)w.batch_size * w.height * w.width >= 0x7fffffffLL) return false;
    }
    return true;
}

int mv3d_launch_roi_grad_tiles(int num_views, const mv3d_roi_grad_view *views, int PH, int PW, const int *tile_override,
                               hipStream_t stream)
{
    RoiTilePack p;
    p.n = num_views; p.PH = PH; p.PW = PW;
    const int nsl = views[0].channels / 64;
    p.wpg = nsl; p.gpt = 1;
    p.magic_phw = rt_magic(PH * PW); p.magic_pw = rt_magic(PW);
    p.dbg = 0; p.trace = nullptr;
#ifdef MV3D_TUNING
    if (getenv("MV3D_RT_DBG")) p.dbg = atoi(getenv("MV3D_RT_DBG"));
    if (getenv("MV3D_RT_TRACE")) p.trace = (long long *)strtoull(getenv("MV3D_RT_TRACE"), nullptr, 10);
    if (getenv("MV3D_RT_WPG")) { p.wpg = atoi(getenv("MV3D_RT_WPG")); p.gpt = nsl / p.wpg; }
#endif
    // the densest view first: its workgroups (the longest lists) are dispatched first
    int ord[MV3D_MAX_ROI_VIEWS] = {0, 1, 2, 3};
    for (int a = 0; a < num_views; ++a)
        for (int b = a + 1; b < num_views; ++b) {
            const mv3d_roi_grad_view &x = views[ord[a]], &y = views[ord[b]];
            const double dx = (double)x.num_rois / ((double)x.batch_size * x.height * x.width);
            const double dy = (double)y.num_rois / ((double)y.batch_size * y.height * y.width);
            if (dy > dx) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
        }
    unsigned tiles = 0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_roi_grad_view &w = views[ord[k]];
        RoiTileViewDev &v = p.v[k];
        v.top_diff = w.top_diff; v.rois = w.bottom_rois; v.argmax = w.argmax_data; v.bottom_diff = w.bottom_diff;
        v.scale = w.spatial_scale;
        v.B = w.batch_size; v.R = w.num_rois; v.H = w.height; v.W = w.width; v.C = w.channels;
        rt_pick_tile(w, PH, PW, v.th, v.twl);
        if (tile_override && tile_override[2 * ord[k]] > 0) { v.th = tile_override[2 * ord[k]]; v.twl = tile_override[2 * ord[k] + 1]; }
        if (v.th > 4 || v.twl > 2) return MV3D_ERR_INVALID_ARG;
        v.tiles_x = (w.width + (1 << v.twl) - 1) >> v.twl;
        v.tiles_y = (w.height + v.th - 1) / v.th;
        v.first_tile = tiles;
        const long long t = (long long)w.batch_size * v.tiles_x * v.tiles_y;
        if ((t + tiles) * p.gpt > 0x7fffffffLL) return MV3D_ERR_INVALID_ARG;
        tiles += (unsigned)t;
    }
    for (int k = num_views; k < MV3D_MAX_ROI_VIEWS; ++k) p.v[k] = p.v[0];
    p.total_tiles = tiles;
    if (tiles == 0) return MV3D_OK;
    hipLaunchKernelGGL(roi_grad_tile_kernel, dim3(tiles * (unsigned)p.gpt), dim3(64 * p.wpg), 0, stream, p);
    return mv3d_launch_status();
}
