// Weight gradient of the 3x3 / stride 1 / SAME convolution for the training trunk (mv3d_tf_amd/trunk_train.py), on the gfx950
// matrix cores: dW[co][tap][ci] = sum over pixels q of dY[q][co] * X[q + shift(tap)][ci], bf16 operands, f32 accumulation.
// What it stands in for: the filter gradient TensorFlow computes for Network.conv (lib/networks/network.py:109-133) when
// lib/fast_rcnn/train_mv.py:138-219 minimises the loss -- here in the mixed-precision trunk, never the fp32 parity contract.
//
// Both operands are the FRAMED NHWC maps of conv3x3_mfma.hip: q runs over the framed pixel index of the whole batch, the zero
// frame of dY makes frame pixels contribute nothing, and a tap is the row shift (dy - 1) * (W + 2) + (dx - 1) of X.
// The contraction index is the PIXEL, which NHWC stores strided; the MFMA wants 8 consecutive k per lane.  So both operand
// tiles are staged as they lie in memory ([pixel][channel] rows, buffer_load ... lds) and read with ds_read_b64_tr_b16: 16 lanes
// fetch a 4-pixel x 16-channel block and each receives ONE channel's 4 pixels (tools/tr_read_probe.hip shows the mapping).
// For X the tap's dx is just a row offset of the read (+ dx rows), so one staged tile of 64 + 2 rows serves the three taps of a
// filter row.
//
// Workgroup = (co tile of 128 | 64) x (64 input channels) x (filter row dy: 3 taps = 192 GEMM columns) x (a range of K steps of
// 64 pixels); 4 waves = 2 (co halves) x 2 (column halves: 3 fragments of 32 columns each).  Partial sums of the K ranges go to
// part[split][tap][ci][co] (f32: a lane's four accumulator rows are four consecutive co = one 16-byte store; with one 4-byte
// store per value the 98 KB of a workgroup's partial sums cost about as long as its K loop at a training batch) and are folded in a
// fixed order by conv3x3_wgrad_reduce_kernel: deterministic, no atomics.
#include <stdlib.h>
#include "common.h"

namespace mv3d_wgrad {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

struct WgradArgs {
    const void *x, *dy;
    float *part, *bpart;          // split-K partial sums: filter [splits][9][Cin][Cout]; bias [splits][Cout] (or nullptr)
    int Wp, Cin, Cout, Q, steps, steps_per_split, splits, ci_tiles;
    unsigned x_bytes, dy_bytes;
};

// Several views (the three trunks at one VGG depth) behind one launch, like ConvGroup of conv3x3_mfma.hip: view k owns the
// workgroups [first[k], first[k + 1]); every view has its own maps, partial sums and number of K splits, the filter shape is shared.
#define WG_MAX_VIEWS 3
struct WgradView {
    const void *x, *dy;
    float *part, *bpart;          // split-K partial sums: filter [splits][9][Cin][Cout]; bias [splits][Cout] (or nullptr)
    float *dw, *db;               // results (the reduce launch)
    int Wp, steps, splits, c_in_real;
    unsigned x_bytes, dy_bytes;
};
// first[k] = view k's first GLOBAL split index (the views' K splits are numbered through), tiles = (co tiles) x (ci tiles) x 3,
// gsplits = all views' splits.  Workgroup -> (global split, tile): the global split is what the XCD index (blockIdx & 7) walks, so
// that ALL tiles of one K range run on ONE XCD: they read the same dY / X pixel windows (a dY window is shared by the 3 x ci-tiles
// workgroups of its co tile, an X window by the co tiles), which then sit in that XCD's L2 once instead of being pulled over the
// fabric into eight L2s by workgroups that the round-robin placement had spread over all XCDs.
struct WgradGroup { WgradView v[WG_MAX_VIEWS]; int n, Cin, Cout, steps_per_split, ci_tiles, c_in_real, tiles, gsplits; int first[WG_MAX_VIEWS]; };
__device__ __forceinline__ WgradArgs wgrad_view_args(const WgradGroup &g, int &id, int &split)
{
    const int row = 8 * g.tiles, b = (int)blockIdx.x;
    int gs = (b / row) * 8 + (b & 7);                             // global split of this workgroup; its XCD = gs % 8
    id = (b % row) >> 3;                                          // tile
    // (3 tiles -- the 64 -> 64 layers: one co tile, one ci tile, the three filter rows -- measured 8 - 15 % SLOWER that way; they keep
    // the plain order, tile fastest)
    if (g.tiles <= 3) { gs = b / g.tiles; id = b % g.tiles; }
    int k = 0;
#pragma unroll
    for (int j = 1; j < WG_MAX_VIEWS; ++j)
        if (j < g.n && gs >= g.first[j]) k = j;
    split = gs < g.gsplits ? gs - g.first[k] : -1;                // (the grid is padded to a multiple of 8 splits: -1 = nothing to do)
    const WgradView &v = g.v[k];
    WgradArgs a;
    a.x = v.x; a.dy = v.dy; a.part = v.part; a.bpart = v.bpart;
    a.Wp = v.Wp; a.Cin = g.Cin; a.Cout = g.Cout; a.Q = 0; a.steps = v.steps; a.steps_per_split = g.steps_per_split; a.splits = v.splits;
    a.ci_tiles = g.ci_tiles; a.x_bytes = v.x_bytes; a.dy_bytes = v.dy_bytes;
    return a;
}

#define WG_PIX 64              // pixels per K step
#define WG_XROWS 72            // staged X rows per step: 64 + 2 (dx = 0..2), padded to 9 pieces of 8 rows
#define WG_OOB 0x7ffffff0      // a byte offset beyond every buffer: the load returns zeros

template <int BMC>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const WgradGroup grp)
{
#if __HIP_DEVICE_COMPILE__
    int id, split;
    const WgradArgs a = wgrad_view_args(grp, id, split);
    if (split < 0) return;
    constexpr int RB = BMC * 2;                                   // bytes of a dY tile row
    constexpr int LPR = RB / 16, RPP = 64 / LPR, DYP = WG_PIX / RPP;   // lanes per row, rows per 1-KB piece, dY pieces
    constexpr int DY_BYTES = WG_PIX * RB, X_BYTES = WG_XROWS * 128, STAGE = DY_BYTES + X_BYTES;
    constexpr int NP = DYP + 9;                                   // DMA pieces per step
    constexpr int FA = BMC / 64;                                  // 32-co fragments per wave
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wb = wave >> 1;
    // workgroup -> (split, dy, ci tile, co tile)
    const int ci_t = id % a.ci_tiles; id /= a.ci_tiles;
    const int dyr = id % 3;
    const int co_t = id / 3;
    const int co0 = co_t * BMC, ci0 = ci_t * 64;
    const int s0 = split * a.steps_per_split, s1 = min(a.steps, s0 + a.steps_per_split);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, (int)a.dy_bytes, 0x00020000);
    // per-lane parts of the DMA source offsets (everything in the vector offset: negative / past-the-end rows must fail the
    // buffer's range check and read as zeros)
    // LDS image: rows of RB bytes whose 64-byte granules are XOR-swizzled by the row (applied HERE, on the source column: the DMA
    // image is lane-linear), so that the 4 rows x 64 B a transposing read touches per pass fall into the four 64-byte bank groups:
    // 256-byte rows: granule ^= row & 3; 128-byte rows: granule ^= (row >> 1) & 1.  Unswizzled the reads measured 64 % conflict cycles.
    const int dy_col = RB == 256 ? ((((lane & 15) >> 2) ^ (lane >> 4)) << 6) | ((lane & 3) << 4)
                                 : ((((lane & 7) >> 2) ^ ((lane >> 4) & 1)) << 6) | ((lane & 3) << 4);
    const int x_col = ((((lane & 7) >> 2) ^ ((lane >> 4) & 1)) << 6) | ((lane & 3) << 4);
    const int dyv = (lane / LPR) * a.Cout * 2 + dy_col + co0 * 2;
    const int xv = (lane >> 3) * a.Cin * 2 + x_col + ci0 * 2;
    const int xshift = (dyr - 1) * a.Wp - 1;
    auto issue = [&](const int step, const int st) __attribute__((always_inline)) {
        const int q0 = step * WG_PIX;
        char *const base = lds + st * STAGE;
#pragma unroll
        for (int p0 = 0; p0 < (NP + 3) / 4; ++p0) {
            const int p = p0 * 4 + wave;
            if (p < DYP) {
                const long long off = (long long)(q0 + p * RPP) * a.Cout * 2 + dyv;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void_t *)(base + p * 1024), 16, off < (long long)a.dy_bytes ? (int)off : WG_OOB, 0, 0, 0);
            } else if (p < NP) {
                const long long off = (long long)(q0 + xshift + (p - DYP) * 8) * a.Cin * 2 + xv;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t *)(base + DY_BYTES + (p - DYP) * 1024), 16,
                                                         (off >= 0 && off < (long long)a.x_bytes) ? (int)off : WG_OOB, 0, 0, 0);
            }
        }
    };

    // transposing reads: lane (t = lane & 15, G = (lane >> 4) & 1, kg = lane >> 5) supplies row 8 kg + 4 h + (t >> 2), channels
    // 16 G + 4 (t & 3) .. + 3 of the block and receives channel 16 G + t, pixels 8 kg + 4 h .. + 3: an MFMA operand (row / column
    // lane & 31, k = 8 (lane >> 5) + 0..7) is two such reads (h = 0, 1)
    const int t = lane & 15, G = (lane >> 4) & 1, kg = lane >> 5;
    auto phys = [](const int row, const int b, const int rb) __attribute__((always_inline)) {      // swizzled byte offset of (row, byte column b)
        const int f = rb == 256 ? (row & 3) : ((row >> 1) & 1);
        return row * rb + ((((b >> 6) ^ f) << 6) | (b & 63));
    };
    int aoff[FA][2], boff[3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * kg + 4 * h + (t >> 2), ch = 16 * G + 4 * (t & 3);
#pragma unroll
        for (int i = 0; i < FA; ++i) aoff[i][h] = phys(row, (wa * (BMC / 2) + 32 * i + ch) * 2, RB);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n0 = 96 * wb + 32 * j;                      // GEMM column of the fragment: tap dx = n0 / 64, channel n0 % 64
            boff[j][h] = DY_BYTES + phys(row + n0 / 64, (n0 % 64 + ch) * 2, 128);
        }
    }

    f32x16 acc[FA][3];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // bias gradient (sum over the pixels of dY): the waves of input-channel tile 0 / filter row 0 / column half 0 add up their dY
    // operand fragments on the vector ALU -- a lane holds 8 pixels of ONE output channel (row lane & 31, k-group lane >> 5).
    // (Round 3 multiplied the fragments by an all-ones operand as well: 33 % more matrix work for those workgroups -- the launch
    // is one round of workgroups, so its time was THEIR time -- and 32 more accumulator registers for everybody: 2 instead of 3
    // workgroups per CU.)
    const bool do_bias = a.bpart != nullptr && ci_t == 0 && dyr == 0 && wb == 0;       // (wave-uniform)
    float bsum[FA];
#pragma unroll
    for (int i = 0; i < FA; ++i) bsum[i] = 0.f;

    if (s0 < s1) issue(s0, 0);
    for (int s = s0; s < s1; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 1 < s1) issue(s + 1, (s + 1 - s0) & 1);
        const char *const st = lds + ((s - s0) & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[FA], fb[3];
#pragma unroll
            for (int i = 0; i < FA; ++i) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(st + ks * 16 * RB + aoff[i][0]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(st + ks * 16 * RB + aoff[i][1]));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fa[i] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(st + ks * 16 * 128 + boff[j][0]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(st + ks * 16 * 128 + boff[j][1]));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fb[j] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < FA; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[i] += (float)fa[i][e];
            }
        }
    }

    if (do_bias) {                                                // the two k-groups of a channel sit in lanes l and l + 32
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32);
            if (lane < 32) a.bpart[(size_t)split * a.Cout + co0 + wa * (BMC / 2) + 32 * i + lane] = tot;
        }
    }
    // partial sums: D layout = lane holds column (GEMM column n) lane & 31, rows (co) 8 q + 4 (lane >> 5) + {0..3}, q = reg >> 2
    float *const out = a.part + (size_t)split * a.Cout * 9 * a.Cin;
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n0 = 96 * wb + 32 * j, tap = 3 * dyr + n0 / 64, ci = ci0 + n0 % 64 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) {                         // a register quad = 4 consecutive output channels: one 16-byte store
                const int co = co0 + wa * (BMC / 2) + 32 * i + 8 * q + 4 * (lane >> 5);
                const f32x4p v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                *(f32x4p *)(out + ((size_t)tap * a.Cin + ci) * a.Cout + co) = v;
            }
        }
#endif
}

// The same weight gradient on f32 maps (the fp32 training trunk): v_mfma_f32_32x32x2_f32 takes ONE f32 per lane and operand
// (row / column lane & 31, k = lane >> 5), so a lane simply reads its (pixel, channel) element of the [pixel][channel] tile with a
// ds_read_b32 -- no transposing read is needed (nor exists for 32-bit elements).  K steps of 32 pixels; lanes 0..31 read one pixel
// row's 32 consecutive channels (128 B), lanes 32..63 the next pixel's: the 128-byte granules of odd rows are XOR-swizzled so that the
// two rows fall on different bank halves.  MFMA-bound by a wide margin (64 matrix cycles per pair of 4-byte LDS reads).
#define WGF_PIX 32
#define WGF_XROWS 40
template <int BMC>
__global__ __launch_bounds__(256) void conv3x3_wgrad_f32_kernel(const WgradGroup grp)
{
#if __HIP_DEVICE_COMPILE__
    int id, split;
    const WgradArgs a = wgrad_view_args(grp, id, split);
    if (split < 0) return;
    constexpr int RB = BMC * 4;                                   // bytes of a dY tile row (512 | 256)
    constexpr int LPR = RB / 16, RPP = 64 / LPR, DYP = WGF_PIX / RPP;
    constexpr int XRB = 256, XRPP = 4, XPCS = WGF_XROWS / XRPP;   // activation rows: 64 channels x 4 B, 4 rows per 1-KB piece
    constexpr int DY_BYTES = WGF_PIX * RB, X_BYTES = WGF_XROWS * XRB, STAGE = DY_BYTES + X_BYTES;
    constexpr int NP = DYP + XPCS;
    constexpr int FA = BMC / 64;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wb = wave >> 1;
    const int ci_t = id % a.ci_tiles; id /= a.ci_tiles;
    const int dyr = id % 3;
    const int co_t = id / 3;
    const int co0 = co_t * BMC, ci0 = ci_t * 64;
    const int s0 = split * a.steps_per_split, s1 = min(a.steps, s0 + a.steps_per_split);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, (int)a.dy_bytes, 0x00020000);
    // DMA lane -> (row of the piece, 16-byte column); the source column's 128-byte granule is XORed with the row's parity
    const int dy_row = lane / LPR, dy_c = (lane % LPR) * 16;
    const int dy_col = (((dy_c >> 7) ^ (dy_row & 1)) << 7) | (dy_c & 127);          // (a piece starts on an even row)
    const int x_row = lane >> 4, x_c = (lane & 15) * 16;
    const int x_col = (((x_c >> 7) ^ (x_row & 1)) << 7) | (x_c & 127);
    const int dyv = dy_row * a.Cout * 4 + dy_col + co0 * 4;
    const int xv = x_row * a.Cin * 4 + x_col + ci0 * 4;
    const int xshift = (dyr - 1) * a.Wp - 1;
    auto issue = [&](const int step, const int st) __attribute__((always_inline)) {
        const int q0 = step * WGF_PIX;
        char *const base = lds + st * STAGE;
#pragma unroll
        for (int p0 = 0; p0 < (NP + 3) / 4; ++p0) {
            const int p = p0 * 4 + wave;
            if (p < DYP) {
                const long long off = (long long)(q0 + p * RPP) * a.Cout * 4 + dyv;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void_t *)(base + p * 1024), 16, off < (long long)a.dy_bytes ? (int)off : WG_OOB, 0, 0, 0);
            } else if (p < NP) {
                const long long off = (long long)(q0 + xshift + (p - DYP) * XRPP) * a.Cin * 4 + xv;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t *)(base + DY_BYTES + (p - DYP) * 1024), 16,
                                                         (off >= 0 && off < (long long)a.x_bytes) ? (int)off : WG_OOB, 0, 0, 0);
            }
        }
    };

    // operand element of a lane: pixel row 2 u + h (+ dx for the activations), channel (lane & 31) of the fragment
    const int h = lane >> 5, c = lane & 31;
    auto phys = [](const int row, const int b, const int rb) __attribute__((always_inline)) { return row * rb + ((((b >> 7) ^ (row & 1)) << 7) | (b & 127)); };
    int aoff[FA], boff[3];
#pragma unroll
    for (int i = 0; i < FA; ++i) aoff[i] = phys(h, (wa * (BMC / 2) + 32 * i + c) * 4, RB);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n0 = 96 * wb + 32 * j;
        boff[j] = DY_BYTES + phys(h + n0 / 64, (n0 % 64 + c) * 4, XRB);          // (2 u rows further per substep: the parity stays)
    }

    f32x16 acc[FA][3];
    float bsum[FA];                                               // bias gradient on the vector ALU (see the bf16 kernel)
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    const bool do_bias = a.bpart != nullptr && ci_t == 0 && dyr == 0 && wb == 0;

    if (s0 < s1) issue(s0, 0);
    for (int s = s0; s < s1; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 1 < s1) issue(s + 1, (s + 1 - s0) & 1);
        const char *const st = lds + ((s - s0) & 1) * STAGE;
#pragma unroll 4
        for (int u = 0; u < WGF_PIX / 2; ++u) {
            float fa[FA], fb[3];
#pragma unroll
            for (int i = 0; i < FA; ++i) fa[i] = *(const float *)(st + 2 * u * RB + aoff[i]);
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[j] = *(const float *)(st + 2 * u * XRB + boff[j]);
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < FA; ++i) bsum[i] += fa[i];
            }
        }
    }

    if (do_bias) {                                                // a lane holds channel lane & 31 of the pixel rows of parity lane >> 5
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32);
            if (lane < 32) a.bpart[(size_t)split * a.Cout + co0 + wa * (BMC / 2) + 32 * i + lane] = tot;
        }
    }
    float *const out = a.part + (size_t)split * a.Cout * 9 * a.Cin;
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n0 = 96 * wb + 32 * j, tap = 3 * dyr + n0 / 64, ci = ci0 + n0 % 64 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) {                         // a register quad = 4 consecutive output channels: one 16-byte store
                const int co = co0 + wa * (BMC / 2) + 32 * i + 8 * q + 4 * (lane >> 5);
                const f32x4p v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                *(f32x4p *)(out + ((size_t)tap * a.Cin + ci) * a.Cout + co) = v;
            }
        }
#endif
}

// folds the split-K partial sums (part[split][tap][ci][co]) and writes the gradient in the framework's filter layout (c_out,
// c_in_real, 3, 3), dropping the padding channels of the input layer (c_in_real <= c_in).  blockIdx.y = view; a workgroup owns 16
// output channels x 4 input channels x 9 taps = 144 16-byte pieces of every split: RED_G thread groups each add a contiguous range of
// the splits in split order (four loads in flight), the groups' sums are added in group order -- a fixed order, the same bits every
// run --, the 16 x 36 results are turned in LDS and leave as 16 contiguous runs of dw (the first version stored every float on its
// own at a stride of 36 bytes and walked the splits one dependent 4-byte load at a time: 43 us per launch).
#define RED_G 4
typedef float f32x4r __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(144 * RED_G) void conv3x3_wgrad_reduce_kernel(const WgradGroup g)
{
    __shared__ f32x4r s_sum[RED_G][144];
    __shared__ float s_out[16 * 36];
    const WgradView &v = g.v[blockIdx.y];
    const int splits = v.splits, c_in = g.Cin, c_in_real = v.c_in_real, c_out = g.Cout, ciq = c_in / 4, cob = c_out / 16;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= cob * ciq) {                           // the bias gradient's partial sums
        const int co = ((int)blockIdx.x - cob * ciq) * (144 * RED_G) + tid;
        if (v.db && co < c_out) {
            // (eight loads in flight, added in split order: the 64-channel layers have 300+ splits and this block was the launch's
            // long pole when it walked them one round trip at a time)
            float s = 0.f;
            int k = 0;
            for (; k + 8 <= splits; k += 8) {
                float b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) b[u] = v.bpart[(long)(k + u) * c_out + co];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += b[u];
            }
            for (; k < splits; ++k) s += v.bpart[(long)k * c_out + co];
            v.db[co] = s;
        }
        return;
    }
    const int co0 = ((int)blockIdx.x / ciq) * 16, ci0 = ((int)blockIdx.x % ciq) * 4;
    const int nreal = min(4, c_in_real - ci0);                    // real input channels of this block
    if (nreal <= 0) return;
    const int item = tid % 144, grp = tid / 144, tap = item / 16, ci_l = (item >> 2) & 3, c4 = item & 3;
    const long n4 = (long)c_out * 9 * c_in / 4;                   // float4 per split
    const f32x4r *p = (const f32x4r *)v.part + (((long)tap * c_in + ci0 + ci_l) * c_out + co0) / 4 + c4;
    const int per = (splits + RED_G - 1) / RED_G, k0 = grp * per, k1 = min(splits, k0 + per);
    f32x4r s = {0.f, 0.f, 0.f, 0.f};
    int k = k0;
    for (; k + 8 <= k1; k += 8) {                                 // eight 16-byte loads in flight (a 64 -> 64 layer: ~80 splits per group)
        f32x4r a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = p[(long)(k + u) * n4];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
    }
    for (; k + 4 <= k1; k += 4) {
        const f32x4r a0 = p[(long)k * n4], a1 = p[(long)(k + 1) * n4], a2 = p[(long)(k + 2) * n4], a3 = p[(long)(k + 3) * n4];
        s += a0; s += a1; s += a2; s += a3;
    }
    for (; k < k1; ++k) s += p[(long)k * n4];
    s_sum[grp][item] = s;
    __syncthreads();
    if (grp == 0) {
        f32x4r t = s_sum[0][item];
#pragma unroll
        for (int j = 1; j < RED_G; ++j) t += s_sum[j][item];
#pragma unroll
        for (int e = 0; e < 4; ++e) s_out[((c4 * 4 + e) * 4 + ci_l) * 9 + tap] = t[e];
    }
    __syncthreads();
    for (int j = tid; j < 16 * 36; j += 144 * RED_G) {
        const int co_l = j / 36, r = j % 36;
        if (r < nreal * 9) v.dw[((long)(co0 + co_l) * c_in_real + ci0) * 9 + r] = s_out[j];
    }
}

// fp32 OIHW filter -> the two packed bf16 forms the trunk's step needs, in one launch: fwd (O, 9 * Ipad), k = tap * Ipad + i (the
// forward convolution) and dgrad (I, 9 * O), W'[i][tap'][o] = W[o][i][8 - tap'] (the data-gradient convolution; nullptr = skip)
__global__ __launch_bounds__(256) void conv3x3_pack_kernel(const float *__restrict__ w, __bf16 *__restrict__ fwd, __bf16 *__restrict__ dgrad, int O, int I,
                                                            int Ipad)
{
    const long n = (long)O * I * 9;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 9);
        const long r = i / 9;
        const int ci = (int)(r % I), o = (int)(r / I);
        const __bf16 v = (__bf16)w[i];
        fwd[((long)o * 9 + tap) * Ipad + ci] = v;
        if (dgrad) dgrad[((long)ci * 9 + (8 - tap)) * O + o] = v;
    }
}

// the same for MANY filters in one launch (every 3x3 layer of the training graph at the top of a step: the packed copies only
// change when the optimiser has stepped), in the trunk's type T (bf16, or f32 for the reference-precision trunk).  A workgroup
// owns a tile of 64 output x 32 input channels x 9 taps: read as it lies in the OIHW filter (runs of 32 x 9 contiguous floats),
// turned in LDS, written as 32-channel runs of the forward packing and 64-filter runs of the data-gradient packing -- the first
// version stored both packings element by element at the strides of the OTHER layout (0.53 ms per step for 44 M weights).
#define PACK_MAX 48
struct PackItem { const float *w; void *fwd, *dgrad; int O, I, Ipad, first; };
struct PackMany { PackItem it[PACK_MAX]; int n, blocks; };
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_pack_many_kernel(const PackMany p)
{
    __shared__ T tile[64][32 * 9 + 2];
    int k = 0;
    for (int j = 1; j < p.n; ++j)
        if ((int)blockIdx.x >= p.it[j].first) k = j;
    const PackItem &t = p.it[k];
    const int O = t.O, I = t.I, Ipad = t.Ipad;
    const int ci_tiles = (I + 31) / 32, b = (int)blockIdx.x - t.first;
    const int o0 = (b / ci_tiles) * 64, c0 = (b % ci_tiles) * 32, nc = min(32, I - c0), run = nc * 9;
    const float *__restrict__ w = t.w;
    for (int idx = threadIdx.x; idx < 64 * run; idx += 256) {
        const int o = idx / run, r = idx - o * run;
        tile[o][r] = (T)w[((long)(o0 + o) * I + c0) * 9 + r];
    }
    __syncthreads();
    T *__restrict__ fwd = (T *)t.fwd;
    for (int idx = threadIdx.x; idx < 64 * run; idx += 256) {                  // (o, tap, ci): ci fastest
        const int ci = idx % nc, t2 = idx / nc, tap = t2 % 9, o = t2 / 9;
        fwd[((long)(o0 + o) * 9 + tap) * Ipad + c0 + ci] = tile[o][ci * 9 + tap];
    }
    T *__restrict__ dgrad = (T *)t.dgrad;
    if (dgrad) {
        for (int idx = threadIdx.x; idx < 64 * run; idx += 256) {              // (ci, tap, o): o fastest
            const int o = idx & 63, t2 = idx >> 6, tap = t2 % 9, ci = t2 / 9;
            dgrad[((long)(c0 + ci) * 9 + (8 - tap)) * O + o0 + o] = tile[o][ci * 9 + tap];
        }
    }
}

}  // namespace mv3d_wgrad
using namespace mv3d_wgrad;

// the workgroups the chip holds at once, for the kernel the plan is made for (registers and LDS decide: asked from the runtime
// once per kernel).  The launch should be ONE round of resident workgroups: an MFMA-bound kernel gains nothing from more, and a
// partly filled second round leaves some CUs with twice the work of others (measured: CUs busy 69 % of the launch).
template <typename K>
static int resident_workgroups(K kernel)
{
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0)
        return 512;
    return per_cu * prop.multiProcessorCount;
}
static int wgrad_target(int bmc, int es)
{
#ifdef MV3D_TUNING
    if (getenv("MV3D_WGRAD_TARGET")) return atoi(getenv("MV3D_WGRAD_TARGET"));
#endif
    static int cache[2][2] = {{0, 0}, {0, 0}};
    int &t = cache[es == 4][bmc == 128];
    if (!t) {
        if (es == 4) t = bmc == 128 ? resident_workgroups(conv3x3_wgrad_f32_kernel<128>) : resident_workgroups(conv3x3_wgrad_f32_kernel<64>);
        else t = bmc == 128 ? resident_workgroups(conv3x3_wgrad_kernel<128>) : resident_workgroups(conv3x3_wgrad_kernel<64>);
    }
    return t;
}

struct WgradPlan { int steps[WG_MAX_VIEWS], splits[WG_MAX_VIEWS], steps_per_split, tiles, bmc; size_t q[WG_MAX_VIEWS], part_off[WG_MAX_VIEWS], bpart_off[WG_MAX_VIEWS], bytes; };

static bool wgrad_plan(int n, const mv3d_wgrad_view *views, int c_in, int c_out, int pix, WgradPlan &P)
{
    if (n <= 0 || n > WG_MAX_VIEWS || !views || c_in <= 0 || c_in % 64 || c_out <= 0 || c_out % 64) return false;
    P.bmc = c_out % 128 == 0 ? 128 : 64;
    P.tiles = (c_out / P.bmc) * (c_in / 64) * 3;
    long total = 0;
    for (int k = 0; k < n; ++k) {
        if (views[k].batch <= 0 || views[k].height <= 0 || views[k].width <= 0) return false;
        P.q[k] = (size_t)views[k].batch * (views[k].height + 2) * (views[k].width + 2);
        P.steps[k] = (int)((P.q[k] + pix - 1) / pix);
        total += P.steps[k];
    }
    // K steps per workgroup: the smallest number (>= 4) with which ALL views' workgroups fit the chip at once -- every view rounds
    // its last split up, so the sum is checked, not estimated: 10 x 96 workgroups instead of 8 x 96 would be a second round
    const int target = wgrad_target(P.bmc, pix == WG_PIX ? 2 : 4);
    long sps = (total * P.tiles + target - 1) / target;
    if (sps < 4) sps = 4;
    for (;; ++sps) {
        long wgs = 0;
        for (int k = 0; k < n; ++k) wgs += (P.steps[k] + sps - 1) / sps;
        if (wgs * P.tiles <= target || wgs <= n) break;
    }
    P.steps_per_split = (int)sps;
    size_t o = 0;
    for (int k = 0; k < n; ++k) {
        P.splits[k] = (P.steps[k] + P.steps_per_split - 1) / P.steps_per_split;
        P.part_off[k] = o; o += (size_t)P.splits[k] * c_out * 9 * c_in * 4;
        P.bpart_off[k] = o; o += ((size_t)P.splits[k] * c_out * 4 + 15) / 16 * 16;
    }
    P.bytes = o;
    return true;
}

static size_t wgrad_ws_bytes(int batch, int height, int width, int c_in, int c_out, int pix)
{
    mv3d_wgrad_view w = {};
    w.batch = batch; w.height = height; w.width = width;
    WgradPlan P;
    return wgrad_plan(1, &w, c_in, c_out, pix, P) ? P.bytes : 0;
}

extern "C" size_t mv3d_conv3x3_wgrad_workspace_bytes(int batch, int height, int width, int c_in, int c_out)
{
    return wgrad_ws_bytes(batch, height, width, c_in, c_out, WG_PIX);
}
extern "C" size_t mv3d_conv3x3_wgrad_f32_workspace_bytes(int batch, int height, int width, int c_in, int c_out)
{
    return wgrad_ws_bytes(batch, height, width, c_in, c_out, WGF_PIX);
}
extern "C" size_t mv3d_conv3x3_wgrad_views_workspace_bytes(int num_views, const mv3d_wgrad_view *views, int c_in, int c_out, int f32_maps)
{
    WgradPlan P;
    return wgrad_plan(num_views, views, c_in, c_out, f32_maps ? WGF_PIX : WG_PIX, P) ? P.bytes : 0;
}

static int wgrad_views_entry(int n, const mv3d_wgrad_view *views, int c_in, int c_in_real, int c_out, void *workspace, size_t workspace_bytes,
                             void *stream, int es)
{
    if (c_in_real <= 0 || c_in_real > c_in || !workspace || ((uintptr_t)workspace & 15)) return MV3D_ERR_INVALID_ARG;
    const int pix = es == 2 ? WG_PIX : WGF_PIX;
    WgradPlan P;
    if (!wgrad_plan(n, views, c_in, c_out, pix, P)) return MV3D_ERR_INVALID_ARG;
    if (workspace_bytes < P.bytes) return MV3D_ERR_WORKSPACE;
    WgradGroup g;
    g.n = n; g.Cin = c_in; g.Cout = c_out; g.steps_per_split = P.steps_per_split; g.ci_tiles = c_in / 64; g.c_in_real = c_in_real;
    int grid = 0;
    const bool want_bias = views[0].db != nullptr;
    for (int k = 0; k < n; ++k) {
        const mv3d_wgrad_view &w = views[k];
        if (!w.x_framed || !w.dy_framed || !w.dw || (want_bias != (w.db != nullptr))) return MV3D_ERR_INVALID_ARG;
        if ((((uintptr_t)w.x_framed | (uintptr_t)w.dy_framed | (uintptr_t)w.dw) & 15) != 0) return MV3D_ERR_INVALID_ARG;
        if (P.q[k] * c_in * es >= 0x7fffff00u || P.q[k] * c_out * es >= 0x7fffff00u) return MV3D_ERR_INVALID_ARG;    // 32-bit buffer offsets
        WgradView &v = g.v[k];
        v.x = w.x_framed; v.dy = w.dy_framed; v.dw = w.dw; v.db = w.db;
        v.part = (float *)((char *)workspace + P.part_off[k]);
        v.bpart = w.db ? (float *)((char *)workspace + P.bpart_off[k]) : nullptr;
        v.Wp = w.width + 2; v.steps = P.steps[k]; v.splits = P.splits[k];
        v.c_in_real = w.c_in_real > 0 ? w.c_in_real : c_in_real;
        if (v.c_in_real > c_in) return MV3D_ERR_INVALID_ARG;
        v.x_bytes = (unsigned)(P.q[k] * c_in * es); v.dy_bytes = (unsigned)(P.q[k] * c_out * es);
        g.first[k] = grid;                                        // (counts global splits here)
        grid += P.splits[k];
    }
    for (int k = n; k < WG_MAX_VIEWS; ++k) { g.v[k] = g.v[0]; g.first[k] = grid; }
    g.tiles = P.tiles; g.gsplits = grid;
    grid = (grid + 7) / 8 * 8 * P.tiles;                          // workgroups: (global splits padded to a multiple of 8) x tiles
    hipStream_t s = (hipStream_t)stream;
    if (es == 2) {
        if (P.bmc == 128) hipLaunchKernelGGL(conv3x3_wgrad_kernel<128>, dim3(grid), dim3(256), 0, s, g);
        else hipLaunchKernelGGL(conv3x3_wgrad_kernel<64>, dim3(grid), dim3(256), 0, s, g);
    } else {
        if (P.bmc == 128) hipLaunchKernelGGL(conv3x3_wgrad_f32_kernel<128>, dim3(grid), dim3(256), 0, s, g);
        else hipLaunchKernelGGL(conv3x3_wgrad_f32_kernel<64>, dim3(grid), dim3(256), 0, s, g);
    }
    const unsigned rblocks = (unsigned)((c_out / 16) * (c_in / 4) + (want_bias ? (c_out + 144 * RED_G - 1) / (144 * RED_G) : 0));
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(rblocks, n), dim3(144 * RED_G), 0, s, g);
    return mv3d_launch_status();
}

static int wgrad_entry(const void *x_framed, const void *dy_framed, float *dw, float *db, int batch, int height, int width, int c_in, int c_in_real,
                       int c_out, void *workspace, size_t workspace_bytes, void *stream, int es)
{
    mv3d_wgrad_view w;
    w.x_framed = x_framed; w.dy_framed = dy_framed; w.dw = dw; w.db = db; w.batch = batch; w.height = height; w.width = width; w.c_in_real = 0;
    return wgrad_views_entry(1, &w, c_in, c_in_real, c_out, workspace, workspace_bytes, stream, es);
}

extern "C" int mv3d_conv3x3_wgrad_bf16(const void *x_framed, const void *dy_framed, float *dw, float *db, int batch, int height, int width,
                                       int c_in, int c_in_real, int c_out, void *workspace, size_t workspace_bytes, void *stream)
{
    return wgrad_entry(x_framed, dy_framed, dw, db, batch, height, width, c_in, c_in_real, c_out, workspace, workspace_bytes, stream, 2);
}
extern "C" int mv3d_conv3x3_wgrad_f32(const void *x_framed, const void *dy_framed, float *dw, float *db, int batch, int height, int width,
                                      int c_in, int c_in_real, int c_out, void *workspace, size_t workspace_bytes, void *stream)
{
    return wgrad_entry(x_framed, dy_framed, dw, db, batch, height, width, c_in, c_in_real, c_out, workspace, workspace_bytes, stream, 4);
}
extern "C" int mv3d_conv3x3_wgrad_views_bf16(int num_views, const mv3d_wgrad_view *views, int c_in, int c_in_real, int c_out, void *workspace,
                                             size_t workspace_bytes, void *stream)
{
    return wgrad_views_entry(num_views, views, c_in, c_in_real, c_out, workspace, workspace_bytes, stream, 2);
}
extern "C" int mv3d_conv3x3_wgrad_views_f32(int num_views, const mv3d_wgrad_view *views, int c_in, int c_in_real, int c_out, void *workspace,
                                            size_t workspace_bytes, void *stream)
{
    return wgrad_views_entry(num_views, views, c_in, c_in_real, c_out, workspace, workspace_bytes, stream, 4);
}

extern "C" int mv3d_conv3x3_pack_bf16(const float *w_oihw, void *fwd_packed, void *dgrad_packed, int c_out, int c_in, int c_in_pad, void *stream)
{
    if (!w_oihw || !fwd_packed || c_out <= 0 || c_in <= 0 || c_in_pad < c_in) return MV3D_ERR_INVALID_ARG;
    const long n = (long)c_out * c_in * 9;
    hipLaunchKernelGGL(conv3x3_pack_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                       (__bf16 *)fwd_packed, (__bf16 *)dgrad_packed, c_out, c_in, c_in_pad);
    return mv3d_launch_status();
}

template <typename T>
static int pack_many_entry(int num_items, const mv3d_pack_item *items, void *stream)
{
    if (num_items <= 0 || !items) return MV3D_ERR_INVALID_ARG;
    for (int i0 = 0; i0 < num_items; i0 += PACK_MAX) {
        PackMany p;
        p.n = num_items - i0 < PACK_MAX ? num_items - i0 : PACK_MAX;
        int blocks = 0;
        for (int k = 0; k < p.n; ++k) {
            const mv3d_pack_item &w = items[i0 + k];
            if (!w.w_oihw || !w.fwd_packed || w.c_out <= 0 || w.c_out % 64 || w.c_in <= 0 || w.c_in_pad < w.c_in) return MV3D_ERR_INVALID_ARG;
            PackItem &t = p.it[k];
            t.w = w.w_oihw; t.fwd = w.fwd_packed; t.dgrad = w.dgrad_packed; t.O = w.c_out; t.I = w.c_in; t.Ipad = w.c_in_pad;
            t.first = blocks;
            blocks += (w.c_out / 64) * ((w.c_in + 31) / 32);
        }
        for (int k = p.n; k < PACK_MAX; ++k) p.it[k] = p.it[0];
        p.blocks = blocks;
        hipLaunchKernelGGL(conv3x3_pack_many_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    }
    return mv3d_launch_status();
}
extern "C" int mv3d_conv3x3_pack_many_bf16(int num_items, const mv3d_pack_item *items, void *stream)
{
    return pack_many_entry<__bf16>(num_items, items, stream);
}
extern "C" int mv3d_conv3x3_pack_many_f32(int num_items, const mv3d_pack_item *items, void *stream)
{
    return pack_many_entry<float>(num_items, items, stream);
}
