// 3x3 / stride 1 / SAME convolution + bias + ReLU of the VGG16 trunks as an implicit GEMM on the gfx950 matrix cores
// (v_mfma_f32_32x32x16_f16: f16 operands, f32 accumulate), and the 2x2 max pool between the stages.
//
// What it stands in for: Network.conv(3, 3, c_o, 1, 1) / Network.max_pool(2, 2, 2, 2, 'VALID') of
// lib/networks/network.py:109-133,182-189 for the layer list of lib/networks/MV3D_train.py:44-81 (conv1_2 .. conv5_3 of the
// BEV and RGB trunks, rpn_conv/3x3).  Lower precision than the reference's fp32 graph: this is the serving trunk
// (BASELINE configs[4] says fp16), never the parity contract; the hot-path layers that consume its maps stay f32.
//
// Data layout (HBM): activations NHWC f16 with a one-pixel zero frame, (B, H + 2, W + 2, C): a tap (dy, dx) of the filter is
// then a plain address offset and no load in the K loop needs a bounds test.  Weights (Cout, 9 * Cin) f16, k = tap * Cin + c.
// GEMM view: D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k], pixels = B*H*W output positions, k = 9 * Cin.
//
// One workgroup = BM pixels x BN couts; a K step is 64 channels of one tap (128 B per pixel / per cout row).  Both operands are
// staged HBM/L2 -> LDS by buffer_load ... lds (16 B per lane, no VGPR round trip) into two LDS stages; the rows are stored with
// their 16-byte k-groups XOR-swizzled by (row >> 1) & 7 (applied on the SOURCE address: the DMA image is lane-linear), which
// makes the ds_read_b128 of the MFMA operands (32 consecutive rows, one k-group) bank-conflict-free per 16 lanes.
// Accumulators go back through LDS so that every global store is a full 16-byte piece of a pixel's channel row.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace mv3d_conv {

template <typename T> struct Vec { typedef T v8 __attribute__((ext_vector_type(8))); typedef T v4 __attribute__((ext_vector_type(4))); };
// the 16-byte piece of a row an MFMA operand read fetches: 8 f16 / bf16 values, or 4 f32 (the exact-f32 variant: v_mfma_f32_32x32x2_f32)
template <typename T> struct OpVec { typedef typename Vec<T>::v8 type; };
template <> struct OpVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

struct ConvArgs {
    const void *x;
    const void *w;
    const float *bias;
    const void *mask;             // optional, same layout as a framed 16-bit y: y = 0 where mask <= 0 (a ReLU's gradient gate)
    void *y;
    int H, W, Cin, Cout, M, HW, Bn;
    int out_pad, relu;
    unsigned x_bytes, w_bytes;
    int m_tiles, n_tiles;
    int Ho, Wo, xtiles;           // POOL mode: pooled size (H / 2, W / 2) and 2-row tiles per pair of rows
};

// Several views (the BEV / image / front-view trunks at one VGG depth: same channel counts, different map sizes) behind ONE
// launch: the grid is the concatenation of the views' tiles (each view's share a multiple of 8 workgroups, so the workgroup ->
// XCD mapping of a view does not depend on its neighbours).  At a training batch of 2 a single view's launch does not fill the
// chip (364 tiles of conv4_x on 512 workgroup slots); three trunks on three streams overlapped them, one grouped launch does it
// on ONE stream -- which is what data parallelism and graph capture need.
#define CONV_MAX_VIEWS 3
struct ConvGroup { ConvArgs v[CONV_MAX_VIEWS]; int n; int first[CONV_MAX_VIEWS]; };

#define BK_BYTES 128     // one K step = 128 bytes of channels of one tap: 64 f16 / bf16 values or 32 f32

// FIRST: the layer fed by the network input (conv1_1: 9 / 3 channels zero-padded to 16 = 32 B per pixel): a K step is FOUR taps
// x 16 channels (12 tap slots, the last 3 with zero weights), so K = 3 steps instead of 9 x 64 mostly-zero channels.
// STAGES: LDS stages of the operand pipeline; the DMA of K step kt + STAGES - 1 is issued while step kt is multiplied.
// T: the operand / activation type: _Float16 (serving), __bf16 (the training trunk: f16's 5-bit exponent would need loss scaling), or
// float = the reference's precision on v_mfma_f32_32x32x2_f32 (exact f32 products and sums at the f32 matrix rate, 1/16 of f16's):
// K steps of 32 channels, f32 maps in and out, MFMA-bound by a wide margin (8x the matrix time per staged byte).
// POOL: the layer's 2x2 / stride 2 max pool (Network.max_pool(2, 2, 2, 2, 'VALID'), network.py:182-189) in the epilogue -- for the
// serving graph's conv1_2 / conv2_2, whose full-size outputs nothing but the pool reads: an M tile is then TWO map rows x BM / 2
// columns (complete pooling windows), the epilogue takes the maximum of every window out of the LDS tile and writes only the
// pooled framed map (a quarter of the bytes; the separate pool launch, its read of the full map and the full map's write are gone).
template <typename T, int BM, int BN, int WP, int WC, int STAGES, bool OUT_F32, bool FIRST, bool POOL = false>
__global__ __launch_bounds__(WP * WC * 64) void conv3x3_f16_kernel(const ConvGroup g)
{
#if __HIP_DEVICE_COMPILE__      // (the LDS address-space casts below do not parse in the host pass, which only needs the stub)
    int view = 0;
#pragma unroll
    for (int j = 1; j < CONV_MAX_VIEWS; ++j)
        if (j < g.n && (int)blockIdx.x >= g.first[j]) view = j;
    const ConvArgs &a = g.v[view];
    constexpr int NW = WP * WC, NT = NW * 64;
    constexpr int TP = BM / WP, TC = BN / WC, FP = TP / 32, FC = TC / 32;
    constexpr int STAGE = (BM + BN) * BK_BYTES;
    constexpr int XCH = BM / 8 / NW, WCH = BN / 8 / NW;          // 8-row DMA pieces per wave and stage
    constexpr int ES = sizeof(T), CPS = BK_BYTES / ES;            // element size, channels per K step
    static_assert(ES == 2 || (OUT_F32 && !FIRST), "f32 operands: f32 maps, no input-layer packing");
    static_assert(!POOL || (!OUT_F32 && !FIRST), "pooled epilogue: 16-bit framed output of an ordinary layer");
    constexpr int TW = BM / 2;                                    // POOL: columns of the two-row tile
    constexpr int ESZ = OUT_F32 ? 4 : 2;
    constexpr int OUT_BYTES = BM * BN * ESZ;
    constexpr int LDS_BYTES = STAGES * STAGE > OUT_BYTES + BM * 4 ? STAGES * STAGE : OUT_BYTES + BM * 4;
    static_assert(LDS_BYTES <= 160 * 1024 && (STAGES == 2 || STAGES == 3) && NW % 2 == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && TP % 32 == 0 && TC % 32 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    unsigned *const s_pix = (unsigned *)(lds + OUT_BYTES);        // (filled after the K loop: the stages may cover it)

    // ---- workgroup -> tile: XCD x (= id & 7) owns a contiguous band of m-tiles (image rows: the tiles above / below a tile --
    // whose taps re-read two of its three input rows -- run on the SAME XCD at about the same time and hit its L2), and the
    // n-tiles of one m-tile run back to back (they share the activation rows).  Round-robin m-tiles over the XCDs instead:
    // conv1_2 6-8 % slower, the rest equal.
    const int id = (int)blockIdx.x - g.first[view], xcd = id & 7, local = id >> 3;
    const int mt = xcd * ((a.m_tiles + 7) >> 3) + local / a.n_tiles, nt = local % a.n_tiles;
    if (mt >= a.m_tiles) return;
    const int m0 = mt * BM, n0 = nt * BN;
    int pb = 0, pyo = 0, pxt = 0;                                 // POOL: frame, pooled row, column tile of this workgroup
    if constexpr (POOL) { pxt = mt % a.xtiles; const int t = mt / a.xtiles; pyo = t % a.Ho; pb = t / a.Ho; }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K2 = FIRST ? 3 * BK_BYTES : 9 * a.Cin * ES;         // bytes of one weight row
    const int Wp = a.W + 2;

    // ---- DMA source addresses: lane -> (row = lane >> 3 of the 8-row piece, 16-byte k-group (lane & 7) ^ swizzle(row))
    const int gsel = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    // (one pair of integer divisions per lane, the other pieces' pixels by walking NW * 8 pixels on: sixteen runtime divisions
    // per lane were a visible part of a short-K workgroup's life)
    int xoff[XCH];
    if constexpr (POOL) {                                         // pixel p of the tile = (row p / TW, column p % TW) of the row pair
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            const int p = (j * NW + wave) * 8 + (lane >> 3);
            const int xx = min(pxt * TW + (p % TW), a.W - 1), yy = 2 * pyo + p / TW;       // (columns past the map: computed, never stored)
            xoff[j] = ((pb * (a.H + 2) + yy) * Wp + xx) * a.Cin * ES + gsel * 16;
        }
    } else {
        int m = m0 + wave * 8 + (lane >> 3);
        int b = m / a.HW, r = m - b * a.HW, yy = r / a.W, xx = r - yy * a.W;
        const int last = ((a.Bn * (a.H + 2) - 3) * Wp + a.W - 1) * a.Cin * ES;     // pixel M - 1: rows past the end read it (never stored)
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            xoff[j] = (m < a.M ? ((b * (a.H + 2) + yy) * Wp + xx) * a.Cin * ES : last) + (FIRST ? (gsel & 1) : gsel) * 16;
            m += NW * 8;
            xx += NW * 8;
            while (xx >= a.W) { xx -= a.W; if (++yy == a.H) { yy = 0; ++b; } }
        }
    }
    const int woff = (lane >> 3) * K2 + gsel * 16;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)a.w_bytes, 0x00020000);
    const int wrow0 = (n0 + wave * 8) * K2;                       // + j * NW * 8 * K2 per piece

    // K-step state (scalar): tap (ty, tx) and the 64-channel slice cc of it
    const int cpt = a.Cin / CPS, KT = FIRST ? 3 : 9 * cpt;
    int ty = 0, tx = 0, cc = 0;
    int sx = 0, sw = 0, tapv = 0;                                 // source offsets of the step being fetched (next())
    char *dst = lds;
    auto next = [&](const int kt) __attribute__((always_inline)) {
        // (readfirstlane: the compiler keeps the loop-carried tap / slice counters in vector registers and, without it, wraps EVERY
        // X-side `buffer_load ... lds` of a K step in a waterfall loop -- readfirstlane, compare, saveexec, load, branch -- to
        // legalise the scalar offset: four such loops per step sat between the barrier and the step's first MFMA)
        sx = FIRST ? 0 : __builtin_amdgcn_readfirstlane(((ty * Wp + tx) * a.Cin + cc * CPS) * ES);
        sw = wrow0 + kt * BK_BYTES;
        dst = lds + (kt % STAGES) * STAGE;
        if (FIRST) {                                              // this lane's tap of the step (per lane, not per step)
            const int t = kt * 4 + (gsel >> 1), t9 = t < 9 ? t : 0, tyy = t9 / 3;
            tapv = (tyy * Wp + (t9 - tyy * 3)) * a.Cin * ES;
        }
        if (++cc == cpt) { cc = 0; if (++tx == 3) { tx = 0; ++ty; } }
    };
    // pieces [q0, q1) of the step's XCH + WCH DMA pieces of this wave
    auto issue = [&](const int q0, const int q1) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < XCH + WCH; ++q) {
            if (q < q0 || q >= q1) continue;
            if (q < XCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t *)(dst + (q * NW + wave) * 1024), 16, xoff[q < XCH ? q : 0] + tapv, sx, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t *)(dst + BM * BK_BYTES + ((q - XCH) * NW + wave) * 1024), 16, woff,
                                                         sw + (q - XCH) * NW * 8 * K2, 0, 0);
        }
    };
    constexpr int NQ = XCH + WCH;

    // ---- MFMA operand reads: lane -> fragment row lane & 31, k-group 2 * ks + (lane >> 5), un-swizzled by (row >> 1) & 7
    const int wp = wave % WP, wc = wave / WP;
    const int h = lane >> 5, sw7 = (lane >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (lane & 31) * BK_BYTES + (((ks * 2 + h) ^ sw7) << 4);
    const int x_lds = wp * TP * BK_BYTES, w_lds = BM * BK_BYTES + wc * TC * BK_BYTES;

    f32x16 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    constexpr int D = STAGES - 1;                                 // prefetch distance in K steps
#pragma unroll
    for (int p = 0; p < D; ++p)
        if (p < KT) { next(p); issue(0, NQ); }
    for (int kt = 0; kt < KT; ++kt) {
        // this wave's pieces of stage kt have landed (the D - 1 younger stages in flight may stay outstanding) ...
        if (D >= 2 && kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NQ) : "memory");   // (D <= 2 supported)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // ... everyone's have, and stage kt - 1 has been consumed
        const bool more = kt + D < KT;
        if (more) next(kt + D);
        if (more) issue(0, NQ);        // (all pieces up front: spreading them between the MFMAs measured 3-7 % slower)
        const char *const st = lds + (kt % STAGES) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            typename OpVec<T>::type fw[FC], fx[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) fw[i] = *(const typename OpVec<T>::type *)(st + w_lds + i * 32 * BK_BYTES + koff[ks]);
#pragma unroll
            for (int j = 0; j < FP; ++j) fx[j] = *(const typename OpVec<T>::type *)(st + x_lds + j * 32 * BK_BYTES + koff[ks]);
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    if constexpr (__is_same(T, _Float16)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fx[j], acc[i][j], 0, 0, 0);
                    else if constexpr (__is_same(T, __bf16)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
                    else {                                        // element e of both 4-vectors is one k (lane half h: channel 8 m + 4 h + e)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[i][e], fx[j][e], acc[i][j], 0, 0, 0);
                    }
                }
        }
    }
    __syncthreads();                                              // every wave is done with the stages: reuse them for the tile
    if constexpr (!POOL) {   // output address of every pixel of the tile: the tile's first pixel by division (wave-uniform), the others by walking
        const int b0 = m0 / a.HW, r0 = m0 - b0 * a.HW, y0 = r0 / a.W, x0 = r0 - y0 * a.W;
        const int Ho = a.H + 2 * a.out_pad, Wo = a.W + 2 * a.out_pad;
        for (int p = tid; p < BM; p += NT) {
            int b = b0, yy = y0, xx = x0 + p;
            while (xx >= a.W) { xx -= a.W; if (++yy == a.H) { yy = 0; ++b; } }
            s_pix[p] = m0 + p < a.M ? (unsigned)(((b * Ho + yy + a.out_pad) * Wo + xx + a.out_pad)) * (unsigned)(a.Cout * ESZ) : 0xFFFFFFFFu;
        }
    }

    // ---- epilogue: + bias, ReLU, -> LDS tile [pixel][cout] (16-byte slots XOR-swizzled by the pixel), -> 16-byte global stores
    // D layout of the 32x32 MFMA: lane holds column (= pixel) lane & 31, rows (= couts) 8 q + 4 (lane >> 5) + {0..3}, q = reg >> 2
    constexpr int ROWB = BN * ESZ, SLOTS = ROWB / 16;
#pragma unroll
    for (int i = 0; i < FC; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = wc * TC + i * 32 + 8 * q + 4 * h;
            const f32x4 bv = *(const f32x4 *)(a.bias + n0 + c0);
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int P = wp * TP + j * 32 + (lane & 31);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][4 * q + e] + bv[e];
                    if (a.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                if constexpr (OUT_F32) {
                    *(f32x4 *)(lds + P * ROWB + ((((c0 >> 2)) ^ (P & (SLOTS - 1))) << 4)) = v;
                } else {
                    typename Vec<T>::v4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (T)v[e];
                    *(typename Vec<T>::v4 *)(lds + P * ROWB + (((c0 >> 3) ^ (P & (SLOTS - 1))) << 4) + (c0 & 7) * 2) = hv;
                }
            }
        }
    }
    __syncthreads();
    char *const yb = (char *)a.y + (size_t)n0 * ESZ;
    if constexpr (POOL) {                                         // the maximum of every 2x2 window, written to the pooled framed map
        typedef typename Vec<T>::v8 V8;
        const size_t row0 = ((size_t)(pb * (a.Ho + 2) + pyo + 1) * (a.Wo + 2) + 1) * (size_t)(a.Cout * ESZ);
        for (int u = tid; u < (TW / 2) * SLOTS; u += NT) {
            const int xo = u / SLOTS, sl = u % SLOTS, xg = pxt * (TW / 2) + xo;
            if (xg >= a.Wo) continue;
            V8 m;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int P = (q >> 1) * TW + 2 * xo + (q & 1);
                const V8 v = *(const V8 *)(lds + P * ROWB + ((sl ^ (P & (SLOTS - 1))) << 4));
                if (q == 0) m = v;
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = (float)v[e] > (float)m[e] ? v[e] : m[e];
                }
            }
            *(V8 *)(yb + row0 + (size_t)xg * (a.Cout * ESZ) + sl * 16) = m;
        }
        return;
    }
#pragma unroll 4
    for (int u = tid; u < BM * SLOTS; u += NT) {
        const int P = u / SLOTS, s = u % SLOTS;
        const unsigned off = s_pix[P];
        u32x4 v = *(const u32x4 *)(lds + P * ROWB + ((s ^ (P & (SLOTS - 1))) << 4));
        if (off == 0xFFFFFFFFu) continue;
        if constexpr (!OUT_F32) if (a.mask) {                     // gate by the sign of the masking map's 8 values of this piece
            typedef typename Vec<T>::v8 V8;
            const V8 m = *(const V8 *)((const char *)a.mask + (size_t)n0 * ESZ + off + s * 16);
            V8 o = __builtin_bit_cast(V8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (float)m[e] > 0.f ? o[e] : (T)0.f;
            v = __builtin_bit_cast(u32x4, o);
        }
        if constexpr (OUT_F32 && ES == 4) if (a.mask) {           // f32 maps: the same gate on the piece's 4 values
            const f32x4 m = *(const f32x4 *)((const char *)a.mask + (size_t)n0 * ESZ + off + s * 16);
            f32x4 o = __builtin_bit_cast(f32x4, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = m[e] > 0.f ? o[e] : 0.f;
            v = __builtin_bit_cast(u32x4, o);
        }
        *(u32x4 *)(yb + off + s * 16) = v;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// The 256 x 256 workgroup with a WAVE-SPECIALISED K loop: no barrier at which all eight waves stand behind one fetch.
//
// The two-stage loop above runs the deep layers at ~1.05 PFLOP/s with its waves parked ~49 % of their cycles at the one
// `vmcnt(0)` + barrier of a K step: ONE workgroup per CU, eight waves that fetch together, wait together and multiply together.
// Here the eight waves are two GROUPS of four (one wave of each group on every SIMD) that run the same program ONE BARRIER
// INTERVAL APART: while a SIMD's group-0 wave is in a matrix segment (8 MFMAs, 256 cycles of the SIMD's matrix pipe), its
// group-1 wave is in a load segment (LDS operand reads of its next segment + its two DMA pieces + its counted wait), and the
// other way round in the next interval -- the matrix pipe of a SIMD always has a wave that is multiplying.
//
//   * K tile = 64 channels of one tap (as above), cut into FOUR half-tiles of 16 KB that are staged and consumed separately:
//     X0 / X1 = the first / second 64 pixels of every wave's 128, W0 / W1 = the first / second 32 couts of every wave's 64.
//     A K tile is four PHASES, each a load segment and a matrix segment over one quadrant (64 pixels x 32 couts) of the wave's
//     128 x 64 output tile:   P1 reads X0 + W0 (12 ds_read_b128) and multiplies (X0, W0);  P2 reads W1 (4), (X0, W1);
//     P3 reads X1 (8), (X1, W1);  P4 reads nothing, (X1, W0) -- W0's fragments stay in registers from P1.  24 operand reads per
//     K tile and wave, 32 MFMAs.
//   * the DMA stream is ONE half-tile per phase, six half-tiles ahead of the phase that reads it, into two K-tile buffers
//     (2 x 64 KB): phase P1 of tile t issues W1[t + 1], P2 X1[t + 1], P3 X0[t + 2], P4 W0[t + 2].  A slot is rewritten at the
//     earliest three barrier intervals after the lagging group's last read of it (whose `lgkmcnt(0)` has retired by then).
//   * waits are COUNTED and never zero in the steady state: a wave waits `vmcnt(8)` (its own two pieces of the four youngest
//     half-tiles stay in flight) in P4 / P1 / P2 -- before the phase's first barrier, for the half-tile that the NEXT phase
//     reads -- so every piece has four phases (~2 k cycles) to land, and the barrier behind the wait publishes it to the readers
//     of both groups (the lagging group's wait precedes the barrier the leading group passes before it reads).
//   * barriers: two per phase (load | matrix), raw `s_barrier`; group 1 takes one extra at the start, group 0 one at the end.
// Same LDS image (rows of 128 B, 16-byte k-groups XOR-swizzled by (row >> 1) & 7 on the DMA's source side), same fragments, same
// accumulator layout and the same epilogue as the kernel above: bit-identical results (the order of a fragment's k steps is
// unchanged: ks ascending inside a K tile, K tiles ascending).
template <typename T>
__global__ __launch_bounds__(512) void conv3x3_pp_kernel(const ConvGroup g)
{
#if __HIP_DEVICE_COMPILE__
    int view = 0;
#pragma unroll
    for (int j = 1; j < CONV_MAX_VIEWS; ++j)
        if (j < g.n && (int)blockIdx.x >= g.first[j]) view = j;
    const ConvArgs &a = g.v[view];
    constexpr int BM = 256, BN = 256, NT = 512, TP = 128, TC = 64, FP = 4, FC = 2;   // 8 waves = 2 pixel halves x 4 cout quarters
    constexpr int ES = sizeof(T), CPS = BK_BYTES / ES;
    static_assert(ES == 2, "16-bit operands");
    constexpr int HALF = 128 * BK_BYTES;                          // one half-tile: 128 rows x 128 B = 16 KB
    constexpr int TILE = 4 * HALF;                                // X0 | W0 | W1 | X1
    constexpr int ESZ = 2, OUT_BYTES = BM * BN * ESZ;
    constexpr int LDS_BYTES = OUT_BYTES + BM * 4 > 2 * TILE ? OUT_BYTES + BM * 4 : 2 * TILE;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    unsigned *const s_pix = (unsigned *)(lds + OUT_BYTES);

    // workgroup -> tile as in the kernel above: XCD x owns a band of m-tiles and runs a tile's n-tiles back to back.  (Measured and
    // dropped, profiles/r05_conv_mfma.txt: the two 256-cout n-tiles of a 512-cout layer on XCD halves, so that each L2 keeps one
    // weight panel -- 3 - 7 % slower: the activation rows then cross the fabric twice.)
    const int id = (int)blockIdx.x - g.first[view], xcd = id & 7, local = id >> 3;
    const int mt = xcd * ((a.m_tiles + 7) >> 3) + local / a.n_tiles, nt = local % a.n_tiles;
    if (mt >= a.m_tiles) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K2 = 9 * a.Cin * ES;                                // bytes of one weight row
    const int Wp = a.W + 2;

    // ---- DMA source offsets.  Piece q (0 / 1) of a half-tile, this wave: LDS rows (q * 8 + wave) * 8 + (lane >> 3).
    //   X half h: LDS row r -> tile pixel (r / 64) * 128 + h * 64 + r % 64      (r / 64 = the pixel half wp of the reading waves)
    //   W half h: LDS row r -> tile cout  (r / 32) * 64 + h * 32 + r % 32       (r / 32 = the cout quarter wc of the reading waves)
    const int gsel = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    const int last = ((a.Bn * (a.H + 2) - 3) * Wp + a.W - 1) * a.Cin * ES;         // pixel M - 1: rows past the end read it (never stored)
    int xoff[2][2], woff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (q * 8 + wave) * 8 + (lane >> 3);
            const int m = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
            const int b = m / a.HW, rr = m - b * a.HW, yy = rr / a.W, xx = rr - yy * a.W;
            xoff[h][q] = (m < a.M ? ((b * (a.H + 2) + yy) * Wp + xx) * a.Cin * ES : last) + gsel * 16;
            woff[h][q] = (n0 + (r >> 5) * 64 + h * 32 + (r & 31)) * K2 + gsel * 16;
        }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)a.w_bytes, 0x00020000);

    // the DMA stream, statically scheduled: X / W half-tile `h` of K tile `t` (scalar tap / slice offset sxt of that tile)
    const int cpt = a.Cin / CPS, KT = 9 * cpt;
    auto issue_x = [&](const int h, const int t, const int sxt) __attribute__((always_inline)) {
        char *const dst = lds + __builtin_amdgcn_readfirstlane((t & 1) * TILE + (h ? 3 * HALF : 0));
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t *)(dst + (q * 8 + wave) * 1024), 16, xoff[h][q], sxt, 0, 0);
    };
    auto issue_w = [&](const int h, const int t) __attribute__((always_inline)) {
        char *const dst = lds + __builtin_amdgcn_readfirstlane((t & 1) * TILE + (1 + h) * HALF);
        const int sw = __builtin_amdgcn_readfirstlane(t * BK_BYTES);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t *)(dst + (q * 8 + wave) * 1024), 16, woff[h][q], sw, 0, 0);
    };
    // scalar source offset of a K tile's tap / channel slice: tiles are walked in order, one cursor that runs two tiles ahead
    int c_ty = 0, c_tx = 0, c_cc = 0;
    auto next_sx = [&]() __attribute__((always_inline)) -> int {
        const int v = __builtin_amdgcn_readfirstlane(((c_ty * Wp + c_tx) * a.Cin + c_cc * CPS) * ES);
        if (++c_cc == cpt) { c_cc = 0; if (++c_tx == 3) { c_tx = 0; ++c_ty; } }
        return v;
    };

    // ---- operand reads
    const int wp = wave & 1, wc = (wave >> 1) & 3;                // pixel half, cout quarter of this wave's 128 x 64 tile
    // (groups: waves {0..3} = group 0, {4..7} = group 1 -- a workgroup's waves go to the SIMDs in a cyclic order, so waves w and
    // w + 4 share a SIMD: one of each group)
    const int h5 = lane >> 5, sw7 = (lane >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (lane & 31) * BK_BYTES + (((ks * 2 + h5) ^ sw7) << 4);
    typedef typename OpVec<T>::type OV;
    const int x_rows = wp * 64 * BK_BYTES, w_rows = wc * 32 * BK_BYTES;

    f32x16 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define PP_SB() __builtin_amdgcn_sched_barrier(0)
#define PP_BAR() do { PP_SB(); __builtin_amdgcn_s_barrier(); PP_SB(); } while (0)
#define PP_MFMA(ACC, FW, FX)                                                                                        \
    do {                                                                                                            \
        if constexpr (__is_same(T, _Float16)) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(FW, FX, ACC, 0, 0, 0);   \
        else ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FW, FX, ACC, 0, 0, 0);                                   \
    } while (0)

    // prologue: half-tiles X0 W0 W1 X1 of tile 0, X0 W0 of tile 1 (stream order; KT >= 9); the first two have to have landed
    const int sx0 = next_sx();
    int sx_a = next_sx(), sx_b = next_sx();                       // tap / slice offsets of tiles kt + 1 and kt + 2
    issue_x(0, 0, sx0); issue_w(0, 0); issue_w(1, 0); issue_x(1, 0, sx0);
    issue_x(0, 1, sx_a); issue_w(0, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    PP_BAR();
    if (wave >= 4) PP_BAR();                                      // group 1 runs one barrier interval behind group 0

    for (int kt = 0; kt < KT; ++kt) {
        const char *const st = lds + (kt & 1) * TILE;
        OV fx[2][4], fw0[4], fw1[4];
        // ---- P1: X0 + W0 -> (X0, W0)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fw0[ks] = *(const OV *)(st + HALF + w_rows + koff[ks]);
        PP_SB();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fx[j][ks] = *(const OV *)(st + x_rows + j * 32 * BK_BYTES + koff[ks]);
        const bool more1 = kt + 1 < KT, more2 = kt + 2 < KT;      // (scalar)
        if (more1) issue_w(1, kt + 1);                            // W1[kt + 1]
        // counted wait: this wave's pieces of the FOUR youngest half-tiles stay in flight (nothing more is issued for the last
        // two tiles: they wait for everything) -> W1[kt] has landed
        if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_BAR();
        __builtin_amdgcn_s_setprio(1);                            // (the compiler's own counted lgkmcnt waits sit in front of each MFMA)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) PP_MFMA(acc[0][j], fw0[ks], fx[j][ks]);
        __builtin_amdgcn_s_setprio(0);
        PP_BAR();
        // ---- P2: W1 -> (X0, W1)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fw1[ks] = *(const OV *)(st + 2 * HALF + w_rows + koff[ks]);
        if (more1) issue_x(1, kt + 1, sx_a);                      // X1[kt + 1]
        if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // X1[kt] has landed
        PP_BAR();
        __builtin_amdgcn_s_setprio(1);                            // (the compiler's own counted lgkmcnt waits sit in front of each MFMA)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) PP_MFMA(acc[1][j], fw1[ks], fx[j][ks]);
        __builtin_amdgcn_s_setprio(0);
        PP_BAR();
        // ---- P3: X1 -> (X1, W1)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fx[j][ks] = *(const OV *)(st + 3 * HALF + x_rows + j * 32 * BK_BYTES + koff[ks]);
        if (more2) issue_x(0, kt + 2, sx_b);                      // X0[kt + 2]
        PP_BAR();
        __builtin_amdgcn_s_setprio(1);                            // (the compiler's own counted lgkmcnt waits sit in front of each MFMA)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) PP_MFMA(acc[1][2 + j], fw1[ks], fx[j][ks]);
        __builtin_amdgcn_s_setprio(0);
        PP_BAR();
        // ---- P4: (X1, W0), W0's fragments still in registers
        if (more2) issue_w(0, kt + 2);                            // W0[kt + 2]
        if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // X0 / W0[kt + 1] have landed
        sx_a = sx_b;
        sx_b = next_sx();
        PP_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) PP_MFMA(acc[0][2 + j], fw0[ks], fx[j][ks]);
        __builtin_amdgcn_s_setprio(0);
        PP_BAR();
    }
    if (wave < 4) PP_BAR();                                       // group 0 waits for group 1's last interval
#undef PP_MFMA
    __syncthreads();                                              // every wave is done with the stages: reuse them for the tile
    {   // output address of every pixel of the tile
        const int b0 = m0 / a.HW, r0 = m0 - b0 * a.HW, y0 = r0 / a.W, x0 = r0 - y0 * a.W;
        const int Ho = a.H + 2 * a.out_pad, Wo = a.W + 2 * a.out_pad;
        for (int p = tid; p < BM; p += NT) {
            int b = b0, yy = y0, xx = x0 + p;
            while (xx >= a.W) { xx -= a.W; if (++yy == a.H) { yy = 0; ++b; } }
            s_pix[p] = m0 + p < a.M ? (unsigned)(((b * Ho + yy + a.out_pad) * Wo + xx + a.out_pad)) * (unsigned)(a.Cout * ESZ) : 0xFFFFFFFFu;
        }
    }
    // ---- epilogue (as above): + bias, ReLU, -> LDS tile [pixel][cout], -> 16-byte global stores [gated by the mask map]
    constexpr int ROWB = BN * ESZ, SLOTS = ROWB / 16;
#pragma unroll
    for (int i = 0; i < FC; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = wc * TC + i * 32 + 8 * q + 4 * h5;
            const f32x4 bv = *(const f32x4 *)(a.bias + n0 + c0);
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int P = wp * TP + j * 32 + (lane & 31);
                typename Vec<T>::v4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][4 * q + e] + bv[e];
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    hv[e] = (T)v;
                }
                *(typename Vec<T>::v4 *)(lds + P * ROWB + (((c0 >> 3) ^ (P & (SLOTS - 1))) << 4) + (c0 & 7) * 2) = hv;
            }
        }
    }
    __syncthreads();
    char *const yb = (char *)a.y + (size_t)n0 * ESZ;
#pragma unroll 4
    for (int u = tid; u < BM * SLOTS; u += NT) {
        const int P = u / SLOTS, s = u % SLOTS;
        const unsigned off = s_pix[P];
        u32x4 v = *(const u32x4 *)(lds + P * ROWB + ((s ^ (P & (SLOTS - 1))) << 4));
        if (off == 0xFFFFFFFFu) continue;
        if (a.mask) {                                             // gate by the sign of the masking map's 8 values of this piece
            typedef typename Vec<T>::v8 V8;
            const V8 m = *(const V8 *)((const char *)a.mask + (size_t)n0 * ESZ + off + s * 16);
            V8 o = __builtin_bit_cast(V8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (float)m[e] > 0.f ? o[e] : (T)0.f;
            v = __builtin_bit_cast(u32x4, o);
        }
        *(u32x4 *)(yb + off + s * 16) = v;
    }
#endif
}

// the pools' views (one launch for the trunks of one depth, like ConvGroup): view k owns workgroups [first[k], first[k + 1])
struct PoolView { const void *x, *g; void *y; int B, H, W, Ho, Wo; };
struct PoolGroup { PoolView v[CONV_MAX_VIEWS]; int n, C8; int first[CONV_MAX_VIEWS + 1]; };
__device__ __forceinline__ int pool_view_of(const PoolGroup &g)
{
    int k = 0;
#pragma unroll
    for (int j = 1; j < CONV_MAX_VIEWS; ++j)
        if (j < g.n && (int)blockIdx.x >= g.first[j]) k = j;
    return k;
}

// 2x2 / stride 2 max pool, VALID (floor), framed NHWC -> framed NHWC; one thread = 8 channels of one output pixel
template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const PoolGroup g)
{
    typedef typename Vec<T>::v8 V;
    const int k = pool_view_of(g);
    const V *__restrict__ x = (const V *)g.v[k].x;
    V *__restrict__ y = (V *)g.v[k].y;
    const int B = g.v[k].B, H = g.v[k].H, W = g.v[k].W, Ho = g.v[k].Ho, Wo = g.v[k].Wo, C8 = g.C8;
    const long total = (long)B * Ho * Wo * C8, nblk = g.first[k + 1] - g.first[k];
    for (long i = (long)(blockIdx.x - g.first[k]) * blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
        const int c = (int)(i % C8);
        long r = i / C8;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho), b = (int)(r / Ho);
        const V *p = x + (((long)b * (H + 2) + 2 * yo + 1) * (W + 2) + 2 * xo + 1) * C8 + c;
        const V v0 = p[0], v1 = p[C8], v2 = p[(long)(W + 2) * C8], v3 = p[(long)(W + 2) * C8 + C8];
        V o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {                 // (through f32: exact for a maximum, and there is no packed bf16 max)
            const float m0 = fmaxf((float)v0[e], (float)v1[e]), m1 = fmaxf((float)v2[e], (float)v3[e]);
            o[e] = (T)fmaxf(m0, m1);
        }
        y[(((long)b * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1) * C8 + c] = o;
    }
}

// Gradient of (ReLU ->) 2x2 max pool: y = the framed pre-pool map (a ReLU output), g = framed gradient w.r.t. the pooled map ->
// gy = framed gradient w.r.t. the PRE-ACTIVATION of y: the window's FIRST maximum (row-major, torch's / MIOpen's choice) gets
// the pooled gradient if it is > 0 (the ReLU mask), the other three positions 0.  Rows / columns the VALID pool drops are not
// written (the owner zeroed the buffer once).  One thread = 8 channels of one pooled pixel.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const PoolGroup pg)
{
    typedef typename Vec<T>::v8 V;
    const int k = pool_view_of(pg);
    const V *__restrict__ y = (const V *)pg.v[k].x;
    const V *__restrict__ g = (const V *)pg.v[k].g;
    V *__restrict__ gy = (V *)pg.v[k].y;
    const int B = pg.v[k].B, H = pg.v[k].H, W = pg.v[k].W, Ho = pg.v[k].Ho, Wo = pg.v[k].Wo, C8 = pg.C8;
    const long total = (long)B * Ho * Wo * C8, nblk = pg.first[k + 1] - pg.first[k];
    for (long i = (long)(blockIdx.x - pg.first[k]) * blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
        const int c = (int)(i % C8);
        long r = i / C8;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho), b = (int)(r / Ho);
        const long base = (((long)b * (H + 2) + 2 * yo + 1) * (W + 2) + 2 * xo + 1) * C8 + c, row = (long)(W + 2) * C8;
        const V v0 = y[base], v1 = y[base + C8], v2 = y[base + row], v3 = y[base + row + C8];
        const V gv = g[(((long)b * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1) * C8 + c];
        V o0, o1, o2, o3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a0 = (float)v0[e], a1 = (float)v1[e], a2 = (float)v2[e], a3 = (float)v3[e];
            const float m = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
            const T gg = m > 0.f ? gv[e] : (T)0.f, z = (T)0.f;
            const int k = a0 == m ? 0 : (a1 == m ? 1 : (a2 == m ? 2 : 3));
            o0[e] = k == 0 ? gg : z; o1[e] = k == 1 ? gg : z; o2[e] = k == 2 ? gg : z; o3[e] = k == 3 ? gg : z;
        }
        gy[base] = o0; gy[base + C8] = o1; gy[base + row] = o2; gy[base + row + C8] = o3;
    }
}

// NHWC f32 (B,H,W,C) -> framed NHWC f16 (B,H+2,W+2,Co), Co >= C: interior pixels, first C channels only (the frame and the
// padding channels are zeroed once by the owner of the buffer)
template <typename T>
__global__ __launch_bounds__(256) void frame_f32_kernel(const float *__restrict__ x, T *__restrict__ y, int B, int H, int W, int C, int Co)
{
    const long total = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int xx = (int)(r % W);
        r /= W;
        const int yy = (int)(r % H), b = (int)(r / H);
        y[(((long)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * Co + c] = (T)x[i];
    }
}

// The same in 16-byte pieces of the framed map: a thread converts the E = 16 / sizeof(T) channels of one piece (channels past C: zeros --
// the padding channels of the input layers' maps are rewritten with the zeros they hold) and writes them with ONE store; consecutive
// threads = consecutive pieces of a pixel, then the next pixel, so a wave reads and writes contiguous runs.  (The element-per-thread kernel
// above -- three 64-bit divisions and one 2-byte store per value -- ran the 608 x 608 x 9 input of a batch of 16 at 2.1 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void frame_pieces_kernel(const float *__restrict__ x, T *__restrict__ y, int B, int H, int W, int C, int Co)
{
    constexpr int E = 16 / (int)sizeof(T);
    const int npc = Co / E;                                          // pieces per pixel
    const unsigned total = (unsigned)B * H * W * npc;                // (the launcher guarantees < 2^32)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned pix = i / (unsigned)npc, k = i - pix * (unsigned)npc;
        const unsigned row = pix / (unsigned)W, xx = pix - row * (unsigned)W;     // row = b * H + yy
        const unsigned b = row / (unsigned)H, yy = row - b * (unsigned)H;
        const float *const src = x + (size_t)pix * C;
        const int c0 = (int)k * E;
        T v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = c0 + e < C ? (T)src[c0 + e] : (T)0.0f;
        u32x4 o;
        __builtin_memcpy(&o, v, 16);
        *(u32x4 *)((char *)y + (((size_t)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * (size_t)Co * sizeof(T) + (size_t)k * 16) = o;
    }
}

template <typename T, int BM, int BN, int WP, int WC, int STAGES, bool FIRST>
int launch_conv(ConvGroup &g, int out_f32, hipStream_t s)
{
    int grid = 0;
    for (int k = 0; k < g.n; ++k) {
        ConvArgs &b = g.v[k];
        b.m_tiles = (b.M + BM - 1) / BM;
        b.n_tiles = b.Cout / BN;
        g.first[k] = grid;
        grid += (b.m_tiles + 7) / 8 * 8 * b.n_tiles;
    }
    for (int k = g.n; k < CONV_MAX_VIEWS; ++k) { g.v[k] = g.v[0]; g.first[k] = grid; }
    if constexpr (sizeof(T) == 4) {                               // f32 operands: f32 maps only
        hipLaunchKernelGGL((conv3x3_f16_kernel<T, BM, BN, WP, WC, STAGES, true, FIRST>), dim3(grid), dim3(WP * WC * 64), 0, s, g);
    } else {
        if (out_f32)
            hipLaunchKernelGGL((conv3x3_f16_kernel<T, BM, BN, WP, WC, STAGES, true, FIRST>), dim3(grid), dim3(WP * WC * 64), 0, s, g);
        else
            hipLaunchKernelGGL((conv3x3_f16_kernel<T, BM, BN, WP, WC, STAGES, false, FIRST>), dim3(grid), dim3(WP * WC * 64), 0, s, g);
    }
    return mv3d_launch_status();
}

// 16-bit output only (tiles whose f32 output image would not fit the LDS)
template <typename T, int BM, int BN, int WP, int WC>
int launch_conv16(ConvGroup &g, hipStream_t s)
{
    int grid = 0;
    for (int k = 0; k < g.n; ++k) {
        ConvArgs &b = g.v[k];
        b.m_tiles = (b.M + BM - 1) / BM;
        b.n_tiles = b.Cout / BN;
        g.first[k] = grid;
        grid += (b.m_tiles + 7) / 8 * 8 * b.n_tiles;
    }
    for (int k = g.n; k < CONV_MAX_VIEWS; ++k) { g.v[k] = g.v[0]; g.first[k] = grid; }
    hipLaunchKernelGGL((conv3x3_f16_kernel<T, BM, BN, WP, WC, 2, false, false>), dim3(grid), dim3(WP * WC * 64), 0, s, g);
    return mv3d_launch_status();
}

// the wave-specialised 256 x 256 workgroup (conv3x3_pp_kernel): 16-bit framed / bare output, ordinary layers
template <typename T>
int launch_conv_pp(ConvGroup &g, hipStream_t s)
{
    int grid = 0;
    for (int k = 0; k < g.n; ++k) {
        ConvArgs &b = g.v[k];
        b.m_tiles = (b.M + 255) / 256;
        b.n_tiles = b.Cout / 256;
        g.first[k] = grid;
        grid += (b.m_tiles + 7) / 8 * 8 * b.n_tiles;
    }
    for (int k = g.n; k < CONV_MAX_VIEWS; ++k) { g.v[k] = g.v[0]; g.first[k] = grid; }
    hipLaunchKernelGGL((conv3x3_pp_kernel<T>), dim3(grid), dim3(512), 0, s, g);
    return mv3d_launch_status();
}

template <typename T, int BM, int BN, int WP, int WC>
int launch_conv_pool(ConvGroup &g, hipStream_t s)
{
    int grid = 0;
    for (int k = 0; k < g.n; ++k) {
        ConvArgs &b = g.v[k];
        b.Ho = b.H / 2; b.Wo = b.W / 2;
        b.xtiles = (2 * b.Wo + BM / 2 - 1) / (BM / 2);
        b.m_tiles = b.Bn * b.Ho * b.xtiles;
        b.n_tiles = b.Cout / BN;
        g.first[k] = grid;
        grid += (b.m_tiles + 7) / 8 * 8 * b.n_tiles;
    }
    for (int k = g.n; k < CONV_MAX_VIEWS; ++k) { g.v[k] = g.v[0]; g.first[k] = grid; }
    hipLaunchKernelGGL((conv3x3_f16_kernel<T, BM, BN, WP, WC, 2, false, false, true>), dim3(grid), dim3(WP * WC * 64), 0, s, g);
    return mv3d_launch_status();
}

}  // namespace mv3d_conv
using namespace mv3d_conv;

// fills one view's arguments; false = invalid
static bool conv_view_args(ConvArgs &a, const mv3d_conv_view &w, int c_in, int c_out, int out_framed, int out_f32, int relu, int es, bool first,
                           bool gated)
{
    if (!w.x_framed || !w.w_packed || !w.bias || !w.y || w.batch <= 0 || w.height <= 0 || w.width <= 0) return false;
    if ((((uintptr_t)w.x_framed | (uintptr_t)w.w_packed | (uintptr_t)w.bias | (uintptr_t)w.y | (uintptr_t)w.gate_framed) & 15) != 0) return false;   // 16-byte pieces
    if (gated != (w.gate_framed != nullptr)) return false;
    const size_t xb = (size_t)w.batch * (w.height + 2) * (w.width + 2) * c_in * es, wb = (size_t)c_out * (first ? 192 : 9 * c_in) * es;
    const size_t yb = (size_t)w.batch * (w.height + 2 * (out_framed != 0)) * (w.width + 2 * (out_framed != 0)) * c_out * (out_f32 ? 4 : 2);
    if (xb >= 0x7fffffffu || wb >= 0x7fffffffu || yb >= 0xffffffffu) return false;                   // 32-bit buffer offsets
    a.x = w.x_framed; a.w = w.w_packed; a.bias = w.bias; a.y = w.y; a.mask = w.gate_framed;
    a.H = w.height; a.W = w.width; a.Cin = c_in; a.Cout = c_out; a.HW = w.height * w.width; a.M = w.batch * w.height * w.width; a.Bn = w.batch;
    a.out_pad = out_framed != 0; a.relu = relu != 0;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    a.m_tiles = a.n_tiles = 0;
    return true;
}

template <typename T>
static int conv3x3_views_entry(int num_views, const mv3d_conv_view *views, int c_in, int c_out, int out_framed, int out_f32, int relu, void *stream)
{
    if (num_views <= 0 || num_views > CONV_MAX_VIEWS || !views) return MV3D_ERR_INVALID_ARG;
    const bool first = c_in == 16;                                // the input layer's packing (see the header)
    if (c_in <= 0 || (c_in % 64 && !first) || c_out <= 0 || c_out % 64) return MV3D_ERR_INVALID_ARG;
    const bool gated = views[0].gate_framed != nullptr;
    if (gated && (!out_framed || out_f32)) return MV3D_ERR_INVALID_ARG;
    ConvGroup g;
    g.n = num_views;
    for (int k = 0; k < num_views; ++k)
        if (!conv_view_args(g.v[k], views[k], c_in, c_out, out_framed, out_f32, relu, 2, first, gated)) return MV3D_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
#ifdef MV3D_TUNING
    static const int input_env = getenv("MV3D_CONV_INPUT") ? atoi(getenv("MV3D_CONV_INPUT")) : 1;
#else
    const int input_env = 1;
#endif
    if (first && c_out == 64 && !out_f32 && !gated && input_env) {   // the input layer's own kernel (conv_input.hip)
        if constexpr (__is_same(T, _Float16)) return mv3d_launch_conv_input_f16(num_views, views, out_framed, relu, s);
        else return mv3d_launch_conv_input_bf16(num_views, views, out_framed, relu, s);
    }
    if (first) return c_out % 128 == 0 ? launch_conv<T, 128, 128, 2, 2, 2, true>(g, out_f32, s) : launch_conv<T, 256, 64, 4, 1, 2, true>(g, out_f32, s);
    // 128x128 / 4 waves / 2 stages, two workgroups per CU.  (256x128 / 8 waves / 3 stages, one workgroup per CU with the DMA two
    // steps ahead, measured equal on the 512-channel layers and 3-5 % slower on the 128 / 256-channel ones: profiles/r03_conv_mfma.txt)
    if (c_out % 128 == 0) {
        // few tiles (one frame: 184 tiles of conv4_x / conv5_x on 512 workgroup slots): half-height tiles put twice as many CUs to
        // work -- the latency of a batch-1 frame (BASELINE configs[1]) is these layers' tile time, not their throughput
        long tiles128 = 0;
        for (int k = 0; k < num_views; ++k) tiles128 += (long)((g.v[k].M + 127) / 128) * (c_out / 128);
#ifdef MV3D_TUNING
        static const int small_max = getenv("MV3D_CONV_SMALL_MAX") ? atoi(getenv("MV3D_CONV_SMALL_MAX")) : 512;
#else
        const int small_max = 512;
#endif
        if (tiles128 < small_max) return launch_conv<T, 64, 128, 2, 2, 2, false>(g, out_f32, s);
        // many tiles and 16-bit maps out: 256 x 256 workgroups of 8 waves, each wave a 128-pixel x 64-cout tile -- per matrix
        // instruction half the DMA pieces and 3/4 of the LDS operand reads of the 128 x 128 / 64 x 64 form (one workgroup per CU:
        // 128 KB of LDS).  Batch-16 serving layers: conv4_2 954 -> 1063 TFLOP/s, conv3_2 848 -> 911, image conv4_2 988 -> 1020;
        // needs enough tiles for several rounds of 256 workgroups (a training batch keeps the 128 x 128 form).
        long tiles256 = 0;
        for (int k = 0; k < num_views; ++k) tiles256 += (long)((g.v[k].M + 255) / 256) * (c_out / 256);
#ifdef MV3D_TUNING
        static const int big_min = getenv("MV3D_CONV_BIG_MIN") ? atoi(getenv("MV3D_CONV_BIG_MIN")) : 640;
        static const int big_wp = getenv("MV3D_CONV_BIG_WP") ? atoi(getenv("MV3D_CONV_BIG_WP")) : 2;
#else
        const int big_min = 640, big_wp = 2;
#endif
        if (!out_f32 && c_out % 256 == 0 && tiles256 >= big_min) {
#ifdef MV3D_TUNING
            static const int pp = getenv("MV3D_CONV_PP") ? atoi(getenv("MV3D_CONV_PP")) : 1;
#else
            const int pp = 1;
#endif
            // the same tile with the wave-specialised K loop (two groups of four waves one barrier interval apart, counted waits)
            if (pp) return launch_conv_pp<T>(g, s);
            return big_wp == 2 ? launch_conv16<T, 256, 256, 2, 4>(g, s) : launch_conv16<T, 256, 256, 4, 2>(g, s);
        }
#ifdef MV3D_TUNING                                                // experiment builds: the 256 x 256 workgroup, 8 waves of 128 x 64
        static const int big = getenv("MV3D_CONV_TILE") ? atoi(getenv("MV3D_CONV_TILE")) : 0;
        if (big == 256 && c_out % 256 == 0 && !out_f32) return launch_conv16<T, 256, 256, 2, 4>(g, s);
        if (big == 2561 && c_out % 128 == 0 && !out_f32) return launch_conv16<T, 256, 128, 2, 2>(g, s);
        if (big == 2562 && c_out % 128 == 0 && !out_f32) return launch_conv16<T, 256, 128, 4, 2>(g, s);
#endif
        return launch_conv<T, 128, 128, 2, 2, 2, false>(g, out_f32, s);
    }
    return launch_conv<T, 256, 64, 4, 1, 2, false>(g, out_f32, s);
}

// convolution + bias + ReLU + 2x2 max pool: y = the POOLED framed map (batch, height / 2 + 2, width / 2 + 2, c_out)
template <typename T>
static int conv3x3_pool_views_entry(int num_views, const mv3d_conv_view *views, int c_in, int c_out, void *stream)
{
    if (num_views <= 0 || num_views > CONV_MAX_VIEWS || !views) return MV3D_ERR_INVALID_ARG;
    if (c_in <= 0 || c_in % 64 || c_out <= 0 || c_out % 64) return MV3D_ERR_INVALID_ARG;
    ConvGroup g;
    g.n = num_views;
    for (int k = 0; k < num_views; ++k) {
        if (views[k].height < 2 || views[k].width < 2 || views[k].gate_framed) return MV3D_ERR_INVALID_ARG;
        if (!conv_view_args(g.v[k], views[k], c_in, c_out, 1, 0, 1, 2, false, false)) return MV3D_ERR_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (c_out % 128 == 0) return launch_conv_pool<T, 128, 128, 2, 2>(g, s);
    return launch_conv_pool<T, 256, 64, 4, 1>(g, s);
}

template <typename T>
static int conv3x3_entry(const void *x_framed, const void *w_packed, const float *bias, void *y, int batch, int height, int width, int c_in,
                         int c_out, int out_framed, int out_f32, int relu, void *stream, const void *mask = nullptr)
{
    mv3d_conv_view w;
    w.x_framed = x_framed; w.w_packed = w_packed; w.bias = bias; w.gate_framed = mask; w.y = y;
    w.batch = batch; w.height = height; w.width = width; w.reserved0 = 0;
    return conv3x3_views_entry<T>(1, &w, c_in, c_out, out_framed, out_f32, relu, stream);
}

// exact-f32 variant: f32 framed activations (c_in a multiple of 32), f32 packed weights, f32 output
static int conv3x3_f32_views_entry(int num_views, const mv3d_conv_view *views, int c_in, int c_out, int out_framed, int relu, void *stream)
{
    if (num_views <= 0 || num_views > CONV_MAX_VIEWS || !views) return MV3D_ERR_INVALID_ARG;
    if (c_in <= 0 || c_in % 32 || c_out <= 0 || c_out % 64) return MV3D_ERR_INVALID_ARG;
    const bool gated = views[0].gate_framed != nullptr;          // (an f32 gate map, framed like the output)
    if (gated && !out_framed) return MV3D_ERR_INVALID_ARG;
    ConvGroup g;
    g.n = num_views;
    long tiles128 = 0;
    for (int k = 0; k < num_views; ++k) {
        if (!conv_view_args(g.v[k], views[k], c_in, c_out, out_framed, 1, relu, 4, false, gated)) return MV3D_ERR_INVALID_ARG;
        tiles128 += (long)((g.v[k].M + 127) / 128) * (c_out / 128);
    }
    hipStream_t s = (hipStream_t)stream;
    // few tiles (a training batch): the f32 kernel is MFMA-bound even with one wave per SIMD, so a CU that holds two workgroups
    // just takes twice as long -- half-size tiles balance the last round (fp32 training step on one stream 58.7 -> 56 ms)
    if (c_out % 128 == 0 && tiles128 < 1024) return launch_conv<float, 64, 128, 2, 2, 2, false>(g, 1, s);
    if (c_out % 128 == 0) return launch_conv<float, 128, 128, 2, 2, 2, false>(g, 1, s);
    return launch_conv<float, 128, 64, 2, 2, 2, false>(g, 1, s);
}

static int conv3x3_f32_entry(const void *x_framed, const void *w_packed, const float *bias, void *y, int batch, int height, int width, int c_in,
                             int c_out, int out_framed, int relu, void *stream)
{
    mv3d_conv_view w;
    w.x_framed = x_framed; w.w_packed = w_packed; w.bias = bias; w.gate_framed = nullptr; w.y = y;
    w.batch = batch; w.height = height; w.width = width; w.reserved0 = 0;
    return conv3x3_f32_views_entry(1, &w, c_in, c_out, out_framed, relu, stream);
}

// x = the map to pool (forward) / the pre-pool map y (backward); g = gradient w.r.t. the pooled map (backward only)
template <typename T, bool BWD>
static int pool_views_entry(int num_views, const mv3d_pool_view *views, int channels, void *stream)
{
    if (num_views <= 0 || num_views > CONV_MAX_VIEWS || !views || channels <= 0 || channels % 8) return MV3D_ERR_INVALID_ARG;
    PoolGroup g;
    g.n = num_views; g.C8 = channels / 8;
    int grid = 0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_pool_view &w = views[k];
        if (!w.x_framed || !w.y_framed || (BWD && !w.g_pooled_framed) || w.batch <= 0 || w.height < 2 || w.width < 2) return MV3D_ERR_INVALID_ARG;
        if ((((uintptr_t)w.x_framed | (uintptr_t)w.y_framed | (uintptr_t)w.g_pooled_framed) & 15) != 0) return MV3D_ERR_INVALID_ARG;
        PoolView &v = g.v[k];
        v.x = w.x_framed; v.g = w.g_pooled_framed; v.y = w.y_framed;
        v.B = w.batch; v.H = w.height; v.W = w.width; v.Ho = w.height / 2; v.Wo = w.width / 2;
        const long total = (long)v.B * v.Ho * v.Wo * g.C8;
        g.first[k] = grid;
        grid += (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    }
    for (int k = num_views; k < CONV_MAX_VIEWS; ++k) { g.v[k] = g.v[0]; g.first[k] = grid; }
    g.first[num_views] = grid;
    g.first[CONV_MAX_VIEWS] = grid;
    if (BWD) hipLaunchKernelGGL(maxpool2x2_bwd_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(maxpool2x2_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g);
    return mv3d_launch_status();
}

template <typename T>
static int maxpool_entry(const void *x_framed, void *y_framed, int batch, int height, int width, int channels, void *stream)
{
    mv3d_pool_view w;
    w.x_framed = x_framed; w.g_pooled_framed = nullptr; w.y_framed = y_framed; w.batch = batch; w.height = height; w.width = width; w.reserved0 = 0;
    return pool_views_entry<T, false>(1, &w, channels, stream);
}

template <typename T>
static int frame_entry(const float *x_nhwc, void *y_framed, int batch, int height, int width, int channels, int channels_out, void *stream)
{
    if (!x_nhwc || !y_framed || batch <= 0 || height <= 0 || width <= 0 || channels <= 0 || channels_out < channels) return MV3D_ERR_INVALID_ARG;
    constexpr int E = 16 / (int)sizeof(T);
    const long pieces = (long)batch * height * width * (channels_out / E);
    if (channels_out % E == 0 && ((uintptr_t)y_framed & 15) == 0 && pieces < 0xffffffffL) {
        const int grid = (int)((pieces + 255) / 256 < 65536 ? (pieces + 255) / 256 : 65536);
        hipLaunchKernelGGL(frame_pieces_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_nhwc, (T *)y_framed, batch, height, width, channels,
                           channels_out);
        return mv3d_launch_status();
    }
    const long total = (long)batch * height * width * channels;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(frame_f32_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_nhwc, (T *)y_framed, batch, height, width, channels,
                       channels_out);
    return mv3d_launch_status();
}

extern "C" int mv3d_conv3x3_f16(const void *x, const void *w, const float *bias, void *y, int batch, int height, int width, int c_in, int c_out,
                                int out_framed, int out_f32, int relu, void *stream)
{
    return conv3x3_entry<_Float16>(x, w, bias, y, batch, height, width, c_in, c_out, out_framed, out_f32, relu, stream);
}
extern "C" int mv3d_conv3x3_bf16(const void *x, const void *w, const float *bias, void *y, int batch, int height, int width, int c_in, int c_out,
                                 int out_framed, int out_f32, int relu, void *stream)
{
    return conv3x3_entry<__bf16>(x, w, bias, y, batch, height, width, c_in, c_out, out_framed, out_f32, relu, stream);
}
extern "C" int mv3d_conv3x3_gated_bf16(const void *x, const void *w, const float *bias, const void *gate_framed, void *y, int batch, int height,
                                       int width, int c_in, int c_out, void *stream)
{
    if (!gate_framed) return MV3D_ERR_INVALID_ARG;
    return conv3x3_entry<__bf16>(x, w, bias, y, batch, height, width, c_in, c_out, 1, 0, 0, stream, gate_framed);
}
extern "C" int mv3d_conv3x3_f32(const void *x, const void *w, const float *bias, void *y, int batch, int height, int width, int c_in, int c_out,
                                int out_framed, int relu, void *stream)
{
    return conv3x3_f32_entry(x, w, bias, y, batch, height, width, c_in, c_out, out_framed, relu, stream);
}
extern "C" int mv3d_conv3x3_views_f16(int num_views, const mv3d_conv_view *views, int c_in, int c_out, int out_framed, int out_f32, int relu,
                                      void *stream)
{
    return conv3x3_views_entry<_Float16>(num_views, views, c_in, c_out, out_framed, out_f32, relu, stream);
}
extern "C" int mv3d_conv3x3_views_bf16(int num_views, const mv3d_conv_view *views, int c_in, int c_out, int out_framed, int out_f32, int relu,
                                       void *stream)
{
    return conv3x3_views_entry<__bf16>(num_views, views, c_in, c_out, out_framed, out_f32, relu, stream);
}
extern "C" int mv3d_conv3x3_views_f32(int num_views, const mv3d_conv_view *views, int c_in, int c_out, int out_framed, int relu, void *stream)
{
    return conv3x3_f32_views_entry(num_views, views, c_in, c_out, out_framed, relu, stream);
}
extern "C" int mv3d_conv3x3_pool_views_f16(int num_views, const mv3d_conv_view *views, int c_in, int c_out, void *stream)
{
    return conv3x3_pool_views_entry<_Float16>(num_views, views, c_in, c_out, stream);
}
extern "C" int mv3d_conv3x3_pool_views_bf16(int num_views, const mv3d_conv_view *views, int c_in, int c_out, void *stream)
{
    return conv3x3_pool_views_entry<__bf16>(num_views, views, c_in, c_out, stream);
}
extern "C" int mv3d_maxpool2x2_f32(const void *x, void *y, int batch, int height, int width, int channels, void *stream)
{
    return maxpool_entry<float>(x, y, batch, height, width, channels, stream);
}
extern "C" int mv3d_frame_nhwc_f32(const float *x, void *y, int batch, int height, int width, int channels, int channels_out, void *stream)
{
    return frame_entry<float>(x, y, batch, height, width, channels, channels_out, stream);
}
extern "C" int mv3d_maxpool2x2_f16(const void *x, void *y, int batch, int height, int width, int channels, void *stream)
{
    return maxpool_entry<_Float16>(x, y, batch, height, width, channels, stream);
}
extern "C" int mv3d_maxpool2x2_bf16(const void *x, void *y, int batch, int height, int width, int channels, void *stream)
{
    return maxpool_entry<__bf16>(x, y, batch, height, width, channels, stream);
}
template <typename T>
static int maxpool_bwd_entry(const void *y_framed, const void *g_pooled_framed, void *gy_framed, int batch, int height, int width, int channels,
                             void *stream)
{
    mv3d_pool_view w;
    w.x_framed = y_framed; w.g_pooled_framed = g_pooled_framed; w.y_framed = gy_framed; w.batch = batch; w.height = height; w.width = width;
    w.reserved0 = 0;
    return pool_views_entry<T, true>(1, &w, channels, stream);
}
extern "C" int mv3d_maxpool2x2_views_f16(int num_views, const mv3d_pool_view *views, int channels, void *stream)
{
    return pool_views_entry<_Float16, false>(num_views, views, channels, stream);
}
extern "C" int mv3d_maxpool2x2_views_bf16(int num_views, const mv3d_pool_view *views, int channels, void *stream)
{
    return pool_views_entry<__bf16, false>(num_views, views, channels, stream);
}
extern "C" int mv3d_maxpool2x2_views_f32(int num_views, const mv3d_pool_view *views, int channels, void *stream)
{
    return pool_views_entry<float, false>(num_views, views, channels, stream);
}
extern "C" int mv3d_maxpool2x2_bwd_views_bf16(int num_views, const mv3d_pool_view *views, int channels, void *stream)
{
    return pool_views_entry<__bf16, true>(num_views, views, channels, stream);
}
extern "C" int mv3d_maxpool2x2_bwd_views_f32(int num_views, const mv3d_pool_view *views, int channels, void *stream)
{
    return pool_views_entry<float, true>(num_views, views, channels, stream);
}
extern "C" int mv3d_maxpool2x2_bwd_bf16(const void *y, const void *g, void *gy, int batch, int height, int width, int channels, void *stream)
{
    return maxpool_bwd_entry<__bf16>(y, g, gy, batch, height, width, channels, stream);
}
extern "C" int mv3d_maxpool2x2_bwd_f32(const void *y, const void *g, void *gy, int batch, int height, int width, int channels, void *stream)
{
    return maxpool_bwd_entry<float>(y, g, gy, batch, height, width, channels, stream);
}
extern "C" int mv3d_frame_nhwc_f16(const float *x, void *y, int batch, int height, int width, int channels, int channels_out, void *stream)
{
    return frame_entry<_Float16>(x, y, batch, height, width, channels, channels_out, stream);
}
extern "C" int mv3d_frame_nhwc_bf16(const float *x, void *y, int batch, int height, int width, int channels, int channels_out, void *stream)
{
    return frame_entry<__bf16>(x, y, batch, height, width, channels, channels_out, stream);
}
