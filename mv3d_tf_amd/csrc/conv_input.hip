// The trunks' INPUT layer (conv1_1: 9 / 3 input channels zero-padded to 16, 64 filters; lib/networks/MV3D_train.py:44) as a kernel of its
// own.  The layer is 1 % of the trunk's FLOPs and 15 % of its bytes: per pixel it reads 32 B and writes 128 B, so it is bound by the
// output stores, and the general implicit-GEMM kernel (conv3x3_mfma.hip: 256 x 64 tiles, operands staged through LDS, K = 3 steps
// between two barriers) spends its time in tile prologues and epilogues (2.1 TB/s of stores, 19 % of the matrix pipes busy).  Here:
//   * the whole filter (64 x 9 taps x 16 channels, 18 KB per view) sits in LDS for the life of a workgroup, laid out as the lanes' A
//     operands of v_mfma_f32_32x32x16 (two 32-filter blocks x nine taps, one K step = one tap's 16 channels): one conflict-free
//     ds_read_b128 per MFMA;
//   * a wave owns a column strip of 32 pixels x 16 map rows at a time (grid-stride over all strips of all views).  The B operand of
//     tap (ty, tx) is ONE 16-byte load per lane straight from the framed input -- lane = (pixel, channel half), 1 KB contiguous per
//     wave -- and a framed row's three operands (tx = 0, 1, 2) serve three output rows: four row slots rotate through registers, so
//     an output row costs three loads (requested one row ahead), not nine; no LDS staging of activations, no barrier in the loop;
//   * 18 MFMAs per 32 pixels, then bias + ReLU + conversion, the 32 x 64 tile turned through 4.25 KB of wave-private LDS so that every
//     global store instruction writes eight pixels' complete 128-byte lines.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace mv3d_conv_input {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Op;
template <> struct Op<_Float16> { typedef _Float16 v8 __attribute__((ext_vector_type(8))); typedef _Float16 v4 __attribute__((ext_vector_type(4))); };
template <> struct Op<__bf16> { typedef __bf16 v8 __attribute__((ext_vector_type(8))); typedef __bf16 v4 __attribute__((ext_vector_type(4))); };

struct InView {
    const char *x;               // framed (B, H + 2, W + 2, 16) of T
    const char *w;               // (64, 192) of T, k = 16 * tap + c
    const float *bias;           // (64)
    char *y;                     // (B, H + 2 pad, W + 2 pad, 64) of T
    int B, H, W, pad;
    int xblocks, strips;         // ceil(W / 32), ceil(H / IN_ROWS)
    unsigned y_bytes;
};
struct InGroup { InView v[3]; int n, relu; int first[4]; int dbg; };   // first[k]: first unit of view k; first[n] = all units

#ifndef IN_ROWS
#define IN_ROWS 16
#endif
#define IN_ROWS_DOC               // map rows per unit (a column strip of 32 pixels x IN_ROWS rows)
// one workgroup of IN_WAVES waves per CU (two waves per SIMD), sharing the filters in LDS
#ifndef IN_WAVES
#define IN_WAVES 8
#endif
#ifndef IN_TILE_ROT
#define IN_TILE_ROT 0
#endif
#ifndef IN_ROWB
#define IN_ROWB 136
#endif
#define IN_ROWB_DOC              // LDS bytes per pixel row of the output tile (128 + 8: the 8-byte writes of 32 pixels spread over the banks)

template <typename T>
__global__ __launch_bounds__(IN_WAVES * 64) void conv3x3_input_kernel(const InGroup g)
{
#if __HIP_DEVICE_COMPILE__
    typedef typename Op<T>::v8 V8;
    typedef typename Op<T>::v4 V4;
    __shared__ __attribute__((aligned(16))) char lds[IN_WAVES][32 * IN_ROWB];
    __shared__ V8 s_w[3][2][9][64];                               // every view's filter as the lanes' A operands: [view][filter block][tap][lane]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;                       // MFMA operand row / column, k half
    char *const tile = lds[(wave + IN_TILE_ROT) & (IN_WAVES - 1)];
    const int wid = blockIdx.x * IN_WAVES + wave, nw = gridDim.x * IN_WAVES, total = g.first[g.n];
    for (int i = threadIdx.x; i < g.n * 18 * 64; i += IN_WAVES * 64) {
        const int l = i & 63, t = (i >> 6) % 9, cb = ((i >> 6) / 9) & 1, k = i / (18 * 64);
        s_w[k][cb][t][l] = *(const V8 *)(g.v[k].w + ((cb * 32 + (l & 31)) * 192 + t * 16 + (l >> 5) * 8) * 2);
    }
    __syncthreads();
    int view = -1;
    f32x16 bias16[2];                                             // the accumulators' initial value: D rows (= filters) 8 q + 4 h + e
    for (int u = wid; u < total; u += nw) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < 3; ++j)
            if (j < g.n && u >= g.first[j]) k = j;
        const InView &a = g.v[k];
        if (k != view) {                                          // (a wave crosses a view boundary at most twice)
            view = k;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bq = *(const f32x4 *)(a.bias + cb * 32 + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bias16[cb][4 * q + e] = bq[e];
                }
        }
        const V8 (*const fw)[9][64] = s_w[k];
        int r = u - g.first[k];
        const int xb = r % a.xblocks; r /= a.xblocks;
        const int ys = r % a.strips, b = r / a.strips;
        const int x0 = xb * 32, y0 = ys * IN_ROWS, Wp = a.W + 2;
        // B operands: pixel x0 + n (clamped inside the row: computed, never stored), tap column tx, channel half h, of framed row fr
        const int xx = min(x0 + n, a.W - 1);
        const char *const px = a.x + (((size_t)b * (a.H + 2) * Wp + xx) * 16 + h * 8) * 2;
        const size_t row_b = (size_t)Wp * 32;
        // output: eight pixels x eight 16-byte pieces per store instruction; columns past the map get an offset past the buffer (dropped
        // by the hardware's range check: no branch, so the compiler's load / store wait counts stay exact)
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, (int)a.y_bytes, 0x00020000);
        const int Wo = a.W + 2 * a.pad;
        unsigned so[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = i * 8 + (lane >> 3);
            so[i] = x0 + p < a.W ? (unsigned)((x0 + p + a.pad) * 128 + (lane & 7) * 16) : 0xfffffff0u;
        }
        // Four row slots rotate: output row y0 + j multiplies framed rows y0 + j .. + 2 (slots j, j + 1, j + 2 mod 4) while framed row
        // y0 + j + 3 -- the only new one the next output row needs -- is on its way into slot j + 3.  Rows past the map's last are
        // clamped onto it and their output rows dropped by the store's range check: every strip runs the same IN_ROWS branch-free steps.
        V8 R[4][3];
        u32x4 sv[2][4] = {};                                        // the stores' data registers, two sets (see IN_STEP)
#define IN_LOAD(slot, fr)                                                                                   \
        {                                                                                                   \
            const char *const q_ = px + (size_t)min((fr), a.H + 1) * row_b;                                 \
            R[slot][0] = *(const V8 *)(q_); R[slot][1] = *(const V8 *)(q_ + 32); R[slot][2] = *(const V8 *)(q_ + 64); \
        }
        IN_LOAD(0, y0) IN_LOAD(1, y0 + 1) IN_LOAD(2, y0 + 2)
#ifdef IN_NO_SCHED_BARRIER
#define IN_SCHED_BARRIER
#else
#define IN_SCHED_BARRIER __builtin_amdgcn_sched_barrier(0);   /* (one step's accumulators at a time: the steps are not interleaved) */
#endif
#if defined(IN_ORDER_CB1_FIRST)
#define IN_MFMA_LOOPS _Pragma("unroll") for (int ty = 0; ty < 3; ++ty) _Pragma("unroll") for (int tx = 0; tx < 3; ++tx) _Pragma("unroll") for (int cb = 1; cb >= 0; --cb)
#elif defined(IN_ORDER_SEQ)
#define IN_MFMA_LOOPS _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) _Pragma("unroll") for (int ty = 0; ty < 3; ++ty) _Pragma("unroll") for (int tx = 0; tx < 3; ++tx)
#else
#define IN_MFMA_LOOPS _Pragma("unroll") for (int ty = 0; ty < 3; ++ty) _Pragma("unroll") for (int tx = 0; tx < 3; ++tx) _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)
#endif
#ifndef IN_STORE_GUARD
#define IN_STORE_GUARD
#endif
#ifndef IN_FENCE
#define IN_FENCE "s_waitcnt lgkmcnt(0)"
#endif
#ifdef IN_NOPS
#define IN_NOP asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#else
#define IN_NOP
#endif
#define IN_STEP(J)                                                                                          \
        {                                                                                                   \
            const bool live_ = y0 + j0 + (J) < a.H;      /* (a row past the map: computed from clamped rows, never stored) */ \
            const int yy = min(y0 + j0 + (J), a.H - 1);                                                     \
            IN_LOAD(((J) + 3) & 3, yy + 3)                                                                  \
            f32x16 acc[2];                                                                                  \
            _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;                            \
            V8 fa[2][9];                                                                                    \
            _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                \
                _Pragma("unroll") for (int t = 0; t < 9; ++t) fa[cb][t] = fw[cb][t][lane];                  \
            IN_MFMA_LOOPS {                                                                                 \
                        if constexpr (__is_same(T, _Float16))                                               \
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cb][ty * 3 + tx], R[((J) + ty) & 3][tx], acc[cb], 0, 0, 0); \
                        else                                                                                \
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][ty * 3 + tx], R[((J) + ty) & 3][tx], acc[cb], 0, 0, 0); \
                    }                                                                                       \
            /* (the A operands stay live until the step's last MFMA) */                                     \
            _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                \
                _Pragma("unroll") for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(fa[cb][t]));           \
            IN_NOP                                                                                          \
            _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
                    V4 hv;                                                                                  \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                         \
                        float v_ = acc[cb][4 * q + e] + bias16[cb][4 * q + e];                              \
                        if (g.relu) v_ = v_ > 0.f ? v_ : 0.f;                                               \
                        hv[e] = (T)v_;                                                                      \
                    }                                                                                       \
                    *(V4 *)(tile + n * IN_ROWB + (cb * 32 + 8 * q + 4 * h) * 2) = hv;                      \
                }                                                                                           \
            /* the tile is written as 8-byte halves vectors and read as 64-bit integers: without a compiler barrier the two sides \
               are free to pass each other (type-based aliasing); the hardware keeps a wave's LDS operations in order */ \
            asm volatile(IN_FENCE ::: "memory");                                                                  \
            const unsigned rowo = (unsigned)((b * (a.H + 2 * a.pad) + yy + a.pad) * Wo) * 128u;             \
            /* The data registers of a store must not be rewritten soon after it: with two waves on a SIMD the younger wave's \
               stores were seen to send values written to their data registers AFTER the store instruction (the next step's LDS \
               addresses appeared in the output, tools/conv_input_check.py) -- the compiler's wait states do not cover the time a \
               128-bit store's data waits for the bus behind the other wave.  So the stores' data registers alternate between two \
               sets, and a set stays live (this empty asm reads it) until the NEXT step's stores are about to be issued. */ \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(sv[((J) + 1) & 1][i]));    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                 \
                const int p = i * 8 + (lane >> 3), s_ = lane & 7;                                           \
                if constexpr (IN_ROWB % 16 == 0) sv[(J) & 1][i] = *(const u32x4 *)(tile + p * IN_ROWB + s_ * 16); \
                else {                                                                                      \
                    const unsigned long long lo = *(const unsigned long long *)(tile + p * IN_ROWB + s_ * 16); \
                    const unsigned long long hi = *(const unsigned long long *)(tile + p * IN_ROWB + s_ * 16 + 8); \
                    sv[(J) & 1][i] = u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)}; \
                }                                                                                           \
            }                                                                                               \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
                __builtin_amdgcn_raw_buffer_store_b128(sv[(J) & 1][i], ry, live_ ? so[i] : 0xfffffff0u, rowo, 0); \
            IN_STORE_GUARD                                                                                  \
            asm volatile(IN_FENCE ::: "memory");                                                                  \
            IN_SCHED_BARRIER                                                                                \
        }
#pragma unroll 1
        for (int j0 = 0; j0 < IN_ROWS; j0 += 4) { IN_STEP(0) IN_STEP(1) IN_STEP(2) IN_STEP(3) }
#undef IN_STEP
#undef IN_LOAD
    }
#endif
}

}  // namespace mv3d_conv_input

// views validated by the caller (conv3x3_views_entry: 16-byte aligned pointers, sizes inside 32-bit offsets); c_in = 16, c_out = 64,
// 16-bit output, no gate
template <typename T>
static int launch_conv_input(int num_views, const mv3d_conv_view *views, int out_framed, int relu, hipStream_t stream)
{
    using namespace mv3d_conv_input;
    InGroup g;
    g.n = num_views; g.relu = relu != 0; g.dbg = 0;
#ifdef MV3D_TUNING
    if (getenv("MV3D_CONV_INPUT_DBG")) g.dbg = atoi(getenv("MV3D_CONV_INPUT_DBG"));
#endif
    long units = 0;
    for (int k = 0; k < num_views; ++k) {
        const mv3d_conv_view &w = views[k];
        InView &v = g.v[k];
        v.x = (const char *)w.x_framed; v.w = (const char *)w.w_packed; v.bias = w.bias; v.y = (char *)w.y;
        v.B = w.batch; v.H = w.height; v.W = w.width; v.pad = out_framed != 0;
        v.xblocks = (w.width + 31) / 32; v.strips = (w.height + IN_ROWS - 1) / IN_ROWS;
        const size_t yb = (size_t)w.batch * (w.height + 2 * v.pad) * (w.width + 2 * v.pad) * 128;
        if (yb >= 0xfffffff0u) return MV3D_ERR_INVALID_ARG;
        v.y_bytes = (unsigned)yb;
        g.first[k] = (int)units;
        units += (long)w.batch * v.strips * v.xblocks;
        if (units > 0x7fffffffL) return MV3D_ERR_INVALID_ARG;
    }
    for (int k = num_views; k < 3; ++k) { g.v[k] = g.v[0]; g.first[k] = (int)units; }
    g.first[num_views] = (int)units;
    if (num_views < 3) g.first[3] = (int)units;
    // persistent waves: one workgroup of IN_WAVES waves per CU
    const long want = (units + IN_WAVES - 1) / IN_WAVES;
    unsigned grid = (unsigned)(want < 256 ? want : 256);
#ifdef MV3D_TUNING
    if (getenv("MV3D_CONV_INPUT_GRID") && (unsigned)atoi(getenv("MV3D_CONV_INPUT_GRID")) < grid) grid = (unsigned)atoi(getenv("MV3D_CONV_INPUT_GRID"));
#endif
    hipLaunchKernelGGL(conv3x3_input_kernel<T>, dim3(grid), dim3(IN_WAVES * 64), 0, stream, g);
    return mv3d_launch_status();
}

int mv3d_launch_conv_input_f16(int num_views, const mv3d_conv_view *views, int out_framed, int relu, hipStream_t stream)
{
    return launch_conv_input<_Float16>(num_views, views, out_framed, relu, stream);
}
int mv3d_launch_conv_input_bf16(int num_views, const mv3d_conv_view *views, int out_framed, int relu, hipStream_t stream)
{
    return launch_conv_input<__bf16>(num_views, views, out_framed, relu, stream);
}
