// Host side of the library: the random subsamplings of the two target layers, drawn in C on numpy's OWN global generator.
//
// The reference draws its fg / bg subsamples with `npr.choice(inds, size=k, replace=False)` on the numpy GLOBAL legacy
// RandomState (lib/rpn_msr/anchor_target_layer_tf.py:146-159,178-183; lib/rpn_msr/proposal_target_layer_tf.py:246-269);
// on a legacy RandomState that is `inds[permutation(len(inds))[:k]]`, and the draws are part of the parity contract (the
// kernels take the permutations as index lists).  numpy's C loop does it at ~4-14 ns per shuffled element behind Python
// calls; a training frame shuffles ~44 k elements (the ~21 k background anchors twice).  This file restates that loop --
//   permutation(n)  = arange(n) shuffled by  for i = n-1 .. 1: j = random_interval(i); swap(x[i], x[j])
//                     (numpy/random/mtrand.pyx RandomState.permutation -> shuffle -> _shuffle_raw)
//   random_interval = smallest all-ones mask >= max, then 32-bit draws & mask until <= max
//                     (numpy/random/src/distributions/distributions.c random_interval)
//   32-bit draw     = MT19937 genrand with tempering (numpy/random/src/mt19937/mt19937.h; the public algorithm of
//                     Matsumoto & Nishimura 1998)
// -- directly on the memory of numpy's generator state ({uint32 key[624]; int pos}: `bit_generator.ctypes.state_address`),
// so the global stream stays bit-compatible with every other user of it without exporting / importing the state.
// numpy is a dependency of the reference that is absent from /root/reference (version unpinned upstream; 2.2.6 here): the
// restatement is pinned against numpy itself in tests/test_legacy_rng.py (same permutations, same generator state after).
#include <stdint.h>
#include <string.h>
#include "../../include/mv3d_hip.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
struct MtState { uint32_t key[MT_N]; int pos; };

inline void mt_gen(MtState *s)
{
    uint32_t *k = s->key;
    int i;
    for (i = 0; i < MT_N - MT_M; ++i) {
        const uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < MT_N - 1; ++i) {
        const uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    const uint32_t y = (k[MT_N - 1] & 0x80000000u) | (k[0] & 0x7fffffffu);
    k[MT_N - 1] = k[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    s->pos = 0;
}

inline uint32_t mt_next(MtState *s)
{
    if (s->pos == MT_N) mt_gen(s);
    uint32_t y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// RandomState.permutation(n) into x[0 .. n).  The reference loop per element is "draw until (draw & mask) <= i, swap": its
// rejection branch (taken ~1 in 4 draws) is unpredictable, so the loop is written per DRAW instead and branch-free: a rejected
// draw swaps x[i] with itself and does not advance i.  Same draws, same swaps, same generator state afterwards.
// Two more things keep the loop-carried chain short (it is what bounds the path on fresh frames: ~45 k shuffled elements per
// frame on ONE generator): the mask only changes when i crosses a power of two, so the loop runs segment by segment with the
// mask loop-invariant (the chain through i is then compare + subtract), and the tempering of a block of generator words is
// done ahead of the shuffle in a loop of its own that the compiler vectorises.
void permutation(MtState *s, int32_t n, int32_t *x)
{
    for (int32_t i = 0; i < n; ++i) x[i] = i;
    int32_t i = n - 1;
    int pos = s->pos;
    uint32_t t[MT_N];
    int t_lo = MT_N;                                                 // t[t_lo .. MT_N) = tempered key[t_lo .. MT_N) of the current block
    while (i >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);          // smallest all-ones mask >= i (i >= 1)
        const int32_t lo = (int32_t)(mask >> 1);                     // the mask serves i in (lo, mask]
        while (i > lo) {
            if (pos == MT_N) { mt_gen(s); pos = 0; t_lo = MT_N; }
            if (t_lo > pos) {
                for (int k = pos; k < MT_N; ++k) {
                    uint32_t y = s->key[k];
                    y ^= (y >> 11);
                    y ^= (y << 7) & 0x9d2c5680u;
                    y ^= (y << 15) & 0xefc60000u;
                    y ^= (y >> 18);
                    t[k] = y;
                }
                t_lo = pos;
            }
            while (pos < MT_N && i > lo) {
                const uint32_t j = t[pos++] & mask;
                const bool ok = j <= (uint32_t)i;
                const uint32_t jj = ok ? j : (uint32_t)i;
                const int32_t a = x[i], b = x[jj];
                x[i] = b; x[jj] = a;
                i -= ok ? 1 : 0;
            }
        }
    }
    s->pos = pos;
}

}  // namespace

extern "C" int mv3d_legacy_permutation(void *mt19937_state, int32_t n, int32_t *out)
{
    if (!mt19937_state || n < 0 || (n > 0 && !out)) return MV3D_ERR_INVALID_ARG;
    MtState *s = (MtState *)mt19937_state;
    if (s->pos < 0 || s->pos > MT_N) return MV3D_ERR_INVALID_ARG;
    permutation(s, n, out);
    return MV3D_OK;
}

extern "C" int mv3d_draw_training_subsamples(void *mt19937_state, int batch, const mv3d_draw_frame *frames,
                                             const mv3d_draw_params *par, int32_t *lists, size_t lists_cap, int32_t *sizes,
                                             int32_t *scratch, size_t scratch_cap)
{
    if (!mt19937_state || batch <= 0 || !frames || !par || !lists || !sizes || !scratch) return MV3D_ERR_INVALID_ARG;
    MtState *s = (MtState *)mt19937_state;
    if (s->pos < 0 || s->pos > MT_N) return MV3D_ERR_INVALID_ARG;
    size_t o = 0;
    // the first `take` entries of permutation(n), appended to the lists; permutation(n) is drawn whenever `draw`
    auto emit = [&](int32_t n, int32_t take, bool draw, int32_t *size_out) -> bool {
        *size_out = 0;
        if (!draw) return true;
        if (n < 0 || (size_t)n > scratch_cap || take < 0 || take > n || o + (size_t)take > lists_cap) return false;
        if (take == n) {                        // (the whole permutation: shuffle in place)
            permutation(s, n, lists + o);
        } else {
            permutation(s, n, scratch);
            memcpy(lists + o, scratch, (size_t)take * sizeof(int32_t));
        }
        *size_out = take;
        o += (size_t)take;
        return true;
    };
    for (int b = 0; b < batch; ++b) {
        const mv3d_draw_frame &f = frames[b];
        int32_t *sz = sizes + 5 * b;
        if (f.n_fg < 0 || f.n_bg < 0 || f.n_low < 0 || f.pt_n_fg < 0 || f.pt_n_bg < 0 || (f.n_fg > 0 && !f.fg_alive))
            return MV3D_ERR_INVALID_ARG;
        // ---- anchor_target_layer_tf.py:146-159: disable all but num_fg foreground / num_bg background anchors
        const int32_t num_fg = par->rpn_num_fg;
        const int32_t *dis_fg = lists + o;
        if (!emit(f.n_fg, f.n_fg - num_fg, f.n_fg > num_fg, &sz[0])) return MV3D_ERR_WORKSPACE;
        const int32_t num_bg = par->rpn_batchsize - (f.n_fg < num_fg ? f.n_fg : num_fg);
        if (!emit(f.n_bg, f.n_bg - num_bg, f.n_bg > num_bg, &sz[1])) return MV3D_ERR_WORKSPACE;
        // ---- :176-183: after the relabel, positives that survive = foreground flags still alive and not disabled above
        int32_t n_pos = 0;
        if (f.n_fg > 0) {
            if ((size_t)f.n_fg > scratch_cap) return MV3D_ERR_WORKSPACE;
            uint8_t *alive = (uint8_t *)scratch;                       // (n_fg bytes of the scratch, free between permutations)
            for (int32_t i = 0; i < f.n_fg; ++i) alive[i] = f.fg_alive[i] ? 1 : 0;
            for (int32_t i = 0; i < sz[0]; ++i) alive[dis_fg[i]] = 0;
            for (int32_t i = 0; i < f.n_fg; ++i) n_pos += alive[i];
        }
        const int32_t num_bg2 = par->rpn_batchsize - n_pos;
        if (!emit(f.n_low, f.n_low - num_bg2, f.n_low > num_bg2, &sz[2])) return MV3D_ERR_WORKSPACE;
        // ---- proposal_target_layer_tf.py:246-269
        const int32_t fg_n = par->roi_fg_max < f.pt_n_fg ? par->roi_fg_max : f.pt_n_fg;
        if (!emit(f.pt_n_fg, fg_n, f.pt_n_fg > 0, &sz[3])) return MV3D_ERR_WORKSPACE;
        const int32_t bg_room = par->rois_per_image - fg_n;
        const int32_t bg_n = bg_room < f.pt_n_bg ? bg_room : f.pt_n_bg;
        if (!emit(f.pt_n_bg, bg_n < 0 ? 0 : bg_n, f.pt_n_bg > 0, &sz[4])) return MV3D_ERR_WORKSPACE;
    }
    return MV3D_OK;
}
