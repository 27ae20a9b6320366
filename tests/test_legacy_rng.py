"""The host-side draws of the library (csrc/legacy_rng.hip) against numpy itself -- numpy's legacy RandomState is a dependency
of the reference (`npr.choice(..., replace=False)` in lib/rpn_msr/anchor_target_layer_tf.py:146-183 and
lib/rpn_msr/proposal_target_layer_tf.py:246-269) that /root/reference does not vendor; the restatement is pinned here:
same permutations, same generator state afterwards, and the batch entry draws exactly what the numpy statement of the two
target layers' draws (train_path.draw_subsamples_host / draw_samples_host) draws.  CPU only: host code of libmv3d_hip.so."""
import ctypes as C

import numpy as np
import numpy.random as npr
import pytest


@pytest.fixture(scope="module")
def L():
    from mv3d_tf_amd import build
    from mv3d_tf_amd._lib import lib
    build.build()
    return lib()


def _state_addr():
    return C.c_void_p(npr.mtrand._rand._bit_generator.ctypes.state_address)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 64, 623, 624, 625, 1000, 21758, 65537, 200001])
def test_permutation_equals_numpy_and_leaves_the_same_state(L, n):
    for seed in (0, 3, 12345):
        for burn in (0, 5, 623):                       # generator positions before / across a block refill
            npr.seed(seed)
            npr.random_sample(burn)
            want = npr.permutation(n)
            after = npr.get_state()
            npr.seed(seed)
            npr.random_sample(burn)
            got = np.full(max(n, 1), -1, np.int32)
            assert L.mv3d_legacy_permutation(_state_addr(), n, got.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(got[:n], want)
            st = npr.get_state()
            assert st[2] == after[2] and np.array_equal(st[1], after[1])
            assert np.array_equal(npr.permutation(11), (npr.set_state(after), npr.permutation(11))[1])


def test_batch_draws_equal_the_numpy_statement(L):
    from mv3d_tf_amd._lib import DrawFrame, DrawParams
    from mv3d_tf_amd.fast_rcnn.config import cfg
    from mv3d_tf_amd.train_path import draw_samples_host, draw_subsamples_host
    rng = np.random.RandomState(1)
    T = cfg.TRAIN
    par = DrawParams(int(T.RPN_BATCHSIZE), int(T.RPN_FG_FRACTION * T.RPN_BATCHSIZE), int(T.BATCH_SIZE),
                     int(np.round(T.FG_FRACTION * T.BATCH_SIZE)))
    # (n_fg, n_bg, n_low, pt_n_fg, pt_n_bg): typical frame, no foreground, few of everything, flood of positives, no candidates
    cases = [(57, 21000, 21400, 40, 1900), (0, 300, 310, 0, 50), (5, 20, 30, 3, 10), (400, 100, 500, 200, 0), (0, 0, 0, 0, 0),
             (33, 97, 128, 32, 96), (1, 1, 1, 1, 1)]
    for trial in range(3):
        B = len(cases)
        heads, frames, keep = [], (DrawFrame * B)(), []
        for b, (n_fg, n_bg, n_low, p_fg, p_bg) in enumerate(cases):
            flags = (rng.random_sample(max(n_fg, 1)) < 0.6).astype(np.uint8)
            keep.append(flags)
            head = np.zeros(32 + max(n_fg, 1), np.uint8)
            head[:16].view(np.int32)[:] = (n_fg + n_bg, n_fg, n_bg, n_low)
            head[32:32 + n_fg] = flags[:n_fg]
            heads.append(head)
            frames[b] = DrawFrame(n_fg, n_bg, n_low, p_fg, p_bg, 0, flags.ctypes.data)
        npr.seed(40 + trial)
        want = []
        for b, c in enumerate(cases):
            dis = draw_subsamples_host(heads[b], lambda: None, 0)
            picks = draw_samples_host(np.array([0, c[3], c[4], 0]))
            want += [np.zeros(0, np.int64) if a is None else a for a in (*dis, *picks)]
        after = npr.get_state()
        npr.seed(40 + trial)
        lists = np.full(sum(len(a) for a in want) + 8, -1, np.int32)
        sizes = (C.c_int32 * (5 * B))()
        scratch = np.zeros(30000, np.int32)
        rc = L.mv3d_draw_training_subsamples(_state_addr(), B, frames, C.byref(par), lists.ctypes.data_as(C.c_void_p), lists.size, sizes,
                                             scratch.ctypes.data_as(C.c_void_p), scratch.size)
        assert rc == 0
        assert list(sizes) == [len(a) for a in want]
        o = 0
        for a in want:
            assert np.array_equal(lists[o:o + len(a)], a)
            o += len(a)
        st = npr.get_state()
        assert st[2] == after[2] and np.array_equal(st[1], after[1])
        # too small a list buffer / scratch is refused, not overrun
        npr.seed(1)
        assert L.mv3d_draw_training_subsamples(_state_addr(), B, frames, C.byref(par), lists.ctypes.data_as(C.c_void_p), 10, sizes,
                                               scratch.ctypes.data_as(C.c_void_p), scratch.size) == 2
        assert L.mv3d_draw_training_subsamples(_state_addr(), B, frames, C.byref(par), lists.ctypes.data_as(C.c_void_p), lists.size, sizes,
                                               scratch.ctypes.data_as(C.c_void_p), 100) == 2


def test_a_replaced_global_bit_generator_is_refused():
    """the C draws write {uint32 key[624]; int pos} at the global generator's state address -- only an MT19937 has
    that layout; any other global bit generator must raise instead of being scribbled over."""
    from mv3d_tf_amd.train_path import _global_mt19937_address
    assert _global_mt19937_address() == npr.mtrand._rand._bit_generator.ctypes.state_address
    if not hasattr(npr, "set_bit_generator"):
        pytest.skip("numpy without set_bit_generator")
    saved = npr.get_bit_generator()
    try:
        npr.set_bit_generator(npr.PCG64(1))
        with pytest.raises(TypeError):
            _global_mt19937_address()
    finally:
        npr.set_bit_generator(saved)
    assert _global_mt19937_address() == saved.ctypes.state_address
