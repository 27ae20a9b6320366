"""No-GPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/mv3d_hip.h declares; argument validation paths that never
touch a device."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def hiplib():
    from mv3d_tf_amd import build
    build.build()
    from mv3d_tf_amd import _lib
    return _lib


def test_header_symbols_are_exported(hiplib):
    hdr = open(os.path.join(ROOT, "include", "mv3d_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mv3d_[A-Za-z0-9_]+|_nms)\s*\(", hdr))
    assert declared, "no declarations parsed"
    handle = hiplib.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/mv3d_hip.h but not exported"
    assert declared == set(hiplib.EXPORTS)


def test_version_and_status_strings(hiplib):
    L = hiplib.lib()
    assert L.mv3d_version() >= 100
    assert L.mv3d_status_string(0) == b"ok"
    assert b"division" in L.mv3d_status_string(hiplib.ERR_ZERO_DIVISION)


def test_workspace_queries_and_argument_validation(hiplib):
    L = hiplib.lib()
    p = hiplib.ProposalParams(8, 12000, 2000, 375, 1242, 50, 0, 0, 0.7, 5.0)
    assert L.mv3d_proposal_3d_capacity(76, 76, C.byref(p)) == 2000
    assert L.mv3d_proposal_3d_workspace_bytes(1, 76, 76, C.byref(p)) > 188 * 189 // 2 * 512   # column-form NMS tiles
    p2 = hiplib.ProposalParams(8, 6000, 300, 375, 1242, 50, 0, 0, 0.7, 5.0)
    assert L.mv3d_proposal_3d_capacity(76, 76, C.byref(p2)) == 300
    assert L.mv3d_proposal_3d_capacity(4, 4, C.byref(p2)) == 64        # fewer anchors than top-N
    assert L.mv3d_nms_workspace_bytes(6000) >= 94 * 95 // 2 * 512
    assert L.mv3d_nms_workspace_bytes(10 ** 6) == 0                      # above the bitmap limit
    assert L.mv3d_anchor_target_workspace_bytes(76, 76, 8) > 0
    # NULL pointers are refused before any HIP call
    assert L.mv3d_proposal_3d(None, None, 1, 76, 76, None, None, C.byref(p), None, None, None, None, None, None, 0,
                              None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_roi_pool_forward(None, 0.125, 1, 4, 8, 8, 16, 7, 7, None, None, None, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_nms_device(None, 10, 0.7, 0, None, None, None, None, 0, None) == hiplib.ERR_INVALID_ARG
    # the MFMA convolution: NULL / misaligned pointers, channel counts the tiles do not cover, buffers beyond 32-bit offsets
    A = 4096                                                             # a non-NULL, 16-byte aligned "pointer" (never dereferenced)
    assert L.mv3d_conv3x3_f16(None, A, A, A, 1, 8, 8, 64, 64, 1, 0, 1, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_conv3x3_f16(A + 2, A, A, A, 1, 8, 8, 64, 64, 1, 0, 1, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_conv3x3_f16(A, A, A, A, 1, 8, 8, 48, 64, 1, 0, 1, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_conv3x3_f16(A, A, A, A, 1, 8, 8, 64, 96, 1, 0, 1, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_conv3x3_f16(A, A, A, A, 64, 608, 608, 64, 64, 1, 0, 1, None) == hiplib.ERR_INVALID_ARG    # 3 GB of activations
    assert L.mv3d_conv3x3_bf16(A, A, A, A, 1, 8, 8, 64, 96, 1, 0, 1, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_conv3x3_gated_bf16(A, A, A, None, A, 1, 8, 8, 64, 64, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_conv3x3_wgrad_workspace_bytes(2, 76, 76, 512, 512) >= 512 * 9 * 512 * 4
    assert L.mv3d_conv3x3_wgrad_workspace_bytes(2, 76, 76, 48, 512) == 0
    assert L.mv3d_conv3x3_wgrad_bf16(A, A, A, None, 2, 76, 76, 512, 512, 512, A, 16, None) == hiplib.ERR_WORKSPACE
    assert L.mv3d_conv3x3_wgrad_bf16(A, A, A, None, 2, 76, 76, 512, 600, 512, A, 1 << 30, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_conv3x3_pack_bf16(None, A, A, 64, 64, 64, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_maxpool2x2_bwd_bf16(A, A, None, 1, 8, 8, 64, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_maxpool2x2_f16(A, A, 1, 8, 8, 12, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_frame_nhwc_f16(A, A, 1, 8, 8, 9, 8, None) == hiplib.ERR_INVALID_ARG


def test_train_path_object_rejects_bad_arguments_before_any_device_call(hiplib):
    """mv3d_train_path_*: configuration and slot tables are checked on the host (no GPU needed to be refused)"""
    import ctypes as C
    L = hiplib.lib()
    conf, slot, h = hiplib.TrainPathConfig(), hiplib.TrainPathSlot(), C.c_void_p()
    assert L.mv3d_train_path_create(C.byref(conf), 1, C.byref(slot), 1, C.byref(h)) == hiplib.ERR_INVALID_ARG       # batch 0
    conf.batch, conf.H, conf.W, conf.num_classes, conf.proposal_cap, conf.anchor_cap, conf.roi_cap, conf.max_gt = 2, 76, 76, 2, 2000, 512, 128, 64
    conf.draw = hiplib.DrawParams(256, 128, 128, 32)
    assert L.mv3d_train_path_create(C.byref(conf), 1, C.byref(slot), 1, C.byref(h)) == hiplib.ERR_INVALID_ARG       # NULL buffers
    conf.draw = hiplib.DrawParams(256, 128, 256, 32)                                                                 # more rows than roi_cap
    assert L.mv3d_train_path_create(C.byref(conf), 1, C.byref(slot), 1, C.byref(h)) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_train_path_create(C.byref(conf), 0, C.byref(slot), 1, C.byref(h)) == hiplib.ERR_INVALID_ARG
    assert not h.value
    assert L.mv3d_train_path_finish(None, 0, None, None, None) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_train_path_submit(None, 0, *([None] * 10)) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_train_path_configure(None, C.byref(conf)) == hiplib.ERR_INVALID_ARG
    assert L.mv3d_train_path_host_seconds(None, None, None) == hiplib.ERR_INVALID_ARG
    L.mv3d_train_path_destroy(None)


def test_missing_library_fails_loudly(hiplib, monkeypatch, tmp_path):
    monkeypatch.setattr(hiplib, "_lib", None)
    monkeypatch.setattr(hiplib, "LIB_PATH", str(tmp_path / "libmv3d_hip.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        hiplib.lib()


def test_config_mirror():
    from mv3d_tf_amd.fast_rcnn import config
    c = config.cfg
    assert c.TRAIN.RPN_PRE_NMS_TOP_N == 12000 and c.TEST.RPN_POST_NMS_TOP_N in (2000, 300)
    assert c.TRAIN.RPN_MIN_SIZE == 5 and c.TRAIN.RPN_NMS_THRESH == 0.7 and c.RNG_SEED == 3
    with pytest.raises(KeyError):
        config._merge({"TRAIN": {"NOT_A_KEY": 1}}, c)
