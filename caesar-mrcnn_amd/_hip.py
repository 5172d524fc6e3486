"""ctypes binding of libmrcnn_hip.so (the C-ABI declared in include/mrcnn_hip.h).

This is the only way the product path reaches the GPU: there is no eager/PyTorch fallback.  If the
library is missing or a call returns a non-zero status a ``HipPathError`` is raised.
PyTorch is used by the callers only as the owner of device memory and of the current HIP stream.
"""
import ctypes as C
import os
import threading

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libmrcnn_hip.so")


class HipPathError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad_t", "pad_l", "OH", "OW",
        "act", "res_mode", "out_mode", "cmod")] + [
        ("out_n_stride", C.c_int64), ("out_h_stride", C.c_int64), ("out_w_stride", C.c_int64)]


class ConvProblem(C.Structure):
    """mrcnn_conv_problem (include/mrcnn_hip.h): one convolution of a mrcnn_conv2d_fwd_multi launch."""
    _fields_ = [("d", ConvDesc)] + [(n, C.c_void_p) for n in ("x", "w", "bias", "scale", "shift", "res", "out", "z_out")]


class WgradProblem(C.Structure):
    """mrcnn_wgrad_problem (include/mrcnn_hip.h)."""
    _fields_ = [("d", ConvDesc), ("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("accumulate", C.c_int32)]


class RoiAlignDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("R", C.c_int32), ("P", C.c_int32), ("C", C.c_int32),
                ("H", C.c_int32 * 4), ("W", C.c_int32 * 4), ("image_area", C.c_float)]


class ProposalDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("A", C.c_int32), ("pre_nms_limit", C.c_int32),
                ("proposal_count", C.c_int32), ("nms_threshold", C.c_float), ("std_dev", C.c_float * 4)]


class DetTargetDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "R", "G", "T", "MH", "MW", "mask_h", "mask_w",
                                          "positive_count")] + [
        ("negative_ratio_r", C.c_float), ("bbox_std_dev", C.c_float * 4), ("use_mini_mask", C.c_int32)]


class RpnTargetDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("A", C.c_int32), ("G", C.c_int32), ("n_train", C.c_int32),
                ("bbox_std_dev", C.c_double * 4)]


class WinoGroup(C.Structure):
    """mrcnn_wino_group (include/mrcnn_hip.h): a group of Winograd tiles of one map."""
    _fields_ = [("oth", C.c_int), ("otw", C.c_int), ("th_n", C.c_int), ("tw_n", C.c_int), ("oh0", C.c_int), ("ow0", C.c_int)]


class WinoFuse(C.Structure):
    """mrcnn_wino_fuse (include/mrcnn_hip.h): the output transform fused into mrcnn_winograd_gemm_fused."""
    _fields_ = [(n, C.c_int32) for n in ("mode", "N", "H", "W", "act")] + [("g", WinoGroup)] + \
               [(n, C.c_void_p) for n in ("out", "z", "bias", "scale", "shift", "below_out", "below_z", "mean", "rstd", "dgamma", "dbeta",
                                          "dbias", "counters")]


class BwdEpilogue(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("out", "z", "scale", "mean", "rstd", "dgamma", "dbeta", "dbias")] + [
        ("act", C.c_int32), ("dy", C.c_void_p)]


class BwdEpilogueH16(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("out", "z", "scale", "mean", "rstd", "dgamma", "dbeta", "dbias")] + [
        ("act", C.c_int32), ("dy", C.c_void_p), ("grad_multiplier", C.c_float)]


class DetectionDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("R", C.c_int32), ("C", C.c_int32), ("max_instances", C.c_int32),
                ("min_confidence", C.c_float), ("nms_threshold", C.c_float), ("bbox_std_dev", C.c_float * 4)]


class LossDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "A", "T", "C", "mask_h", "mask_w", "max_rpn_pos",
                                          "mask_loss_dice")] + [("w", C.c_float * 5)]


ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
RES_NONE, RES_SAME, RES_UP2 = 0, 1, 2
OUT_NHWC, OUT_DECONV2 = 0, 1

_P = C.c_void_p
_SIGNATURES = {
    "mrcnn_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc)] + [_P] * 9),
    "mrcnn_conv2d_fwd_ws": (C.c_int, [C.POINTER(ConvDesc)] + [_P] * 9 + [C.c_size_t, _P]),
    "mrcnn_conv2d_fwd_workspace": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "mrcnn_conv2d_fwd_multi_workspace": (C.c_size_t, [C.POINTER(ConvProblem), C.c_int]),
    "mrcnn_conv2d_fwd_multi": (C.c_int, [C.POINTER(ConvProblem), C.c_int, _P, C.c_size_t, _P]),
    "mrcnn_conv2d_wgrad_multi_workspace": (C.c_size_t, [C.POINTER(WgradProblem), C.c_int]),
    "mrcnn_conv2d_wgrad_multi": (C.c_int, [C.POINTER(WgradProblem), C.c_int, _P, C.c_size_t, _P]),
    "mrcnn_conv2d_wgrad": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, C.c_size_t, C.c_int, _P]),
    "mrcnn_conv2d_wgrad_workspace": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "mrcnn_weight_flip_transpose": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_weight_flip_transpose_batched": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "mrcnn_bn_fold": (C.c_int, [_P, _P, _P, _P, C.c_float, _P, _P, _P, C.c_int64, _P]),
    "mrcnn_epilogue_bwd": (C.c_int, [_P] * 11 + [C.c_int64, C.c_int, C.c_int, _P]),
    "mrcnn_maxpool3x3s2_fwd": (C.c_int, [_P, _P, _P] + [C.c_int] * 8 + [_P]),
    "mrcnn_maxpool3x3s2_bwd": (C.c_int, [_P, _P, _P] + [C.c_int] * 6 + [_P]),
    "mrcnn_subsample2_fwd": (C.c_int, [_P, _P] + [C.c_int] * 4 + [_P]),
    "mrcnn_subsample2_bwd_acc": (C.c_int, [_P, _P] + [C.c_int] * 4 + [_P]),
    "mrcnn_upsample2_bwd": (C.c_int, [_P, _P] + [C.c_int] * 5 + [_P]),
    "mrcnn_pixel_unshuffle2": (C.c_int, [_P, _P] + [C.c_int] * 4 + [_P]),
    "mrcnn_mask_out_bwd": (C.c_int, [_P] * 8 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_copy2d": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_size_t, _P]),
    "mrcnn_fill_zero": (C.c_int, [_P, C.c_size_t, _P]),
    "mrcnn_unpack_mask_bits": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_unmold_masks_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mrcnn_unmold_masks": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P,
                                     C.c_size_t, _P]),
    "mrcnn_fits_workspace": (C.c_size_t, [C.c_int, C.c_int]),
    "mrcnn_fits_to_rgb": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _P, _P, C.c_size_t, _P]),
    "mrcnn_mold_image_u8": (C.c_int, [_P] + [C.c_int] * 9 + [C.POINTER(C.c_double), _P, _P, C.c_size_t, _P]),
    "mrcnn_add_inplace": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mrcnn_softmax_rows": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "mrcnn_roialign_fwd": (C.c_int, [C.POINTER(RoiAlignDesc)] + [_P] * 8),
    "mrcnn_roialign_bwd": (C.c_int, [C.POINTER(RoiAlignDesc)] + [_P] * 7),
    "mrcnn_roialign_bwd_gather": (C.c_int, [C.POINTER(RoiAlignDesc)] + [_P] * 7),
    "mrcnn_roialign_fwd_h16": (C.c_int, [C.POINTER(RoiAlignDesc), C.c_int] + [_P] * 7),
    "mrcnn_roialign_bwd_h16": (C.c_int, [C.POINTER(RoiAlignDesc), C.c_int, _P, _P, C.c_float] + [_P] * 5),
    "mrcnn_proposal_workspace": (C.c_size_t, [C.POINTER(ProposalDesc)]),
    "mrcnn_proposal_status_offset": (C.c_size_t, [C.POINTER(ProposalDesc), _P, C.POINTER(C.c_size_t)]),
    "mrcnn_proposal_fwd": (C.c_int, [C.POINTER(ProposalDesc)] + [_P] * 8 + [C.c_size_t, _P]),
    "mrcnn_detection_targets": (C.c_int, [C.POINTER(DetTargetDesc)] + [_P] * 12),
    "mrcnn_conv2d_fwd_h16": (C.c_int, [C.POINTER(ConvDesc), C.c_int] + [_P] * 8),
    "mrcnn_conv2d_fwd_h16_res": (C.c_int, [C.POINTER(ConvDesc), C.c_int] + [_P] * 9),
    "mrcnn_conv2d_fwd_h16_supported": (C.c_int, [C.POINTER(ConvDesc), C.c_int]),
    "mrcnn_conv2d_dgrad_ep_h16": (C.c_int, [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, _P, C.POINTER(BwdEpilogueH16), _P]),
    "mrcnn_conv2d_dgrad_ep": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, C.POINTER(BwdEpilogue), _P, C.c_size_t, _P]),
    "mrcnn_conv2d_wgrad_h16_workspace": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "mrcnn_conv2d_wgrad_h16": (C.c_int, [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, _P, C.c_size_t, C.c_int, C.c_float, _P]),
    "mrcnn_mask_out_fwd_h16": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "mrcnn_mask_out_bwd_h16": (C.c_int, [C.c_int] + [_P] * 8 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "mrcnn_weights_to_h16": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_weights_to_h16_batched": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_cast_to_h16": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "mrcnn_epilogue_bwd_h16": (C.c_int, [C.c_int] + [_P] * 10 + [C.c_int64, C.c_int, C.c_int, C.c_float, _P]),
    "mrcnn_epilogue_bwd_h16_dy": (C.c_int, [C.c_int] + [_P] * 11 + [C.c_int64, C.c_int, C.c_int, C.c_float, _P]),
    "mrcnn_cast_from_h16": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "mrcnn_axpy_from_h16": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "mrcnn_rpn_targets_workspace": (C.c_size_t, [C.POINTER(RpnTargetDesc)]),
    "mrcnn_rpn_targets": (C.c_int, [C.POINTER(RpnTargetDesc)] + [_P] * 7 + [C.c_size_t, _P]),
    "mrcnn_detection_workspace": (C.c_size_t, [C.POINTER(DetectionDesc)]),
    "mrcnn_detection_fwd": (C.c_int, [C.POINTER(DetectionDesc)] + [_P] * 6 + [C.c_size_t, _P]),
    "mrcnn_losses_workspace": (C.c_size_t, [C.POINTER(LossDesc)]),
    "mrcnn_losses_fwd_bwd": (C.c_int, [C.POINTER(LossDesc)] + [_P] * 18 + [C.c_size_t, _P]),
    "mrcnn_grad_prepare": (C.c_int, [_P, _P, C.c_float, _P, C.c_int64, _P, _P]),
    "mrcnn_sumsq": (C.c_int, [_P, C.c_int64, _P, _P]),
    "mrcnn_sgd_momentum": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P, C.c_int64, _P]),
    "mrcnn_sgd_momentum_guarded": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P, C.c_int64, _P, _P]),
    "mrcnn_winograd_buffer_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mrcnn_winograd_weights": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_winograd_input": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_gemm_batched_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_winograd_gemm": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_winograd_output": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_winograd_group_floats": (C.c_size_t, [C.POINTER(WinoGroup), C.c_int, C.c_int]),
    "mrcnn_winograd_input_g": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(WinoGroup), _P]),
    "mrcnn_winograd_weights_g": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_winograd_output_g": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(WinoGroup), _P]),
    "mrcnn_winograd_output_bwd_g": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(WinoGroup), _P]),
    "mrcnn_winograd_output_bwd_zmask_g": (C.c_int, [_P] * 10 + [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(WinoGroup), _P]),
    "mrcnn_winograd_gemm_fused": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(WinoFuse), _P]),
    "mrcnn_winograd_dy_g": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(WinoGroup), _P]),
    "mrcnn_winograd_dw_g": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_deconv2x2_gemm": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_winograd_dy": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_winograd_dw": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_winograd_output_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mrcnn_allreduce_load": (C.c_int, [C.c_char_p]),
    "mrcnn_allreduce_unique_id": (C.c_int, [_P]),
    "mrcnn_allreduce_init": (C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_int]),
    "mrcnn_allreduce_scratch": (C.c_size_t, [C.c_int, C.c_int64, C.c_int]),
    "mrcnn_allreduce_grad": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int, _P, C.c_size_t, _P]),
    "mrcnn_allreduce_direct_plan": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, _P, _P, _P, _P]),
    "mrcnn_allreduce_direct_simulate": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int64, C.c_int64, _P]),
    "mrcnn_allreduce_destroy": (C.c_int, [_P]),
    "mrcnn_allreduce_last_error": (C.c_char_p, []),
    "mrcnn_tuning_set": (C.c_int, [C.c_char_p, C.c_longlong]),
    "mrcnn_hip_version": (C.c_char_p, []),
}

_lib = None
_proxy = None
_tape = None          # a list while a step is being recorded (engine.step_taped), else None
_tape_owner = None    # the thread that opened it: calls from any other thread (loaders, another engine) are not part of the step

# entry points that launch nothing (pure queries / process state): never recorded
# (mrcnn_allreduce_grad is appended by its caller, parallel.RcclComm.reduce: the other mrcnn_allreduce_* calls are set-up)
_NO_RECORD = ("_workspace", "_supported", "_status_offset", "_version", "mrcnn_allreduce_", "_scratch", "_floats")


class _LibProxy(object):
    """What lib() hands out: the ctypes library with every launching entry point wrapped so that, while a tape is open,
    the call (function, arguments) is appended to it after it ran.  Replaying the tape issues exactly the same launches
    without re-running the Python that built their arguments (descriptors, workspace look-ups, shape logic): a step's
    ~700-1000 launches cost 14-21 ms of Python when issued from the engine and ~2-3 ms from the tape."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        real = getattr(self._real, name)
        if any(k in name for k in _NO_RECORD):
            f = real
        else:
            def f(*args, _real=real):
                rc = _real(*args)
                t = _tape
                if t is not None and not rc and _tape_owner == threading.get_ident():
                    t.append((_real, args))     # (a refusal -- MRCNN_ERR_UNSUPPORTED: the caller falls back -- launched nothing)
                return rc
        setattr(self, name, f)              # next look-up skips __getattr__
        return f


def tape_begin():
    global _tape, _tape_owner
    assert _tape is None, "a launch tape is already being recorded"
    _tape, _tape_owner = [], threading.get_ident()
    return _tape


def tape_end():
    global _tape, _tape_owner
    t, _tape, _tape_owner = _tape, None, None
    return t


def tape_replay(tape):
    """Issue the recorded calls again.  C entry points return a status (0 = ok); torch stream / event methods return None."""
    for f, a in tape:
        rc = f(*a)
        if rc:
            raise HipPathError("replayed launch %s failed with status %r" % (getattr(f, "__name__", f), rc))


def ev_record(stream):
    """A new event recorded on `stream` (torch objects); part of the tape while one is open."""
    import torch
    ev = torch.cuda.Event()
    ev.record(stream)
    if _tape is not None:
        _tape.append((ev.record, (stream,)))
    return ev


def ev_wait(stream, ev):
    stream.wait_event(ev)
    if _tape is not None:
        _tape.append((stream.wait_event, (ev,)))


def stream_wait(a, b):
    """a waits for everything enqueued on b so far."""
    a.wait_stream(b)
    if _tape is not None:
        _tape.append((a.wait_stream, (b,)))


def exported_symbols():
    """Names every build of the library must export (mirrors include/mrcnn_hip.h)."""
    return sorted(_SIGNATURES)


def lib():
    """Load (once) and return the kernel library (behind the recording proxy); never falls back to anything else."""
    global _lib, _proxy
    if _lib is None:
        why = ""
        if not os.path.exists(LIB_PATH):
            # not built yet (fresh checkout on a box with the toolchain): build once, in-tree; still no fallback.
            # build() takes a file lock and links to a temporary name, so the ranks of a torchrun job that all arrive
            # here together neither compile twice nor load a half-written library.
            try:
                from . import build as _build
                _build.build(verbose=False)
            except Exception as e:
                why = " Building it here failed: %s" % (e,)
        if not os.path.exists(LIB_PATH):
            raise HipPathError(
                "libmrcnn_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'`; there is no CPU fallback for the hot path.%s" % (LIB_PATH, why))
        # torch first: its wheel carries its own HIP runtime, and the streams / device pointers we are handed belong
        # to that one.  Loaded before torch, this library would pull in the system runtime instead and every launch
        # on a torch stream would fail (seen as status -2 from the first kernel when build() and smoke() share a process).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)   # AttributeError here == symbol missing == broken build
            fn.restype = res
            fn.argtypes = args
        _lib = l
        _proxy = _LibProxy(l)
    return _proxy


def check(status, what):
    if status != 0:
        raise HipPathError("%s failed with status %d" % (what, status))


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    return None if t is None else t.data_ptr()


_dev_index = None


def set_device_index(index):
    """Device whose current stream the launches go to (set by the engine; default: torch's current device
    at the first launch)."""
    global _dev_index
    _dev_index = int(index)


def current_stream():
    """Raw hipStream_t of torch's current stream on the engine's device (the C-level getter: the Python
    `torch.cuda.current_stream()` object costs ~7 us per call, two per launch)."""
    import torch
    global _dev_index
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_dev_index)
