"""ctypes binding of libmadnet_hip.so (C-ABI declared in include/madnet_hip.h; tuning hooks in include/madnet_hip_tune.h).

The reference binds its native op with tf.load_op_library (Nets/sharedLayers.py:11-17);
cffi is not available here, stdlib ctypes is.  PyTorch tensors are storage only: every call
passes raw device pointers + sizes + a hipStream_t.

The product path FAILS LOUDLY if the HIP library is missing or no GPU is visible -- there is no
CPU fallback.  (tests/emul builds a *separate* CPU functional emulator of the same kernel
sources; it is only ever loaded explicitly by tests through `Lib(path)`.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MADNET_HIP_LIB: another build of the same library (A/B runs of compiler options); default = the in-tree build
LIB_PATH = os.environ.get("MADNET_HIP_LIB") or os.path.join(_HERE, "libmadnet_hip.so")


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "Hi", "Wi", "Ho", "Wo", "K", "N", "kh", "kw", "stride", "dil", "pad_t", "pad_l",
                 "mode", "w_trans", "in_ld", "out_ld", "mask_ld", "accumulate")] + \
               [("alpha", C.c_float), ("mask_alpha", C.c_float), ("mask_c0", C.c_int32), ("mask_c1", C.c_int32), ("precision", C.c_int32)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("i", C.c_int32 * 27), ("f", C.c_float * 4),
                ("p", C.c_void_p * 12), ("n", C.c_int64)]


(OP_CONV, OP_WGRAD, OP_CORR_FWD, OP_CORR_BWD, OP_WARP_FWD, OP_WARP_BWD, OP_RESIZE_FWD, OP_RESIZE_BWD,
 OP_PAD_REFLECT, OP_LOSS, OP_METRICS, OP_MOMENTUM, OP_COPY_CH, OP_LEAKY_BWD, OP_FILL, OP_BIAS_GRAD,
 OP_WGRAD_PARTIAL, OP_WGRAD_REDUCE, OP_PROXY_LOSS, OP_SUPERVISED_LOSS, OP_ADAM, OP_ADAM_ADVANCE, OP_RESIZE_IMAGE, OP_LEVEL_FRONT, OP_RESERVED_25, OP_PACK_W, OP_CORR_WARP_BWD,
 OP_SHADOW_CAST, OP_WGRAD_STREAM, OP_HEAD_BWD, OP_HEAD_FWD, OP_CONV_PLANES, OP_PLANE_SPLIT, OP_STAMP, OP_CONV_PLANES_BWD, OP_DET_FLUSH, OP_CONV_IMAGE, OP_ALLREDUCE, OP_FETCH_INPUTS) = range(1, 40)


COMM_ID_BYTES = 128            # MH_COMM_ID_BYTES
ALLREDUCE_MAX_BUFS = 8         # MH_ALLREDUCE_MAX_BUFS
FETCH_MAX = 4                  # MH_FETCH_MAX


class InputTable(C.Structure):     # mh_input_table
    _fields_ = [("src", C.c_void_p * FETCH_MAX), ("u8", C.c_int32 * FETCH_MAX)]
OP_JOIN = 0x100
OP_NODEFER = 0x200
MAX_LANES = 5


class WgradSeg(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dst", C.c_void_p), ("size", C.c_int32), ("splits", C.c_int32),
                ("blk0", C.c_int32), ("accumulate", C.c_int32)]

class PackSeg(C.Structure):          # mh_pack_seg
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("taps", C.c_int32), ("K", C.c_int32), ("N", C.c_int32),
                ("planes", C.c_int32), ("blk0", C.c_int32), ("trans", C.c_int32), ("kc16", C.c_int32), ("reserved", C.c_int32)]


class WgradItem(C.Structure):        # mh_wgrad_item
    _fields_ = [("d", ConvDesc), ("inp", C.c_void_p), ("dout", C.c_void_p), ("ws", C.c_void_p), ("db", C.c_void_p),
                ("dout_ld", C.c_int32), ("splits", C.c_int32), ("group_max_m", C.c_int32), ("reserved", C.c_int32)]

class ShadowSeg(C.Structure):        # mh_shadow_seg
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("npix", C.c_int64), ("C", C.c_int32), ("src_ld", C.c_int32),
                ("dst_ld", C.c_int32), ("blk0", C.c_int32)]


class PlaneSeg(C.Structure):         # mh_plane_seg
    _fields_ = [("src", C.c_void_p), ("hi", C.c_void_p), ("lo", C.c_void_p), ("npix", C.c_int64), ("C", C.c_int32), ("src_ld", C.c_int32),
                ("dst_ld", C.c_int32), ("blk0", C.c_int32), ("src2", C.c_void_p), ("C2", C.c_int32), ("src2_ld", C.c_int32)]


class PlanRef(C.Structure):          # mh_plan_ref
    _fields_ = [("ops", C.c_void_p), ("nops", C.c_int32), ("reserved", C.c_int32)]


class HeadBwdDesc(C.Structure):      # mh_head_bwd_desc
    _fields_ = [(n, C.c_int32) for n in ("kind", "B", "H", "W", "N", "Hr", "Wr", "cy", "cx", "Ho", "Wo")] + [("mul", C.c_float)] + \
               [(n, C.c_int32) for n in ("src0_ld", "src1_ld", "dx_ld", "mask_ld", "accumulate_dx")] + [("mask_alpha", C.c_float)]


class WgsLayer(C.Structure):         # mh_wgs_layer
    _fields_ = [("x", C.c_void_p), ("dz", C.c_void_p), ("ws", C.c_void_p), ("db", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("B", "H", "W", "K", "N", "dil", "x_ld", "dz_ld", "ktiles", "ntiles", "splits", "blk0", "stride", "reserved")]


_P = C.c_void_p
_I = C.c_int32
_F = C.c_float
_L = C.c_int64

# name -> (restype, argtypes); every symbol include/madnet_hip.h declares
SIGNATURES = {
    "mh_last_error": (C.c_char_p, []),
    "mh_last_kernel": (C.c_char_p, []),
    "mh_abi_version": (_I, []),
    "mh_planes_kc16": (_I, [_I]),
    "mh_crc32c": (C.c_uint32, [C.c_char_p, _L, C.c_uint32]),
    "mh_device_count": (_I, []),
    "mh_init": (_I, []),
    "mh_tune_conv_tile": (_I, [_I, _I]),
    "mh_tune_conv_thin": (_I, [_I]),
    "mh_tune_conv_patch": (_I, [_I]),
    "mh_tune_wgrad_wgs": (_I, [_I]),
    "mh_tune_corr": (_I, [_I]),
    "mh_tune_corr_row": (_I, [_I]),
    "mh_conv2d": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    "mh_conv2d_wgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _I, _P, _P, _P]),
    "mh_conv2d_wgrad_partial": (_I, [C.POINTER(ConvDesc), _P, _P, _I, _P, C.POINTER(C.c_int32), _P, _P]),
    "mh_corr_warp_bwd": (_I, [_P, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mh_conv2d_wb": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "mh_conv2d_sh": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "mh_pack_weights": (_I, [_P, _I, _I, _P]),
    "mh_pack_bytes": (_L, [_I, _I, _I, _I]),
    "mh_pack32_bytes": (_L, [_I, _I, _I]),
    "mh_conv2d_planes_ok": (_I, [C.POINTER(ConvDesc)]),
    "mh_conv2d_planes": (_I, [C.POINTER(ConvDesc), _P, _P, _I, _P, _P, _P, _P, _P, _I, _P]),
    "mh_plane_split": (_I, [_P, _I, _I, _P]),
    "mh_conv2d_planes_bwd_ok": (_I, [C.POINTER(ConvDesc)]),
    "mh_conv2d_planes_bwd": (_I, [C.POINTER(ConvDesc), _P, _I, _P, _P, _I, _P, _P, _I, _P]),
    "mh_tune_conv_planes": (_I, [_I]),
    "mh_tune_conv_bank": (_I, [_I]),
    "mh_tune_wgrad_image": (_I, [_I]),
    "mh_tune_conv_bank_tile": (_I, [_I]),
    "mh_tune_conv_bank_small": (_I, [_I]),
    "mh_tune_conv_rows": (_I, [_I]),
    "mh_tune_wgrad_target_pct": (_I, [_I]),
    "mh_tune_conv_x3_igemm": (_I, [_I]),
    "mh_conv2d_wgrad_partial_group": (_I, [C.POINTER(WgradItem), _I, _P]),
    "mh_wgrad_reduce": (_I, [_P, _I, _I, _P]),
    "mh_shadow_cast": (_I, [_P, _I, _I, _P]),
    "mh_plans_prepare": (_I, [_I]),
    "mh_plans_run": (_I, [C.POINTER(PlanRef), _I, _P]),
    "mh_conv2d_sh2": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mh_conv2d_sh3": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "mh_conv2d_takes_shadows": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "mh_conv2d_sh4": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mh_level_front_fwd_planes": (_I, [_P, _I, _I, _F, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "mh_level_front_head_fwd": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _F, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "mh_level_front_head_ok": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "mh_conv_image_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _I, _I, _I, _I, _F, _P, _I, _P, _I, _P]),
    "mh_conv_image_ok": (_I, [_I, _I, _I, _I, _I]),
    "mh_conv2d_head": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    "mh_head_bwd": (_I, [C.POINTER(HeadBwdDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mh_wgrad_stream_plan": (_I, [C.POINTER(WgsLayer), _I, _I, _I, C.POINTER(C.c_int32)]),
    "mh_wgrad_stream": (_I, [_P, _I, _I, _I, _I, _P]),
    "mh_tune_wgrad_stream": (_I, [_I]),
    "mh_proxy_ws_floats": (_L, [_I, _I, _I]),
    "mh_proxy_loss": (_I, [_P, _P, _P, _P, _P, _F, _F, _I, _I, _I, _P]),
    "mh_supervised_loss": (_I, [_P, _P, _P, _P, _P, _F, _F, _F, _I, _I, _I, _P]),
    "mh_adam": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _P]),
    "mh_adam_advance": (_I, [_P, _F, _F, _P]),
    "mh_corr_fwd": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mh_corr_fwd_prec": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mh_level_front_fwd": (_I, [_P, _I, _I, _F, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mh_corr_bwd": (_I, [_P, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mh_corr_bwd_prec": (_I, [_P, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mh_shift_corr": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mh_shift_corr_grad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mh_warp_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mh_warp_bwd": (_I, [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "mh_resize_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "mh_resize_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "mh_resize_image_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mh_resize_image_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mh_bilinear_sampler_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mh_bilinear_sampler_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mh_u8_to_f32": (_I, [_P, _P, _L, _P]),
    "mh_fetch_inputs": (_I, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _I, _P]),
    "mh_host_device_pointer": (_I, [_P, C.POINTER(C.c_void_p)]),
    "mh_pad_reflect": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "mh_loss_ws_floats": (_L, [_I, _I, _I]),
    "mh_reprojection_loss": (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _P]),
    "mh_reprojection_loss_phase": (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P]),
    "mh_metrics_ws_floats": (_L, [_I, _I, _I]),
    "mh_metrics": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _P]),
    "mh_momentum": (_I, [_P, _P, _P, _L, _F, _F, _F, _P]),
    "mh_copy_channels": (_I, [_P, _I, _P, _I, _L, _I, _F, _I, _P]),
    "mh_leaky_bwd": (_I, [_P, _I, _P, _I, _L, _I, _F, _P]),
    "mh_fill": (_I, [_P, _L, _F, _P]),
    "mh_stamp": (_I, [_P, _P]),
    "mh_deterministic_add": (_I, [_P, C.c_int64, _P]),
    "mh_deterministic_remove": (_I, [_P]),
    "mh_deterministic_ranges": (_I, []),
    "mh_deterministic_overflow": (_I, []),
    "mh_det_flush": (_I, [_P, _P, C.c_int64, _P]),
    "mh_stamp_rate_khz": (_L, []),
    "mh_bias_grad": (_I, [_P, _I, _L, _I, _P, _P]),
    "mh_bias_grad_blocks": (_I, [_L, _I]),
    "mh_comm_available": (_I, []),
    "mh_comm_unique_id": (_I, [_P]),
    "mh_comm_init": (_I, [_P, _I, _I, C.POINTER(C.c_void_p)]),
    "mh_comm_destroy": (_I, [_P]),
    "mh_comm_info": (_I, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mh_allreduce_sum": (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _I, _P, _P]),
    "mh_bias_grad_partial": (_I, [_P, _I, _L, _I, _P, _I, _P]),
    "mh_plan_run": (_I, [C.POINTER(Op), _I, _P]),
    "mh_graph_begin": (_I, [_P]),
    "mh_graph_end": (_I, [_P, C.POINTER(_P)]),
    "mh_graph_launch": (_I, [_P, _P]),
    "mh_graph_destroy": (_I, [_P]),
    "mh_event_create": (_I, [C.POINTER(_P)]),
    "mh_event_record": (_I, [_P, _P]),
    "mh_event_elapsed_ms": (_I, [_P, _P, C.POINTER(_F)]),
    "mh_event_destroy": (_I, [_P]),
    "mh_stream_sync": (_I, [_P]),
}
_NO_STATUS = {"mh_comm_available", "mh_deterministic_overflow", "mh_bias_grad_blocks", "mh_tune_conv_bank_small", "mh_conv_image_ok", "mh_level_front_head_ok", "mh_deterministic_ranges", "mh_planes_kc16", "mh_conv2d_planes_bwd_ok", "mh_stamp_rate_khz", "mh_pack32_bytes", "mh_conv2d_planes_ok", "mh_tune_conv_planes", "mh_tune_wgrad_target_pct", "mh_tune_wgrad_image", "mh_conv2d_takes_shadows", "mh_tune_conv_bank_tile", "mh_tune_conv_rows", "mh_last_error", "mh_last_kernel", "mh_tune_conv_bank", "mh_pack_bytes", "mh_abi_version", "mh_tune_conv_patch", "mh_crc32c", "mh_device_count", "mh_loss_ws_floats", "mh_metrics_ws_floats", "mh_proxy_ws_floats"}


class MadnetHipError(RuntimeError):
    pass


class Lib(object):
    """Typed view of the shared library; status-returning calls raise MadnetHipError."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise MadnetHipError(
                "HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C real-time-self-adaptive-deep-stereo_amd/csrc`). There is no CPU fallback." % path)
        self.path = path
        self._dll = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self._dll, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_raw_" + name, fn)
            if name in _NO_STATUS:
                setattr(self, name[3:], fn)
            else:
                setattr(self, name[3:], self._checked(name, fn))
        self._inited = set()

    def ensure_init(self, device_index=None):
        """mh_init (the > 64 KiB dynamic-LDS opt-in of every kernel instantiation, the plan executor's side streams) applies to the CURRENT device:
        once per device a process uses (engines call it with their device current)"""
        if device_index not in self._inited:
            self.init()
            self._inited.add(device_index)

    def _checked(self, name, fn):
        def call(*a):
            rc = fn(*a)
            if rc != 0:
                msg = self._dll.mh_last_error()
                raise MadnetHipError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))
            return 0
        call.__name__ = name
        return call


_lib = None


def lib():
    """The product library.  Raises if it is not built or no HIP device is visible."""
    global _lib
    if _lib is None:
        l = Lib(LIB_PATH)
        n = l.device_count()
        if n <= 0:
            raise MadnetHipError("libmadnet_hip.so loaded but no HIP device is visible (mh_device_count=%d); "
                                 "the MI355X path has no CPU fallback" % n)
        l.ensure_init()
        _lib = l
    return _lib
