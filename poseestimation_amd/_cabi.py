"""ctypes binding of libmvx.so, the C ABI declared in include/mvx.h.

The library is built in-tree (``make -C poseestimation_amd/csrc`` or ``__graft_entry__.build()``).
There is no fallback: if the library is missing, or a tensor is not a float32 HIP device tensor, the
calls below raise.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmvx.so")

MVX_OK, MVX_ERR_INVALID_ARGUMENT, MVX_ERR_HIP, MVX_ERR_WORKSPACE, MVX_ERR_UNSUPPORTED = range(5)
AGG_SUM, AGG_CONF, AGG_MAX, AGG_SOFTMAX = range(4)
SELECT_NO_UNPROJECT_BRICKS = 1
SELECT_NO_DLT_ROWS = 2
SELECT_NO_SOFT_ARGMAX_RESIDENT = 4
SELECT_SOFT_ARGMAX_HELP_ALWAYS = 8
SELECT_NO_BACKWARD_IN_LDS = 16

_c_float_p = ctypes.c_void_p  # device pointers travel as integers
_i64, _int, _f32, _vp, _sz = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/mvx.h declares
SIGNATURES = {
    "mvx_version": (_int, []),
    "mvx_last_error": (ctypes.c_char_p, []),
    "mvx_set_kernel_selection": (None, [ctypes.c_uint]),
    "mvx_soft_argmax_2d": (_int, [_vp, _i64, _int, _int, _f32, _int, _vp, _vp, _vp]),
    "mvx_triangulate_dlt": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp]),
    "mvx_algebraic_glue_dlt": (_int, [_vp, _vp, _vp, _i64, _int, _int, _f32, _f32, _vp, _vp, _vp, _vp]),
    "mvx_algebraic_forward": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _int, _f32, _f32, _vp, _vp, _vp,
                                     _vp, _vp, _vp]),
    "mvx_project_points": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "mvx_unproject_workspace_bytes": (_sz, [_i64, _int, _int, _int, _int]),
    "mvx_unproject": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _sz,
                             _vp]),
    "mvx_unproject_channels_last": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _vp,
                                           _vp]),
    "mvx_unproject_grid": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _int, _vp, _i64, _int, _int, _int, _int,
                                  _int, _int, _int, _int, _vp, _vp, _sz, _vp]),
    "mvx_unproject_backward_workspace_bytes": (_sz, [_i64, _int, _int, _int, _int, _int, _int, _int, _int]),
    "mvx_unproject_backward": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _vp, _vp,
                                      _vp, _sz, _vp]),
    "mvx_soft_argmax_3d_backward_workspace_bytes": (_sz, [_i64, _int, _int, _int, _int]),
    "mvx_soft_argmax_3d_backward": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _f32, _int, _vp, _vp, _vp, _vp, _sz,
                                           _vp]),
    "mvx_build_coord_volumes": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _f32, _f32, _f32, _int, _vp, _vp]),
    "mvx_soft_argmax_3d_workspace_bytes": (_sz, [_i64, _int, _int, _int, _int]),
    "mvx_soft_argmax_3d": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _f32, _int, _vp, _vp, _vp, _sz, _vp]),
    "mvx_soft_argmax_3d_grid": (_int, [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _int, _i64, _int, _int, _int, _int, _f32, _int,
                                       _vp, _vp, _vp, _sz, _vp]),
    "mvx_soft_argmax_3d_static": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _f32, _int, _vp, _vp, _vp]),
    "mvx_heatmap_argmax": (_int, [_vp, _i64, _int, _int, _f32, _f32, _vp, _vp]),
    "mvx_triangulate_ransac": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, ctypes.c_double, _int, _vp, _vp, _vp]),
    "mvx_draw_view_pairs": (_int, [_vp, _vp, _int, _i64, _vp]),
    "mvx_savgol_filter": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp]),
    "mvx_unity_points": (_int, [_vp, _vp, _i64, _int, _f32, _vp, _vp]),
    "mvx_realtime_tail": (_int, [_vp, _vp, _int, _int, _int, _vp, _int, _vp, _f32, _int, _vp, _vp]),
}

_lib = None


class MvxError(RuntimeError):
    pass


def load():
    """Loads libmvx.so once and declares every entry point; raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MvxError(
            "poseestimation_amd: %s is missing; build it with `make -C %s` (needs hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback for the triangulation kernels." % (LIB_PATH, os.path.join(_HERE, "csrc")))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status, what):
    if status == MVX_OK:
        return
    msg = load().mvx_last_error().decode("utf-8", "replace")
    if status == MVX_ERR_INVALID_ARGUMENT:
        raise ValueError("%s: %s" % (what, msg))
    raise MvxError("%s failed (status %d): %s" % (what, status, msg))


def device_f32(t, name):
    """Returns `t` as a contiguous float32 tensor on its HIP device; raises for anything that is not on a GPU."""
    if not torch.is_tensor(t):
        raise TypeError("%s must be a torch tensor, got %s" % (name, type(t).__name__))
    if not t.is_cuda:
        raise MvxError("%s is on %s: the mvx kernels run on a HIP device only (no CPU path)" % (name, t.device))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def same_device(ref, ref_name, **tensors):
    """Raises unless every tensor given by keyword (None entries are skipped) lives on `ref`'s device.  The kernels
    receive raw addresses and run on `ref`'s GPU: a tensor from another GPU would be read as a foreign pointer
    (the reference's torch ops raise a RuntimeError for mixed devices; MvxError is one)."""
    for name, t in tensors.items():
        if t is not None and t.device != ref.device:
            raise MvxError("%s is on %s but %s is on %s: all tensors of one call must be on the same device"
                           % (name, t.device, ref_name, ref.device))


def needs_grad(*tensors):
    """True when autograd is recording and one of `tensors` requires grad."""
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors)


def forward_only(what, *tensors):
    """For operations (or arguments of operations) that have no backward kernel.  Raises when autograd is recording
    and one of `tensors` requires grad: returning a detached result there would drop the gradient without a word."""
    if needs_grad(*tensors):
        raise MvxError("%s is forward-only (no backward kernel): call it under torch.no_grad() or detach its inputs; "
                       "training through this operation needs the reference's differentiable torch ops" % what)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def raw_stream(device_index):
    """Handle of torch's current stream on device `device_index` as a plain integer (no Stream object built)."""
    return torch._C._cuda_getCurrentRawStream(device_index)
