"""ctypes binding of libm2t.so (include/m2t.h).

The product path has NO fallback: if the HIP library is missing, or a call fails, this
module raises.  Build it with ``python -m m2trans_amd.build`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libm2t.so")

F32, BF16 = 0, 1

_vp, _i, _f, _d, _ll = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_longlong

# name -> (restype, argtypes); must list every symbol include/m2t.h declares
SIGNATURES = {
    "m2t_version": (_i, []),
    "m2t_last_error_string": (C.c_char_p, []),
    "m2t_plan_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i]),
    "m2t_plan_destroy": (None, [_vp]),
    "m2t_plan_query": (_ll, [_vp, C.c_char_p]),
    "m2t_set_option": (_i, [_vp, C.c_char_p, _ll]),
    "m2t_stream_wait_bucket": (_i, [_vp, _i, _vp]),
    "m2t_plan_init_workspace": (_i, [_vp, _vp, _vp]),
    "m2t_forward": (_i, [_vp, _vp, _vp, _vp, _f, _i, _vp, _vp]),
    "m2t_l1_loss": (_i, [_vp, _vp, _f, _d, _f, _vp, _vp, _vp]),
    "m2t_l1_loss_deferred": (_i, [_vp, _vp, _f, _d, _f, _vp, _vp, _vp]),
    "m2t_set_output_grad": (_i, [_vp, _vp, _f, _vp, _vp]),
    "m2t_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "m2t_adam_step": (_i, [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _i, _f, _vp]),
    "m2t_profile_enable": (_i, [C.c_ulonglong]),
    "m2t_profile_read": (_i, [_i, C.POINTER(_d), C.POINTER(_ll)]),
    "m2t_profile_sample_every": (_i, [_i]),
    "m2t_swin_create": (_i, [C.POINTER(_vp), _i, _i]),
    "m2t_swin_destroy": (None, [_vp]),
    "m2t_swin_query": (_ll, [_vp, C.c_char_p]),
    "m2t_swin_param_name": (C.c_char_p, [_vp, _i]),
    "m2t_swin_load_weights": (_i, [_vp, _vp, _vp, _vp]),
    "m2t_swin_encode": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_i), _i, _vp, _vp, _vp]),
    "m2t_swin_encode_pair": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, C.POINTER(_i), _i, _vp, _vp, _vp]),
    "m2t_semantic_loss": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "m2t_bicubic_resize": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "m2t_text_create": (_i, [C.POINTER(_vp), _i, _i, _i]),
    "m2t_text_destroy": (None, [_vp]),
    "m2t_text_query": (_ll, [_vp, C.c_char_p]),
    "m2t_text_param_name": (C.c_char_p, [_vp, _i]),
    "m2t_text_load_weights": (_i, [_vp, _vp, _vp, _vp]),
    "m2t_text_encode": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), _i, _i, _vp, _vp, _vp]),
    "m2t_transblock_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "m2t_transblock_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "m2t_eval_metrics_scratch_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "m2t_eval_metrics": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "m2t_eval_gmsd_scratch_bytes": (C.c_size_t, [_i]),
    "m2t_eval_gmsd": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "m2t_eval_fsim_scratch_bytes": (C.c_size_t, [_i, _i]),
    "m2t_eval_fsim": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "m2t_crop_patches": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "m2t_image_to_tensor": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "m2t_box_mix": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "m2t_dwt": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "m2t_iwt": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "m2t_pixel_shuffle": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "m2t_pixel_unshuffle": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "m2t_to_nhwc": (_i, [_i, _vp, _vp, _i, _i, _i, _vp]),
    "m2t_to_nchw": (_i, [_i, _vp, _vp, _i, _i, _i, _vp]),
    "m2t_window_attention_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "m2t_window_attention_bwd_scratch_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "m2t_window_attention_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
}

_lib = None


class M2TError(RuntimeError):
    pass


def load():
    """Load libm2t.so once; raises M2TError when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise M2TError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m m2trans_amd.build` "
            "(needs hipcc). There is no CPU fallback for this path.")
    # torch must bring in ITS HIP runtime (torch/lib/libamdhip64.so) first: libm2t.so then binds
    # to that already-loaded runtime, so streams and device pointers are shared with torch.
    # Loaded the other way round the process ends up with two runtimes (and ours sees no device).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().m2t_last_error_string()
        raise M2TError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
