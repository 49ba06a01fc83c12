"""ctypes binding of libstabstitch_hip.so (the C ABI declared in include/stabstitch_hip.h).

There is no CPU fallback: if the shared library is missing or a tensor is not a contiguous fp32
CUDA(HIP) tensor, the call raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libstabstitch_hip.so')

c_fp = ctypes.c_void_p        # device pointer
c_i = ctypes.c_int
c_f = ctypes.c_float
c_ll = ctypes.c_longlong
c_st = ctypes.c_void_p        # hipStream_t

# name -> (restype, argtypes); mirrors include/stabstitch_hip.h one to one
SIGNATURES = {
    'ss_version': (c_i, []),
    'ss_error_string': (ctypes.c_char_p, [c_i]),
    'ss_nchw_to_nhwc': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_nhwc_to_nchw': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_conv_workspace_floats': (c_ll, []),
    'ss_conv_nhwc': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp] + [c_i] * 15 + [c_i, c_ll, c_ll, c_ll, c_fp, c_ll, c_st]),
    'ss_maxpool_nhwc': (c_i, [c_fp, c_fp] + [c_i] * 7 + [c_st]),
    'ss_maxpool_nhwc_split': (c_i, [c_fp, c_fp, c_fp] + [c_i] * 7 + [c_st]),
    'ss_linear': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    'ss_ccl_workspace_floats': (c_ll, [c_i, c_i, c_i, c_i]),
    'ss_ccl': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_fp, c_st]),
    'ss_cost_volume': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_tensor_dlt': (c_i, [c_fp, c_fp, c_fp, c_i, c_st]),
    'ss_spatial_decompose': (c_i, [c_fp, c_fp, c_fp, c_i, c_f, c_f, c_st]),
    'ss_spatial_meshes': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_f, c_st]),
    'ss_homo_warp_nhwc': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_homo_warp_nchw': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_tps_solve': (c_i, [c_fp, c_fp, c_fp, c_i, c_st]),
    'ss_tps_points': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_st]),
    'ss_tsmotion_workspace_floats': (c_ll, [c_i]),
    'ss_tsmotion': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_f, c_fp, c_st]),
    'ss_tps_warp_nchw': (c_i, [c_fp, c_fp, c_fp, c_fp] + [c_i] * 7 + [c_st]),
    'ss_tps_warp_mask_nchw': (c_i, [c_fp, c_fp, c_fp, c_fp] + [c_i] * 7 + [c_st]),
    'ss_tps_warp_views': (c_i, [ctypes.POINTER(c_fp), c_fp, c_fp, c_fp] + [c_i] * 6 + [c_st]),
    'ss_add_mul': (c_i, [c_fp, c_fp, c_f, c_f, c_ll, c_st]),
    'ss_mask_union': (c_i, [c_fp, c_fp, c_fp, c_ll, c_st]),
    'ss_ingest_u8': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_canvas_to_u8': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_st]),
    'ss_render_average': (c_i, [ctypes.POINTER(c_fp), c_fp, c_fp, c_fp] + [c_i] * 6 + [c_st]),
    'ss_linear_blend_workspace_floats': (c_ll, [c_i, c_i]),
    'ss_linear_blend': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_st]),
    'ss_mesh_bbox': (c_i, [c_fp, c_i, c_f, c_f, c_fp, c_i, c_st]),
    'ss_mesh_normalize': (c_i, [c_fp, c_fp, c_fp, c_i, c_f, c_f, c_st]),
    'ss_smooth_embed': (c_i, [c_fp] * 9 + [c_i] * 4 + [c_st]),
    'ss_smooth_finalize': (c_i, [c_fp] * 13 + [c_i] * 4 + [c_st]),
    'ss_alignment_psnr_ssim': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_st]),
    'ss_stability_score': (c_i, [c_fp, c_fp, c_i, c_st]),
    'ss_distortion_score': (c_i, [c_fp, c_fp, c_fp, c_i, c_st]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'stabstitch2_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
                'or stabstitch2_amd/csrc/build.sh (hipcc --offload-arch=gfx950). There is no CPU fallback.'
                % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class HipError(RuntimeError):
    pass


def check(code, what):
    if code != 0:
        raise HipError('%s failed: %s (%d)' % (what, lib().ss_error_string(code).decode(), code))


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t, allow_none=False):
    """Device pointer of a contiguous fp32 device tensor."""
    if t is None:
        if allow_none:
            return None
        raise ValueError('null tensor passed to a HIP kernel')
    if not t.is_cuda:
        raise HipError('stabstitch2_amd kernels need device tensors (got %s); there is no CPU path' % t.device)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise HipError('stabstitch2_amd kernels need contiguous float32 tensors (got %s, contiguous=%s)'
                       % (t.dtype, t.is_contiguous()))
    return ctypes.c_void_p(t.data_ptr())


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
