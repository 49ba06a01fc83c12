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
    'ss_conv_workspace_need': (c_ll, [c_i] * 14),
    'ss_conv_nhwc': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp] + [c_i] * 15 + [c_i, c_ll, c_ll, c_ll, c_fp, c_ll, c_st]),
    'ss_conv_pool2_nhwc': (c_i, [c_fp, c_fp, c_fp, c_fp] + [c_i] * 13 + [c_ll, c_ll, c_ll, c_fp, c_ll, c_st]),
    'ss_nchw_to_nhwc3_padded': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_st]),
    'ss_conv_stem3': (c_i, [c_fp, c_fp, c_fp, c_fp] + [c_i] * 7 + [c_ll] * 3 + [c_st]),
    'ss_stem_pool_packed_floats': (c_ll, [c_i]),
    'ss_stem_pool_pack': (c_i, [c_fp, c_fp, c_i, c_st]),
    'ss_stem_pool': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_ll, c_st]),
    'ss_wino_packed_floats': (c_ll, [c_i, c_i]),
    'ss_wino_packed3_floats': (c_ll, [c_i, c_i]),
    'ss_wino_pack': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_st]),
    'ss_wino_pack3': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_st]),
    'ss_conv_uses_winograd': (c_i, [c_i] * 9),
    'ss_conv3x3_wino_nhwc': (c_i, [c_fp] * 5 + [c_i] * 8 + [c_ll] * 3 + [c_st]),
    'ss_conv3x3_wino3_nhwc': (c_i, [c_fp] * 5 + [c_i] * 8 + [c_ll] * 3 + [c_st]),
    'ss_wino43_packed_floats': (c_ll, [c_i, c_i]),
    'ss_conv_uses_wino43': (c_i, [c_i] * 13),
    'ss_wino43_pack': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_st]),
    'ss_conv3x3_wino43_nhwc': (c_i, [c_fp] * 5 + [c_i] * 8 + [c_ll] * 3 + [c_st]),
    'ss_conv3x3_wino_pool2_nhwc': (c_i, [c_fp] * 4 + [c_i] * 8 + [c_ll] * 3 + [c_st]),
    'ss_maxpool_nhwc': (c_i, [c_fp, c_fp] + [c_i] * 7 + [c_st]),
    'ss_maxpool_nhwc_split': (c_i, [c_fp, c_fp, c_fp] + [c_i] * 7 + [c_st]),
    'ss_linear': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_st]),
    'ss_linear_grouped': (c_i, [c_fp, c_ll, c_fp, c_fp, ctypes.POINTER(c_fp), c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_ccl_workspace_floats': (c_ll, [c_i, c_i, c_i, c_i]),
    'ss_ccl': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_fp, c_st]),
    'ss_l2norm_nhwc': (c_i, [c_fp, c_fp, c_ll, c_i, c_st]),
    'ss_cost_volume': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_cost_volume_bidir': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_cost_volume_shifted': (c_i, [c_fp, c_fp, c_fp] + [c_i] * 8 + [c_st]),
    'ss_cost_volume_set_tile': (c_i, [c_i]),
    'ss_wino43_set_persistent': (c_i, [c_i]),
    'ss_tensor_dlt': (c_i, [c_fp, c_fp, c_fp, c_i, c_st]),
    'ss_spatial_decompose': (c_i, [c_fp, c_fp, c_fp, c_i, c_f, c_f, c_st]),
    'ss_spatial_meshes': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_f, c_st]),
    'ss_homo_warp_nhwc': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_homo_warp_pair_nhwc': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_homo_warp_nchw': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_tps_solve': (c_i, [c_fp, c_fp, c_fp, c_i, c_st]),
    'ss_tps_solve_shared_target': (c_i, [c_fp, c_fp, c_fp, c_i, c_st]),
    'ss_tps_points': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_st]),
    'ss_tsmotion_workspace_floats': (c_ll, [c_i]),
    'ss_tps_inverse': (c_i, [c_fp, c_fp, c_st]),
    'ss_tsmotion': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_f, c_fp, c_fp, c_st]),
    'ss_tsmotion_lag': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_f, c_f, c_fp, c_fp, c_st]),
    'ss_tps_warp_nchw': (c_i, [c_fp, c_fp, c_fp, c_fp] + [c_i] * 7 + [c_st]),
    'ss_tps_warp_mask_nchw': (c_i, [c_fp, c_fp, c_fp, c_fp] + [c_i] * 7 + [c_st]),
    'ss_tps_warp_views': (c_i, [ctypes.POINTER(c_fp), c_fp, c_fp, c_fp] + [c_i] * 6 + [c_st]),
    'ss_add_mul': (c_i, [c_fp, c_fp, c_f, c_f, c_ll, c_st]),
    'ss_mask_union': (c_i, [c_fp, c_fp, c_fp, c_ll, c_st]),
    'ss_ingest_u8': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_st]),
    'ss_canvas_to_u8': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_st]),
    'ss_render_average': (c_i, [ctypes.POINTER(c_fp), c_fp, c_fp, c_fp, c_ll, c_fp] + [c_i] * 6 + [c_st]),
    'ss_render_average_u8': (c_i, [ctypes.POINTER(c_fp), c_fp, c_fp, c_fp, c_ll, c_fp] + [c_i] * 6 + [c_st]),
    'ss_render_average_clip': (c_i, [ctypes.POINTER(c_fp), c_fp, c_fp, c_fp, c_ll, c_fp] + [c_i] * 7 + [c_st]),
    'ss_render_average_clip_u8': (c_i, [ctypes.POINTER(c_fp), c_fp, c_fp, c_fp, c_ll, c_fp] + [c_i] * 7 + [c_st]),
    'ss_render_footprint_floats': (c_ll, [c_i, c_i, c_i]),
    'ss_render_footprints': (c_i, [c_fp, c_fp, c_fp] + [c_i] * 6 + [c_st]),
    'ss_render_footprints_watch': (c_i, [c_fp, c_fp, c_fp] + [c_i] * 6 + [c_f, c_fp, c_fp, c_st]),
    'ss_linear_blend_workspace_floats': (c_ll, [c_i, c_i]),
    'ss_linear_blend': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_st]),
    'ss_linear_clip_workspace_floats': (c_ll, [c_i, c_i, c_i, c_i]),
    'ss_linear_clip_set_rows': (c_i, [c_i]),
    'ss_render_linear_clip': (c_i, [ctypes.POINTER(c_fp), c_fp, c_fp, c_fp, c_fp] + [c_i] * 7 + [c_fp, c_st]),
    'ss_render_linear_clip_u8': (c_i, [ctypes.POINTER(c_fp), c_fp, c_fp, c_fp, c_fp] + [c_i] * 7 + [c_fp, c_st]),
    'ss_mesh_bbox': (c_i, [c_fp, c_i, c_f, c_f, c_fp, c_i, c_st]),
    'ss_mesh_normalize': (c_i, [c_fp, c_fp, c_fp, c_i, c_f, c_f, c_st]),
    'ss_canvas_watch': (c_i, [c_fp, c_i, c_i, c_f, c_fp, c_fp, c_st]),
    'ss_stream_normalize_watch': (c_i, [ctypes.c_void_p, c_i, c_ll, c_fp, c_i, c_fp, c_i, c_f, c_f, c_f, c_fp, c_fp, c_st]),
    'ss_mesh_normalize_views': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_f, c_f, c_st]),
    'ss_mesh_normalize_views_boxes': (c_i, [c_fp, c_ll, c_fp, c_fp, c_i, c_i, c_i, c_f, c_f, c_st]),
    'ss_fill_f32': (c_i, [c_fp, c_f, c_ll, c_st]),
    'ss_h2mesh': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_st]),
    'ss_three_view_align': (c_i, [c_fp] * 9 + [c_i, c_f, c_f, c_st]),
    'ss_three_view_finish': (c_i, [c_fp] * 7 + [c_ll, c_st]),
    'ss_three_view_splines': (c_i, [c_fp] * 4 + [c_ll] + [c_fp] * 8 + [c_i, c_f, c_f, c_st]),
    'ss_stream_splines': (c_i, [ctypes.c_void_p, c_i, c_ll, c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_f, c_f, c_st]),
    'ss_three_view_normalize': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_ll, c_st]),
    'ss_smooth_embed': (c_i, [c_fp] * 9 + [c_i] * 4 + [c_st]),
    'ss_smooth_finalize': (c_i, [c_fp] * 13 + [c_i] * 4 + [c_st]),
    'ss_smooth_stitch': (c_i, [c_fp] * 11 + [c_i] * 2 + [c_st]),
    'ss_window_push': (c_i, [c_fp, c_fp, ctypes.c_void_p, c_i, c_i, c_i, c_fp, c_i, c_i, c_ll, c_ll, c_st]),
    'ss_window_push_groups': (c_i, [c_fp, c_fp, ctypes.c_void_p, c_i, c_i, c_i, c_i, c_fp, c_i, c_i, c_ll, c_ll, c_st]),
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
    code = 0


def check(code, what):
    if code != 0:
        err = HipError('%s failed: %s (%d)' % (what, lib().ss_error_string(code).decode(), code))
        err.code = code
        raise err


class DevPtr(ctypes.c_void_p):
    """Device pointer that remembers which GPU it lives on (so that `call` can pick that GPU's stream)."""
    dev = None


class _CurrentStream:
    """Placeholder for "the current HIP stream of the device the pointer arguments live on"; resolved by `call`."""


_STREAM = _CurrentStream()


def stream():
    return _STREAM


def dptr(t, allow_none=False, dtype=torch.float32):
    """Device pointer of a contiguous device tensor of `dtype` (fp32 unless stated)."""
    if t is None:
        if allow_none:
            return None
        raise ValueError('null tensor passed to a HIP kernel')
    if not t.is_cuda:
        raise HipError('stabstitch2_amd kernels need device tensors (got %s); there is no CPU path' % t.device)
    if t.dtype != dtype or not t.is_contiguous():
        raise HipError('stabstitch2_amd kernels need contiguous %s tensors (got %s, contiguous=%s)'
                       % (str(dtype).replace('torch.', ''), t.dtype, t.is_contiguous()))
    p = DevPtr(t.data_ptr())
    p.dev = t.device.index
    return p


def ptr_array(tensors, dtype=torch.float32):
    """`const float* const*` argument: array of device pointers (all tensors checked like `dptr`); carries the device."""
    ps = [dptr(t, dtype=dtype) for t in tensors]
    arr = (ctypes.c_void_p * len(ps))(*[p.value for p in ps])
    arr.dev = ps[0].dev
    if any(p.dev != arr.dev for p in ps):
        raise HipError('tensors of one launch live on different GPUs')
    return arr


def call(name, *args):
    """Launch on the GPU that owns the pointer arguments, on THAT device's current stream (not the current device's):
    nets moved to cuda:1 keep working while cuda:0 is current, and launches stay ordered with the torch ops that
    produced their inputs."""
    dev = None
    for a in args:
        d = getattr(a, 'dev', None)
        if d is not None:
            if dev is None:
                dev = d
            elif d != dev:
                raise HipError('%s: tensors of one launch live on different GPUs (cuda:%d and cuda:%d)' % (name, dev, d))
    fn = getattr(lib(), name)
    if dev is None:                          # no device pointers (workspace queries, version)
        return check(fn(*args), name)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    args = tuple(st if a is _STREAM else a for a in args)
    if dev == torch.cuda.current_device():
        return check(fn(*args), name)
    with torch.cuda.device(dev):
        return check(fn(*args), name)
