"""ctypes binding of libmvipnerf.so (the C ABI declared in include/mvip_nerf.h).

The product path has no CPU fallback: if the library is missing or a call fails, this raises.
PyTorch is used above this layer only for device memory, streams and autograd plumbing.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVIP_LIB_PATH: another build of the same library (tools/: the -DMVIP_EXPERIMENT_* timing builds live in lib_experiment/)
LIB_PATH = os.environ.get('MVIP_LIB_PATH') or os.path.join(_HERE, 'lib', 'libmvipnerf.so')

_c_f = ctypes.c_void_p      # device pointers travel as integers
_i64 = ctypes.c_int64
_int = ctypes.c_int
_flt = ctypes.c_float

_SIGNATURES = {
    'mvip_abi_version': (_int, []),
    'mvip_build_is_experiment': (_int, []),
    'mvip_strerror': (ctypes.c_char_p, [_int]),
    'mvip_last_hip_error': (ctypes.c_char_p, []),
    'mvip_device_info': (_int, [ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.c_char_p, _int]),
    'mvip_get_rays': (_int, [_c_f, _int, _int, _flt, _int, _int, _int, _int, _c_f, _c_f, _c_f]),
    'mvip_ray_rows': (_int, [_c_f, _c_f, _c_f, _flt, _flt, _i64, _c_f, _c_f]),
    'mvip_ray_rows_from_pose': (_int, [_c_f, _int, _int, _flt, _flt, _flt, _c_f, _i64, _c_f, _c_f]),
    'mvip_stratified_z': (_int, [_c_f, _int, _i64, _int, _c_f, _int, _c_f, _c_f, _c_f]),
    'mvip_posenc': (_int, [_c_f, _i64, _int, _c_f, _c_f]),
    'mvip_mlp_packed_floats': (_i64, []),
    'mvip_mlp_pack': (_int, [ctypes.POINTER(ctypes.c_void_p), _c_f, _c_f]),
    'mvip_mlp_pack_f16x3': (_int, [ctypes.POINTER(ctypes.c_void_p), _c_f, _c_f, _c_f]),
    'mvip_mlp_forward_rays_f16x3': (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, _c_f]),
    'mvip_mlp_forward_points_f16x3': (_int, [_c_f, _c_f, _c_f, _i64, _c_f, _c_f]),
    'mvip_mlp_pack_f16x3_w16': (_int, [ctypes.POINTER(ctypes.c_void_p), _c_f, _c_f, _c_f]),
    'mvip_mlp_forward_rays_f16x3_w16': (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, _c_f]),
    'mvip_mlp_forward_points_f16x3_w16': (_int, [_c_f, _c_f, _c_f, _i64, _c_f, _c_f]),
    'mvip_mlp_forward_rays_stash_f16x3_w16': (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, _c_f, _c_f]),
    'mvip_mlp_forward_rays': (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, _int, _c_f]),
    'mvip_mlp_forward_points': (_int, [_c_f, _c_f, _c_f, _i64, _c_f, _int, _c_f]),
    'mvip_mlp_pack16': (_int, [ctypes.POINTER(ctypes.c_void_p), _c_f, _c_f, _c_f]),
    'mvip_mlp_forward_rays16': (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, _c_f]),
    'mvip_mlp_forward_rays_stash16': (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, _c_f, _c_f]),
    'mvip_mlp_forward_points16': (_int, [_c_f, _c_f, _c_f, _i64, _c_f, _c_f]),
    'mvip_render_coarse_fused': (_int, [_c_f, _c_f, _i64, _c_f, _int, _c_f, _c_f, _c_f, _int, _int, _int, _c_f, _c_f, _c_f, _c_f,
                                        _c_f, _c_f, _c_f, _c_f, _c_f]),
    'mvip_render_fine_fused': (_int, [_c_f, _c_f, _c_f, _i64, _c_f, _int, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'mvip_mlp_backward_workspace_bytes': (_i64, [_i64]),
    'mvip_mlp_backward_rays': (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, ctypes.POINTER(ctypes.c_void_p), _c_f, _i64,
                                      _int, _c_f]),
    'mvip_mlp_backward_points': (_int, [_c_f, _c_f, _c_f, _i64, _c_f, ctypes.POINTER(ctypes.c_void_p), _c_f, _i64, _int,
                                        _c_f]),
    'mvip_mlp_stash_floats': (_i64, [_i64]),
    'mvip_mlp_forward_rays_stash': (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, _c_f, _int, _c_f]),
    'mvip_mlp_forward_points_stash': (_int, [_c_f, _c_f, _c_f, _i64, _c_f, _c_f, _int, _c_f]),
    'mvip_mlp_backward_stash': (_int, [_c_f, _c_f, _i64, _c_f, ctypes.POINTER(ctypes.c_void_p), _c_f, _i64, _int, _c_f]),
    'mvip_mlp_unpack_grads': (_int, [_c_f, ctypes.POINTER(ctypes.c_void_p), _int, _c_f]),
    'mvip_composite_forward': (_int, [_c_f, _c_f, _c_f, _int, _c_f, _i64, _int, _int, _c_f, _c_f, _c_f, _c_f,
                                      _c_f, _c_f, _c_f]),
    'mvip_composite_backward': (_int, [_c_f, _c_f, _c_f, _int, _c_f, _i64, _int, _int, _c_f, _c_f, _c_f, _c_f,
                                       _c_f, _c_f, _c_f, _c_f]),
    'mvip_sample_pdf_merge': (_int, [_c_f, _c_f, _c_f, _int, _i64, _int, _int, _c_f, _c_f, _c_f, _c_f, _c_f,
                                     _c_f]),
    'mvip_sample_pdf': (_int, [_c_f, _c_f, _c_f, _int, _i64, _int, _int, _c_f, _c_f, _c_f, _c_f]),
    'mvip_depth2xyz': (_int, [_c_f, _int, _int, _flt, _flt, _flt, _flt, _c_f, _c_f]),
    'mvip_depth2xyz_backward': (_int, [_c_f, _int, _int, _flt, _flt, _flt, _flt, _c_f, _c_f]),
    'mvip_normal_fit_forward': (_int, [_c_f, _int, _int, _int, _c_f, _c_f, _c_f, _c_f]),
    'mvip_normal_fit_backward': (_int, [_c_f, _c_f, _c_f, _c_f, _int, _int, _int, _c_f, _c_f, _c_f]),
    'mvip_sds_add_noise': (_int, [_c_f, _c_f, _flt, _flt, _i64, _c_f, _c_f]),
    'mvip_sds_grad': (_int, [_c_f, _c_f, _c_f, _flt, _flt, _i64, _int, _c_f, _c_f]),
    'mvip_sds_add_noise_dev': (_int, [_c_f, _c_f, _c_f, _i64, _c_f, _c_f]),
    'mvip_sds_grad_dev': (_int, [_c_f, _c_f, _c_f, _flt, _c_f, _i64, _int, _c_f, _c_f]),
    'mvip_vae_sample': (_int, [_c_f, _c_f, _flt, _i64, _i64, _i64, _c_f, _c_f]),
    'mvip_vae_sample_backward': (_int, [_c_f, _c_f, _c_f, _flt, _i64, _i64, _i64, _c_f, _c_f]),
    'mvip_timestep_sincos': (_int, [_c_f, _c_f, _i64, _i64, _c_f, _c_f]),
    'mvip_resize_bilinear': (_int, [_c_f, _i64, _i64, _i64, _i64, _i64, _c_f, _c_f]),
    'mvip_resize_bilinear_backward': (_int, [_c_f, _i64, _i64, _i64, _i64, _i64, _c_f, _c_f]),
    'mvip_groupnorm_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'mvip_groupnorm_forward': (_int, [_c_f, _c_f, _c_f, _i64, _i64, _i64, _int, _flt, _int, _int, _c_f, _c_f, _c_f, _c_f,
                                      _c_f]),
    'mvip_groupnorm_stats': (_int, [_c_f, _i64, _i64, _i64, _int, _flt, _int, _c_f, _c_f, _c_f, _c_f]),
    'mvip_conv3x3_supported': (_int, [_i64, _i64, _i64, _i64]),
    'mvip_conv3x3_packed_bytes': (_i64, [_i64, _i64]),
    'mvip_conv3x3_pack': (_int, [_c_f, _i64, _i64, _int, _c_f, _c_f]),
    'mvip_packed_weights_two_product': (_int, [_c_f, _i64, ctypes.POINTER(_int), _c_f]),
    'mvip_absmax_scale': (_int, [_c_f, _i64, _c_f, _c_f, _c_f]),
    'mvip_split_planes': (_int, [_c_f, _i64, _i64, _i64, _c_f, _c_f, _int, _c_f]),
    'mvip_groupnorm_split_planes': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _int, _int, _c_f, _int, _c_f]),
    'mvip_conv3x3_f16x3': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _i64, _c_f, _int, _c_f]),
    'mvip_conv3x3_workspace_bytes': (_i64, [_i64, _i64, _i64, _i64, _i64]),
    'mvip_conv3x3_f16x3_ws': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _i64, _c_f, _c_f, _int, _c_f]),
    'mvip_hashgrid_forward': (_int, [_c_f, _c_f, _c_f, _i64, _flt, _c_f, _c_f]),
    'mvip_hashgrid_backward': (_int, [_c_f, _c_f, _c_f, _i64, _flt, _c_f, _c_f]),
    'mvip_sh4': (_int, [_c_f, _i64, _c_f, _c_f]),
    'mvip_hashgrid_backward_half2': (_int, [_c_f, _c_f, _c_f, _i64, _flt, _c_f, _c_f, _c_f, _c_f]),
    'mvip_hashgrid_mlp_packed_floats': (_i64, []),
    'mvip_hashgrid_mlp_pack': (_int, [_c_f, _c_f, _c_f, _c_f]),
    'mvip_hashgrid_nerf_forward': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _i64, _flt, _c_f, _c_f]),
    'mvip_skinny_wgrad_slabs': (_i64, [_i64]),
    'mvip_skinny_wgrad': (_int, [_c_f, _c_f, _i64, _i64, _i64, _c_f, _c_f]),
    'mvip_skinny_linear': (_int, [_c_f, _i64, _i64, _c_f, _i64, _i64, _i64, _int, _c_f, _c_f]),
    'mvip_gemm_packed_bytes': (_i64, [_i64, _i64]),
    'mvip_gemm_pack_a': (_int, [_c_f, _i64, _i64, _i64, _i64, _c_f, _c_f]),
    'mvip_split_planes_strided': (_int, [_c_f, _i64, _i64, _i64, _i64, _i64, _i64, _c_f, _c_f, _int, _c_f]),
    'mvip_split_planes_upsample2': (_int, [_c_f, _i64, _i64, _i64, _i64, _c_f, _c_f, _int, _c_f]),
    'mvip_softmax_rows': (_int, [_c_f, _i64, _i64, _flt, _c_f, _c_f]),
    'mvip_softmax_rows_backward': (_int, [_c_f, _c_f, _i64, _i64, _flt, _c_f, _c_f]),
    'mvip_gemm_f16x3': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _c_f, _int, _c_f]),
    'mvip_gemm_geglu_f16x3': (_int, [_c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _i64, _c_f, _c_f, _c_f, _int, _c_f]),
    'mvip_im2col_split_planes': (_int, [_c_f, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _i64, _i64, _i64, _i64,
                                        _c_f, _c_f, _int, _c_f]),
    'mvip_col2im': (_int, [_c_f, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _i64, _i64, _i64, _i64, _c_f, _c_f]),
    'mvip_gemm_workspace_bytes': (_i64, [_i64, _i64, _i64, _i64]),
    'mvip_gemm_f16x3_ws': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _c_f, _c_f, _int, _c_f]),
    'mvip_gemm_ln_segments': (_i64, [_i64, _i64, _i64, _i64, _int]),
    'mvip_gemm_f16x3_ws_ln': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _c_f, _c_f, _c_f, _int, _c_f]),
    'mvip_layernorm_split_planes_stats': (_int, [_c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _i64, _flt, _flt, _c_f, _int, _c_f]),
    'mvip_gemm_f16x3_cfg': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _c_f, _int, _int, _c_f]),
    'mvip_attention_supported': (_int, [_i64]),
    'mvip_attention_v_bytes': (_i64, [_i64, _i64, _i64, _i64]),
    'mvip_attention_pack_v': (_int, [_c_f, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _c_f, _c_f, _int, _c_f]),
    'mvip_absmax_scale_sections': (_int, [_c_f, _i64, _i64, _i64, _c_f, _c_f, _c_f]),
    'mvip_attention_f16x3': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _flt, _int,
                                    _c_f, _int, _c_f]),
    'mvip_gemm_f16x3_sinks': (_int, [_c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _int, ctypes.POINTER(_i64),
                                     ctypes.POINTER(_int), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_flt), _int, _int, _c_f]),
    'mvip_gemm_f16x3_planes_ws': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _c_f, _flt, _c_f, _int, _c_f]),
    'mvip_gemm_geglu_f16x3_sink': (_int, [_c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _i64, _c_f, _flt, _int, _c_f]),
    'mvip_attention_f16x3_sink': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                         _i64, _i64, _flt, _int, _c_f, _int, _c_f]),
    'mvip_layernorm_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'mvip_layernorm_split_planes': (_int, [_c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _flt, _flt, _c_f, _c_f, _int, _c_f]),
    'mvip_geglu': (_int, [_c_f, _i64, _i64, _i64, _i64, _c_f, _c_f, _c_f, _c_f]),
    'mvip_linear_small': (_int, [_c_f, _c_f, _c_f, _i64, _i64, _i64, _int, _c_f, _c_f]),
    'mvip_linear_small_grouped': (_int, [_c_f, _c_f, _c_f, _i64, _i64, _i64, _int, _c_f, _i64, _c_f, _c_f]),
    'mvip_groupnorm_backward': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _int, _int, _int, _c_f, _c_f,
                                       _c_f]),
    'mvip_conv3x3_row_moments_doubles': (_i64, [_i64, _i64, _i64, _i64, _i64]),
    'mvip_conv3x3_tile_moments_scratch_bytes': (_i64, [_i64, _i64, _i64, _i64, _i64]),
    'mvip_conv3x3_f16x3_tile_moments': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _i64, _c_f, _c_f, _c_f,
                                               _int, _c_f]),
    'mvip_conv3x3_f16x3_ws_moments': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _i64, _i64, _c_f, _c_f, _c_f,
                                             _int, _c_f]),
    'mvip_groupnorm_backward_maxima': (_i64, [_i64, _i64, _i64]),
    'mvip_groupnorm_backward_fused': (_int, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _i64, _i64, _int, _int, _int, _c_f, _c_f,
                                             _c_f, _c_f, _c_f]),
    'mvip_absmax_scale_from_maxima': (_int, [_c_f, _i64, _c_f, _c_f]),
    'mvip_groupnorm_split_planes_moments': (_int, [_c_f, _c_f, _c_f, _c_f, _flt, _i64, _i64, _i64, _int, _int, _c_f, _int,
                                                   _c_f]),
    'mvip_groupnorm_split_planes_moments_out': (_int, [_c_f, _c_f, _c_f, _c_f, _flt, _i64, _i64, _i64, _int, _int, _c_f, _c_f, _c_f,
                                                       _int, _c_f]),
}

# every symbol include/mvip_nerf.h declares; tests check the built library exports all of them
DECLARED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
ABI_VERSION = 5


class MvipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MvipError(
            f'{LIB_PATH} not found: build it with `python -m mvip_nerf_amd.csrc.build` '
            '(or __graft_entry__.build()).  There is no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.mvip_abi_version() != ABI_VERSION:
        raise MvipError('libmvipnerf.so ABI version mismatch')
    if lib.mvip_build_is_experiment() and os.environ.get('MVIP_ALLOW_EXPERIMENT_BUILD') != '1':
        raise MvipError(f'{LIB_PATH} was compiled with -DMVIP_EXPERIMENT_* (timing experiment, wrong results): rebuild with '
                        '`python -m mvip_nerf_amd.csrc.build`, or set MVIP_ALLOW_EXPERIMENT_BUILD=1 for the experiment itself')
    _lib = lib
    return lib


def check(code, what=''):
    if code != 0:
        lib = load()
        msg = lib.mvip_strerror(code).decode()
        if code == -2:
            msg += ': ' + lib.mvip_last_hip_error().decode()
        raise MvipError(f'{what or "mvip call"} failed: {msg}')


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32):
    """Device pointer of a dense tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise MvipError('expected a CUDA/HIP tensor (the HIP path has no CPU fallback)')
    if t.dtype != dtype:
        raise MvipError(f'expected dtype {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise MvipError('expected a contiguous tensor')
    return ctypes.c_void_p(t.data_ptr())


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t).value
    return arr


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
