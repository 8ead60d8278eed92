"""Error behaviour of the C-ABI entry points, without a GPU: every entry validates its arguments before the first HIP call,
so a malformed call returns MVIP_EINVAL (-1), an empty call (no rays / no samples / no images) returns MVIP_OK with null
pointers, and a well-formed shape with a null operand returns MVIP_EINVAL -- the same three answers the Python mirror turns
into `MvipError` (tests/test_abi.py).  No compute call is made here.

Argument order follows include/mvip_nerf.h; pointers travel as integers (0 = NULL)."""
import ctypes

import pytest

from mvip_nerf_amd import _lib

OK, EINVAL = 0, -1
P0 = None            # NULL


def raw(name, *args):
    return getattr(_lib.load(), name)(*args)


# (entry point, malformed call, empty call with null operands or None, well-formed shape with null operands)
CASES = [
    # rays / samples / compositing / MLP
    ('mvip_sample_pdf', (P0, P0, P0, 0, 8, 1, 64, P0, P0, P0, P0), (P0, P0, P0, 0, 0, 63, 64, P0, P0, P0, P0),
     (P0, P0, P0, 0, 8, 63, 64, P0, P0, P0, P0)),
    ('mvip_mlp_forward_rays', (P0, P0, P0, 4, 0, P0, 0, P0), (P0, P0, P0, 0, 64, P0, 0, P0), (P0, P0, P0, 4, 64, P0, 0, P0)),
    ('mvip_mlp_forward_rays_f16x3_w16', (P0, P0, P0, 4, 0, P0, P0), (P0, P0, P0, 0, 64, P0, P0), (P0, P0, P0, 4, 64, P0, P0)),
    ('mvip_mlp_forward_rays_stash_f16x3_w16', (P0, P0, P0, 4, 0, P0, P0, P0), (P0, P0, P0, 0, 64, P0, P0, P0), (P0, P0, P0, 4, 64, P0, P0, P0)),
    # SDS operand producers and contractions (rows a14-a16)
    ('mvip_resize_bilinear', (P0, 3, 0, 8, 16, 16, P0, P0), (P0, 0, 8, 8, 16, 16, P0, P0), (P0, 3, 8, 8, 16, 16, P0, P0)),
    ('mvip_absmax_scale', (P0, -1, P0, P0, P0), None, (P0, 16, P0, P0, P0)),
    ('mvip_vae_sample', (P0, P0, 0.18215, 1, 0, 64, P0, P0), (P0, P0, 0.18215, 0, 4, 64, P0, P0), (P0, P0, 0.18215, 1, 4, 64, P0, P0)),
    ('mvip_vae_sample_backward', (P0, P0, P0, 0.18215, -1, 4, 64, P0, P0), (P0, P0, P0, 0.18215, 1, 4, 0, P0, P0),
     (P0, P0, P0, 0.18215, 1, 4, 64, P0, P0)),
    ('mvip_timestep_sincos', (P0, P0, 1, 0, P0, P0), (P0, P0, 0, 160, P0, P0), (P0, P0, 2, 160, P0, P0)),
    ('mvip_split_planes', (P0, 1, 24, 64, P0, P0, 0, P0), (P0, 0, 32, 64, P0, P0, 0, P0), (P0, 1, 32, 64, P0, P0, 0, P0)),
    ('mvip_split_planes', (P0, 1, 32, 64, P0, P0, 3, P0), None, (P0, 1, 32, 64, P0, P0, 2, P0)),     # prec is 0 (f16x3), 1 (fp16 mode) or 2 (= 0 for a producer)
    ('mvip_packed_weights_two_product', (P0, 0, None, P0), None, (P0, 1024, None, P0)),
    ('mvip_groupnorm_stats', (P0, 1, 30, 64, 32, 1e-6, 0, P0, P0, P0, P0), (P0, 0, 64, 64, 32, 1e-6, 0, P0, P0, P0, P0),
     (P0, 1, 64, 64, 32, 1e-6, 0, P0, P0, P0, P0)),
    # 3x3 convolution: 33 output channels / a 12 x 12 image are not tileable
    ('mvip_conv3x3_f16x3_ws', (P0, P0, P0, P0, P0, P0, 1, 32, 33, 64, 64, P0, P0, 0, P0),
     (P0, P0, P0, P0, P0, P0, 0, 32, 32, 64, 64, P0, P0, 1, P0), (P0, P0, P0, P0, P0, P0, 1, 32, 32, 64, 64, P0, P0, 1, P0)),
    ('mvip_conv3x3_f16x3_ws', (P0, P0, P0, P0, P0, P0, 1, 32, 32, 12, 12, P0, P0, 0, P0), None, None),
    ('mvip_conv3x3_f16x3_ws', (P0, P0, P0, P0, P0, P0, 1, 32, 32, 64, 64, P0, P0, 3, P0), None, None),
    # GEMM: K must be a multiple of 32, M of 32, P of the pixel tile
    ('mvip_gemm_f16x3_ws', (P0, P0, P0, P0, P0, P0, 1, 48, 64, 256, P0, P0, 0, P0), (P0, P0, P0, P0, P0, P0, 0, 64, 64, 256, P0, P0, 0, P0),
     (P0, P0, P0, P0, P0, P0, 1, 64, 64, 256, P0, P0, 1, P0)),
    ('mvip_gemm_f16x3_ws', (P0, P0, P0, P0, P0, P0, 1, 64, 40, 256, P0, P0, 0, P0), None, None),
    # GEMMs / attention whose epilogue writes the next contraction's operands: sections are multiples of 64 rows that add up
    # to M, a V-fragment section is the last one, scales are positive
    ('mvip_gemm_f16x3_sinks', (P0, P0, P0, P0, 1, 64, 128, 256, 0, None, None, None, None, 1, 0, P0), None, None),
    ('mvip_gemm_f16x3_planes_ws', (P0, P0, P0, P0, P0, 1, 64, 48, 256, P0, 1.0, P0, 0, P0), (P0, P0, P0, P0, P0, 0, 64, 64, 256, P0, 1.0, P0, 0, P0),
     (P0, P0, P0, P0, P0, 1, 64, 64, 256, P0, 1.0, P0, 1, P0)),
    ('mvip_gemm_f16x3_planes_ws', (P0, P0, P0, P0, P0, 1, 64, 64, 256, P0, 0.0, P0, 0, P0), None, None),      # out_scale must be > 0
    ('mvip_gemm_geglu_f16x3_sink', (P0, P0, P0, P0, 1, 64, 96, 256, 256, P0, 1.0, 0, P0), (P0, P0, P0, P0, 0, 64, 128, 256, 256, P0, 1.0, 0, P0),
     (P0, P0, P0, P0, 1, 64, 128, 256, 256, P0, 1.0, 0, P0)),
    ('mvip_attention_f16x3_sink', (P0, P0, P0, P0, P0, P0, 1, 8, 40, 256, 256, 256, 256, 128, 256, 16, 0.158, 0, P0, 0, P0), None,
     (P0, P0, P0, P0, P0, P0, 1, 8, 40, 256, 256, 256, 256, 256, 256, 16, 0.158, 0, P0, 0, P0)),                 # q_stride < Lq
    ('mvip_im2col_split_planes', (P0, 1, 4, 16, 16, 3, 3, 0, 1, 1, 8, 8, 48, 64, P0, P0, 0, P0), None,
     (P0, 1, 4, 16, 16, 3, 3, 2, 1, 1, 8, 8, 48, 64, P0, P0, 0, P0)),
    ('mvip_col2im', (P0, 1, 4, 16, 16, 3, 3, 2, 1, 1, 8, 8, 16, 64, P0, P0), (P0, 0, 4, 16, 16, 3, 3, 2, 1, 1, 8, 8, 48, 64, P0, P0),
     (P0, 1, 4, 16, 16, 3, 3, 2, 1, 1, 8, 8, 48, 64, P0, P0)),
    ('mvip_layernorm_split_planes', (P0, P0, P0, 1, 100, 77, 256, 1e-5, 1.0, P0, P0, 0, P0), (P0, P0, P0, 0, 320, 77, 256, 1e-5, 1.0, P0, P0, 0, P0),
     (P0, P0, P0, 1, 320, 77, 256, 1e-5, 1.0, P0, P0, 1, P0)),
    # hash-grid model's small layers (row f4): at most 64 x 64, points in fours
    ('mvip_skinny_linear', (P0, 1, 1, P0, 65, 16, 128, 0, P0, P0), (P0, 1, 1, P0, 16, 16, 0, 0, P0, P0), (P0, 1, 1, P0, 16, 16, 128, 0, P0, P0)),
    ('mvip_skinny_linear', (P0, 1, 1, P0, 16, 16, 130, 0, P0, P0), None, None),
]


@pytest.mark.parametrize('name,bad,empty,null', CASES, ids=[f'{c[0]}-{i}' for i, c in enumerate(CASES)])
def test_entry_point_argument_checks(name, bad, empty, null):
    assert len(bad) == len(_lib._SIGNATURES[name][1]), 'the case must follow the binding in _lib.py'
    assert raw(name, *bad) == EINVAL
    if empty is not None:
        assert raw(name, *empty) == OK
    if null is not None:
        assert raw(name, *null) == EINVAL


def test_workspace_queries_answer_zero_for_unsupported_shapes():
    lib = _lib.load()
    assert lib.mvip_conv3x3_supported(32, 32, 64, 64) == 1
    assert lib.mvip_conv3x3_supported(32, 32, 8, 8) == 1             # the UNet's innermost level
    assert lib.mvip_conv3x3_supported(33, 32, 64, 64) == 0
    assert lib.mvip_conv3x3_supported(32, 24, 64, 64) == 0
    assert lib.mvip_conv3x3_supported(32, 32, 12, 12) == 0
    assert lib.mvip_conv3x3_workspace_bytes(1, 32, 33, 64, 64) == 0
    assert lib.mvip_conv3x3_workspace_bytes(0, 32, 32, 64, 64) == 0
    # a small grid with a long channel range is split over workgroups and needs room for the partial sums
    assert lib.mvip_conv3x3_workspace_bytes(2, 1280, 1280, 16, 16) > 0
    assert lib.mvip_conv3x3_workspace_bytes(1, 128, 128, 512, 512) == 0
    assert lib.mvip_conv3x3_packed_bytes(0, 32) == 0


def test_error_strings():
    lib = _lib.load()
    assert lib.mvip_strerror(0).decode()
    assert lib.mvip_strerror(EINVAL).decode().startswith('invalid')
    with pytest.raises(_lib.MvipError, match='invalid'):
        _lib.check(EINVAL, 'mvip_probe')
    assert isinstance(ctypes.c_void_p(0).value, type(None))          # NULL travels as None
