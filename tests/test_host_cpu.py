"""CPU tests of host-side logic that needs no kernel launch: module construction, state-dict keys,
checkpoint format round trip (with and without the reference's `module.` prefix), PNG writer."""
import os
import types
import zlib

import numpy as np
import torch


def make_args(tmp):
    return types.SimpleNamespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=64, alpha_model_path=None,
        netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=3e-3, basedir=str(tmp),
        expname='exp', ft_path=None, no_reload=False, perturb=1., N_samples=64, white_bkgd=True, raw_noise_std=1.,
        dataset_type='llff', no_ndc=True, lindisp=True, sigma_loss=False)


def test_checkpoint_round_trip_with_module_prefix(tmp_path):
    from mvip_nerf_amd import run
    os.makedirs(tmp_path / 'exp')
    args = make_args(tmp_path)
    tr, te, start, grad_vars, opt = run.create_nerf(args, device=torch.device('cpu'))
    assert start == 0
    with torch.no_grad():
        for k, p in enumerate(grad_vars):
            p.fill_(0.001 * (k + 1))
    run.save_checkpoint(str(tmp_path / 'exp' / '000007.tar'), 7, tr, opt)
    ck = torch.load(str(tmp_path / 'exp' / '000007.tar'))
    assert set(ck) == {'global_step', 'network_fn_state_dict', 'network_fine_state_dict', 'optimizer_state_dict'}
    assert all(k.startswith('module.') for k in ck['network_fn_state_dict'])          # the reference's key form
    assert list(ck['network_fn_state_dict'])[0] == 'module.pts_linears.0.weight'
    tr2, _, start2, gv2, _ = run.create_nerf(args, device=torch.device('cpu'))         # auto-reload newest *.tar
    assert start2 == 7
    for a, b in zip(grad_vars, gv2):
        assert torch.equal(a, b)
    # and files written without the prefix load as well
    run.save_checkpoint(str(tmp_path / 'exp' / '000009.tar'), 9, tr, opt, module_prefix=False)
    assert run.create_nerf(args, device=torch.device('cpu'))[2] == 9


def test_nerf_module_matches_reference_layout():
    from mvip_nerf_amd.run_nerf_helpers import NeRF
    from mvip_nerf_amd import ops
    from oracle.weights import seeded_state_dict
    m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    sd = seeded_state_dict(0)
    assert list(m.state_dict()) == list(sd)
    assert [tuple(v.shape) for v in m.state_dict().values()] == [v.shape for v in sd.values()]
    assert tuple(ops.PARAM_ORDER) == tuple(sd) and ops.PARAM_SHAPES == tuple(v.shape for v in sd.values())
    assert sum(p.numel() for p in m.parameters()) == 595844


def test_png_writer(tmp_path):
    from mvip_nerf_amd.run import _write_png
    img = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    _write_png(str(tmp_path / 'a.png'), img)
    b = open(tmp_path / 'a.png', 'rb').read()
    assert b[:8] == b'\x89PNG\r\n\x1a\n' and b[12:16] == b'IHDR'
    i = b.index(b'IDAT')
    n = int.from_bytes(b[i - 4:i], 'big')
    raw = zlib.decompress(b[i + 4:i + 4 + n])
    rows = np.frombuffer(raw, np.uint8).reshape(5, 1 + 21)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(5, 7, 3), img)
