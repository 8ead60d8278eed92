"""Deterministic synthetic inputs shared by the golden generator and the tests (TEST INFRASTRUCTURE).

`np.random.RandomState` is numpy's frozen legacy generator: the same seed gives the same bits on
every numpy version, so fixtures can carry a seed instead of 2.4 MB of weights per network.
"""
import numpy as np

_LAYERS = ([(f'pts_linears.{i}', 256, 63 if i == 0 else (319 if i == 5 else 256)) for i in range(8)]
           + [('views_linears.0', 128, 283), ('feature_linear', 256, 256),
              ('alpha_linear', 1, 256), ('rgb_linear', 3, 128)])


def seeded_state_dict(seed, gain=1.0):
    """State dict of the 8x256 NeRF MLP (keys as DS_NeRF/run_nerf_helpers.py:86-100 creates
    them), U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like nn.Linear's default, as float32 numpy."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, fan_out, fan_in in _LAYERS:
        b = gain / np.sqrt(fan_in)
        sd[name + '.weight'] = rs.uniform(-b, b, size=(fan_out, fan_in)).astype(np.float32)
        sd[name + '.bias'] = rs.uniform(-b, b, size=(fan_out,)).astype(np.float32)
    return sd


def bench_like_rays(n, seed, near=1.2, far=7.74):
    """[n, 11] ray rows (o, d, near, far, viewdirs) shaped like the SURVEY.md §8(d) workload:
    camera near the origin looking down -z with a 378x504 / f=383.65 field of view."""
    rs = np.random.RandomState(seed)
    px = rs.uniform(0, 504, size=n)
    py = rs.uniform(0, 378, size=n)
    d = np.stack([(px - 252.) / 383.65, -(py - 189.) / 383.65, -np.ones(n)], -1)
    th = rs.uniform(0, 2 * np.pi)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    d = d @ R.T
    o = np.broadcast_to(np.array([0.3 * np.sin(th), 0., 0.3 * np.cos(th)]), (n, 3))
    v = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rows = np.concatenate([o, d, np.full((n, 1), near), np.full((n, 1), far), v], -1)
    return rows.astype(np.float32)


def seeded_clip_text_state(seed, vocab=49408, d=768, layers=12, ctx=77):
    """State dict of the CLIP ViT-L/14 TEXT tower in `mvip_nerf_amd.guidance.sd_nets.CLIPTextModel`'s key names
    (fp32 numpy, deterministic in `seed`): N(0, 0.02) matrices and embeddings, LayerNorm gains 1 + N(0, 0.1),
    biases N(0, 0.02) -- non-trivial everywhere so a wrong mapping or a missing bias shows up."""
    rs = np.random.RandomState(seed)
    sd = {}

    def mat(name, *shape):
        sd[name] = (rs.standard_normal(size=shape) * 0.02).astype(np.float32)

    def ln(name):
        sd[name + '.weight'] = (1.0 + 0.1 * rs.standard_normal(size=(d,))).astype(np.float32)
        sd[name + '.bias'] = (0.02 * rs.standard_normal(size=(d,))).astype(np.float32)
    mat('token_embedding.weight', vocab, d)
    mat('position_embedding.weight', ctx, d)
    for i in range(layers):
        ln(f'layers.{i}.layer_norm1')
        ln(f'layers.{i}.layer_norm2')
        for p in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
            mat(f'layers.{i}.{p}.weight', d, d)
            mat(f'layers.{i}.{p}.bias', d)
        mat(f'layers.{i}.fc1.weight', 4 * d, d)
        mat(f'layers.{i}.fc1.bias', 4 * d)
        mat(f'layers.{i}.fc2.weight', d, 4 * d)
        mat(f'layers.{i}.fc2.bias', d)
    ln('final_layer_norm')
    return sd


def clip_key_to_transformers(k):
    """Our CLIPTextModel key -> the key of transformers' CLIPTextModel (what the reference instantiates,
    DS_NeRF/guidance/sd_utils.py:69-74 through the diffusers pipeline)."""
    if k.startswith(('token_embedding', 'position_embedding')):
        return 'embeddings.' + k
    if k.startswith('final_layer_norm'):
        return k
    _, i, rest = k.split('.', 2)
    if rest.startswith(('q_proj', 'k_proj', 'v_proj', 'out_proj')):
        return f'encoder.layers.{i}.self_attn.{rest}'
    if rest.startswith(('fc1', 'fc2')):
        return f'encoder.layers.{i}.mlp.{rest}'
    return f'encoder.layers.{i}.{rest}'
