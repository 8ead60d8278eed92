"""Loading a diffusers-layout Stable-Diffusion checkpoint DIRECTORY into `sd_nets.SDNetworks` (host logic only).

The reference obtains its networks with
    StableDiffusionInpaintPipeline.from_pretrained("runwayml/stable-diffusion-inpainting", revision="fp16", torch_dtype=...)
(DS_NeRF/guidance/sd_utils.py:69-74) and tokenizes / encodes prompts through `pipe._encode_prompt` (:317-326).  Neither the
libraries' hub access nor the weights exist offline, so `from_pretrained` by NAME cannot be mirrored; what can be is the layout
such a download has on disk:

    <dir>/unet/diffusion_pytorch_model[.fp16].safetensors | .bin          UNet2DConditionModel   (686 tensors)
    <dir>/vae/diffusion_pytorch_model[.fp16].safetensors  | .bin          AutoencoderKL          (248 tensors)
    <dir>/text_encoder/model[.fp16].safetensors | pytorch_model[.fp16].bin  CLIPTextModel        (196 tensors)
    <dir>/tokenizer/vocab.json + merges.txt                               CLIP BPE tokenizer (optional)

`StableDiffusion(device, fp16, vram_O, hf_key=<dir>)` calls `load_into(networks, <dir>)`:
  * every file is read (safetensors, or torch.load(weights_only=True) for .bin), keys are mapped onto the module names of
    sd_nets (they ARE diffusers' names for the UNet / VAE; the text tower maps transformers' `text_model.*` names; the VAE
    attention block's pre-0.15 names query / key / value / proj_attn are accepted), and loaded with STRICT key and shape
    equality against the committed manifest `sd_checkpoint_manifest.json` (names + shapes only, generated from the published
    architecture as sd_nets builds it): a missing, extra or mis-shaped tensor refuses the whole directory with the lists;
  * the values are kept in fp32 containers (what the reference's default mode does with its fp16 files); whether EVERY UNet /
    VAE weight is an exact fp16 value is recorded in `networks.fp16_weights` -- that is what selects the two-product
    contractions (ops.TWO_PRODUCT); a non-fp16 checkpoint keeps three products;
  * with tokenizer files present the prompt goes through `transformers.CLIPTokenizer` (pad to 77 with the model's pad token,
    truncate), else the byte-level stand-in stays, with a warning.
"""
import json
import os
import re
import warnings

import torch

MANIFEST_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sd_checkpoint_manifest.json')

_WEIGHT_FILES = {
    'unet': ('diffusion_pytorch_model.safetensors', 'diffusion_pytorch_model.fp16.safetensors',
             'diffusion_pytorch_model.bin', 'diffusion_pytorch_model.fp16.bin'),
    'vae': ('diffusion_pytorch_model.safetensors', 'diffusion_pytorch_model.fp16.safetensors',
            'diffusion_pytorch_model.bin', 'diffusion_pytorch_model.fp16.bin'),
    'text_encoder': ('model.safetensors', 'model.fp16.safetensors', 'pytorch_model.bin', 'pytorch_model.fp16.bin'),
}


class CheckpointError(RuntimeError):
    pass


# ---- key maps: checkpoint name -> sd_nets name --------------------------------------------------------------------------
_VAE_ATTN_OLD = {'query': 'to_q', 'key': 'to_k', 'value': 'to_v', 'proj_attn': 'to_out.0'}


def map_key(component, k):
    """Name of checkpoint tensor `k` inside the sd_nets module of `component`, or None for a tensor that carries no weight
    (transformers' `position_ids` buffer)."""
    if component == 'vae':
        m = re.match(r'^((?:encoder|decoder)\.mid_block\.attentions\.0\.)(query|key|value|proj_attn)\.(weight|bias)$', k)
        if m:
            return f'{m.group(1)}{_VAE_ATTN_OLD[m.group(2)]}.{m.group(3)}'
        return k
    if component == 'text_encoder':
        if k.endswith('position_ids'):
            return None
        k = k[len('text_model.'):] if k.startswith('text_model.') else k
        if k.startswith('embeddings.'):
            return k[len('embeddings.'):]
        m = re.match(r'^encoder\.layers\.(\d+)\.(?:self_attn\.|mlp\.)?(.+)$', k)
        if m:
            return f'layers.{m.group(1)}.{m.group(2)}'
        return k
    return k


def unmap_key(component, k):
    """The checkpoint's (current diffusers / transformers) name of sd_nets parameter `k`: what the manifest lists."""
    if component != 'text_encoder':
        return k
    if k.startswith(('token_embedding', 'position_embedding')):
        return 'text_model.embeddings.' + k
    m = re.match(r'^layers\.(\d+)\.(.+)$', k)
    if m:
        rest = m.group(2)
        if rest.startswith(('q_proj', 'k_proj', 'v_proj', 'out_proj')):
            return f'text_model.encoder.layers.{m.group(1)}.self_attn.{rest}'
        if rest.startswith(('fc1', 'fc2')):
            return f'text_model.encoder.layers.{m.group(1)}.mlp.{rest}'
        return f'text_model.encoder.layers.{m.group(1)}.{rest}'
    return 'text_model.' + k


def build_manifest():
    """{component: {checkpoint key: [shape]}} from the modules sd_nets builds (meta device: no memory, no arithmetic)."""
    from . import sd_nets
    with torch.device('meta'):
        mods = {'unet': sd_nets.UNet2DConditionModel(), 'vae': sd_nets.AutoencoderKL(), 'text_encoder': sd_nets.CLIPTextModel()}
    return {c: {unmap_key(c, k): list(v.shape) for k, v in m.state_dict().items()} for c, m in mods.items()}


def manifest():
    with open(MANIFEST_PATH) as f:
        return json.load(f)


# ---- reading ------------------------------------------------------------------------------------------------------------
def find_weight_file(root, component):
    d = os.path.join(root, component)
    for name in _WEIGHT_FILES[component]:
        p = os.path.join(d, name)
        if os.path.isfile(p):
            return p
    raise CheckpointError(f'{d}: none of {list(_WEIGHT_FILES[component])} found -- not a diffusers-layout checkpoint directory')


def read_state(path):
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        return load_file(path, device='cpu')
    sd = torch.load(path, map_location='cpu', weights_only=True)
    if isinstance(sd, dict) and 'state_dict' in sd and isinstance(sd['state_dict'], dict):
        sd = sd['state_dict']
    return sd


def check_against_manifest(component, state, man=None):
    """Strict: the mapped key set and every shape must equal the manifest's.  Returns {sd_nets key: tensor}."""
    man = (manifest() if man is None else man)[component]
    want = {map_key(component, k): tuple(shp) for k, shp in man.items()}
    got = {}
    for k, v in state.items():
        mk = map_key(component, k)
        if mk is None:
            continue
        if mk in got:
            raise CheckpointError(f'{component}: two tensors map to {mk!r}')
        if mk != k and component == 'vae' and v.dim() == 4 and v.shape[2:] == (1, 1):
            v = v[:, :, 0, 0]                               # the oldest files hold the attention projections as 1x1 convolutions
        got[mk] = v
    missing = sorted(set(want) - set(got))
    unexpected = sorted(set(got) - set(want))
    shapes = sorted(f'{k}: {tuple(got[k].shape)} != {want[k]}' for k in set(got) & set(want) if tuple(got[k].shape) != want[k])
    if missing or unexpected or shapes:
        def few(xs):
            return ', '.join(xs[:6]) + (f', ... ({len(xs)} in all)' if len(xs) > 6 else '')
        raise CheckpointError(f'{component}: checkpoint does not match the SD-1.5-inpainting architecture'
                              + (f'; missing: {few(missing)}' if missing else '')
                              + (f'; unexpected: {few(unexpected)}' if unexpected else '')
                              + (f'; shapes: {few(shapes)}' if shapes else ''))
    return got


def is_fp16_exact(t):
    t = t.detach()
    if t.dtype == torch.float16:
        return True
    if not t.is_floating_point():
        return True
    f = t.float()
    return bool(torch.equal(f.half().float(), f))


def load_component(module, component, path, man=None):
    """Strict load of one weight file into `module` (values cast to the module's parameter dtype); returns whether every
    tensor was an exact fp16 value."""
    state = check_against_manifest(component, read_state(path), man)
    exact = all(is_fp16_exact(v) for v in state.values())
    own = module.state_dict()
    if set(own) != set(state):                             # the manifest and the module are generated from the same code
        raise CheckpointError(f'{component}: module / manifest key sets differ (stale sd_checkpoint_manifest.json?)')
    with torch.no_grad():
        for k, dst in own.items():
            dst.copy_(state[k].to(dtype=dst.dtype))
    return exact


class CLIPBPETokenizer:
    """transformers.CLIPTokenizer behind the call the SDS code makes: prompt -> [1, 77] int64 ids (BOS ... EOS, padded with
    the tokenizer's pad token, truncated), as `pipe.tokenizer(prompt, padding='max_length', max_length=77, truncation=True)`
    inside `_encode_prompt` (DS_NeRF/guidance/sd_utils.py:317)."""
    CTX = 77

    def __init__(self, tokenizer_dir):
        from transformers import CLIPTokenizer
        self.tok = CLIPTokenizer(os.path.join(tokenizer_dir, 'vocab.json'), os.path.join(tokenizer_dir, 'merges.txt'))

    def __call__(self, prompt):
        enc = self.tok(prompt, padding='max_length', max_length=self.CTX, truncation=True, return_tensors='pt')
        return enc['input_ids'].to(torch.long)


def load_into(networks, root):
    """Fill `networks` (sd_nets.SDNetworks) from the checkpoint directory `root`; returns a report dict.  Refuses (raises
    CheckpointError, nothing half-loaded is left in use: the caller builds `networks` for this call) on any mismatch."""
    root = os.fspath(root)
    if not os.path.isdir(root):
        raise CheckpointError(f'{root}: not a directory (hub names cannot be resolved offline: pass a local diffusers-layout directory)')
    man = manifest()
    report = {'root': root, 'files': {}, 'fp16_exact': {}}
    for comp, mod in (('unet', networks.unet), ('vae', networks.vae), ('text_encoder', networks.text_encoder)):
        path = find_weight_file(root, comp)
        report['files'][comp] = os.path.relpath(path, root)
        report['fp16_exact'][comp] = load_component(mod, comp, path, man)
    # two-product contractions need EVERY frozen UNet / VAE weight to be an exact fp16 value (the packers check again per image)
    networks.fp16_weights = bool(report['fp16_exact']['unet'] and report['fp16_exact']['vae'])
    tdir = os.path.join(root, 'tokenizer')
    if os.path.isfile(os.path.join(tdir, 'vocab.json')) and os.path.isfile(os.path.join(tdir, 'merges.txt')):
        networks.tokenizer = CLIPBPETokenizer(tdir)
        report['tokenizer'] = 'CLIPTokenizer (tokenizer/vocab.json + merges.txt)'
    else:
        warnings.warn(f'{root}: no tokenizer/vocab.json + merges.txt -- prompts are tokenized by the byte-level stand-in, '
                      'which does NOT produce CLIP token ids; embeddings of real prompts will be wrong', RuntimeWarning)
        report['tokenizer'] = 'ByteTokenizer (stand-in)'
    for m in (networks.vae, networks.unet, networks.text_encoder):
        for p in m.parameters():
            p.requires_grad_(False)
    networks._cache.clear()                                   # prompt embeddings of the previous weights
    networks.checkpoint = report
    return report


if __name__ == '__main__':                                     # regenerate the manifest: python -m mvip_nerf_amd.guidance.sd_checkpoint
    man_ = build_manifest()
    with open(MANIFEST_PATH, 'w') as f_:
        json.dump(man_, f_, indent=0, sort_keys=True)
    print({c: len(v) for c, v in man_.items()}, os.path.getsize(MANIFEST_PATH), 'bytes')
