"""Golden vectors for the CLIP text tower: the reference encodes its prompts with transformers' `CLIPTextModel`
(DS_NeRF/guidance/sd_utils.py:69-74, :317 through the diffusers pipeline's `_encode_prompt`).  That class IS importable
in the build container (transformers 5.15), so this one network BODY of the SDS path can be pinned: the library model is
built from the published ViT-L/14 text configuration, filled with seeded weights (no checkpoint exists offline), run on
two token sequences, and its last hidden state stored.  tests/test_host_cpu.py loads the same seeded weights into
`sd_nets.CLIPTextModel` and compares.  (The BPE tokenizer's vocabulary files are not available offline: token ids come
from the byte-level stand-in; the UNet / VAE bodies stay unpinned -- diffusers is absent.)

    python oracle/gen_golden_clip.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def main():
    from transformers import CLIPTextConfig, CLIPTextModel
    from oracle.weights import seeded_clip_text_state, clip_key_to_transformers
    from mvip_nerf_amd.guidance.sd_nets import ByteTokenizer
    cfg = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                         num_attention_heads=12, max_position_embeddings=77, hidden_act='quick_gelu', layer_norm_eps=1e-5,
                         projection_dim=768, bos_token_id=49406, eos_token_id=49407, pad_token_id=1)
    model = CLIPTextModel(cfg).eval()
    seed = 4242
    ours = seeded_clip_text_state(seed)
    mapped = {clip_key_to_transformers(k): torch.from_numpy(v) for k, v in ours.items()}
    missing, unexpected = model.load_state_dict(mapped, strict=False)
    assert not unexpected and all('position_ids' in m for m in missing), (missing, unexpected)
    tok = ByteTokenizer()
    ids = torch.cat([tok('a stone bench in a park'), tok('')], 0)
    with torch.no_grad():
        out = model(input_ids=ids)[0]
    path = os.path.join(ROOT, 'tests', 'golden', 'clip_text.npz')
    np.savez_compressed(path, seed=seed, ids=ids.numpy(), last_hidden_state=out.numpy().astype(np.float32))
    print('clip_text:', os.path.getsize(path) / 1024, 'KB', out.shape, float(out.abs().max()))


if __name__ == '__main__':
    main()
