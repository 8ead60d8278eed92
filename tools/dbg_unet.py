import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvip_nerf_amd.guidance import sd_nets, transformer_cm
cuda = torch.device('cuda:0')
torch.manual_seed(0)
unet = sd_nets.UNet2DConditionModel().to(cuda).eval()
for p in unet.parameters():
    p.requires_grad_(False)
x = torch.randn(2, 9, 64, 64, device=cuda)
ctx = torch.randn(2, 77, 768, device=cuda)
t = torch.tensor(417, device=cuda)
outs = {}
for sinks in (False, True):
    transformer_cm.USE_SINKS = sinks
    bad = []
    hooks = []
    for name, m in unet.named_modules():
        if isinstance(m, sd_nets.Transformer2DModel):
            hooks.append(m.register_forward_hook(lambda mod, i, o, name=name: bad.append((name, tuple(o.shape), bool(torch.isfinite(o).all()), float(o.abs().max())))))
    with torch.no_grad():
        outs[sinks] = unet(x, t, encoder_hidden_states=ctx)[0]
    for h in hooks:
        h.remove()
    print('sinks', sinks, 'finite', bool(torch.isfinite(outs[sinks]).all()))
    for b in bad:
        print('   ', b)
d = (outs[True] - outs[False]).abs().max() / outs[False].abs().max()
print('rel diff sinks vs not', float(d))
for name, m in unet.named_modules():
    pk = m.__dict__.get('_mvip_cm')
    if pk is not None:
        print(name, 'C', pk.C, 'scales q k v q2 act', pk.s_q1, pk.s_k1, pk.s_v1, pk.s_q2, pk.s_act)
