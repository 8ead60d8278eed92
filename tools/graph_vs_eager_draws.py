"""Do the graphed and the eager SDS step draw the SAME random numbers from the same seed?  Records every _randn result of
one eager step and of the first graph replay after re-seeding (full-size networks), prints per-draw equality and the
relative difference of the image gradients."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd.guidance.sd_utils import StableDiffusion      # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
sd = StableDiffusion(dev, False, False, use_graphs=False)
g = torch.Generator(device=dev).manual_seed(2)
base = torch.rand(1, 3, 378, 504, device=dev, generator=g)
mask = torch.zeros(1, 1, 378, 504, device=dev)
mask[:, :, 137:241, 196:307] = 1
rec = []
orig = StableDiffusion._randn


def spy(self, shape, dtype=torch.float32):
    t = orig(self, shape, dtype)
    rec.append(t)
    return t


StableDiffusion._randn = spy
out = {}
res = {}
for private in (False, True):
    for mode in ('eager', 'graph', 'graph2'):
        sd.use_graphs = mode != 'eager'
        if private:
            sd.seed_generator(77)
        else:
            sd.generator = None
            torch.cuda.manual_seed(77)
        rec.clear()
        pred = base.clone().requires_grad_(True)
        (1e-4 * sd.train_step_sd(1000, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
        torch.cuda.synchronize()
        key = ('private_' if private else 'default_') + mode
        # in graph mode the recorded tensors of the capture are the graph's static buffers: after the replay they hold this
        # replay's draws; a replay without capture records nothing new (graph2): reuse the capture's list
        if rec:
            res[key + '_draws'] = [t.clone() for t in rec[-4:]]
            keep = list(rec[-4:])
        else:
            res[key + '_draws'] = [t.clone() for t in keep]
        res[key + '_grad'] = pred.grad.clone()
    for mode in ('graph', 'graph2'):
        k = ('private_' if private else 'default_')
        e, gdr = res[k + 'eager_draws'], res[k + mode + '_draws']
        out[k + mode] = {'draws_equal': [bool(torch.equal(a, b)) for a, b in zip(e, gdr)],
                         'grad_rel_diff': float((res[k + mode + '_grad'] - res[k + 'eager_grad']).norm() / res[k + 'eager_grad'].norm())}
print(json.dumps(out, indent=1))
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/r4_graph_vs_eager_draws.json', 'w'), indent=1)
