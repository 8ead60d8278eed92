"""Mirror of the live part of `DS_NeRF/nerf/utils.py`: `Pretrain_Model.cal_loss`, the dispatcher that
sums the RGB / collaborative / normal SDS terms (DS_NeRF/nerf/utils.py:174-311).

The reference also draws `rand_poses` each call and uses only `phis` to compute a value that is
never read (utils.py:239-254).  The pose algebra is dead code and is not restated, but its RANDOM DRAWS are
consumed in the same order and shapes when `reference_rng` is on (the default: `guidance['SD'].reference_rng`),
so that a seeded run sees the same device random stream as the reference: torch.rand(B) for the radius, one
python `random.random()` for the branch, then either three torch.randn(B) (uniform-sphere branch) or two
torch.rand(B) (utils.py:119-135).  The Perp-Neg / text-direction helpers nothing calls (utils.py:8-98) are not
restated.  `global_step` is counted as in the reference.
"""
import random

import torch


class Pretrain_Model(object):
    def __init__(self, opt, device, guidance):
        self.opt = opt
        self.device = device
        self.global_step = 0
        self.guidance = guidance              # {'SD': StableDiffusion}
        self.embeddings = {}
        if self.guidance is not None:
            for key in self.guidance:
                for p in self.guidance[key].parameters():
                    p.requires_grad = False
                self.embeddings[key] = {}

    def _reference_rng(self):
        sd = self.guidance.get('SD') if isinstance(self.guidance, dict) or hasattr(self.guidance, 'get') else None
        return bool(getattr(sd, 'reference_rng', True))

    def _consume_rand_poses_draws(self, B):
        """The draws of rand_poses(B, ...) (DS_NeRF/nerf/utils.py:119-135), values discarded."""
        dev = self.device
        torch.rand(B, device=dev)
        if random.random() < float(getattr(self.opt, 'uniform_sphere_rate', 0)):
            for _ in range(3):
                torch.randn(B, device=dev)
        else:
            torch.rand(B, device=dev)
            torch.rand(B, device=dev)

    def cal_loss(self, i, rgbs4_tensor, pre_normal_map, pred_depth, pred_rgb, rgb, masks, mask4, B=1):
        """Signature and term order of DS_NeRF/nerf/utils.py:222-311."""
        opt = self.opt
        self.rgb, self.pred_rgb, self.pred_depth = rgb, pred_rgb, pred_depth
        self.pre_normal_map, self.rgbs4_tensor = pre_normal_map, rgbs4_tensor
        self.B, self.masks = B, masks
        if self._reference_rng():
            self._consume_rand_poses_draws(B)
        self.global_step += 1
        loss = 0
        if 'SD' in self.guidance:
            sd = self.guidance['SD']
            terms = []                                  # in the reference's order (its random draws are made in this order)
            if opt.is_rgb_guidance:
                terms.append(lambda: sd.train_step_sd(i, masks, opt.text, self.pred_rgb, as_latent=True,
                                                      guidance_scale=opt.rgb_guidance_scale,
                                                      grad_scale=opt.lambda_guidance,
                                                      save_guidance_path=getattr(opt, 'save_guidance_path', None)))
            if opt.is_colla_guidance and i > 0:
                terms.append(lambda: sd.train_step_colla_sds(i, mask4, opt.text, self.rgbs4_tensor, as_latent=True,
                                                             guidance_scale=opt.colla_guidance_scale,
                                                             grad_scale=opt.lambda_guidance,
                                                             save_guidance_path=getattr(opt, 'save_guidance_path', None)))
            if opt.is_normal_guidance and i > opt.normal_start:
                terms.append(lambda: sd.train_step_sd_normal(
                    i, masks, opt.text_normal, self.pre_normal_map, as_latent=True,
                    guidance_scale=opt.normal_guidance_scale, normal_start=opt.normal_start,
                    grad_scale=opt.lambda_guidance, save_guidance_path=getattr(opt, 'save_guidance_path', None)))
            streams = self._term_streams(sd, len(terms))
            if streams is None:
                # a model configured for SEVERAL terms keeps even its single-term iterations (before normal_start, i == 0) off the
                # default stream: a captured step replayed there would cost every later multi-term iteration its concurrency
                # (guidance/sd_utils._OffDefaultStream; a one-term model stays on the default stream, which is faster for it)
                hop = None
                if ((getattr(opt, 'is_normal_guidance', False) or getattr(opt, 'is_colla_guidance', False))
                        and getattr(sd, 'use_graphs', False) and torch.device(self.device).type == 'cuda'):
                    from ..guidance.sd_utils import _OffDefaultStream
                    hop = _OffDefaultStream(self.device)
                if hop is None:
                    for fn in terms:
                        loss = loss + fn()
                else:
                    with hop:
                        outs = [fn() for fn in terms]
                        hop.keep(*outs)
                    for out in outs:
                        loss = loss + out
            else:
                # The terms are independent diffusion-prior evaluations (the reference runs them one after the other,
                # DS_NeRF/nerf/utils.py:280-302): each replays its captured step on a stream of its own -- thousands of small
                # launches whose boundaries and bandwidth-bound passes fill each other's gaps -- and the sum joins them.  The host
                # issues them in the reference's order, so the random draws keep their order.  MVIP_SDS_TERM_STREAMS=0: in line.
                cur = torch.cuda.current_stream(self.device)
                outs = []
                for fn, st in zip(terms, streams):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        out = fn()
                    if torch.is_tensor(out):
                        out.record_stream(cur)
                    outs.append(out)
                for st in streams:
                    cur.wait_stream(st)
                for out in outs:
                    loss = loss + out
        return loss

    def _term_streams(self, sd, n):
        """One stream per SDS term when the terms are captured steps on a GPU (n >= 2), else None."""
        import os
        if n < 2 or os.environ.get('MVIP_SDS_TERM_STREAMS', '1') == '0' or not getattr(sd, 'use_graphs', False):
            return None
        if torch.device(self.device).type != 'cuda':
            return None
        from .. import streams as _streams               # a process-wide pool: see streams.py (hardware queues are few)
        pool = _streams.term_streams(self.device, n)
        self.__dict__['_streams'] = pool
        return pool
