"""Mirror of the live part of `DS_NeRF/nerf/utils.py`: `Pretrain_Model.cal_loss`, the dispatcher that
sums the RGB / collaborative / normal SDS terms (DS_NeRF/nerf/utils.py:174-311).

The reference also draws `rand_poses` each call and uses only `phis` to compute a value that is
never read (utils.py:239-254); that dead code -- and the Perp-Neg / text-direction helpers nothing
calls (utils.py:8-98) -- is not restated.  `global_step` is still counted.
"""


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

    def cal_loss(self, i, rgbs4_tensor, pre_normal_map, pred_depth, pred_rgb, rgb, masks, mask4, B=1):
        """Signature and term order of DS_NeRF/nerf/utils.py:222-311."""
        opt = self.opt
        self.rgb, self.pred_rgb, self.pred_depth = rgb, pred_rgb, pred_depth
        self.pre_normal_map, self.rgbs4_tensor = pre_normal_map, rgbs4_tensor
        self.B, self.masks = B, masks
        self.global_step += 1
        loss = 0
        if 'SD' in self.guidance:
            sd = self.guidance['SD']
            if opt.is_rgb_guidance:
                loss = loss + sd.train_step_sd(i, masks, opt.text, self.pred_rgb, as_latent=True,
                                               guidance_scale=opt.rgb_guidance_scale,
                                               grad_scale=opt.lambda_guidance,
                                               save_guidance_path=getattr(opt, 'save_guidance_path', None))
            if opt.is_colla_guidance and i > 0:
                loss = loss + sd.train_step_colla_sds(i, mask4, opt.text, self.rgbs4_tensor, as_latent=True,
                                                      guidance_scale=opt.colla_guidance_scale,
                                                      grad_scale=opt.lambda_guidance,
                                                      save_guidance_path=getattr(opt, 'save_guidance_path', None))
            if opt.is_normal_guidance and i > opt.normal_start:
                loss = 1.0 * loss + 1.0 * sd.train_step_sd_normal(
                    i, masks, opt.text_normal, self.pre_normal_map, as_latent=True,
                    guidance_scale=opt.normal_guidance_scale, normal_start=opt.normal_start,
                    grad_scale=opt.lambda_guidance, save_guidance_path=getattr(opt, 'save_guidance_path', None))
        return loss
