"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement of FreeFine's sampling loops and task entry points (/root/reference/src/demo/model.py):
  invert :816-925, forward_sampling :476-622, forward_sampling_background_gen :656-812,
  forward_sampling_compose :301-435, DDIM_inversion_func :1341-1364, Details_Preserving_regeneration* :1640-1804,
  FreeFine_generation :1012-1049, FreeFine_background_generation :1088-1118.
Quirks kept on purpose: the reference-stream latent at denoise step i is one level cleaner
than t_i in the edit/compose loops (model.py:582, 394) but aligned in bg-gen (:756); `local_edit_text` is always True
on the edit path (model.py:1692 passes it under the wrong keyword); masks stay uint8.

Pinned by tests/golden/g5_*.npz (trajectories produced by the imported reference driving the same oracle UNet
modules through its own hooks).
"""
from copy import deepcopy

import numpy as np
import torch

from . import masks as M
from . import scheduler as Sch
from .attention_modulation import Modulator


class OraclePipeline:
    def __init__(self, unet, vae, text_embed, sched=None):
        """unet: oracle.sd_unet.UNet2DConditionModel; vae: oracle.sd_vae.AutoencoderKL; text_embed(list[str]) -> [N,77,D]"""
        self.unet, self.vae, self.text_embed = unet, vae, text_embed
        self.sched = sched or Sch.DDIMSchedule()
        self.modulator = Modulator("edit", num_att_layers=len(unet.attention_modules()))
        unet.set_modulator(self.modulator)

    def set_hook(self, hook):  # register_attention_control{,_4bggen,_compose}
        self.modulator.hook = hook

    # -- VAE bracket -----------------------------------------------------------------------------------------
    @staticmethod
    def preprocess_image(img_u8):
        return (torch.from_numpy(img_u8).float() / 127.5 - 1).permute(2, 0, 1)[None]

    def image2latent(self, x):
        return self.vae.encode_mean(x) * 0.18215

    def latent2image(self, z):
        return (self.vae.decode(z / 0.18215) / 2 + 0.5).clamp(0, 1)

    # -- loops -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def invert(self, image, prompt, num_inference_steps, num_actual_inference_steps):
        text = self.text_embed([prompt] * image.shape[0])
        latents = self.image2latent(image)
        self.sched.set_timesteps(num_inference_steps)
        lst = [latents]
        for i, t in enumerate(reversed(self.sched.timesteps)):
            if i >= num_actual_inference_steps:
                continue
            eps = self.unet(latents, t, text)
            latents, _ = Sch.inv_step(self.sched, eps, t, latents)
            lst.append(latents)
        return latents, lst

    def _configure(self, method_type):
        c = self.modulator
        if method_type == "tca":
            c.use_tca, c.layer_idx, c.method = True, list(range(10, 16)), "tca"
        elif method_type in ("mmsa", "mmsa_es"):
            c.use_tca, c.layer_idx, c.method = True, list(range(10, 16)), "mmsa"
        elif method_type in ("ssa", "sdsa"):
            c.use_style_align, c.method = True, method_type

    def _step_schedule(self, method_type, i, start_step, end_step, n, end_scale):
        if method_type == "tca":
            self.modulator.context_guidance = Sch.linear_param(i, start_step, end_step, n, end_scale=end_scale)
        elif method_type == "mmsa_es" and i >= end_step:
            self.modulator.use_tca = False

    @torch.no_grad()
    def forward_sampling(self, prompts, refer_latents, latents, end_step, num_inference_steps, num_actual_inference_steps,
                         guidance_scale, eta, end_scale, local_var_reg, completion_mask_cfg, method_type="tca",
                         local_perturbation=True, mode="edit"):
        """mode 'edit' (model.py:476-622) or 'bggen' (:656-812)."""
        assert guidance_scale > 1.0
        self._configure(method_type)
        self.modulator.local_edit = True
        text = torch.cat([self.text_embed([""] * len(prompts)), self.text_embed(prompts)], dim=0)
        self.sched.set_timesteps(num_inference_steps)
        start_step = num_inference_steps - num_actual_inference_steps
        lst = [latents]
        for i, t in enumerate(self.sched.timesteps):
            if i < start_step:
                continue
            if mode == "edit":
                ref = refer_latents[i - start_step + 1][1]
                if latents.shape[0] > 1:
                    latents[1:] = ref
                else:
                    latents = torch.cat([latents, ref])
            else:
                ref = refer_latents[i - start_step]
                if latents.shape[0] > 1:
                    latents = latents[0].unsqueeze(0)
                latents = torch.cat([latents, ref], dim=0)
            self._step_schedule(method_type, i, start_step, end_step, num_inference_steps, end_scale)
            eps = self.unet(torch.cat([latents] * 2), t, text)
            eu, ec = eps.chunk(2, dim=0)
            eps = eu + guidance_scale * (ec - eu) * completion_mask_cfg
            mask = local_var_reg if local_perturbation else torch.ones_like(local_var_reg)
            noise = torch.randn(eps.shape) if eta > 0 else None
            latents = Sch.ctrl_step(self.sched, eps, t, latents, mask, eta, noise)[0]
            lst.append(latents if mode == "edit" else latents[0])
        return self.latent2image(latents), lst

    @torch.no_grad()
    def forward_sampling_compose(self, prompts, refer_latents, latents, end_step, num_inference_steps,
                                 num_actual_inference_steps, guidance_scale, eta, end_scale, local_var_reg, cfg_masks_tensor,
                                 method_type="tca", local_perturbation=True):
        assert guidance_scale > 1.0
        self._configure(method_type)
        self.modulator.local_edit = True
        prompts = list(prompts) + [""]
        self.modulator.prompt_length = len(prompts)
        text = torch.cat([self.text_embed([""] * latents.shape[0]), self.text_embed(prompts)], dim=0)
        self.sched.set_timesteps(num_inference_steps)
        start_step = num_inference_steps - num_actual_inference_steps
        lst = [latents]
        for i, t in enumerate(self.sched.timesteps):
            if i < start_step:
                continue
            ref = refer_latents[i - start_step + 1][1:]
            if latents.shape[0] > 1:
                latents[1:] = ref
            else:
                latents = torch.cat([latents, ref])
            self._step_schedule(method_type, i, start_step, end_step, num_inference_steps, end_scale)
            eps = self.unet(torch.cat([latents, latents[0][None]]), t, text)
            eu, ec = eps[0][None], eps[-1][None]
            eps = eu + guidance_scale * (ec - eu) * cfg_masks_tensor
            mask = local_var_reg if local_perturbation else torch.ones_like(local_var_reg)
            noise = torch.randn(eps.shape) if eta > 0 else None
            latents = Sch.ctrl_step(self.sched, eps, t, latents[0][None], mask, eta, noise)[0]
            lst.append(latents[0])
        return self.latent2image(latents)[0], lst

    # -- task entry points -----------------------------------------------------------------------------------
    def ddim_inversion(self, img, ref_imgs, num_step, start_step):
        src = self.preprocess_image(img)
        for r in ref_imgs:
            src = torch.cat((src, self.preprocess_image(r)))
        _, lst = self.invert(src, "", num_step, num_step - start_step)
        self.modulator.reset()
        return lst

    def freefine_generation(self, ori_img, ori_mask, coarse_input, target_mask, guidance_text, guidance_scale, eta,
                            end_step=10, num_step=50, start_step=25, method_type="tca", local_perturbation=True, seed=42,
                            draw_mask=None, use_auto_draw=False, cons_area=None, reduce_inp_artifacts=False, end_scale=0.5):
        torch.manual_seed(seed)
        red = lambda m: m[:, :, 0] if (m is not None and m.ndim == 3) else m
        ori_mask, target_mask, draw_mask = red(ori_mask), red(target_mask), red(draw_mask)
        self.set_hook("edit")
        lst = self.ddim_inversion(coarse_input, [ori_img], num_step, start_step)
        start = deepcopy(lst[-1])
        H, W = coarse_input.shape[:2]
        # NB the float mask DDIM_inversion_func returns (model.py:1364) is dropped: regeneration receives target_mask (:1031-1033)
        fg, shifted_t, ori_t, cfg_m, var_m = M.prepare_various_mask(target_mask, ori_mask, draw_mask, H, W, tuple(start.shape[2:]),
                                                                    use_auto_draw=use_auto_draw, cons_area=cons_area,
                                                                    reduce_inp_artifacts=reduce_inp_artifacts)
        c = self.modulator
        c.fg_retain_mask, c.fg_retain_mask_st2, c.fg_ref_mask, c.local_edit_region = fg, shifted_t, ori_t, fg
        c.reset()
        imgs, traj = self.forward_sampling([guidance_text, ""], lst[::-1], start, end_step, num_step, num_step - start_step,
                                           guidance_scale, eta, end_scale, var_m, cfg_m, method_type, local_perturbation, "edit")
        c.reset()
        to_u8 = lambda im: (im.permute(1, 2, 0).numpy() * 255).astype(np.uint8)
        return to_u8(imgs[0]), to_u8(imgs[1]), traj

    def freefine_background_generation(self, ori_img, ori_mask, guidance_text, guidance_scale, eta, end_step=10, num_step=50,
                                       start_step=25, method_type="tca", local_perturbation=True, seed=42, end_scale=0.5):
        torch.manual_seed(seed)
        if ori_mask.ndim == 3:
            ori_mask = ori_mask[:, :, 0]
        self.set_hook("bggen")
        lst = self.ddim_inversion(ori_img, [], num_step, start_step)
        start = deepcopy(lst[-1])
        H, W = ori_img.shape[:2]
        mask_t, var_m = M.prepare_mask_bggen(ori_mask, H, W, tuple(start.shape[2:]))
        c = self.modulator
        c.fg_retain_mask, c.local_edit_region = mask_t, mask_t
        c.reset()
        imgs, traj = self.forward_sampling([guidance_text, ""], lst[::-1], start, end_step, num_step, num_step - start_step,
                                           guidance_scale, eta, end_scale, var_m, var_m, method_type, local_perturbation, "bggen")
        c.reset()
        return (imgs[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8), traj

    def freefine_compose(self, img_lists, ori_mask_lists, tgt_mask_lists, coarse_input, guidance_text_list, guidance_scale, eta,
                         end_step=10, num_step=50, start_step=25, method_type="tca", local_perturbation=True, seed=42,
                         draw_mask=None, end_scale=0.5, dil_completion=False, dil_factor=15, appearance_transfer=False):
        torch.manual_seed(seed)
        red = lambda m: m[:, :, 0] if m.ndim == 3 else m
        ori_mask_lists, tgt_mask_lists = [red(m) for m in ori_mask_lists], [red(m) for m in tgt_mask_lists]
        self.set_hook("compose")
        lst = self.ddim_inversion(coarse_input, img_lists, num_step, start_step)
        start = deepcopy(lst[-1])
        H, W = coarse_input.shape[:2]
        tgt_t, ori_t, lp, cfg_m = M.prepare_composition_masks(ori_mask_lists, tgt_mask_lists, H, W, tuple(start.shape[2:]),
                                                              dil_completion, dil_factor, draw_mask, appearance_transfer)
        c = self.modulator
        c.src_masks, c.tgt_masks = ori_t, tgt_t
        c.reset()
        img, traj = self.forward_sampling_compose(list(guidance_text_list), lst[::-1], start, end_step, num_step,
                                                  num_step - start_step, guidance_scale, eta, end_scale, lp, cfg_m, method_type,
                                                  local_perturbation)
        c.reset()
        return (img.permute(1, 2, 0).numpy() * 255).astype(np.uint8), traj
