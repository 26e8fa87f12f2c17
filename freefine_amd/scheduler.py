"""DDIM schedule with the surface the reference touches on diffusers' DDIMScheduler (/root/reference/src/demo/model.py:
123-127, 157-160, 200-209, 384): `.alphas_cumprod`, `.final_alpha_cumprod`, `.config.num_train_timesteps`,
`.num_inference_steps`, `.set_timesteps(n)`, `.timesteps`, `DDIMScheduler.from_config(cfg)`.
Defaults are SD's scheduler_config.json (scaled_linear 0.00085 -> 0.012, 1000 steps, steps_offset=1,
set_alpha_to_one=False, leading spacing): timesteps for n=50 are 981, 961, ..., 1."""
from types import SimpleNamespace

import numpy as np
import torch


class DDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1, set_alpha_to_one=False, clip_sample=False, prediction_type="epsilon", **kw):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(beta_schedule)
        if prediction_type != "epsilon":
            raise ValueError("FreeFine's inv_step/ctrl_step assume epsilon prediction (SD-2.1-base, not the 768-v model)")
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, steps_offset=steps_offset, set_alpha_to_one=set_alpha_to_one,
                                      clip_sample=clip_sample, prediction_type=prediction_type)
        self.num_inference_steps = None
        self.timesteps = None

    @classmethod
    def from_config(cls, config, **kw):
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        d.update(kw)
        return cls(**{k: v for k, v in d.items() if not k.startswith("_")})

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)
