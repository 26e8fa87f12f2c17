"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement of the DDIM schedule and of FreeFine's two scheduler steps:
  inv_step      /root/reference/src/demo/model.py:109-132
  ctrl_step     /root/reference/src/demo/model.py:134-198  (+ _get_variance :200-209)
  linear_param  /root/reference/src/demo/model.py:438-455
The schedule is diffusers' DDIMScheduler as configured by SD's scheduler_config.json (scaled_linear betas 0.00085 ->
0.012, 1000 train steps, steps_offset=1, set_alpha_to_one=False, "leading" spacing; in-tree corroboration:
evaluation/DragDiffusion/geobench_eval.py:77-79).
"""
from types import SimpleNamespace

import numpy as np
import torch


class DDIMSchedule:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, set_alpha_to_one=False):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset)
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.config.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)


def inv_step(sched, eps, timestep, x):
    next_step = int(timestep)
    t = min(next_step - sched.config.num_train_timesteps // sched.num_inference_steps, 999)
    a_t = sched.alphas_cumprod[t] if t >= 0 else sched.final_alpha_cumprod
    a_next = sched.alphas_cumprod[next_step]
    pred_x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    x_next = a_next ** 0.5 * pred_x0 + (1 - a_next) ** 0.5 * eps
    return x_next, pred_x0


def get_variance(sched, t, prev_t):
    a_t = sched.alphas_cumprod[t]
    a_prev = sched.alphas_cumprod[prev_t] if prev_t >= 0 else sched.final_alpha_cumprod
    return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)


def ctrl_step(sched, eps, timestep, x, mask, eta, noise=None):
    """mask: [h,w] tensor in its pipeline dtype (uint8 in the reference); noise: the randn draw of this step (same
    shape as eps) -- the reference draws it from the global generator iff eta > 0 (model.py:185-188)."""
    t = int(timestep)
    prev_t = t - sched.config.num_train_timesteps // sched.num_inference_steps
    a_t = sched.alphas_cumprod[t]
    a_prev = sched.alphas_cumprod[prev_t] if prev_t > 0 else sched.final_alpha_cumprod
    pred_x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    std = eta * get_variance(sched, t, prev_t).to(eps.dtype) ** 0.5
    if eps.shape[0] == 2:
        std = torch.cat((std[None], torch.zeros_like(std)[None]))[:, None, None, None]
        mask = mask.repeat(1, 4, 1, 1)
        mask = torch.cat((mask, torch.ones_like(mask)))
    pred_dir = (1 - a_prev) ** 0.5 * eps * (1 - mask) + (1 - a_prev - std ** 2) ** 0.5 * eps * mask
    x_prev = a_prev ** 0.5 * pred_x0 + pred_dir
    if eta > 0:
        x_prev = x_prev + std * noise * mask
    return x_prev, pred_x0


def linear_param(t, t1, t0, t2, end_scale=0.5):
    if t < t1 or t > t2:
        raise ValueError(f"t must be in [{t1}, {t2}]")
    if t <= t0:
        return 1 + (end_scale - 1) / (t0 - t1) * (t - t1)
    return end_scale + (-end_scale / (t2 - t0)) * (t - t0)
