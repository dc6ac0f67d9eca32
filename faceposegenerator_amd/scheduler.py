"""Host-side mirror of ``diffusers.DDPMScheduler`` as the reference uses it.

Reference call sites: ``DDPMScheduler.from_pretrained(model, subfolder="scheduler")``
(inference_ID-Booth.py:104, train_ID-Booth.py:615), ``.step(...).pred_original_sample``
(train_ID-Booth.py:1081), ``.add_noise`` (:1018), ``.get_velocity`` (:1058), ``.config.*``
(:147-153, :1011, :1055).  Restated from diffusers 0.32.2 ``schedulers/scheduling_ddpm.py``
(SURVEY.md §3.3): scaled_linear betas, leading spacing, steps_offset, fixed_small variance, no clipping.

Tables and coefficients are host fp32 (30 scalars).  ``step`` on GPU tensors runs the fused HIP
kernel ``idb_cfg_ddpm_step``; the pipeline's sampling loop calls the same kernel directly with CFG
fused in.
"""
from __future__ import annotations

from dataclasses import asdict
from types import SimpleNamespace
from typing import List, Optional, Tuple, Union

import torch

from . import spec as S
from . import weights as W


class DDPMSchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample

    def __iter__(self):                     # tuple-style unpacking (return_dict=False callers index [0])
        return iter((self.prev_sample, self.pred_original_sample))


class DDPMScheduler:
    order = 1

    def __init__(self, config: S.SchedulerConfig = S.SD21_SCHED):
        if config.beta_schedule != "scaled_linear":
            raise ValueError(f"beta_schedule {config.beta_schedule!r} is not supported (SD-2.x uses scaled_linear)")
        if config.variance_type != "fixed_small" or config.clip_sample:
            raise ValueError("only variance_type='fixed_small' without clipping is supported")
        if config.timestep_spacing != "leading":
            raise ValueError("only timestep_spacing='leading' is supported")
        self._cfg = config
        self.config = SimpleNamespace(**asdict(config))
        self.betas = torch.linspace(config.beta_start ** 0.5, config.beta_end ** 0.5,
                                    config.num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.arange(config.num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    # ---- construction -----------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, **kw) -> "DDPMScheduler":
        """Local directory only (model names cannot be resolved offline)."""
        return cls(W.load_scheduler_config(path, subfolder))

    @classmethod
    def from_config(cls, config, **kw) -> "DDPMScheduler":
        if isinstance(config, S.SchedulerConfig):
            return cls(config)
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        keys = S.SchedulerConfig.__dataclass_fields__.keys()
        return cls(S.SchedulerConfig(**{k: v for k, v in d.items() if k in keys}))

    # ---- schedule ---------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        n_train = self._cfg.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError(f"num_inference_steps {num_inference_steps} > num_train_timesteps {n_train}")
        self.num_inference_steps = num_inference_steps
        ratio = n_train // num_inference_steps
        ts = [int(round(i * ratio)) + self._cfg.steps_offset for i in range(num_inference_steps - 1, -1, -1)]
        self.timesteps = torch.tensor(ts, dtype=torch.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def previous_timestep(self, timestep: int) -> int:
        if self.num_inference_steps:
            ts = self.timesteps.tolist()
            idx = ts.index(int(timestep))
            return ts[idx + 1] if idx + 1 < len(ts) else -1
        return int(timestep) - 1

    def step_coefficients(self, timestep: int) -> Tuple[float, float, float, float, float]:
        """(sqrt(abar_t), sqrt(1-abar_t), c_x0, c_x, sigma) in fp32, computed with the same fp32
        tensor ops as upstream so that the HIP step and the oracle see identical scalars."""
        t = int(timestep)
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c_x0 = (a_prev ** 0.5 * cur_b) / b_t
        c_x = cur_a ** 0.5 * b_prev / b_t
        sigma = torch.clamp(b_prev / b_t * cur_b, min=1e-20) ** 0.5 if t > 0 else torch.tensor(0.0)
        return float(a_t ** 0.5), float(b_t ** 0.5), float(c_x0), float(c_x), float(sigma)

    # ---- step -------------------------------------------------------------------------
    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None,
             return_dict: bool = True, variance_noise: Optional[torch.Tensor] = None):
        t = int(timestep)
        sa, sb, c_x0, c_x, sigma = self.step_coefficients(t)
        noise = None
        if t > 0:
            noise = variance_noise
            if noise is None:
                gdev = generator.device if generator is not None else torch.device("cpu")
                noise = torch.randn(model_output.shape, generator=generator, device=gdev,
                                    dtype=torch.float32).to(sample.device)
        vpred = self._cfg.prediction_type == "v_prediction"
        if self._cfg.prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError(f"prediction_type {self._cfg.prediction_type!r} is not supported")
        if sample.is_cuda:
            from . import engine as E
            prev, x0 = E.ddpm_step_device(model_output, sample, noise, (sa, sb, c_x0, c_x, sigma), vpred)
        else:
            # host tensors: the same scalar formulas in torch (the scheduler is device-agnostic
            # host logic upstream as well); the GPU pipeline never takes this branch.
            mo, x = model_output.float(), sample.float()
            x0 = (sa * x - sb * mo) if vpred else (x - sb * mo) / sa
            prev = c_x0 * x0 + c_x * x
            if noise is not None:
                prev = prev + sigma * noise.float()
        if not return_dict:
            return (prev, x0)
        return DDPMSchedulerOutput(prev, x0)

    # ---- training-side helpers kept for API completeness --------------
    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(original_samples.device)
        sa = (ac[timesteps] ** 0.5).flatten()
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.ndim < original_samples.ndim:
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise

    def get_velocity(self, sample, noise, timesteps):
        ac = self.alphas_cumprod.to(sample.device)
        sa = (ac[timesteps] ** 0.5).flatten()
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.ndim < sample.ndim:
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * noise - sb * sample

    def __len__(self):
        return self._cfg.num_train_timesteps


class DPMSolverMultistepScheduler:
    """DPM-Solver++ (2M, midpoint), the sampler the reference switches to for its validation images:
    ``DPMSolverMultistepScheduler.from_config(pipeline.scheduler.config, **scheduler_args)`` (train_ID-Booth.py:155), i.e.
    upstream's defaults on SD-2.1's scheduler config: algorithm_type "dpmsolver++", solver_order 2, solver_type "midpoint",
    lower_order_final, final_sigmas_type "zero", no Karras sigmas, no thresholding, "leading" spacing with steps_offset.
    Every update is  x_next = c_x * x + c_0 * x0_hat(current) + c_1 * x0_hat(previous); the per-step scalars come from
    ``step_coefficients(i)`` and the device step is ``idb_cfg_ddpm_step`` with the x0 history as its third operand
    (engine.HipEngine.sample(multistep=True)).  ``step`` is the stateful host form of the same update."""
    order = 1

    def __init__(self, config: S.SchedulerConfig = S.SD21_SCHED, solver_order: int = 2):
        if config.beta_schedule != "scaled_linear":
            raise ValueError(f"beta_schedule {config.beta_schedule!r} is not supported (SD-2.x uses scaled_linear)")
        if config.timestep_spacing != "leading":
            raise ValueError("only timestep_spacing='leading' is supported")
        if config.prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError(f"prediction_type {config.prediction_type!r} is not supported")
        if solver_order not in (1, 2):
            raise ValueError("solver_order must be 1 or 2")
        self._cfg = config
        self.solver_order = solver_order
        self.config = SimpleNamespace(**asdict(config), solver_order=solver_order, algorithm_type="dpmsolver++",
                                      solver_type="midpoint", lower_order_final=True, final_sigmas_type="zero")
        betas = torch.linspace(config.beta_start ** 0.5, config.beta_end ** 0.5, config.num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.arange(config.num_train_timesteps - 1, -1, -1, dtype=torch.int64)
        self.sigmas = None
        self._reset()

    def _reset(self):
        self._step_index = 0
        self._hist = []                                   # x0 predictions, newest last

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, **kw) -> "DPMSolverMultistepScheduler":
        return cls(W.load_scheduler_config(path, subfolder), **{k: v for k, v in kw.items() if k == "solver_order"})

    @classmethod
    def from_config(cls, config, **kw) -> "DPMSolverMultistepScheduler":
        """Accepts another scheduler's ``.config`` (namespace / dict) or a SchedulerConfig; ``variance_type`` and other keys the
        solver does not use are ignored, as upstream's from_config does."""
        order = kw.pop("solver_order", 2)
        if isinstance(config, S.SchedulerConfig):
            return cls(config, order)
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        d.update({k: v for k, v in kw.items() if k in S.SchedulerConfig.__dataclass_fields__})
        keys = S.SchedulerConfig.__dataclass_fields__.keys()
        return cls(S.SchedulerConfig(**{k: v for k, v in d.items() if k in keys}), order)

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        n_train = self._cfg.num_train_timesteps
        if num_inference_steps >= n_train:
            raise ValueError(f"num_inference_steps {num_inference_steps} >= num_train_timesteps {n_train}")
        self.num_inference_steps = num_inference_steps
        ratio = n_train // (num_inference_steps + 1)                       # upstream: last_timestep // (N + 1), lambda_min_clipped = -inf
        ts = [int(round(i * ratio)) + self._cfg.steps_offset for i in range(num_inference_steps, 0, -1)]
        self.timesteps = torch.tensor(ts, dtype=torch.int64)
        ac = self.alphas_cumprod.double()
        sig = ((1 - ac) / ac) ** 0.5
        self.sigmas = torch.cat([sig[self.timesteps], torch.zeros(1, dtype=torch.float64)]).float()   # final_sigmas_type "zero"
        self._reset()

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma: torch.Tensor):
        alpha_t = 1.0 / (sigma ** 2 + 1.0) ** 0.5
        return alpha_t, sigma * alpha_t

    def step_coefficients(self, i: int):
        """(alpha_s, sigma_s, c_0, c_x, c_1) of step index i in fp32: x0 = (x - sigma_s eps) / alpha_s (or alpha_s x - sigma_s v),
        x_next = c_0 x0 + c_x x + c_1 x0_prev."""
        n = len(self.timesteps)
        sg = self.sigmas
        a_t, s_t = self._alpha_sigma(sg[i + 1])
        a_s0, s_s0 = self._alpha_sigma(sg[i])
        lam = lambda a, s_: torch.log(a) - torch.log(s_)
        h = lam(a_t, s_t) - lam(a_s0, s_s0)
        c_x = s_t / s_s0
        base = -(a_t * (torch.exp(-h) - 1.0))
        final = i == n - 1                                                # final sigma is zero: lower_order_final
        second_last_low = i == n - 2 and n < 15                           # lower_order_second applies to solver_order 3 only
        first_order = self.solver_order == 1 or i == 0 or final
        del second_last_low
        if first_order:
            c_0, c_1 = base, torch.tensor(0.0)
        else:
            a_s1, s_s1 = self._alpha_sigma(sg[i - 1])
            h_0 = lam(a_s0, s_s0) - lam(a_s1, s_s1)
            r0 = h_0 / h
            # x_t = c_x x + base * D0 + 0.5 * base * D1,  D0 = m0, D1 = (m0 - m1) / r0
            c_0 = base + 0.5 * base / r0
            c_1 = -0.5 * base / r0
        return float(a_s0), float(s_s0), float(c_0), float(c_x), float(c_1)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None, return_dict: bool = True, **kw):
        i = self._step_index
        if self.sigmas is None or i >= len(self.timesteps):
            raise ValueError("call set_timesteps first / too many steps")
        if int(timestep) != int(self.timesteps[i]):
            raise ValueError(f"step {i} expects timestep {int(self.timesteps[i])}, got {int(timestep)}")
        a_s, s_s, c_0, c_x, c_1 = self.step_coefficients(i)
        mo, x = model_output.float(), sample.float()
        x0 = (a_s * x - s_s * mo) if self._cfg.prediction_type == "v_prediction" else (x - s_s * mo) / a_s
        prev = c_0 * x0 + c_x * x
        if c_1 != 0.0:
            prev = prev + c_1 * self._hist[-1]
        self._hist = (self._hist + [x0])[-2:]
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return DDPMSchedulerOutput(prev, x0)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(original_samples.device)
        sa, sb = (ac[timesteps] ** 0.5).flatten(), ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.ndim < original_samples.ndim:
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise

    def __len__(self):
        return self._cfg.num_train_timesteps
