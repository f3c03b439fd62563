"""PyramidFlowMatchEulerDiscreteScheduler -- drop-in for
diffusion_schedulers/scheduling_flow_matching.py:27-297 (same constructor, attributes, set_timesteps /
step signatures, stateful _step_index).  Tables are host float64/float32 numpy exactly as the
reference computes them; the per-step update inside the sampler's hot loop is the fused HIP kernel
pf_cfg_euler_step (this class only hands it sigma_next - sigma via `dsigma()`); the generic
``step()`` is kept for API compatibility on arbitrary tensors.
"""
import math
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch


@dataclass
class FlowMatchEulerDiscreteSchedulerOutput:
    prev_sample: torch.Tensor


class PyramidFlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, shift=1.0, stages=3, stage_range=None, gamma=1 / 3):
        if stage_range is None:
            stage_range = [0, 1 / 3, 2 / 3, 1]
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift, stages=stages,
                                      stage_range=list(stage_range), gamma=gamma)
        self.timestep_ratios, self.timesteps_per_stage, self.sigmas_per_stage = {}, {}, {}
        self.start_sigmas, self.end_sigmas, self.ori_start_sigmas = {}, {}, {}
        self.init_sigmas_for_each_stage()
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()
        self.gamma = gamma

    # -- global table (:70-88), float32 like the reference
    def init_sigmas(self):
        n, shift = self.config.num_train_timesteps, self.config.shift
        t = torch.from_numpy(np.linspace(1, n, n, dtype=np.float32)[::-1].copy())
        sig = t / n
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * n
        self.sigmas = sig
        self._step_index = None
        self._begin_index = None

    # -- per-stage tables (:90-149)
    def init_sigmas_for_each_stage(self):
        self.init_sigmas()
        c = self.config
        n = c.num_train_timesteps
        dist = []
        for i in range(c.stages):
            a = max(int(c.stage_range[i] * n), 0)
            b = min(int(c.stage_range[i + 1] * n), n)
            start = self.sigmas[a].item()
            end = self.sigmas[b].item() if b < n else 0.0
            self.ori_start_sigmas[i] = start
            if i != 0:
                ori = 1 - start
                start = 1 - (1 / (math.sqrt(1 + (1 / c.gamma)) * (1 - ori) + ori)) * ori
            dist.append(start - end)
            self.start_sigmas[i], self.end_sigmas[i] = start, end
        tot = sum(dist)
        for i in range(c.stages):
            r0 = 0.0 if i == 0 else sum(dist[:i]) / tot
            r1 = 1.0 if i == c.stages - 1 else sum(dist[:i + 1]) / tot
            self.timestep_ratios[i] = (r0, r1)
        for i in range(c.stages):
            r0, r1 = self.timestep_ratios[i]
            tmax = self.timesteps[int(r0 * n)].item()
            tmin = self.timesteps[min(int(r1 * n), n - 1)].item()
            self.timesteps_per_stage[i] = torch.from_numpy(np.linspace(np.float32(tmax), np.float32(tmin), n + 1)[:-1].copy())
            self.sigmas_per_stage[i] = torch.from_numpy(np.linspace(1, 0, n + 1)[:-1].copy())

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index=0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps, stage_index, device=None):
        # :179-206
        self.num_inference_steps = num_inference_steps
        self.init_sigmas()
        st = self.timesteps_per_stage[stage_index]
        ts = np.linspace(st[0].item(), st[-1].item(), num_inference_steps)
        self.timesteps = torch.from_numpy(ts).to(device=device)
        ss = self.sigmas_per_stage[stage_index]
        sig = torch.from_numpy(np.linspace(ss[0].item(), ss[-1].item(), num_inference_steps)).to(device=device)
        self.sigmas = torch.cat([sig, torch.zeros(1, dtype=sig.dtype, device=sig.device)])
        self._sigmas_host = np.concatenate([np.linspace(ss[0].item(), ss[-1].item(), num_inference_steps), [0.0]])
        self._timesteps_host = ts
        self._step_index = None

    def dsigma(self):
        """sigma_next - sigma of the pending step (float64 host value), then advance the counter."""
        if self._step_index is None:
            self._step_index = 0
        i = self._step_index
        self._step_index += 1
        return float(self._sigmas_host[i + 1] - self._sigmas_host[i])

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        # :230-294
        if isinstance(timestep, int) or isinstance(timestep, (torch.IntTensor, torch.LongTensor)):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to"
                             " `EulerDiscreteScheduler.step()` is not supported. Make sure to pass"
                             " one of the `scheduler.timesteps` as a timestep.")
        if self._step_index is None:
            self._step_index = 0
        sample = sample.to(torch.float32)
        sigma = self.sigmas[self._step_index]
        sigma_next = self.sigmas[self._step_index + 1]
        prev = (sample + (sigma_next - sigma) * model_output).to(model_output.dtype)
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return FlowMatchEulerDiscreteSchedulerOutput(prev_sample=prev)

    def __len__(self):
        return self.config.num_train_timesteps
