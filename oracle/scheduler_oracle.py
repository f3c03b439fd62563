"""TEST INFRASTRUCTURE ONLY (oracle) -- CPU restatement of
``diffusion_schedulers/scheduling_flow_matching.py`` (reference file:line cited
per function).  Pinned against the imported reference in
``tests/test_oracle_vs_reference.py`` and against SURVEY Appendix B values in
``tests/test_scheduler.py``.  Never imported by the product path.
"""
import math

import numpy as np
import torch


class SchedulerOracle:
    """scheduling_flow_matching.py:27-297."""

    def __init__(self, num_train_timesteps=1000, shift=1.0, stages=3,
                 stage_range=(0, 1 / 3, 2 / 3, 1), gamma=1 / 3):
        self.n = num_train_timesteps
        self.shift = shift
        self.stages = stages
        self.stage_range = list(stage_range)
        self.gamma = gamma
        self.timestep_ratios, self.timesteps_per_stage, self.sigmas_per_stage = {}, {}, {}
        self.start_sigmas, self.end_sigmas, self.ori_start_sigmas = {}, {}, {}
        self._init_stages()
        self._step_index = None

    def _global(self):
        # :70-88  (float32 arithmetic, as the reference)
        t = np.linspace(1, self.n, self.n, dtype=np.float32)[::-1].copy()
        t = torch.from_numpy(t).to(torch.float32)
        s = t / self.n
        s = self.shift * s / (1 + (self.shift - 1) * s)
        return s * self.n, s

    def _init_stages(self):
        # :90-149
        timesteps, sigmas = self._global()
        dist = []
        for i in range(self.stages):
            a = max(int(self.stage_range[i] * self.n), 0)
            b = min(int(self.stage_range[i + 1] * self.n), self.n)
            start = sigmas[a].item()
            end = sigmas[b].item() if b < self.n else 0.0
            self.ori_start_sigmas[i] = start
            if i != 0:
                ori = 1 - start
                corrected = (1 / (math.sqrt(1 + (1 / self.gamma)) * (1 - ori) + ori)) * ori
                start = 1 - corrected
            dist.append(start - end)
            self.start_sigmas[i], self.end_sigmas[i] = start, end
        tot = sum(dist)
        for i in range(self.stages):
            r0 = 0.0 if i == 0 else sum(dist[:i]) / tot
            r1 = 1.0 if i == self.stages - 1 else sum(dist[:i + 1]) / tot
            self.timestep_ratios[i] = (r0, r1)
        for i in range(self.stages):
            r0, r1 = self.timestep_ratios[i]
            tmax = timesteps[int(r0 * self.n)]
            tmin = timesteps[min(int(r1 * self.n), self.n - 1)]
            ts = np.linspace(tmax, tmin, self.n + 1)
            # np.linspace on 0-dim torch endpoints may hand back a Tensor (reference handles both, :142)
            self.timesteps_per_stage[i] = ts[:-1] if isinstance(ts, torch.Tensor) else torch.from_numpy(ts[:-1])
            self.sigmas_per_stage[i] = torch.from_numpy(np.linspace(1, 0, self.n + 1)[:-1])

    def set_timesteps(self, num_inference_steps, stage_index):
        # :179-206
        st = self.timesteps_per_stage[stage_index]
        self.timesteps = torch.from_numpy(np.linspace(st[0].item(), st[-1].item(), num_inference_steps))
        ss = self.sigmas_per_stage[stage_index]
        sig = torch.from_numpy(np.linspace(ss[0].item(), ss[-1].item(), num_inference_steps))
        self.sigmas = torch.cat([sig, torch.zeros(1, dtype=sig.dtype)])
        self._step_index = None

    def step(self, model_output, sample):
        # :230-294 ; product (sigma_next - sigma) * v takes v's dtype (0-dim f64 x tensor)
        if self._step_index is None:
            self._step_index = 0
        sample = sample.to(torch.float32)
        d = self.sigmas[self._step_index + 1] - self.sigmas[self._step_index]
        prev = (sample + d * model_output).to(model_output.dtype)
        self._step_index += 1
        return prev

    def renoise_coeffs(self, stage):
        # pipeline.py:735-738
        ori = 1 - self.ori_start_sigmas[stage]
        alpha = 1 / (math.sqrt(1 + (1 / self.gamma)) * (1 - ori) + ori)
        beta = alpha * (1 - ori) / math.sqrt(self.gamma)
        return alpha, beta
