"""TEST INFRASTRUCTURE ONLY -- builds tiny seeded instances of the UNMODIFIED
reference classes (dev container only; needs /root/reference) so that the
oracle restatement can be pinned against the reference itself and golden
fixtures can be generated (oracle/gen_golden.py).  Recipe: SURVEY.md section 8c.
"""
import types

import torch

from . import shims

TINY_DIT = dict(num_layers=2, num_single_layers=2, num_attention_heads=4, attention_head_dim=64,
                joint_attention_dim=32, pooled_projection_dim=16, in_channels=64,
                axes_dims_rope=[16, 24, 24])
TINY_VAE = dict(encoder_out_channels=16, decoder_in_channels=16,
                encoder_block_out_channels=(32, 32, 64, 64), decoder_block_out_channels=(32, 32, 64, 64),
                encoder_layers_per_block=(1, 1, 1, 1), decoder_layers_per_block=(2, 2, 2, 2))


def seed_weights(module, seed, std=0.05):
    """N(0,std) for matrices/biases, N(1,0.1) for norm gains -- NOT the zero-init of flux:168-183."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.ndim == 1 and ("norm" in n and n.endswith("weight")):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(std * torch.randn(p.shape, generator=g))
    return module


def build_ref_dit(cfg=None, seed=1234):
    ref = shims.load_reference()
    m = ref.PyramidFluxTransformer(**(cfg or TINY_DIT)).eval()
    return seed_weights(m, seed)


def build_ref_vae(cfg=None, seed=4321):
    ref = shims.load_reference()
    v = ref.CausalVideoVAE(**(cfg or TINY_VAE)).eval()
    return seed_weights(v, seed)


class StubTextEncoder:
    """(prompt, device) -> (embeds[1,Lt,C], mask[1,Lt] long, pooled[1,Cp]); deterministic per prompt."""

    def __init__(self, Lt=16, C=32, Cp=16):
        self.Lt, self.C, self.Cp = Lt, C, Cp

    def to(self, *_a, **_k):
        return self

    def __call__(self, prompt, device):
        s = sum(ord(ch) for ch in (prompt if isinstance(prompt, str) else prompt[0])) % 1000
        g = torch.Generator().manual_seed(s)
        e = torch.randn(1, self.Lt, self.C, generator=g)
        p = torch.randn(1, self.Cp, generator=g)
        n_valid = 5 + s % (self.Lt - 5)
        m = torch.zeros(1, self.Lt, dtype=torch.long)
        m[:, :n_valid] = 1
        return e, m, p


def build_ref_pipeline(dit, vae, stages=(1, 2, 4), text_encoder=None):
    """object.__new__ construction (the real __init__ needs checkpoints)."""
    ref = shims.load_reference()
    P = ref.PyramidDiTForVideoGeneration
    pipe = object.__new__(P)
    pipe.stages = list(stages)
    pipe.sample_ratios = [1] * len(stages)
    pipe.corrupt_ratio = 1 / 3
    pipe.dit, pipe.vae = dit, vae
    pipe.text_encoder = text_encoder or StubTextEncoder()
    pipe.load_text_encoder, pipe.load_vae = True, True
    pipe.model_name = "pyramid_flux"
    pipe.vae_shift_factor, pipe.vae_scale_factor = -0.04, 1 / 1.8726
    pipe.vae_video_shift_factor, pipe.vae_video_scale_factor = -0.2343, 1 / 3.0986
    pipe.downsample = 8
    pipe.frame_per_unit = 1
    pipe.max_temporal_length = 31
    rng = [0, 1] if len(stages) == 1 else [i / len(stages) for i in range(len(stages) + 1)]
    pipe.scheduler = ref.PyramidFlowMatchEulerDiscreteScheduler(stages=len(stages), stage_range=rng, gamma=1 / 3)
    pipe.sequential_offload_enabled = False
    pipe.cfg_rate = 0.1
    pipe.return_log = True
    pipe.use_flash_attn = False
    return pipe


class NoiseStream:
    """Pre-drawn standard-normal stream handed identically to reference and oracle/HIP."""

    def __init__(self, seed=1):
        self.g = torch.Generator().manual_seed(seed)

    def block_noise(self, bs, ch, t, h, w, gamma=1 / 3):
        from .pipeline_oracle import block_noise_from_normal
        n = bs * ch * t * (h // 2) * (w // 2)
        eps = torch.randn(n, 4, generator=self.g)
        return block_noise_from_normal(eps, bs, ch, t, h, w, gamma)


def patch_block_noise(pipe, stream):
    def f(self, bs, ch, temp, height, width):
        return stream.block_noise(bs, ch, temp, height, width, self.scheduler.config.gamma)
    pipe.sample_block_noise = types.MethodType(f, pipe)
    return pipe
