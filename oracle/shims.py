"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stub modules that let the UNMODIFIED reference (``/root/reference``) import in
this container, where ``diffusers``, ``timm``, ``torchvision``, ``tensorboardX``
and ``IPython`` are absent.  Only ``tests/`` (dev container, where the reference
is mounted), ``oracle/gen_golden.py`` and the CPU-baseline leg of ``bench.py``
may use this.  ``/root/reference`` does not exist on the GPU box: everything
here degrades to ``available() == False`` there.

The arithmetic that lives in the missing third-party packages is restated
here from their published behaviour (pins: ``requirements.txt:6``
``diffusers>=0.30.1``):
  * ``diffusers.models.activations.GELU``  = Linear + F.gelu(approximate=...)
  * ``diffusers.models.attention_processor.Attention`` restricted to the VAE
    mid-block use (``video_vae/modeling_block.py:413-427``): GroupNorm ->
    q,k,v Linear -> 1-head softmax(qk^T/sqrt(C)) v -> Linear -> +residual.
  * ``diffusers.utils.torch_utils.randn_tensor`` = torch.randn(shape, generator=...)
"""
import importlib.machinery
import inspect
import math
import os
import sys
import types
from collections import OrderedDict
from dataclasses import fields, is_dataclass

REFERENCE_ROOT = os.environ.get("PYFLOW_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pyramid_dit"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=True)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


_INSTALLED = False


def install():
    """Register the stub modules and put the reference on sys.path. Idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    # real packages that must be imported BEFORE fake torchvision exists
    import transformers  # noqa: F401
    from transformers import (CLIPTextModel, CLIPTextModelWithProjection,  # noqa: F401
                              CLIPTokenizer, T5EncoderModel, T5TokenizerFast)
    import accelerate  # noqa: F401

    # ---------------- diffusers ----------------
    class BaseOutput(OrderedDict):
        def __post_init__(self):
            if is_dataclass(self):
                for f in fields(self):
                    v = getattr(self, f.name)
                    if v is not None:
                        self[f.name] = v

        def __getitem__(self, k):
            if isinstance(k, str):
                return dict(self.items())[k]
            return self.to_tuple()[k]

        def to_tuple(self):
            return tuple(self[k] for k in self.keys())

    class _Logger:
        def __getattr__(self, _):
            return lambda *a, **k: None

    logging = types.SimpleNamespace(get_logger=lambda *_a, **_k: _Logger())

    def deprecate(*_a, **_k):
        return None

    def is_torch_version(*_a, **_k):
        return True

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        gen_device = generator.device if generator is not None else device
        t = torch.randn(tuple(shape), generator=generator, device=gen_device, dtype=dtype)
        return t.to(device) if device is not None else t

    class _Config(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

    class ConfigMixin:
        @property
        def config(self):
            return self.__dict__.get("_internal_dict", _Config())

        def register_to_config(self, **kw):
            d = self.__dict__.setdefault("_internal_dict", _Config())
            d.update(kw)

    def register_to_config(init):
        sig = inspect.signature(init)

        def wrapped(self, *args, **kwargs):
            ba = sig.bind(self, *args, **kwargs)
            ba.apply_defaults()
            cfg = {k: v for k, v in ba.arguments.items() if k not in ("self", "kwargs")}
            object.__setattr__(self, "_internal_dict", _Config(cfg)) if not isinstance(self, nn.Module) \
                else self.__dict__.__setitem__("_internal_dict", _Config(cfg))
            init(self, *args, **kwargs)
        wrapped.__wrapped__ = init
        return wrapped

    class ModelMixin(nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

        @property
        def dtype(self):
            return next(self.parameters()).dtype

    class SchedulerMixin:
        pass

    class GELU(nn.Module):
        def __init__(self, dim_in, dim_out, approximate="none", bias=True):
            super().__init__()
            self.proj = nn.Linear(dim_in, dim_out, bias=bias)
            self.approximate = approximate

        def forward(self, x):
            return F.gelu(self.proj(x), approximate=self.approximate)

    class _Unused(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError("not on the hot path")

    class FP32SiLU(nn.Module):
        def forward(self, x):
            return F.silu(x.float()).to(x.dtype)

    def get_activation(name):
        name = name.lower()
        if name in ("silu", "swish"):
            return nn.SiLU()
        if name == "gelu":
            return nn.GELU()
        if name == "relu":
            return nn.ReLU()
        if name == "mish":
            return nn.Mish()
        raise ValueError(name)

    class Attention(nn.Module):
        """diffusers Attention restricted to the deprecated-attn-block VAE use."""

        def __init__(self, query_dim, heads=8, dim_head=64, rescale_output_factor=1.0, eps=1e-5,
                     norm_num_groups=None, spatial_norm_dim=None, residual_connection=False,
                     bias=False, upcast_softmax=False, _from_deprecated_attn_block=False, **_k):
            super().__init__()
            assert spatial_norm_dim is None
            inner = heads * dim_head
            self.heads = heads
            self.scale = dim_head ** -0.5
            self.rescale_output_factor = rescale_output_factor
            self.residual_connection = residual_connection
            self.upcast_softmax = upcast_softmax
            self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True) \
                if norm_num_groups is not None else None
            self.to_q = nn.Linear(query_dim, inner, bias=bias)
            self.to_k = nn.Linear(query_dim, inner, bias=bias)
            self.to_v = nn.Linear(query_dim, inner, bias=bias)
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

        def forward(self, hidden_states, temb=None, **_k):
            residual = hidden_states
            b, c, h, w = hidden_states.shape
            x = hidden_states.view(b, c, h * w).transpose(1, 2)
            if self.group_norm is not None:
                x = self.group_norm(x.transpose(1, 2)).transpose(1, 2)
            q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)
            hd = q.shape[-1] // self.heads

            def split(t):
                return t.view(b, -1, self.heads, hd).transpose(1, 2)
            q, k, v = split(q), split(k), split(v)
            s = torch.matmul(q, k.transpose(-1, -2)) * self.scale
            if self.upcast_softmax:
                s = s.float()
            p = s.softmax(dim=-1).to(v.dtype)
            o = torch.matmul(p, v).transpose(1, 2).reshape(b, -1, self.heads * hd)
            o = self.to_out[1](self.to_out[0](o))
            o = o.transpose(-1, -2).reshape(b, c, h, w)
            if self.residual_connection:
                o = o + residual
            return o / self.rescale_output_factor

    from dataclasses import dataclass

    @dataclass
    class AutoencoderKLOutput(BaseOutput):
        latent_dist: "object" = None

    _mod("diffusers")
    _mod("diffusers.utils", BaseOutput=BaseOutput, is_torch_version=is_torch_version,
         logging=logging, deprecate=deprecate)
    _mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.models")
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin)
    _mod("diffusers.models.activations", GELU=GELU, GEGLU=_Unused, ApproximateGELU=_Unused,
         SwiGLU=_Unused, FP32SiLU=FP32SiLU, get_activation=get_activation)
    _mod("diffusers.models.attention_processor", Attention=Attention, SpatialNorm=_Unused,
         AttentionProcessor=object, AttnProcessor=object, AttnAddedKVProcessor=object,
         ADDED_KV_ATTENTION_PROCESSORS=(), CROSS_ATTENTION_PROCESSORS=())
    _mod("diffusers.models.lora", LoRACompatibleConv=nn.Conv2d, LoRACompatibleLinear=nn.Linear)
    _mod("diffusers.models.normalization", AdaGroupNorm=_Unused)
    _mod("diffusers.models.modeling_outputs", AutoencoderKLOutput=AutoencoderKLOutput)

    # ---------------- timm ----------------
    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", trunc_normal_=trunc_normal_, drop_path=lambda x, *a, **k: x,
         to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x)
    _mod("timm.models.hub", download_cached_file=lambda *a, **k: None, get_cache_dir=lambda *a, **k: "/tmp")

    # ---------------- torchvision (transforms restated: pipeline.py:906-909) ----------------
    import numpy as np

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor:
        def __call__(self, pic):
            a = np.asarray(pic)
            if a.ndim == 2:
                a = a[:, :, None]
            t = torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous()
            return t.float().div(255) if t.dtype == torch.uint8 else t.float()

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            m = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
            s = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
            return (t - m) / s

    _mod("torchvision")
    _mod("torchvision.transforms", Compose=Compose, ToTensor=ToTensor, Normalize=Normalize)
    _mod("torchvision.models", vgg16=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))

    # ---------------- misc ----------------
    _mod("tensorboardX", SummaryWriter=object)
    _mod("IPython", embed=lambda *a, **k: None)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _INSTALLED = True


def load_reference():
    """Import the unmodified reference classes. Returns a namespace."""
    install()
    from pyramid_dit import PyramidDiTForVideoGeneration
    from pyramid_dit.flux_modules.modeling_pyramid_flux import PyramidFluxTransformer
    from pyramid_dit.mmdit_modules.modeling_pyramid_mmdit import PyramidDiffusionMMDiT
    from video_vae import CausalVideoVAE
    from diffusion_schedulers import PyramidFlowMatchEulerDiscreteScheduler
    return types.SimpleNamespace(
        PyramidDiTForVideoGeneration=PyramidDiTForVideoGeneration,
        PyramidFluxTransformer=PyramidFluxTransformer,
        PyramidDiffusionMMDiT=PyramidDiffusionMMDiT,
        CausalVideoVAE=CausalVideoVAE,
        PyramidFlowMatchEulerDiscreteScheduler=PyramidFlowMatchEulerDiscreteScheduler,
    )
