"""The `nn.Module`-shaped surface of the reference's model objects that its callers touch
(`inference_multigpu.py:52-55`: `model.vae.to(device)`, `model.dit.to(device)`, `model.text_encoder.to(device)`,
`model.vae.enable_tiling()`; `pyramid_dit_for_video_gen_pipeline.py:1261-1267`: `.device` / `.dtype` of the dit).

The engines here are not nn.Modules: weights are packed into MFMA-friendly layouts on ONE device at construction and
never move.  `.to()` therefore validates and returns self -- a request for a different device type fails loudly
(there is no CPU path), a dtype request is ignored (bf16 storage / fp32 accumulate is fixed by the kernels)."""
import json
import os

import torch


class DeviceModuleAPI:
    dev = None                                   # torch.device, set by the engine's constructor

    def to(self, *args, **kwargs):
        dev = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, torch.device)):
                dev = a
        if dev is not None:
            dev = torch.device(dev)
            mine = torch.device(self.dev)
            if dev.type != mine.type or (dev.index is not None and mine.index is not None and dev.index != mine.index):
                raise RuntimeError(f"{type(self).__name__} lives on {mine} (packed MI355X layouts); it cannot move to {dev}")
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device) if device is not None else "cuda")

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("inference-only engine (training is out of scope)")
        return self

    def requires_grad_(self, requires_grad=False):
        return self

    @property
    def device(self):
        return torch.device(self.dev)

    @property
    def dtype(self):
        return torch.bfloat16


def load_diffusers_dir(path):
    """`ModelMixin.from_pretrained` directory layout (pipeline.py:73,81,156): config.json + *.safetensors"""
    from safetensors.torch import load_file
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    sd = {}
    for fn in sorted(os.listdir(path)):
        if fn.endswith(".safetensors"):
            sd.update(load_file(os.path.join(path, fn)))
    if not sd:
        raise FileNotFoundError(f"no .safetensors weights under {path}")
    return sd, {k: v for k, v in cfg.items() if not k.startswith("_")}
