import torch


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def round_sd(sd):
    """weights rounded to bf16 on both sides of a parity comparison (SURVEY 8c tolerance statement)."""
    return {k: bf16_round(v.float()) for k, v in sd.items()}
