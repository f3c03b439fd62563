"""Temporal context parallelism for the causal VAE: reference-named helpers on top of SPComm.

Mirrors video_vae/context_parallel_ops.py:14-167 and utils.py:47-105 (group bookkeeping): `conv_scatter_to_
context_parallel_region` (local slice keeping the first `kernel_size` frames on rank 0), `cp_pass_from_previous_rank`
(halo: the last k-1 frames travel to the next rank, rank 0 prepends zeros = the causal padding), and
`conv_gather_from_context_parallel_region`.  These operate on plain tensors (any device) and exist for API parity and
for tests; the decode path itself (`CausalVideoVAE.decode_context_parallel`) exchanges the halo directly between the
channels-last activation buffers (`PBuf.exchange_halo`) and supports uneven frame ranges, which the reference's split
(`(T - k) % P == 0`) cannot express.
"""
import torch

from .sp import SPComm

_CP = None


def initialize_context_parallel(context_parallel_size=None, group=None):
    """utils.py:61-86: consecutive-rank groups; default = the whole world."""
    global _CP
    import torch.distributed as dist
    world = dist.get_world_size()
    size = context_parallel_size or world
    assert world % size == 0
    if group is None and size != world:
        rank = dist.get_rank()
        for g0 in range(0, world, size):
            grp = dist.new_group(list(range(g0, g0 + size)))
            if g0 <= rank < g0 + size:
                group = grp
    _CP = SPComm(group)
    return _CP


def is_context_parallel_initialized():
    return _CP is not None


def get_context_parallel_comm():
    return _CP


def get_context_parallel_world_size():
    return _CP.world if _CP else 1


def get_context_parallel_rank():
    return _CP.rank if _CP else 0


def reset_context_parallel():
    """test hook: forget the group"""
    global _CP
    _CP = None


def conv_scatter_to_context_parallel_region(input_, dim=2, kernel_size=1):
    """context_parallel_ops.py:14-38, 158-159"""
    P, r = get_context_parallel_world_size(), get_context_parallel_rank()
    if P == 1:
        return input_
    n = (input_.size(dim) - kernel_size) // P
    x = input_.transpose(dim, 0)
    x = x[: n + kernel_size] if r == 0 else x[r * n + kernel_size:(r + 1) * n + kernel_size]
    return x.transpose(dim, 0).contiguous()


def cp_pass_from_previous_rank(input_, dim, kernel_size):
    """context_parallel_ops.py:76-114: returns [halo (k-1 frames) | input_] along `dim`."""
    if kernel_size == 1 or _CP is None:
        return input_
    x = input_.transpose(0, dim)
    k1 = kernel_size - 1
    recv = torch.zeros_like(x[-k1:]).contiguous()          # rank 0 keeps zeros: the causal padding
    _CP.shift(x[-k1:].contiguous(), recv)
    return torch.cat([recv, x], dim=0).transpose(0, dim).contiguous()


def conv_gather_from_context_parallel_region(input_, dim=2, kernel_size=1):
    """context_parallel_ops.py:41-73: concatenation of the per-rank chunks along `dim` on every rank (uneven chunk
    sizes allowed; the reference relies on NCCL's list all_gather for that)."""
    P, r = get_context_parallel_world_size(), get_context_parallel_rank()
    if P == 1:
        return input_
    import torch.distributed as dist
    x = input_.transpose(0, dim).contiguous()
    sizes = [torch.zeros(1, dtype=torch.long) for _ in range(P)]
    dist.all_gather(sizes, torch.tensor([x.shape[0]]), group=_CP.group)
    parts = []
    for p in range(P):
        part = x if p == r else torch.empty((int(sizes[p]),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _CP.broadcast(part, p)
        parts.append(part)
    return torch.cat(parts, dim=0).transpose(0, dim).contiguous()
