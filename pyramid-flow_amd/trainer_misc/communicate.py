"""`all_to_all(input_, process_group, world_size, scatter_dim, gather_dim, concat_output)` with the reference's
semantics (trainer_misc/communicate.py:7-66): split `input_` into `world_size` equal parts along `scatter_dim`,
exchange part p with rank p, concatenate what arrives along `gather_dim` (or keep the first part).  Inference only (no
autograd function).  The DiT engines do NOT go through this form -- they exchange pre-packed, unevenly split buffers
with one `all_to_all_single` (pyflow_hip/flux_sp.py) -- it exists for callers of the reference API and as the
reference point of the sequence-parallel tests.  Transport: one `all_to_all_single` over RCCL; over gloo (no
all_to_all) grouped point-to-point."""
import torch
import torch.distributed as dist


def all_to_all(input_, process_group, world_size=1, scatter_dim=2, gather_dim=1, concat_output=True):
    if world_size == 1:
        return input_
    parts = [t.contiguous() for t in torch.tensor_split(input_, world_size, scatter_dim)]
    assert all(p.shape == parts[0].shape for p in parts), "scatter_dim must divide evenly (communicate.py:17-18)"
    send = torch.stack(parts)
    recv = torch.empty_like(send)
    if dist.get_backend(process_group) == "nccl":
        dist.all_to_all_single(recv, send, group=process_group)
    else:
        rank = dist.get_rank(process_group)
        ops = []
        for p in range(world_size):
            if p == rank:
                recv[p].copy_(send[p])
                continue
            gp = dist.get_global_rank(process_group, p) if process_group is not None else p
            ops.append(dist.P2POp(dist.isend, send[p], gp, process_group))
            ops.append(dist.P2POp(dist.irecv, recv[p], gp, process_group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if concat_output:
        return torch.cat(list(recv), dim=gather_dim).contiguous()
    return recv[0]          # multi-GPU inference: the latents of every rank are equal, keep the first (:23-25)
