"""Drop-in for the context-parallel helpers of the reference's top-level `utils` module (utils.py:18-105): group
bookkeeping for the temporal context parallelism of the causal VAE.  State and communicator live in pyflow_hip.cp; the
rest of the reference's `utils.py` (LPIPS / VGG checkpoint download helpers, md5 checks) belongs to training and is out
of scope."""
import torch.distributed as dist

from pyflow_hip import cp as _cp


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def is_context_parallel_initialized():
    return _cp.is_context_parallel_initialized()


def set_context_parallel_group(size, group):
    """utils.py:54-58"""
    _cp.initialize_context_parallel(size, group=group)


def initialize_context_parallel(context_parallel_size):
    """utils.py:61-76: consecutive-rank groups of `context_parallel_size`"""
    assert not _cp.is_context_parallel_initialized(), "context parallel group is already initialized"
    _cp.initialize_context_parallel(context_parallel_size)


def get_context_parallel_group():
    comm = _cp.get_context_parallel_comm()
    assert comm is not None, "context parallel group is not initialized"
    return comm.group if comm.group is not None else dist.group.WORLD


def get_context_parallel_world_size():
    assert _cp.is_context_parallel_initialized(), "context parallel size is not initialized"
    return _cp.get_context_parallel_world_size()


def get_context_parallel_rank():
    assert _cp.is_context_parallel_initialized(), "context parallel size is not initialized"
    return get_rank() % _cp.get_context_parallel_world_size()


def get_context_parallel_group_rank():
    assert _cp.is_context_parallel_initialized(), "context parallel size is not initialized"
    return get_rank() // _cp.get_context_parallel_world_size()
