"""Sequence-parallel group bookkeeping under the reference's names (trainer_misc/sp_utils.py:14-98); the state and
the communicator live in pyflow_hip.sp (one SPComm per process), these functions are views of it."""
import torch.distributed as dist

from pyflow_hip import sp as _sp
from .utils import get_rank


def is_sequence_parallel_initialized():
    return _sp.is_sequence_parallel_initialized()


def init_sequence_parallel_group(args, guidance_parallel=False):
    """:21-47: consecutive-rank groups of `args.sp_group_size` over the first `args.sp_proc_num` processes
    (-1 = all of them); must run before the model is built (the engines pick their sequence-parallel form then).
    guidance_parallel (not a reference option, a world of two only): the ranks split the classifier-free-guidance pair
    instead of the sequence (pyflow_hip/flux_cfg.py)."""
    assert not _sp.is_sequence_parallel_initialized(), "sequence parallel group is already initialized"
    assert dist.is_available() and dist.is_initialized(), "The pytorch distributed should be initialized"
    print(f"Setting the Sequence Parallel Size {args.sp_group_size}")
    _sp.init_sequence_parallel_group(args, guidance_parallel=guidance_parallel)


def get_sequence_parallel_group():
    comm = _sp.get_sequence_parallel_comm()
    assert comm is not None, "sequence parallel group is not initialized"
    return comm.group if comm.group is not None else dist.group.WORLD


def get_sequence_parallel_world_size():
    assert _sp.is_sequence_parallel_initialized(), "sequence parallel size is not initialized"
    return _sp.get_sequence_parallel_world_size()


def get_sequence_parallel_rank():
    assert _sp.is_sequence_parallel_initialized(), "sequence parallel size is not initialized"
    return get_rank() % _sp.get_sequence_parallel_world_size()


def get_sequence_parallel_group_rank():
    assert _sp.is_sequence_parallel_initialized(), "sequence parallel size is not initialized"
    return get_rank() // _sp.get_sequence_parallel_world_size()


def get_sequence_parallel_proc_num():
    return _sp.get_sequence_parallel_proc_num()
