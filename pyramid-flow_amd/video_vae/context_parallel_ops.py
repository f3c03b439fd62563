"""Drop-in module path for video_vae/context_parallel_ops.py:158-167 -- the three tensor-level primitives of the
temporal context parallelism (inference semantics; the reference wraps them in autograd Functions for training):
  conv_scatter_to_context_parallel_region(input_, dim, kernel_size)   local slice, rank 0 keeps the first k frames
  conv_gather_from_context_parallel_region(input_, dim, kernel_size)  all ranks get the concatenation
  cp_pass_from_previous_rank(input_, dim, kernel_size)                [halo of k-1 frames | input_], zeros on rank 0
Implemented in pyflow_hip/cp.py on the same communicator the HIP decode uses for its halo exchange
(CausalVideoVAE.decode_context_parallel exchanges the halo directly between channels-last activation buffers and
also handles the uneven frame ranges the reference's split cannot express)."""
from pyflow_hip.cp import (  # noqa: F401
    conv_scatter_to_context_parallel_region,
    conv_gather_from_context_parallel_region,
    cp_pass_from_previous_rank,
)
