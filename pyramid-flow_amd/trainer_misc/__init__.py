"""Drop-in for the inference-side names of the reference's `trainer_misc` package (trainer_misc/__init__.py:1-28):
the process-group set-up `inference_multigpu.py:10-11, 34-39` calls, the sequence-parallel getters
(trainer_misc/sp_utils.py:14-98) and `all_to_all` (trainer_misc/communicate.py:55-66).  The training half of the
package (optimizer / scheduler factories, FSDP and DDP trainers, checkpoint savers) is out of scope."""
from .utils import (  # noqa: F401
    get_rank,
    get_world_size,
    is_main_process,
    is_dist_avail_and_initialized,
    init_distributed_mode,
    setup_for_distributed,
)
from .sp_utils import (  # noqa: F401
    is_sequence_parallel_initialized,
    init_sequence_parallel_group,
    get_sequence_parallel_group,
    get_sequence_parallel_world_size,
    get_sequence_parallel_rank,
    get_sequence_parallel_group_rank,
    get_sequence_parallel_proc_num,
)
from .communicate import all_to_all  # noqa: F401
