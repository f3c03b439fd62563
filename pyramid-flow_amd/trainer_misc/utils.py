"""Process-group helpers with the reference's names and environment contract (trainer_misc/utils.py:27-106).

`init_distributed_mode(args)` reads OMPI_COMM_WORLD_{RANK,LOCAL_RANK,SIZE} or RANK / WORLD_SIZE / LOCAL_RANK, fills
`args.rank / .world_size / .gpu / .distributed / .dist_backend / .dist_url`, binds the process to its GPU and
initialises the default group over env:// -- backend "nccl" (= RCCL over xGMI on MI355X), one process per GPU.  On a
host without a GPU (the CPU test tier) the backend is "gloo"; nothing else differs."""
import datetime
import os

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def setup_for_distributed(is_master):
    """trainer_misc/utils.py:56-68: print only on the master process unless force=True is passed"""
    import builtins
    builtin_print = builtins.print
    if getattr(builtin_print, "_pyflow_wrapped", False):
        builtin_print = builtin_print._pyflow_orig

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    print._pyflow_wrapped = True
    print._pyflow_orig = builtin_print
    builtins.print = print


def init_distributed_mode(args, init_pytorch_ddp=True):
    if int(os.getenv("OMPI_COMM_WORLD_SIZE", "0")) > 0:
        os.environ["LOCAL_RANK"] = os.environ["OMPI_COMM_WORLD_LOCAL_RANK"]
        os.environ["RANK"] = os.environ["OMPI_COMM_WORLD_RANK"]
        os.environ["WORLD_SIZE"] = os.environ["OMPI_COMM_WORLD_SIZE"]
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", "0"))
    else:
        print("Not using distributed mode")
        args.distributed = False
        return
    args.distributed = True
    have_gpu = torch.cuda.is_available()
    args.dist_backend = "nccl" if have_gpu else "gloo"
    args.dist_url = "env://"
    print("| distributed init (rank {}): {}, gpu {}".format(args.rank, args.dist_url, args.gpu), flush=True)
    if init_pytorch_ddp:
        if have_gpu:
            torch.cuda.set_device(args.gpu)
        if not dist.is_initialized():
            dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                                    rank=args.rank, timeout=datetime.timedelta(days=365))
        dist.barrier()
        setup_for_distributed(args.rank == 0)
