"""The drop-in boundary: module paths, names and call signatures the reference's own callers use.

* `test_reference_caller_binds` (dev container only): parses the UNMODIFIED /root/reference/inference_multigpu.py with
  `ast` and checks that every name it imports from `pyramid_dit` / `trainer_misc`, every attribute chain it touches on
  the model object and every keyword it passes to `generate` / `generate_i2v` / the constructor exists here and binds
  (`inspect.signature(...).bind`).  Same for the reference package `__init__` files (all re-exported inference names).
* trainer_misc / utils / video_vae.context_parallel_ops over gloo, world size 2: the environment contract of
  `init_distributed_mode`, group bookkeeping getters, `all_to_all` against its definition
  (tensor_split -> exchange -> cat, trainer_misc/communicate.py:7-26).
"""
import ast
import inspect
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REF = "/root/reference"
_DROPIN = ("pyramid_dit", "trainer_misc", "utils", "video_vae", "diffusion_schedulers")


@pytest.fixture(autouse=True)
def _isolate_dropin_modules():
    """the drop-in packages carry the reference's top-level names: forget them afterwards so that the tests which
    import the real reference from /root/reference (tests/test_oracle_vs_reference.py) see their own modules"""
    import sys
    before = {k for k in sys.modules if k.split(".")[0] in _DROPIN}
    yield
    for k in [k for k in sys.modules if k.split(".")[0] in _DROPIN and k not in before]:
        del sys.modules[k]


def _calls_on(tree, root):
    """(attr chain tuple, keyword names) for every call whose function is an attribute chain starting at `root`"""
    out = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call):
            chain, f = [], node.func
            while isinstance(f, ast.Attribute):
                chain.append(f.attr)
                f = f.value
            if isinstance(f, ast.Name) and f.id == root and chain:
                out.append((tuple(reversed(chain)), [k.arg for k in node.keywords], len(node.args)))
    return out


@pytest.mark.reference
def test_reference_caller_binds():
    import pyramid_dit
    import trainer_misc
    src = open(os.path.join(REF, "inference_multigpu.py")).read()
    tree = ast.parse(src)
    mods = {"pyramid_dit": pyramid_dit, "trainer_misc": trainer_misc}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module in mods:
            for a in node.names:
                assert hasattr(mods[node.module], a.name), f"{node.module}.{a.name} missing"
    cls = pyramid_dit.PyramidDiTForVideoGeneration
    # constructor call: PyramidDiTForVideoGeneration(args.model_path, model_dtype, model_name=..., model_variant=...)
    ctor = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name)
            and n.func.id == "PyramidDiTForVideoGeneration"]
    assert ctor
    for c in ctor:
        inspect.signature(cls.__init__).bind(None, *([0] * len(c.args)), **{k.arg: 0 for k in c.keywords})
    from pyflow_hip.flux import FluxEngine
    from pyflow_hip.vae import CausalVideoVAE
    from pyflow_hip.text_encoder import FluxTextEncoderWithMask
    owners = {"vae": CausalVideoVAE, "dit": FluxEngine, "text_encoder": FluxTextEncoderWithMask}
    seen = set()
    for chain, kws, nargs in _calls_on(tree, "model"):
        seen.add(chain)
        if len(chain) == 1:
            fn = getattr(cls, chain[0])
        else:
            fn = getattr(owners[chain[0]], chain[1])
        inspect.signature(fn).bind(None, *([0] * nargs), **{k: 0 for k in kws})
    assert {("vae", "to"), ("dit", "to"), ("text_encoder", "to"), ("vae", "enable_tiling"), ("generate",),
            ("generate_i2v",)} <= seen


@pytest.mark.reference
def test_reference_package_exports_exist():
    """every inference-side name the reference's package __init__ files re-export is importable from the same path"""
    import importlib
    skip = {  # training half (out of scope)
        "create_optimizer", "cosine_scheduler", "constant_scheduler", "NativeScalerWithGradNormCount", "auto_load_model",
        "save_model", "init_sync_input_group", "get_sync_input_group", "train_one_epoch_with_fsdp", "train_one_epoch",
        "CausalVideoVAELossWrapper", "LPIPSWithDiscriminator",
        "DDPMCosineScheduler"}          # scheduling_cosine_ddpm.py: not used by the sampling pipeline
    for pkg in ("pyramid_dit", "pyramid_dit.flux_modules", "pyramid_dit.mmdit_modules", "video_vae", "diffusion_schedulers",
                "trainer_misc"):
        tree = ast.parse(open(os.path.join(REF, pkg.replace(".", "/"), "__init__.py")).read())
        mine = importlib.import_module(pkg)
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom):
                for a in node.names:
                    if a.name in skip:
                        continue
                    assert hasattr(mine, a.name), f"{pkg}.{a.name} missing"
    import utils
    import video_vae.context_parallel_ops as cpo
    for n in ("initialize_context_parallel", "is_context_parallel_initialized", "get_context_parallel_group",
              "get_context_parallel_world_size", "get_context_parallel_rank", "get_context_parallel_group_rank"):
        assert callable(getattr(utils, n))
    for n in ("conv_scatter_to_context_parallel_region", "conv_gather_from_context_parallel_region",
              "cp_pass_from_previous_rank"):
        assert callable(getattr(cpo, n))


def test_dit_forward_signature_matches_reference_kwargs():
    from pyramid_dit.flux_modules import PyramidFluxTransformer
    from pyramid_dit.mmdit_modules import PyramidDiffusionMMDiT
    for cls in (PyramidFluxTransformer, PyramidDiffusionMMDiT):
        sig = inspect.signature(cls.forward)
        assert list(sig.parameters)[1:] == ["sample", "encoder_hidden_states", "encoder_attention_mask",
                                             "pooled_projections", "timestep_ratio"]        # flux:392-399, mmdit:420-427
        assert callable(cls.from_pretrained) and callable(cls.to) and callable(cls.eval)


def test_module_api_is_device_checked():
    from pyflow_hip.refapi import DeviceModuleAPI
    m = DeviceModuleAPI()
    m.dev = torch.device("cuda:0")
    assert m.to("cuda") is m and m.to(torch.device("cuda", 0)) is m and m.to(torch.bfloat16) is m and m.eval() is m
    assert m.device == torch.device("cuda:0") and m.dtype == torch.bfloat16
    with pytest.raises(RuntimeError):
        m.to("cpu")
    with pytest.raises(NotImplementedError):
        m.train()


def _tm_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import trainer_misc
    import utils
    from video_vae.context_parallel_ops import cp_pass_from_previous_rank
    args = types.SimpleNamespace(sp_group_size=world, sp_proc_num=-1)
    trainer_misc.init_distributed_mode(args)                       # trainer_misc/utils.py:71-106
    ok = [args.distributed, args.rank == rank, args.world_size == world, args.gpu == rank, dist.is_initialized(),
          not trainer_misc.is_sequence_parallel_initialized()]
    trainer_misc.init_sequence_parallel_group(args)                # sp_utils.py:21-47
    ok += [trainer_misc.is_sequence_parallel_initialized(), trainer_misc.get_sequence_parallel_world_size() == world,
           trainer_misc.get_sequence_parallel_rank() == rank, trainer_misc.get_sequence_parallel_group_rank() == 0,
           trainer_misc.get_sequence_parallel_proc_num() == world, trainer_misc.get_rank() == rank,
           trainer_misc.get_world_size() == world, trainer_misc.is_main_process() == (rank == 0)]
    grp = trainer_misc.get_sequence_parallel_group()
    # all_to_all: heads scattered (dim 2), sequence gathered (dim 1)  (flux_block.py:286-296 usage)
    full = torch.arange(2 * 3 * world * 4 * world * 2, dtype=torch.float32).reshape(2, 3 * world, 4 * world, 2)
    mine = full[:, rank * 3:(rank + 1) * 3]                                     # my sequence chunk, all heads
    out = trainer_misc.all_to_all(mine.contiguous(), grp, world, scatter_dim=2, gather_dim=1)
    ok.append(torch.equal(out, full[:, :, rank * 4:(rank + 1) * 4]))            # all rows, my heads
    first = trainer_misc.all_to_all(mine.contiguous(), grp, world, scatter_dim=2, gather_dim=1, concat_output=False)
    ok.append(torch.equal(first, full[:, :3, rank * 4:(rank + 1) * 4]))
    # context parallel bookkeeping + halo pass through the reference's module paths
    utils.initialize_context_parallel(world)
    ok += [utils.is_context_parallel_initialized(), utils.get_context_parallel_world_size() == world,
           utils.get_context_parallel_rank() == rank, utils.get_context_parallel_group_rank() == 0]
    x = torch.full((1, 2, 3, 2, 2), float(rank + 1))
    h = cp_pass_from_previous_rank(x, 2, 3)
    exp_front = 0.0 if rank == 0 else float(rank)
    ok.append(h.shape[2] == 5 and bool((h[:, :, :2] == exp_front).all()) and bool((h[:, :, 2:] == rank + 1).all()))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_misc_and_utils_over_gloo():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(all(r[1]) for r in res), res


def test_init_distributed_mode_without_env(monkeypatch):
    import trainer_misc
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMPI_COMM_WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    args = types.SimpleNamespace()
    trainer_misc.init_distributed_mode(args)
    assert args.distributed is False
