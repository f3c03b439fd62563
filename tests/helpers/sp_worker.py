"""worker of tests/test_sp_gpu.py: one rank of a P-process sequence-parallel miniFLUX forward.
All ranks share cuda:0 (the GPU box has one GPU); the exchange runs over gloo staged through the host, i.e. the
same SPComm code path as RCCL except for the transport.  Rank 0 also runs the single-process engine and compares."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    heads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    variant = sys.argv[3] if len(sys.argv) > 3 else "flux"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from pyflow_hip import synth
    from pyflow_hip.flux import FluxEngine
    from pyflow_hip.flux_sp import FluxEngineSP
    from pyflow_hip.sp import init_sequence_parallel_group
    from util import rel_l2, round_sd
    comm = init_sequence_parallel_group(sp_group_size=world)
    if variant == "mmdit":
        cfg = dict(synth.tiny_mmdit_cfg(), num_attention_heads=heads, caption_projection_dim=heads * 64)
        sd = round_sd(synth.mmdit_state_dict(cfg, seed=3, std=0.05, lively=True))
        sd["pos_embed.pos_embed"] = synth.mmdit_state_dict(cfg, seed=3)["pos_embed.pos_embed"]
    else:
        cfg = dict(synth.TINY_FLUX, num_attention_heads=heads)
        if heads > 10:          # released width: one double + one single block (8 rank processes build it on one GPU)
            cfg.update(num_layers=1, num_single_layers=1)
        # (the released width -- 30 heads, d = 1920 -- keeps activations in range with the released init scale)
        sd = round_sd(synth.random_state_dict(synth.flux_param_shapes(cfg), seed=3, std=0.05 if heads <= 10 else 0.02, lively=True))
    g = torch.Generator().manual_seed(0)
    shapes = [(2, 4, 8), (1, 8, 16), (1, 16, 32), (1, 16, 32)]
    clips = [torch.randn(2, 16, *s, generator=g).to(torch.bfloat16).float().cuda() for s in shapes]
    enc = torch.randn(2, 16, 32, generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(2, 16, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(2, 16, generator=g)
    t = [704.0, 704.0]
    eng = FluxEngineSP(sd, cfg, "cuda", comm=comm)
    plan = eng.make_plan(shapes, mask)
    eng.encode_context(enc)
    v = eng.forward_tokens(plan, clips, t, pooled).clone()
    # the same forward with the rank's small image GEMMs NOT split along K (FluxEngineSP.split_small = False): the summation
    # order of the single-rank engine -> the tight bound below; the default (split) is a perf choice with a measured error
    eng.split_small = False
    v_ns = eng.forward_tokens(plan, clips, t, pooled).clone()
    eng.split_small = True
    # round 5: with torch.distributed collectives the kernels between two exchanges replay from launch-list segments
    # (flux_sp._SegmentedProgram); the eager issue of the same sequence must give the same bits, also on a second replay
    assert eng.launch_mode == "list" and not getattr(eng.comm, "recordable", False)
    seg = getattr(plan, "_sp_list", None)
    assert seg is not None and type(seg[1]).__name__ == "_SegmentedProgram" and len(seg[1]) > 8, (seg, len(seg[1]))
    v_again = eng.forward_tokens(plan, clips, t, pooled).clone()
    eng.launch_mode = "eager"
    v_eager = eng.forward_tokens(plan, clips, t, pooled).clone()
    eng.launch_mode = "list"
    torch.cuda.synchronize()
    ok = bool(torch.equal(v, v_again) and torch.equal(v, v_eager))
    if rank == 0:
        ref_eng = FluxEngine(sd, cfg, "cuda")
        ref_eng.encode_context(enc)
        ref = ref_eng.forward_tokens(plan, clips, t, pooled).clone()
        err = rel_l2(v.cpu(), ref.cpu())
        err_ns = rel_l2(v_ns.cpu(), ref.cpu())
        # ... and against the CPU oracle (the reference's arithmetic), not only against the single-rank HIP engine
        if variant == "mmdit":
            from oracle.mmdit_oracle import mmdit_forward as oracle_forward
        else:
            from oracle.flux_oracle import flux_forward as oracle_forward
        o_ref = oracle_forward(sd, cfg, [c.cpu() for c in clips], enc, mask, pooled, torch.tensor(t))
        tcur, hcur, wcur = plan.cur
        Cc = eng.w.out_cols // 4
        x = v.cpu()[:, :, :eng.w.out_cols].reshape(2, tcur, hcur // 2, wcur // 2, 2, 2, Cc)
        x = x.permute(0, 1, 2, 4, 3, 5, 6).reshape(2, tcur, hcur, wcur, Cc).permute(0, 4, 1, 2, 3)
        err_oracle = rel_l2(x, o_ref)
        # (vs the single-rank engine: two bf16 evaluations whose small GEMMs split K differently since round 4)
        ok = ok and err < 5e-3 and err_ns < 2e-3 and err_oracle < 2e-2
        with open(out_path, "w") as f:
            f.write(f"vs single-rank HIP {err:.3e} (small GEMMs unsplit: {err_ns:.3e}), vs oracle {err_oracle:.3e} {int(ok)} world={world} heads={heads} "
                    f"lay_rows={eng.layout(plan).rows} lay_heads={eng.layout(plan).heads}\n")
    # all ranks must hold the same replicated result
    gathered = [torch.empty_like(v.cpu()) for _ in range(world)]
    dist.all_gather(gathered, v.cpu())
    same = all(torch.equal(gathered[0], x) for x in gathered)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if (ok and same) else 1)


if __name__ == "__main__":
    main()
