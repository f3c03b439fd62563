"""GROUPED launch of the persistent GEMM (pf_gemm_desc.A2 ..., ABI 6): the second problem's tiles (the 128 text rows of a
double-stream block: flux_block.py:816-835, 868-872, 1022-1036 run every projection once per stream with other weights) ride in
the first problem's launch.  A tile is computed by one workgroup in K order whether or not the launch is grouped, so both
problems' results must equal the two separate launches BIT FOR BIT -- every flavour the double blocks use (K|V|Q with the
QK-RMSNorm + RoPE epilogue and per-problem gains / RoPE rows, GELU, gate * x + residual in place), batch 2 with strides and
row offsets, with and without the tail split; a shape whose first problem does not take the persistent kernel runs as two
launches (same bits again); and a whole miniFLUX forward with grouped double blocks against the two-stream form."""
import ctypes as C

import pytest
import torch

from util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _rope_table(rows, seed):
    ang = torch.rand(rows, 32, generator=torch.Generator().manual_seed(seed)) * 6.28
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous().to(DEV)


@pytest.mark.parametrize("L_img", [4400, 3008, 200])          # 200: the first problem is too small for the persistent kernel
def test_grouped_launch_equals_two_launches(L_img):
    from pyflow_hip import ops
    from pyflow_hip.ops import GEMM_GATE_RES
    d, B, Lt = 1920, 2, 128
    L = Lt + L_img
    ws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    x = _mk((B, L, d), 1).to(torch.bfloat16).to(DEV)
    W_img = _mk((4 * d, d), 2, 0.03).to(torch.bfloat16).to(DEV)
    W_txt = _mk((4 * d, d), 3, 0.03).to(torch.bfloat16).to(DEV)
    b_img, b_txt = _mk((4 * d,), 4).to(DEV), _mk((4 * d,), 5).to(DEV)
    gate = _mk((B, 12 * d), 6).to(DEV)
    wq, wk, wq2, wk2 = [(1.0 + 0.3 * _mk((64,), 7 + i)).to(DEV) for i in range(4)]
    rope = _rope_table(L, 11)
    which = ops.L.load().pf_gemm_which(L_img, B, d, d)
    assert (which == 8) == (L_img >= 3008)

    def run(grouped, kind):
        out = torch.zeros(B, L, 4 * d, dtype=torch.bfloat16, device=DEV)
        hid = x.clone()
        if kind == "kvq":            # N = 3d, QK epilogue: image rows use (wq, wk) and RoPE rows Lt + m, text rows (wq2, wk2) and rows m
            qk = dict(rope=rope, wq=wq, wk=wk, d=d, k_col0=0, q_col0=2 * d, row0=Lt, eps=1e-6, q_scale=0.18)
            kw = dict(bias=b_img, batch=B, strideA=L * d, strideC=L * 4 * d, a_off=Lt * d, c_off=Lt * 4 * d, qk=qk)
            sec = dict(M=Lt, W=W_txt, bias=b_txt, wq=wq2, wk=wk2, row0=0)
            args = (x, W_img, out, L_img, 3 * d, d, d, d, 4 * d)
        elif kind == "gelu":         # N = 4d, GELU on every column
            kw = dict(bias=b_img, batch=B, strideA=L * d, strideC=L * 4 * d, a_off=Lt * d, c_off=Lt * 4 * d, gelu_from=0,
                      tail_workspace=ws)
            sec = dict(M=Lt, W=W_txt, bias=b_txt)
            args = (x, W_img, out, L_img, 4 * d, d, d, d, 4 * d)
        else:                        # N = d: hidden = hidden + gate * (x W^T + b), in place, per-stream gate columns
            kw = dict(bias=b_img, res=hid, gate=gate, gate_off=2 * d, ldr=d, batch=B, strideA=L * d, strideC=L * d, strideR=L * d,
                      gate_stride=12 * d, flags=GEMM_GATE_RES, a_off=Lt * d, c_off=Lt * d, r_off=Lt * d, tail_workspace=ws)
            sec = dict(M=Lt, W=W_txt, bias=b_txt, gate_off=8 * d)
            args = (x, W_img, hid, L_img, d, d, d, d, d)
        if grouped:
            ops.gemm(*args, **kw, second=sec)
        else:
            ops.gemm(*args, **kw)
            kw2 = dict(kw)
            kw2.update(a_off=sec.get("a_off", 0), c_off=sec.get("c_off", 0), bias=sec["bias"])
            if "r_off" in kw2:
                kw2.update(r_off=0, gate_off=sec["gate_off"])
            if "qk" in kw2:
                kw2["qk"] = dict(kw["qk"], wq=sec["wq"], wk=sec["wk"], row0=sec["row0"])
            # the scratch the grouped descriptor carries (ops.gemm hands `tail_workspace` over when the FIRST problem takes the
            # persistent kernel) is what the library's two-launch fallback gives the second problem too
            had_ws = "tail_workspace" in kw and ops.L.load().pf_gemm_which(args[3], B, args[4], args[5]) == 8
            kw2.pop("tail_workspace", None)
            a2 = list(args)
            a2[1], a2[3] = sec["W"], sec["M"]
            ops.gemm(*a2, **kw2, workspace=ws if had_ws else None)
        torch.cuda.synchronize()
        return hid if kind == "res" else out

    for kind in ("kvq", "gelu", "res"):
        if which == 8:
            # with K-split scratch the tail plan depends on the launch's tile count, i.e. other tiles are split when the text
            # tiles join the list: equal to one bf16 ulp (fp32 summation order), like every tail-split comparison ...
            g, s = run(True, kind), run(False, kind)
            assert rel_l2(g.float(), s.float()) < 3e-3
            assert (g.float() - s.float()).abs().max() <= 2 ** -6 * s.float().abs().max()
        ops.gemm_set_policy(-4)          # ... and bit for bit on whole tiles
        g, s = run(True, kind), run(False, kind)
        if which == 8:
            # image rows: same kernel, same tiles -> same bits; text rows: persistent kernel in both (the separate text launch of
            # this test is forced through it below) -- compare against the library's own two-launch fallback instead
            assert torch.equal(g[:, Lt:], s[:, Lt:]), kind
            ops.gemm_set_policy(8)
            ops.gemm_set_policy(-4)
            try:
                s8 = run(False, kind)
            finally:
                ops.gemm_set_policy(0)
            assert torch.equal(g[:, :Lt], s8[:, :Lt]), f"{kind}: text rows differ from the separate persistent launch"
            assert rel_l2(g[:, :Lt].float(), s[:, :Lt].float()) < 3e-3          # vs the 128 x 128 kernel: summation order only
        else:
            assert torch.equal(g, s), kind                                       # not grouped by the library: two launches, same bits
        ops.gemm_set_policy(0)
        assert g.abs().max() > 0


def test_forward_with_grouped_double_blocks():
    """miniFLUX (2 double + 1 single block) at L = 3 008 / 4 208: the grouped form against the two-stream form -- image and text
    rows from the same persistent kernel vs text rows from the 128 x 128 kernel with a K split: fp32 summation order only."""
    from pyflow_hip import synth
    from pyflow_hip.flux import FluxEngine
    cfg = dict(synth.MINIFLUX, num_layers=2, num_single_layers=1)
    g = torch.Generator().manual_seed(3)
    sd = synth.random_state_dict(synth.flux_param_shapes(cfg), seed=5, std=0.02, lively=True)
    for shapes in ([(3, 24, 40), (1, 24, 40), (1, 48, 80), (1, 48, 80)], [(5, 24, 40), (1, 48, 80), (1, 48, 80), (1, 48, 80)]):
        clips = [torch.randn(1, 16, *s_, generator=g).to(DEV) for s_ in shapes]
        enc = torch.randn(2, 128, 4096, generator=g).to(torch.bfloat16)
        mask = torch.zeros(2, 128, dtype=torch.long)
        mask[0, :40] = 1
        mask[1, :96] = 1
        pooled = torch.randn(2, 768, generator=g)
        eng = FluxEngine(sd, cfg, DEV)
        plan = eng.make_plan(shapes, mask)
        eng.encode_context(enc)
        outs = {}
        for grp in (True, False, True):
            eng.group_text = grp
            for mode in ("eager", "graph"):
                eng.launch_mode = mode
                v = eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True).clone()
                assert torch.isfinite(v).all()
                outs.setdefault((grp, mode), v)
                assert torch.equal(outs[(grp, mode)], v)                       # repeatable
        assert torch.equal(outs[(True, "eager")], outs[(True, "graph")])
        # two bf16 evaluations of one forward that differ in fp32 summation order: the text rows come from the persistent
        # kernel (K in one piece) instead of the 128 x 128 kernel with a K split, and the image GEMMs' tail plans see 2 x N / 256
        # more tiles (other tiles are split along K).  Measured 2.9e-3 / 2.2e-3 over 3 blocks; the same class of difference as
        # "tail split on / off" (tests/test_fullsize_gpu.py: 2e-3 over one block of each kind at L = 15 488)
        e = rel_l2(outs[(True, "eager")].cpu(), outs[(False, "eager")].cpu())
        # ... and with whole tiles only (no tail split) what is left is the text rows' kernel
        from pyflow_hip import ops
        ops.gemm_set_policy(-4)
        try:
            eng.launch_mode = "eager"
            eng.group_text = True
            a = eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True).clone()
            eng.group_text = False
            b = eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True).clone()
        finally:
            ops.gemm_set_policy(4)
        e_whole = rel_l2(a.cpu(), b.cpu())
        print(f"L = {plan.L}: grouped vs two-stream double blocks rel-L2 {e:.3e}; whole tiles only {e_whole:.3e}")
        assert e < 5e-3 and e_whole < 5e-3
