"""Product scheduler vs the committed reference tables (bit-exact) and SURVEY Appendix B known answers."""
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "scheduler_tables.pt")


def _mk(name):
    from pyflow_hip.scheduler import PyramidFlowMatchEulerDiscreteScheduler as S
    return S() if name == "default" else S(stages=1, stage_range=[0, 1])


@pytest.mark.parametrize("name", ["default", "one_stage"])
def test_tables_bit_exact(name):
    g = torch.load(GOLD)[name]
    s = _mk(name)
    assert s.start_sigmas == g["start"] and s.end_sigmas == g["end"] and s.ori_start_sigmas == g["ori"]
    assert s.timestep_ratios == g["ratios"]
    for (st, n), (ts, sg) in g["tables"].items():
        s.set_timesteps(n, st)
        assert torch.equal(s.timesteps, ts), (st, n)
        assert torch.equal(s.sigmas, sg), (st, n)
        assert s.timesteps.dtype == ts.dtype and s.sigmas.dtype == sg.dtype


@pytest.mark.parametrize("name", ["default", "one_stage"])
def test_oracle_scheduler_matches_golden(name):
    from oracle.scheduler_oracle import SchedulerOracle
    g = torch.load(GOLD)[name]
    s = SchedulerOracle() if name == "default" else SchedulerOracle(stages=1, stage_range=[0, 1])
    assert s.start_sigmas == g["start"] and s.end_sigmas == g["end"]
    for (st, n), (ts, sg) in g["tables"].items():
        s.set_timesteps(n, st)
        assert torch.equal(s.timesteps, ts) and torch.equal(s.sigmas, sg)


def test_appendix_b_known_answers():
    s = _mk("default")
    assert s.start_sigmas == {0: 1.0, 1: 0.8002399489209289, 2: 0.5007496155411024}
    assert s.end_sigmas == {0: 0.6669999957084656, 1: 0.33399999141693115, 2: 0.0}
    s.set_timesteps(10, 1)
    assert abs(s.timesteps[0].item() - 744.0) < 1e-3 and abs(s.timesteps[1].item() - 704.262) < 1e-3
    assert s.timesteps[1].to(torch.bfloat16).item() == 704.0
    assert abs(s.sigmas[1].item() - 0.889) < 1e-9 and s.sigmas[-1].item() == 0.0
    s.set_timesteps(20, 0)
    assert abs(s.timesteps[-1].item() - 744.256) < 1e-3


def test_step_semantics_and_dsigma():
    s = _mk("default")
    s.set_timesteps(10, 1)
    x = torch.randn(1, 16, 1, 4, 4)
    v = torch.randn(1, 16, 1, 4, 4).to(torch.bfloat16)
    out = s.step(v, s.timesteps[0], x.to(torch.bfloat16)).prev_sample
    d = s.sigmas[1] - s.sigmas[0]
    exp = (x.to(torch.bfloat16).float() + (d * v)).to(torch.bfloat16)     # product rounded to bf16 first
    assert torch.equal(out, exp)
    assert (d * v).dtype == torch.bfloat16
    with pytest.raises(ValueError):
        s.step(v, 3, x)
    s.set_timesteps(10, 1)
    assert abs(s.dsigma() - float(d)) < 1e-15 and s.step_index == 1
