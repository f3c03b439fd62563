"""The generated schedules of the lab kernels are what their generators emit (lab/gen_gemm4w_body.py -> lab/gemm4w_lab.hip,
lab/gen_attn128_body.py -> lab/attn128_pipe.h): regenerating into a copy changes nothing.  The generators also carry the
consistency checks of the schedules (steady state reached, counted s_waitcnt values from a replay of the in-order queues),
which run as part of this."""
import importlib.util
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("gen,target", [("gen_gemm4w_body.py", "gemm4w_lab.hip"), ("gen_attn128_body.py", "attn128_pipe.h")])
def test_regenerating_is_idempotent(tmp_path, monkeypatch, gen, target):
    lab = os.path.join(ROOT, "lab")
    shutil.copy(os.path.join(lab, target), tmp_path / target)
    mod = _load(os.path.join(lab, gen), gen[:-3])
    monkeypatch.setattr(mod, "HERE", str(tmp_path))
    monkeypatch.setattr(sys, "argv", [gen])
    mod.main()
    assert open(tmp_path / target).read() == open(os.path.join(lab, target)).read()


def test_gemm4w_waits_are_counted_from_the_issue_order():
    """a needed load with n later vector-memory operations in flight is waited for with vmcnt(n) (6-bit counter: at most 63), a
    needed fragment read with n later LDS operations with lgkmcnt(n)"""
    mod = _load(os.path.join(ROOT, "lab", "gen_gemm4w_body.py"), "gen_gemm4w_body")
    for nb in (4, 3):
        mod.NB = nb
        ev, marks = mod.run_events(12, stores_after=5)
        waits = mod.annotate(ev)
        nmem = 8 + 2 * nb
        issued = []
        for n, e in enumerate(ev):
            if e[0] in ("ld", "st"):
                issued.append(n)
            if e[0] == "wr":
                kind, cnt = waits[n]
                src = max(k for k in range(n) if ev[k][0] == "ld" and ev[k][1:3] == e[1:3])
                later = sum(1 for k in issued if k > src)
                assert kind == "vm" and cnt == min(later, 63)
        assert sum(1 for e in ev if e[0] == "ld") == 12 * nmem and sum(1 for e in ev if e[0] == "wr") == 12 * nmem
