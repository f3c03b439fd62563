"""ISA lint of the hand-scheduled kernels (CPU tier: hipcc cross-compiles gfx950 without a GPU).

The persistent GEMM and the halo convolution order their vector-memory traffic with COUNTED `s_waitcnt vmcnt(N)` written as
inline asm.  Two things the compiler can silently add break that scheme without breaking a single result:
  * a scratch access (spill) -- a vector-memory operation the counts do not know about;
  * a wait of its own in front of the first use of a register that a global load filled: where that use has sunk behind the
    epilogue's stores, it becomes an `s_waitcnt vmcnt(0)` that waits for the tile's whole store burst (rounds 2-4 shipped
    exactly that: gemm8p's residual flavour issued its 16 stores as 16 load + store round trips, the bias-folding flavours
    drained every tile's stores before touching the next tile; round 5 found it in the ISA, not in a profile).
This test compiles the two sources to ISA and checks, per kernel: no scratch, and the tile's stores form runs of 16
`global_store_dwordx4` (32 for the park path / the fp32 flavour) with no `s_waitcnt vmcnt` and no load between them and no
such wait directly behind them.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyramid-flow_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _isa(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                    "-Wno-unused-value", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):.*?^\s*\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        body = text[m.start():m.start(2)]
        meta = m.group(2)
        kernels[m.group(1)] = (body, meta)
    return kernels


def _store_runs(body):
    """lengths of the runs of global_store_dwordx4 that no s_waitcnt vmcnt and no load interrupts, and per run whether a
    vmcnt wait follows it directly.  Barriers do not end a run: the listing is textual, and hipcc lays the block that holds a
    tile's last store out behind the loop's barrier (round 6: `S15 B S1 B`)."""
    ops = []
    for ln in body.splitlines():
        t = ln.split(";")[0].strip()
        if t.startswith("global_store_dwordx4"):
            ops.append("S")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            ops.append("W")
        elif t.startswith(("global_load", "global_atomic", "buffer_", "scratch_")):
            ops.append("L")
    runs, clean = [], []
    for m in re.finditer(r"S+", "".join(ops)):
        runs.append(len(m.group(0)))
        clean.append("".join(ops)[m.end():m.end() + 1] != "W")
    return runs, clean


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemm8p_epilogue_stores_are_not_guarded_by_compiler_waits(tmp_path):
    ks = {k: v for k, v in _isa("gemm8p.hip", tmp_path).items() if "gemm8p_kernel" in k}
    assert len(ks) >= 6                      # conv, QK (12), residual (1), fp32 (2), GELU (4), plain (0)
    for name, (body, meta) in ks.items():
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta), f"{name}: scratch in a kernel with counted waits"
        assert "scratch_" not in body, name
        runs, clean = _store_runs(body)
        # the tile's stores go out in whole blocks of 16 (one per (row fragment, column half)): a shorter run is a store that
        # something -- a wait, a load -- separates from its neighbours
        assert runs and all(r % 16 == 0 for r in runs), (name, runs)
        assert all(clean), f"{name}: a vmcnt wait directly behind a run of stores {list(zip(runs, clean))}"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_halo_conv_epilogue_loads_before_stores(tmp_path):
    ks = {k: v for k, v in _isa("convhalo.hip", tmp_path).items() if "conv_halo128_kernel" in k}
    assert ks
    for name, (body, meta) in ks.items():
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta), name
        runs, _ = _store_runs(body)
        assert runs and all(r % 16 == 0 for r in runs), (name, runs)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src, marker, n_min", [("gemm.hip", "gemm_kernel", 4), ("gemm256.hip", "gemm256_kernel", 6),
                                                 ("attention.hip", "attn", 8)])
def test_no_scratch_in_the_other_kernels_with_counted_waits(tmp_path, src, marker, n_min):
    """the 128 x 128 kernel's three-stage ring (vmcnt(8)), the 256-row ping-pong kernel and the attention kernels count their
    LDS-DMA pieces too: a spill would be a vector-memory operation those counts do not know about"""
    ks = {k: v for k, v in _isa(src, tmp_path).items() if marker in k}
    assert len(ks) >= n_min, sorted(ks)
    for name, (body, meta) in ks.items():
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta), f"{name}: scratch"
        assert "scratch_" not in body, name
    if src == "gemm.hip":
        # the three-stage main loop: its only vector-memory waits are the two the source writes (the counted one, and the full
        # drain of the last K-step) -- the compiler adds none in front of the LDS reads
        for name, (body, _) in ks.items():
            if "ELi3EE" not in name:
                continue
            lines = [ln.split(";")[0].strip() for ln in body.splitlines()]
            first = next(i for i, t in enumerate(lines) if t.startswith("s_waitcnt vmcnt(8)"))
            last_mfma = max(i for i, t in enumerate(lines) if t.startswith("v_mfma"))
            between = [t for t in lines[first + 1:last_mfma] if t.startswith("s_waitcnt") and "vmcnt" in t]
            assert between == [], (name, between)
