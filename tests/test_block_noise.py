"""Default (vectorised) `sample_block_noise` of the drop-in pipeline vs the reference's definition
(pyramid_dit_for_video_gen_pipeline.py:697-703): every 2x2 block of every (b, c, t) plane is one draw of
N(0, (1+g) I - g 11^T) laid out `(b c t h w) (p q) -> b c t (h p) (w q)`.  The product draws all blocks with one
randn(N, 4) @ L^T (closed-form factor, `block_noise_cholesky`); the reference loops MultivariateNormal.sample().
Same distribution, different random stream -> checked statistically and structurally.  CPU-only (host arithmetic)."""
import math

import torch

from pyflow_hip.pipeline import PyramidDiTForVideoGeneration, block_noise_cholesky


def _pipe(gamma=1 / 3):
    p = object.__new__(PyramidDiTForVideoGeneration)          # sample_block_noise needs the scheduler config only
    p.block_noise_fn = None
    p.scheduler = type("S", (), {"config": type("C", (), {"gamma": gamma})()})()
    return p


def test_cholesky_factor_reproduces_the_covariance():
    for g in (1 / 3, 0.2, 0.05):
        L = block_noise_cholesky(g).double()
        cov = torch.eye(4, dtype=torch.float64) * (1 + g) - torch.ones(4, 4, dtype=torch.float64) * g
        assert (L @ L.T - cov).abs().max() < 1e-6
        assert torch.equal(L, torch.tril(L))
    # g = 1/3: the covariance is singular (block sums vanish); torch's fp32 Cholesky leaves a 4.2e-4 round-off pivot
    # where the closed form gives exactly 0 -- both satisfy L L^T = cov to fp32 accuracy
    from oracle.pipeline_oracle import block_noise_cholesky as torch_factor
    assert (block_noise_cholesky(1 / 3) - torch_factor(1 / 3)).abs().max() < 5e-4


def test_default_block_noise_statistics_and_layout():
    torch.manual_seed(123)
    g = 1 / 3
    bs, ch, t, h, w = 1, 16, 1, 96, 160
    z = _pipe(g).sample_block_noise(bs, ch, t, h, w)
    assert z.shape == (bs, ch, t, h, w) and z.dtype == torch.float32
    # layout: (p q) of block (i, j) sits at rows 2i..2i+1, cols 2j..2j+1
    blocks = z.reshape(bs, ch, t, h // 2, 2, w // 2, 2).permute(0, 1, 2, 3, 5, 4, 6).reshape(-1, 4).double()
    n = blocks.shape[0]
    assert n == bs * ch * t * (h // 2) * (w // 2)
    # singular direction: the four values of a block sum to zero (g = 1/3)
    assert blocks.sum(1).abs().max() < 1e-5
    cov = blocks.T @ blocks / n
    want = torch.eye(4, dtype=torch.float64) * (1 + g) - torch.ones(4, 4, dtype=torch.float64) * g
    tol = 6.0 * (1 + g) / math.sqrt(n)                       # ~6 sigma of a sample covariance entry
    assert (cov - want).abs().max() < tol
    assert blocks.mean(0).abs().max() < 6.0 * math.sqrt((1 + g) / n)
    # different blocks are independent: neighbouring blocks' first entries are uncorrelated
    zz = z[0, 0, 0].double()
    a, b = zz[0::2, 0::2][:, :-1].reshape(-1), zz[0::2, 0::2][:, 1:].reshape(-1)
    assert abs((a * b).mean()) < 6.0 / math.sqrt(a.numel())


def test_default_block_noise_matches_reference_stream_transform():
    """the same standard-normal draws pushed through the reference's (p q) rearrange give the same tensor"""
    from oracle.pipeline_oracle import block_noise_from_normal
    bs, ch, t, h, w = 2, 3, 2, 8, 12
    torch.manual_seed(7)
    z = _pipe().sample_block_noise(bs, ch, t, h, w)
    torch.manual_seed(7)
    eps = torch.randn(bs * ch * t * (h // 2) * (w // 2), 4)
    ref = block_noise_from_normal(eps, bs, ch, t, h, w)
    assert (z - ref).abs().max() < 2e-3          # factors differ by the 4.2e-4 round-off pivot only


def test_injected_block_noise_fn_is_used():
    p = _pipe()
    p.block_noise_fn = lambda *s: torch.full(s, 3.0)
    assert torch.equal(p.sample_block_noise(1, 2, 1, 4, 4), torch.full((1, 2, 1, 4, 4), 3.0))


def test_default_draw_is_the_matrix_product_it_replaces():
    """round 6: the draw is written as numpy multiply-adds on strided views (two OpenMP regions per stage boundary were the
    host's longest pause); it must stay the product it replaces -- ONE randn(N, 4) from the global generator, times L^T, laid out
    `(b c t h w) (p q) -> b c t (h p) (w q)` -- to fp32 rounding"""
    bs, ch, t, h, w = 2, 16, 1, 24, 40
    torch.manual_seed(11)
    z = _pipe().sample_block_noise(bs, ch, t, h, w)
    torch.manual_seed(11)
    eps = torch.randn(bs * ch * t * (h // 2) * (w // 2), 4)
    L = block_noise_cholesky(1 / 3)
    ref = (eps.double() @ L.double().T).reshape(bs, ch, t, h // 2, w // 2, 2, 2).permute(0, 1, 2, 3, 5, 4, 6).reshape(bs, ch, t, h, w)
    assert z.is_contiguous() and (z.double() - ref).abs().max() < 1e-6
    # and the next draw continues the same global stream (nothing else consumed the generator)
    a = torch.randn(3)
    torch.manual_seed(11)
    torch.randn(bs * ch * t * (h // 2) * (w // 2), 4)
    assert torch.equal(a, torch.randn(3))
