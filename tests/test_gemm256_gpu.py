"""GPU parity of the 256 x BN ping-pong GEMM kernel (gemm256.hip) against torch fp32 and against the 128 x 128
kernel, forced through pf_gemm_set_policy so that small shapes reach it too."""
import pytest
import torch
import torch.nn.functional as F

from util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.fixture
def policy():
    from pyflow_hip import ops
    yield ops.gemm_set_policy
    ops.gemm_set_policy(0)


@pytest.mark.parametrize("bn,M,N,K", [
    (128, 256, 128, 64), (128, 700, 384, 192), (192, 256, 192, 64), (192, 1000, 384, 1920), (192, 77, 576, 128),
    (256, 513, 512, 256), (256, 3000, 256, 640), (192, 2048 + 5, 1920, 320),
    (256, 700, 5760, 128), (256, 300, 328, 192), (256, 1024, 13440, 64),          # N tail of the 256-wide tile
    (192, 600, 384, 7680), (256, 520, 512, 9600),       # long reductions (ff2 / proj_out of the single blocks)
    (192, 5120 + 7, 2688, 192), (256, 4100, 4352, 128), (128, 4100, 2304, 64),   # > 256 tiles: several tiles per persistent workgroup
])
def test_gemm256_bias(policy, bn, M, N, K):
    from pyflow_hip import ops
    policy(bn)
    A = _mk((M, K), 1).to(torch.bfloat16).to(DEV)
    W = _mk((N, K), 2, 0.05).to(torch.bfloat16).to(DEV)
    W[0, :] += 1.0          # asymmetric: a transposed C write cannot pass
    bias = _mk((N,), 3).to(DEV)
    C = torch.zeros(M + 3, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(A, W, C, M, N, K, K, K, N, bias=bias)
    ref = A.float() @ W.float().T + bias
    assert rel_l2(C[:M].float(), ref) < 5e-3
    assert (C[:M].float() - ref).abs().max() <= 2 ** -6 * ref.abs().max()
    assert C[M:].abs().max() == 0          # rows beyond M untouched


@pytest.mark.parametrize("bn", [128, 192, 256])
def test_gemm256_identity_layout(policy, bn):
    from pyflow_hip import ops
    policy(bn)
    M, N, K = 512, 2 * bn, 512
    A = torch.eye(M, dtype=torch.bfloat16, device=DEV)
    W = (torch.arange(N)[:, None] * 0.5 + torch.arange(K)[None, :] * 0.001).to(torch.bfloat16).to(DEV)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(A, W, C, M, N, K, K, K, N)
    assert torch.equal(C.float(), W.float().T.contiguous())


@pytest.mark.parametrize("bn,d", [(192, 384), (256, 256), (128, 128)])
def test_gemm256_batched_strided_gelu_gate(policy, bn, d):
    from pyflow_hip import ops
    policy(bn)
    B, Lr = 2, 600
    L = Lr + 16
    x = _mk((B, L, d), 4).to(torch.bfloat16).to(DEV)
    W = _mk((2 * d, d), 5, 0.06).to(torch.bfloat16).to(DEV)
    bias = _mk((2 * d,), 6, 0.1).to(DEV)
    out = torch.zeros(B, L, 2 * d, dtype=torch.bfloat16, device=DEV)
    ops.gemm(x, W, out, Lr, 2 * d, d, d, d, 2 * d, bias=bias, batch=B, strideA=L * d, strideC=L * 2 * d,
             gelu_from=d, a_off=16 * d, c_off=16 * 2 * d)
    ref = x[:, 16:].float() @ W.float().T + bias
    ref[..., d:] = F.gelu(ref[..., d:], approximate="tanh")
    assert rel_l2(out[:, 16:].float(), ref) < 5e-3
    assert out[:, :16].abs().max() == 0
    hid = _mk((B, L, d), 7).to(torch.bfloat16).to(DEV)
    hid0 = hid.clone()
    W2 = _mk((d, 2 * d), 8, 0.05).to(torch.bfloat16).to(DEV)
    b2 = _mk((d,), 9, 0.1).to(DEV)
    gate = _mk((B, 3 * d), 10).to(DEV)
    ops.gemm(out, W2, hid, Lr, d, 2 * d, 2 * d, 2 * d, d, bias=b2, res=hid, gate=gate, gate_off=d, ldr=d, batch=B,
             strideA=L * 2 * d, strideC=L * d, strideR=L * d, gate_stride=3 * d, flags=ops.GEMM_GATE_RES,
             a_off=16 * 2 * d, c_off=16 * d, r_off=16 * d)
    ref2 = hid0[:, 16:].float() + gate[:, None, d:2 * d] * (out[:, 16:].float() @ W2.float().T + b2)
    assert rel_l2(hid[:, 16:].float(), ref2) < 5e-3
    assert torch.equal(hid[:, :16], hid0[:, :16])


def test_gemm256_out_f32_and_long_k(policy):
    from pyflow_hip import ops
    policy(192)
    M, N, K = 300, 192, 64 * 37            # odd tile count exercises every ring phase
    A = _mk((M, K), 11).to(torch.bfloat16).to(DEV)
    W = _mk((N, K), 12, 0.05).to(torch.bfloat16).to(DEV)
    C = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A, W, C, M, N, K, K, K, N, flags=ops.GEMM_OUT_F32)
    assert rel_l2(C, A.float() @ W.float().T) < 1e-5


@pytest.mark.parametrize("nk", [1, 2, 3, 4, 5])
def test_gemm256_short_k_pipeline(policy, nk):
    """prologue / drain paths of the LDS-DMA ring (K = 64 .. 320)."""
    from pyflow_hip import ops
    policy(256)
    M, N, K = 1024, 256, 64 * nk
    A = _mk((M, K), 20 + nk).to(torch.bfloat16).to(DEV)
    W = _mk((N, K), 30 + nk, 0.1).to(torch.bfloat16).to(DEV)
    C = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A, W, C, M, N, K, K, K, N, flags=ops.GEMM_OUT_F32)
    assert rel_l2(C, A.float() @ W.float().T) < 1e-5


def test_gemm256_matches_128_kernel_bitwise_class(policy):
    """both kernels accumulate K in the same 64-wide order per MFMA chain: results agree to fp32 rounding."""
    from pyflow_hip import ops
    M, N, K = 4096, 768, 1920
    A = _mk((M, K), 41).to(torch.bfloat16).to(DEV)
    W = _mk((N, K), 42, 0.03).to(torch.bfloat16).to(DEV)
    outs = []
    for pol in (-1, 192, 256, 128):
        policy(pol)
        C = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        ops.gemm(A, W, C, M, N, K, K, K, N, flags=ops.GEMM_OUT_F32)
        outs.append(C)
    for o in outs[1:]:
        assert rel_l2(o, outs[0]) < 1e-6


def test_vae_decode_with_forced_256_kernel(policy):
    """implicit-GEMM conv path (pixel-shuffle / depth-to-time store maps, residual add) through gemm256."""
    from pyflow_hip import synth
    from pyflow_hip.vae import CausalVideoVAE
    from oracle.vae_oracle import vae_decode
    from util import round_sd
    policy(128)
    cfg = synth.TINY_VAE
    sd = round_sd(synth.random_state_dict(synth.vae_decoder_param_shapes(cfg), seed=5, std=0.05, lively=True))
    z = torch.randn(1, 16, 3, 6, 10, generator=torch.Generator().manual_seed(2))
    ocfg = dict(decoder_block_out_channels=cfg["block_out_channels"], decoder_layers_per_block=cfg["layers_per_block"],
                decoder_spatial_up_sample=cfg["spatial_up_sample"], decoder_temporal_up_sample=cfg["temporal_up_sample"])
    ref = vae_decode(sd, ocfg, z)
    vae = CausalVideoVAE(sd, cfg, "cuda")
    out = vae.decode(z.cuda(), temporal_chunk=True, window_size=1).sample.float().cpu()
    assert rel_l2(out, ref) < 3e-2
