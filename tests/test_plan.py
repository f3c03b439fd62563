"""Host sequence plan: implicit mask == the reference mask (as restated by the oracle), RoPE table, counts."""
import numpy as np
import pytest
import torch

from oracle.flux_oracle import build_mask, rope_table, sequence_geometry


def _case(clip_shapes, Lt, valid):
    B = len(valid)
    mask = torch.zeros(B, Lt, dtype=torch.long)
    for b, v in enumerate(valid):
        mask[b, :v] = 1
    clips = [torch.zeros(B, 16, *s) for s in clip_shapes]
    return mask, clips


@pytest.mark.parametrize("clip_shapes,Lt,valid", [
    ([(1, 16, 32)], 16, (5, 12)),
    ([(2, 4, 8), (1, 8, 16), (1, 16, 32), (1, 16, 32)], 16, (5, 12)),
    ([(5, 6, 10), (1, 12, 20), (1, 24, 40), (1, 24, 40)], 128, (40, 96)),
    ([(1, 8, 8), (1, 8, 8)], 16, (16, 1)),
])
def test_mask_and_rope(clip_shapes, Lt, valid):
    from pyflow_hip.plan import SequencePlan
    mask, clips = _case(clip_shapes, Lt, valid)
    plan = SequencePlan(clip_shapes, mask, [16, 24, 24], "cpu")
    ids, frame_t = sequence_geometry(clips)
    ref_mask = build_mask(mask, frame_t)[:, 0].numpy()
    assert np.array_equal(plan.dense_mask(), ref_mask)
    assert plan.useful_pairs() == int(ref_mask.sum())
    # K5 closed form: v(v+n0) + p^2 + sum_k n_k (v + sum_{j<=k} n_j)
    counts = np.bincount(frame_t.numpy().astype(int))
    for b, v in enumerate(valid):
        p = Lt - v
        exp = v * (v + counts[0]) + p * p + sum(n * (v + counts[:k + 1].sum()) for k, n in enumerate(counts))
        assert ref_mask[b].sum() == exp
    all_ids = torch.cat([torch.zeros(Lt, 3), ids], 0)
    rt = rope_table(all_ids, [16, 24, 24])                 # [L,32,2,2] = [[c,-s],[s,c]]
    assert torch.equal(plan.rope[..., 0], rt[:, :, 0, 0]) and torch.equal(plan.rope[..., 1], rt[:, :, 1, 0])
    # tile ends cover every visible key and never exceed L
    te = plan.host["tile_kv_end"]
    assert te.max() <= plan.L and te.min() >= Lt


def test_interpolated_positions():
    from pyflow_hip.plan import image_token_ids
    ids = image_token_ids([(1, 4, 4), (1, 8, 8), (1, 16, 16)])
    assert ids[:4, 2].tolist() == [1.5, 5.5, 1.5, 5.5]                      # f = 4 -> 4k + 1.5 (2x2 grid)
    assert ids[4:8, 2].tolist() == [0.5, 2.5, 4.5, 6.5]                     # f = 2
    assert ids[:, 0].unique().tolist() == [0.0, 1.0, 2.0]


def test_non_prefix_mask_rejected():
    from pyflow_hip.plan import SequencePlan
    m = torch.tensor([[1, 0, 1, 0]])
    with pytest.raises(NotImplementedError):
        SequencePlan([(1, 4, 4)], m, [16, 24, 24], "cpu")


def test_checkpoint_key_remap_and_diffusers_dir(tmp_path):
    """raw-.pth key handling of load_checkpoint (pipeline.py:213-224) and the diffusers directory loader (:73,156)."""
    import json
    import torch
    from safetensors.torch import save_file
    from pyflow_hip.pipeline import PyramidDiTForVideoGeneration as P, _load_diffusers_dir
    ck = {"dit.proj_out.weight": torch.ones(2, 2), "proj_out.bias": torch.zeros(2), "vae.decoder.x": torch.ones(1),
          "text_encoder.y": torch.ones(1)}
    out = P.remap_dit_checkpoint(ck)
    assert set(out) == {"proj_out.weight", "proj_out.bias"}
    d = tmp_path / "diffusion_transformer_768p"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"num_layers": 8, "axes_dims_rope": [16, 24, 24]}))
    save_file({"proj_out.weight": torch.arange(4.0).reshape(2, 2)}, str(d / "diffusion_pytorch_model.safetensors"))
    sd, cfg = _load_diffusers_dir(str(d))
    assert cfg["num_layers"] == 8 and torch.equal(sd["proj_out.weight"], torch.arange(4.0).reshape(2, 2))
