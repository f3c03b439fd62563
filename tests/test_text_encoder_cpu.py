"""Host logic of the prompt encoders against transformers (the third-party code the reference's wrappers call)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "helpers"))


@pytest.mark.parametrize("L,nb,md", [(128, 32, 128), (77, 32, 128), (256, 32, 128), (200, 16, 64)])
def test_relative_position_buckets_match_transformers(L, nb, md):
    from transformers.models.t5.modeling_t5 import T5Attention
    from pyflow_hip.text_encoder import t5_relative_position_buckets
    ctx = torch.arange(L)[:, None]
    mem = torch.arange(L)[None, :]
    ref = T5Attention._relative_position_bucket(mem - ctx, bidirectional=True, num_buckets=nb, max_distance=md)
    assert torch.equal(t5_relative_position_buckets(L, nb, md), ref)


def test_stub_tokenizer_shapes():
    from hf_text import StubTokenizer
    t = StubTokenizer(512)
    o = t(["a cat", "two dogs running"], max_length=16)
    assert o.input_ids.shape == (2, 16) and o.attention_mask.sum(1).tolist() == [6, 17 if 17 < 16 else 16]
    assert (o.input_ids.argmax(-1) == o.attention_mask.sum(1) - 1).all()
