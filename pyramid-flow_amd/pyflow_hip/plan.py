"""Per-(unit, stage) sequence plan: token geometry, 3-axis RoPE table and the implicit form of the
reference's block-causal attention mask.  Pure host arithmetic (numpy / torch-CPU), computed once
per (unit, stage) and uploaded; replaces the per-forward `_prepare_pyramid_image_ids`, `EmbedND`
and `[B,1,L,L]` mask construction of modeling_pyramid_flux.py:186-237, 266-270, 318-350.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _axis_pos(n, train_n):
    # modeling_pyramid_flux.py:193-204: linear interpolation of arange(train_n) onto n samples
    if n == train_n:
        return torch.arange(train_n).float()
    return F.interpolate(torch.arange(train_n)[None, None].float(), n, mode="linear")[0, 0]


def image_token_ids(clip_shapes):
    """clip_shapes: list of (t, h, w) LATENT dims, oldest -> newest. Returns ids [L_img,3] float32."""
    th, tw = clip_shapes[-1][1] // 2, clip_shapes[-1][2] // 2
    out, start = [], 0
    for (t, h, w) in clip_shapes:
        h2, w2 = h // 2, w // 2
        ids = torch.zeros(t, h2, w2, 3)
        ids[..., 0] += torch.arange(start, start + t)[:, None, None]
        ids[..., 1] += _axis_pos(h2, th)[None, :, None]
        ids[..., 2] += _axis_pos(w2, tw)[None, None, :]
        out.append(ids.reshape(-1, 3))
        start += t
    return torch.cat(out, 0)


def rope_cos_sin(ids, axes_dim, theta=10000):
    """modeling_pyramid_flux.py:28-57 -> [L, sum(axes)/2, 2] float32 (cos, sin), angles in float64."""
    parts = []
    for i, dim in enumerate(axes_dim):
        scale = torch.arange(0, dim, 2, dtype=torch.float64) / dim
        omega = 1.0 / (theta ** scale)
        ang = ids[:, i].double()[:, None] * omega[None]
        parts.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1))
    return torch.cat(parts, dim=1).float()


class SequencePlan:
    """Geometry of one stage sequence [text(Lt) | clips...] for a CFG batch of B rows."""

    QTILE = 128

    def __init__(self, clip_shapes, enc_mask, axes_dim, device):
        enc_mask = np.asarray(enc_mask.cpu() if isinstance(enc_mask, torch.Tensor) else enc_mask)
        B, Lt = enc_mask.shape
        self.B, self.Lt = B, Lt
        self.clip_shapes = list(clip_shapes)
        self.clip_tokens = [t * (h // 2) * (w // 2) for (t, h, w) in clip_shapes]
        self.L_img = sum(self.clip_tokens)
        self.L = Lt + self.L_img
        self.Lp = (self.L + 63) // 64 * 64
        t, h, w = clip_shapes[-1]
        self.cur = (t, h, w)
        self.n_cur = self.clip_tokens[-1]
        ids = image_token_ids(clip_shapes)
        all_ids = torch.cat([torch.zeros(Lt, 3), ids], 0)
        self.rope = rope_cos_sin(all_ids, axes_dim).contiguous().to(device)
        # ---- implicit mask: valid text must be a prefix (tokenizer pads at the end)
        frame_t = ids[:, 0].numpy().astype(np.int64)
        n_frames = int(frame_t.max()) + 1 if frame_t.size else 0
        counts = np.bincount(frame_t, minlength=n_frames)
        frame_end = Lt + np.cumsum(counts)                       # exclusive end position of frame f
        a_lo = np.zeros((B, self.L), np.int32)
        a_hi = np.zeros((B, self.L), np.int32)
        b_hi = np.zeros((B, self.L), np.int32)
        for b in range(B):
            v = int(enc_mask[b].sum())
            if not (enc_mask[b, :v] == 1).all():
                raise NotImplementedError("encoder_attention_mask must be a prefix of ones (tokenizer right-padding)")
            # valid text rows: valid text + frame(s) with t == 0 ; padded text rows: padded text only
            a_lo[b, :v], a_hi[b, :v], b_hi[b, :v] = 0, v, (frame_end[0] if n_frames else Lt)
            a_lo[b, v:Lt], a_hi[b, v:Lt], b_hi[b, v:Lt] = v, Lt, Lt
            a_lo[b, Lt:], a_hi[b, Lt:] = 0, v
            b_hi[b, Lt:] = frame_end[frame_t]
        nqt = (self.L + self.QTILE - 1) // self.QTILE
        tile_end = np.zeros((B, nqt), np.int32)
        for qt in range(nqt):
            sl = slice(qt * self.QTILE, min((qt + 1) * self.QTILE, self.L))
            tile_end[:, qt] = np.maximum(b_hi[:, sl].max(axis=1), Lt)
        self.a_lo = torch.from_numpy(a_lo).to(device)
        self.a_hi = torch.from_numpy(a_hi).to(device)
        self.b_hi = torch.from_numpy(b_hi).to(device)
        self.tile_kv_end = torch.from_numpy(tile_end).to(device)
        self.host = dict(a_lo=a_lo, a_hi=a_hi, b_hi=b_hi, tile_kv_end=tile_end)

    def dense_mask(self):
        """[B, L, L] bool -- test helper: the mask the intervals stand for."""
        j = np.arange(self.L)[None, None, :]
        a_lo, a_hi, b_hi = (self.host[k][:, :, None] for k in ("a_lo", "a_hi", "b_hi"))
        return np.where(j < self.Lt, (j >= a_lo) & (j < a_hi), j < b_hi)

    def useful_pairs(self, q_row_begin=0):
        """number of unmasked (q, k) pairs summed over the batch for query rows >= q_row_begin (FLOP accounting)."""
        cache = self.__dict__.setdefault("_pairs", {})
        if q_row_begin not in cache:
            h = self.host
            r = q_row_begin
            cache[q_row_begin] = int((h["a_hi"][:, r:] - h["a_lo"][:, r:]).astype(np.int64).sum()
                                     + (h["b_hi"][:, r:] - self.Lt).astype(np.int64).sum())
        return cache[q_row_begin]
