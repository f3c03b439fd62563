"""Guidance-parallel DiT forward for TWO ranks: each rank evaluates ONE branch of the classifier-free-guidance pair.

The reference evaluates `torch.cat([latents] * 2)` with `[negative | positive]` prompt rows in one forward
(pyramid_dit_for_video_gen_pipeline.py:747-776); its sequence parallelism for miniFLUX likewise ends, between the double- and
the single-stream blocks, in a scatter of that batch of 2 over the two ranks (modeling_pyramid_flux.py:471-485).  Here the
two branches never meet before the guidance combine: rank r runs the complete single-GPU engine (all rows, all 30 heads,
batch 1) on prompt row r, and the only exchange per step is the current frame's velocity tokens (n_cur x 64 fp32 ~ 1 MB,
summed into a replicated [2][n_cur][npad] buffer: disjoint slots + zeros, exact), after which both ranks perform the
identical CFG + Euler update -- no all-to-all at all, where the Ulysses exchange of two ranks would push half of every
K|V|Q and attention-output matrix through the ONE xGMI link between them (tools/rank_shape_bench.py: 7.5-10.6 s per video
exposed at P = 2).  Same arithmetic per sample as the batch-of-2 forward up to the GEMMs' fp32 summation order (the tile
walk depends on the row count).  Selected with init_sequence_parallel_group(..., guidance_parallel=True) on a world of 2.
"""
import torch

from .flux import FluxEngine
from .sp import pair_gather


class FluxEngineCFG(FluxEngine):
    def __init__(self, state_dict, cfg, device="cuda", comm=None):
        super().__init__(state_dict, cfg, device)
        assert comm is not None and comm.world == 2, "guidance parallelism splits the CFG pair over exactly two ranks"
        self.comm = comm
        self._row_cache = {}          # id(tensor) -> (tensor kept alive, version, my row): stable objects for the caches below

    def _my_row(self, t):
        """row `rank` of a [2, ...] tensor as ONE object per source tensor (FluxEngine.conditioning keys its cache on the
        identity of `pooled`; a fresh slice per step would defeat it)"""
        ent = self._row_cache.get(id(t))
        if ent is None or ent[0] is not t or ent[1] != t._version:
            if len(self._row_cache) > 16:
                self._row_cache.clear()
            r = self.comm.rank
            ent = (t, t._version, t[r:r + 1].contiguous())
            self._row_cache[id(t)] = ent
        return ent[2]

    def make_plan(self, clip_shapes, enc_mask, cfg_pair=False):
        # the caller SAYS that the two rows are one sample's guidance pair (the pipeline does, under classifier-free
        # guidance); a genuine batch of 2 handed to the engine-level API takes the replicated base path on every rank
        if not cfg_pair:
            return super().make_plan(clip_shapes, enc_mask)
        assert enc_mask.shape[0] == 2, "a guidance pair has exactly two prompt rows"
        plan = super().make_plan(clip_shapes, enc_mask[self.comm.rank:self.comm.rank + 1])
        plan.cfg_pair = True
        return plan

    def encode_context(self, enc, cfg_pair=False):
        if not cfg_pair:
            return super().encode_context(enc)
        assert enc.shape[0] == 2, "a guidance pair has exactly two prompt rows"
        return super().encode_context(enc[self.comm.rank:self.comm.rank + 1])

    def forward_tokens(self, plan, clips, timesteps, pooled, ctx=None, shared_clips=False, debug=None):
        if not getattr(plan, "cfg_pair", False):          # no guidance pair (guidance scale 1): every rank computes the same
            return super().forward_tokens(plan, clips, timesteps, pooled, ctx, shared_clips, debug)
        r = self.comm.rank
        assert len(timesteps) == 2 and shared_clips, "guidance parallelism: the CFG duplicate of one latent (pipeline.py:747)"
        v = super().forward_tokens(plan, clips, timesteps[r:r + 1], self._my_row(pooled), ctx, True, debug)   # [1, n_cur, npad]
        n = v.shape[1] * v.shape[2]
        out = self._buf("vtok_pair", 2 * n, torch.float32)[:2 * n]
        return pair_gather(self.comm, v, out).view(2, v.shape[1], v.shape[2])
