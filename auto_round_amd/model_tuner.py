"""The caller of the hot path, for standalone use: the per-block driver loop of the reference's orchestrator
(`CompressionOrchestrator._quantize_blocks`, auto_round/compressors/orchestrator.py:176-388) reduced to what the path
needs -- walk the decoder blocks in order, tune each one, hand its outputs to the next:

    for block k:  fp_out, q_out, best = quantizer.compress_block(block_k, fp_in, input_others, q_in)
                  [pack immediately]                                      (orchestrator.py:327-337, immediate_pack)
                  fp_in <- fp_out ;  q_in <- q_out   (enable_quanted_input, composer.py:460,476-481)

Model loading, calibration-data capture and checkpoint writing stay with the reference.
`tune_blocks_sharded` is the multi-GPU form (blocks independent on the fp chain, see sharding.py)."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch

from .export import pack_block
from .quantizer import BlockContext, SignRoundQuantizer, stack_samples


def tune_blocks(blocks: Sequence[torch.nn.Module], block0_inputs, input_others: dict, quantizer: SignRoundQuantizer,
                input_ids=None, pack: bool = False, block_names: Optional[List[str]] = None,
                on_block_done: Optional[Callable] = None, shard_writer=None) -> List[Dict]:
    """Sequentially tune `blocks` (already on the quantizer's device, or movable to it).  Returns one record per block:
    {"name", "stats", "best_params", "packed" (if pack)}.  Weights are baked in place like the reference does."""
    device = quantizer.device
    fp_in = stack_samples(block0_inputs, device)
    q_in = None
    out = []
    n = len(blocks)
    for k, block in enumerate(blocks):
        block.to(device)
        ctx = BlockContext(block_index=k, block_cnt=n, block_name=(block_names[k] if block_names else str(k)))
        fp_out, q_out, best = quantizer.compress_block(block, fp_in, input_others, q_in, ctx, input_ids=input_ids)
        rec = {"name": ctx.block_name, "stats": dict(quantizer.last_stats), "best_params": best}
        if pack or shard_writer is not None:
            rec["packed"] = pack_block(block)
            if shard_writer is not None:      # immediate saving: stream the finished block out (orchestrator.py:340-353)
                shard_writer.write_block(ctx.block_name, rec["packed"])
        out.append(rec)
        if on_block_done is not None:
            on_block_done(k, block, rec)
        fp_in = fp_out
        q_in = q_out if quantizer.config.enable_quanted_input else None
    return out


def tune_blocks_sharded(blocks, block0_inputs, input_others, quantizer, seed: int = 42, policy: str = "round_robin",
                        group=None, input_ids=None, block_names: Optional[List[str]] = None, pack: bool = False,
                        on_block_done: Optional[Callable] = None, shard_writer=None) -> Dict[int, Dict]:
    """Multi-GPU form of `tune_blocks`: requires enable_quanted_input=False (blocks are only independent on the fp chain).  Every
    owned block goes through the same pre-tuning calibration and loss mask as `tune_blocks` (sharding.tune_sharded); the blocks
    this rank does not own are never touched.  `block0_inputs` as for `tune_blocks` (a list of per-sample tensors or one [N, S, H]
    tensor; every rank passes the same shape, rank 0's values are the ones used).  -> {block index: the record `tune_blocks` makes
    for that block} for the blocks THIS rank tuned; with `shard_writer` every finished block is packed and streamed out at once."""
    from . import sharding

    if quantizer.config.enable_quanted_input:
        raise ValueError("block sharding needs enable_quanted_input=False: with quantised-input chaining block k+1 "
                         "depends on the tuned block k (SURVEY 8e) -- run replicas, or data_parallel=True inside each block")
    device = quantizer.device
    x0 = stack_samples(block0_inputs, device)
    owned = sharding.assign_blocks(len(blocks), sharding.dp_world(group)[1], policy)[sharding.dp_world(group)[0]]
    for k in owned:
        blocks[k].to(device)
    out: Dict[int, Dict] = {}

    def done(k, block, rec):
        name = block_names[k] if block_names else str(k)
        r = {"name": name, "stats": rec["stats"], "best_params": rec["best_params"]}
        if pack or shard_writer is not None:
            r["packed"] = pack_block(block)
            if shard_writer is not None:
                shard_writer.write_block(name, r["packed"])
        out[k] = r
        if on_block_done is not None:
            on_block_done(k, block, r)

    sharding.tune_sharded(blocks, x0, input_others, quantizer, seed=seed, policy=policy, group=group, input_ids=input_ids,
                          on_block_done=done)
    return out
