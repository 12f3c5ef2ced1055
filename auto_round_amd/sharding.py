"""Block-level sharding of the tuning path across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).  The reference has no block-level parallelism
(SURVEY 2.3); this is the MI355X-first strategy of SURVEY 8e.

Independent units: with `enable_quanted_input=False` block k is tuned on the fp activation chain only
(fp_in[k] -> fp_out[k] = fp_in[k+1]), so blocks are independent once their (input, target) pair exists.

Data path (no per-iteration collective anywhere):
  1. `broadcast_calibration`   one RCCL broadcast of the block-0 input [nsamples, seq, hidden] from the root
     (2.15 GB for Llama-3-8B; xGMI is point-to-point, the root drives its 7 links concurrently).
  2. `relay_fp_chain`          owner(k) computes fp_out[k] with the SAME in-loop forward path it later tunes with
     (pre-cached vs in-loop forwards are not numerically identical, reference utils/resume.py:14-23) and sends it
     point-to-point to owner(k+1); each owner keeps (fp_in, fp_out) of its blocks in HBM.
  3. every rank tunes its own blocks with the index schedule it would have had in a sequential run
     (`replay_index_schedules`: the Python `random` stream is replayed per block on every rank).
  4. `gather_results`          packed buffers / stats are gathered on the root (or each rank writes its own shard).
"""
from __future__ import annotations

import random
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def assign_blocks(n_blocks: int, world_size: int, policy: str = "round_robin") -> List[List[int]]:
    """Block indices owned by each rank.  round_robin keeps the fp-chain relay flowing rank r -> r+1."""
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    if policy == "round_robin":
        return [list(range(r, n_blocks, world_size)) for r in range(world_size)]
    if policy == "contiguous":
        per, extra = divmod(n_blocks, world_size)
        out, start = [], 0
        for r in range(world_size):
            cnt = per + (1 if r < extra else 0)
            out.append(list(range(start, start + cnt)))
            start += cnt
        return out
    raise ValueError(f"unknown policy {policy}")


def owner_of(block: int, n_blocks: int, world_size: int, policy: str = "round_robin") -> int:
    if policy == "round_robin":
        return block % world_size
    for r, blocks in enumerate(assign_blocks(n_blocks, world_size, policy)):
        if block in blocks:
            return r
    raise IndexError(block)


def replay_index_schedules(seed: int, n_blocks: int, nsamples: int, batch_size: int, iters: int,
                           gradient_accumulate_steps: int = 1) -> List[List[List[int]]]:
    """The minibatch index schedule of EVERY block exactly as a sequential reference run draws it: the reference seeds
    Python's global `random` once (transformers.set_seed -> random.seed, compressors/base.py:360) and IndexSampler is
    its only consumer (compressors/utils.py:420-438), one sampler per block in block order.  Every rank replays the
    whole stream (cheap: n_blocks * (1 + iters*bs/nsamples) shuffles) and keeps the rows of its own blocks."""
    from .quantizer import IndexSampler

    state = random.getstate()
    try:
        random.seed(seed)
        gbs = min(nsamples, batch_size * gradient_accumulate_steps)
        out = []
        for _ in range(n_blocks):
            s = IndexSampler(nsamples, gbs)
            out.append([s.next_batch() for _ in range(iters)])
        return out
    finally:
        random.setstate(state)


def broadcast_calibration(x: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """In-place broadcast of the shared calibration activations (allocated with the same shape on every rank)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(x, src=src, group=group)
    return x


def relay_fp_chain(blocks: Sequence[Optional[torch.nn.Module]], x0: torch.Tensor, forward_all: Callable,
                   policy: str = "round_robin", group=None) -> Dict[int, tuple]:
    """Run the fp chain once across ranks.  `blocks[k]` is the module on its owner and may be None elsewhere.
    `forward_all(block, x) -> y` must be the tuning-time forward (same batching/autocast).  Returns
    {k: (fp_in[k], fp_out[k])} for the blocks this rank owns."""
    n = len(blocks)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine: Dict[int, tuple] = {}
    cur = x0 if owner_of(0, n, world, policy) == rank else None
    for k in range(n):
        own = owner_of(k, n, world, policy)
        nxt = owner_of(k + 1, n, world, policy) if k + 1 < n else None
        if own == rank:
            y = forward_all(blocks[k], cur)
            mine[k] = (cur, y)
            if nxt is not None and nxt != rank:
                dist.send(y.contiguous(), dst=nxt, group=group)
                cur = None
            else:
                cur = y
        elif nxt == rank:
            # the next block is mine: receive its input from the current owner
            buf = torch.empty_like(x0)
            dist.recv(buf, src=own, group=group)
            cur = buf
    return mine


def tune_sharded(blocks: Sequence[Optional[torch.nn.Module]], x0: torch.Tensor, input_others: dict, quantizer,
                 seed: int = 42, policy: str = "round_robin", group=None, input_ids=None, pipelined: bool = True,
                 on_block_done: Optional[Callable] = None) -> Dict[int, dict]:
    """Shard `blocks` over the ranks and tune the local ones against the fp chain.  Every rank must pass the same
    x0 buffer shape; rank `src=0` holds the data.  Returns {block index: stats/best_params} for local blocks.

    The relay forward of a block IS its calibration forward (`quantizer.calibrate_block`: NVFP4 global-scale unification,
    act_max hooks + fill-in for idle experts, imatrix hooks of the algorithm extension), and the loss mask (`input_ids`)
    reaches `quantize_block`, so a sharded block is prepared and tuned exactly like the sequential `compress_block` does
    with `enable_quanted_input=False`.

    pipelined (default): every rank walks ITS blocks in order -- receive the block's input from the previous owner, run the
    calibration forward, hand the output to the next owner with an asynchronous send, tune -- so that the fp sweep of the
    blocks downstream runs on the other GPUs while this one tunes.  With round-robin ownership rank r's next input arrives
    (N-1 forwards later) while it is still tuning, and its receive was posted before the tuning started, so neither the sweep
    nor the transfers sit on the critical path: after the pipeline has filled (rank r idles r forwards once), a round of N
    blocks costs one forward + one tuning run per GPU.  pipelined=False keeps the two-phase form (whole sweep first, then all
    tuning), whose sweep is a serial prefix of n_blocks forwards.  Both give bit-identical blocks (same inputs, same schedule)."""
    n = len(blocks)
    cfg = quantizer.config
    if cfg.enable_quanted_input:
        raise ValueError("block sharding tunes every block against the fp activation chain: set enable_quanted_input=False "
                         "(with quantised-input chaining blocks are sequential; use data_parallel=True instead)")
    broadcast_calibration(x0, 0, group)
    scheds = replay_index_schedules(seed, n, x0.shape[0], cfg.batch_size, cfg.iters, cfg.gradient_accumulate_steps)
    out = {}

    def tune(k, xin, yout):
        best = quantizer.quantize_block(blocks[k], xin, input_others, yout, None, None, input_ids=input_ids,
                                        index_schedule=scheds[k])
        out[k] = {"best_params": best, "stats": dict(quantizer.last_stats)}
        if on_block_done is not None:
            on_block_done(k, blocks[k], out[k])

    if not pipelined:
        pairs = relay_fp_chain(blocks, x0, lambda b, x: quantizer.calibrate_block(b, x, input_others), policy, group)
        for k, (xin, yout) in pairs.items():
            tune(k, xin, yout)
        return out

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = [k for k in range(n) if owner_of(k, n, world, policy) == rank]
    pending_recv: Dict[int, tuple] = {}
    sends = []

    def post_recv(k):
        """Post the receive of block k's input (if it comes from another rank) ahead of time."""
        if k is None or k == 0 or k in pending_recv:
            return
        src = owner_of(k - 1, n, world, policy)
        if src != rank:
            buf = torch.empty_like(x0)
            pending_recv[k] = (buf, dist.irecv(buf, src=src, group=group))

    carried = None                                  # output of my previous block when I also own the next one
    if mine:
        post_recv(mine[0])
    for i, k in enumerate(mine):
        if k == 0:
            xin = x0
        elif k in pending_recv:
            buf, work = pending_recv.pop(k)
            work.wait()
            xin = buf
        else:
            xin = carried
        yout = quantizer.calibrate_block(blocks[k], xin, input_others)
        carried = None
        if k + 1 < n:
            nxt = owner_of(k + 1, n, world, policy)
            if nxt != rank:
                sends.append((yout, dist.isend(yout.contiguous(), dst=nxt, group=group)))
            else:
                carried = yout
        post_recv(mine[i + 1] if i + 1 < len(mine) else None)      # arrives while this block is being tuned
        tune(k, xin, yout)
        sends = [(t, w) for t, w in sends if not w.is_completed()]
    for _, w in sends:
        w.wait()
    return out


def gather_results(local: dict, dst: int = 0, group=None):
    """Gather per-block python results (stats, CPU tensors) on `dst`; returns the merged dict there, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dict(local)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bucket = [None] * world if rank == dst else None
    dist.gather_object(local, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    return merged


# ----------------------------------------------------------------------------------------------------------------------
# data parallelism INSIDE one block (for the reference's default quantised-input chaining, where blocks are sequential)
# ----------------------------------------------------------------------------------------------------------------------
def dp_world(group=None):
    """(rank, world) of the data-parallel group, (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def all_reduce_sum_(t: torch.Tensor, group=None, async_op: bool = False):
    """In-place SUM all-reduce over RCCL (backend "nccl" on ROCm).  The weight-gradient buffers are bf16; gloo (CPU-side
    test backend) has no bf16 reduction, so there the sum runs on an fp32 copy."""
    if dist.get_backend(group) == "gloo" and t.dtype in (torch.bfloat16, torch.float16):
        f = t.to(torch.float32)
        dist.all_reduce(f, op=dist.ReduceOp.SUM, group=group)
        t.copy_(f)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def enable_overlapped_sync(block, arenas, group=None):
    """Bucketed gradient all-reduce overlapped with the backward pass: every wrapped layer starts the SUM all-reduce of its own
    slice of the arena's dWq buffer as soon as its weight-gradient GEMM has been issued (async on RCCL's stream), and
    `sync_block_gradients` only waits for them.  Collectives must be issued in the same order on every rank, so this is only
    enabled for blocks whose layers all run in every iteration in a data-independent order: it is refused (returns False) for
    blocks with expert sub-modules, where participation depends on the routing of each rank's tokens."""
    if any(".experts." in n or n.endswith(".experts") for n, _ in block.named_modules()):
        return False
    pending = []
    for a in arenas:
        a._dp_pending = pending
        for lyr in a.layers:
            flat = a.dWq[lyr._off:lyr._off + lyr.numel]

            def start(flat=flat):
                w = all_reduce_sum_(flat, group, async_op=True)
                if w is not None:
                    pending.append(w)

            lyr._post_dw = start
            lyr._dp_overlapped = True
    return True


def sync_block_gradients(arenas, total_loss: torch.Tensor, group=None, average_loss: bool = True):
    """What the reference's DDP / `_all_reduce_model_grads` does per iteration (utils/distributed.py:30-140), on this
    layout: ONE bucket per arena -- the block-wide dWq buffer (bf16, 2 B/weight instead of the reference's 4 B/weight of
    fp32 parameter gradients; dV, d min_scale and d max_scale are linear in dWq, so reducing dWq before the fused
    backward + sign-SGD kernel is equivalent to reducing them) -- plus the scalar loss.  Idle layers are zeroed first so
    every rank contributes a defined value."""
    _, world = dp_world(group)
    if world == 1:
        return
    for a in arenas:
        if getattr(a, "_dp_pending", None) is not None:      # per-layer buckets were started during the backward pass
            for w in a._dp_pending:
                w.wait()
            a._dp_pending.clear()
            continue
        for lyr in a.layers:
            if not lyr._dw_accum[0]:
                lyr.weight_grad.zero_()
                lyr._dw_accum[0] = True
        all_reduce_sum_(a.dWq, group)
    dist.all_reduce(total_loss, op=dist.ReduceOp.SUM, group=group)
    if average_loss:        # "mean" losses: the global mean is the mean of the equally sized local means
        total_loss.div_(world)
