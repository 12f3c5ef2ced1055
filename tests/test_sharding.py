"""Multi-GPU block sharding covered on CPU with world_size-2 `gloo` process groups (tier brief (5)): assignment,
sequential-equivalent index-schedule replay, calibration broadcast, point-to-point fp-chain relay, result gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from auto_round_amd import sharding as sh

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_assign_blocks_policies():
    assert sh.assign_blocks(5, 2) == [[0, 2, 4], [1, 3]]
    assert sh.assign_blocks(5, 2, "contiguous") == [[0, 1, 2], [3, 4]]
    assert sh.assign_blocks(32, 8)[3] == [3, 11, 19, 27]
    for n, w, pol in ((7, 3, "round_robin"), (7, 3, "contiguous"), (80, 8, "round_robin")):
        owned = sh.assign_blocks(n, w, pol)
        assert sorted(sum(owned, [])) == list(range(n))
        for r, blocks in enumerate(owned):
            for b in blocks:
                assert sh.owner_of(b, n, w, pol) == r


def test_replayed_schedules_equal_the_sequential_reference_stream():
    """Block 0 and block 1 of a seed-42 run, as the reference's IndexSampler drew them (tests/golden/sampler.npz)."""
    z = np.load(os.path.join(GOLDEN, "sampler.npz"))
    import random

    import auto_round_amd.quantizer  # noqa: F401  (first import of transformers consumes global random numbers)

    random.seed(123)
    before = random.random()
    random.seed(123)
    sched = sh.replay_index_schedules(42, 2, 128, 8, 200)
    assert random.random() == before, "replay must not disturb the caller's random state"
    assert np.array_equal(np.array(sched[0]), z["n128_b8"])
    assert np.array_equal(np.array(sched[1]), z["n128_b8_block2"])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeQuantizer:
    """Stands in for SignRoundQuantizer in the CPU test: same surface tune_sharded uses."""

    class _Cfg:
        batch_size, iters, gradient_accumulate_steps, enable_quanted_input = 2, 6, 1, False

    def __init__(self):
        self.config = self._Cfg()
        self.last_stats = {}

    def forward_all(self, block, x, others):
        with torch.no_grad():
            return torch.cat([block(x[i:i + 2]) for i in range(0, x.shape[0], 2)])

    calibrated = None

    def calibrate_block(self, block, x, others):          # the relay's forward IS the block's calibration forward
        self.calibrated = (self.calibrated or 0) + 1
        return self.forward_all(block, x, others)

    def quantize_block(self, block, xin, others, yout, q_inputs, ctx, input_ids=None, index_schedule=None):
        assert input_ids == "ids"                         # the loss mask reaches the sharded blocks
        self.last_stats = {"sched_sum": int(np.array(index_schedule).sum()), "in_sum": float(xin.sum()), "out_sum": float(yout.sum()),
                           "calibrated": self.calibrated}
        return {"dummy": torch.tensor(float(len(index_schedule)))}


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_blocks, N, H = 5, 8, 16
        torch.manual_seed(0)   # identical weights on every rank (a real run loads only the owned blocks)
        blocks = [torch.nn.Sequential(torch.nn.Linear(H, H), torch.nn.Tanh()) for _ in range(n_blocks)]
        x0 = torch.zeros(N, 4, H)
        if rank == 0:
            x0.copy_(torch.randn(N, 4, H, generator=torch.Generator().manual_seed(7)))
        q = _FakeQuantizer()
        mine = [b if sh.owner_of(k, n_blocks, world) == rank else None for k, b in enumerate(blocks)]
        res = sh.tune_sharded(mine, x0, {}, q, seed=42, input_ids="ids")
        # sequential ground truth
        xs = [torch.randn(N, 4, H, generator=torch.Generator().manual_seed(7))]
        for b in blocks:
            xs.append(q.forward_all(b, xs[-1], {}))
        sched = sh.replay_index_schedules(42, n_blocks, N, 2, 6)
        assert sorted(res) == sh.assign_blocks(n_blocks, world)[rank]
        for k, r in res.items():
            assert abs(r["stats"]["in_sum"] - float(xs[k].sum())) < 1e-4, (rank, k)
            assert abs(r["stats"]["out_sum"] - float(xs[k + 1].sum())) < 1e-4, (rank, k)
            assert r["stats"]["sched_sum"] == int(np.array(sched[k]).sum())
            # every owned block went through calibrate_block before it was tuned (pipelined: one by one, in block order)
            assert r["stats"]["calibrated"] == sorted(res).index(k) + 1, (rank, k, r["stats"])
        # the two-phase form (whole sweep first) gives the same pairs and schedules
        q2 = _FakeQuantizer()
        res2 = sh.tune_sharded(mine, x0.clone() if rank == 0 else torch.zeros_like(x0), {}, q2, seed=42, input_ids="ids", pipelined=False)
        for k, r in res.items():
            assert res2[k]["stats"]["in_sum"] == r["stats"]["in_sum"] and res2[k]["stats"]["sched_sum"] == r["stats"]["sched_sum"]
            assert res2[k]["stats"]["calibrated"] == len(res)
        merged = sh.gather_results({k: v["stats"] for k, v in res.items()}, dst=0)
        if rank == 0:
            assert sorted(merged) == list(range(n_blocks))
            open(os.path.join(tmp, "ok"), "w").write("1")
        else:
            assert merged is None
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_chain_relay_and_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


@pytest.mark.skipif(not os.path.isdir("/root/reference/auto_round"), reason="reference tree not present (GPU box)")
def test_plugin_registers_with_the_reference_registry():
    import sys

    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, "/root/reference"):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round.algorithms.registry import resolve_algorithm_alias, resolve_pipeline_member

    import auto_round_amd.plugin as plugin

    Cfg, Q = plugin.register()
    assert plugin.register() == (Cfg, Q)          # idempotent
    assert resolve_algorithm_alias("mi355x") == "mi355x_signround"
    cfg = Cfg(iters=7)
    assert resolve_pipeline_member(cfg) is Q
    from auto_round.algorithms.registry import normalize_algorithm_config

    assert type(normalize_algorithm_config(cfg)) is Cfg  # not coerced to the V2/Adam variants
    from auto_round_amd.export import QuantLinearPlain, QuantLinearZP

    assert plugin.packing_quant_linear("auto_round:auto_gptq", 4, 128, True) is QuantLinearZP
    assert plugin.packing_quant_linear("auto_round", 4, 128, False) is QuantLinearPlain


def _dp_sync_worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        class Lyr:
            def __init__(self, grad, ran):
                self.weight_grad, self._dw_accum = grad, [ran]

        class Arena:
            pass

        a = Arena()
        a.dWq = torch.arange(12, dtype=torch.float32).mul(rank + 1).to(torch.bfloat16)   # rank r holds (r+1) * [0..11]
        # layer 0 ran on both ranks; layer 1 (an idle MoE expert here) only on rank 1: rank 0's stale slice must count as 0
        a.layers = [Lyr(a.dWq[:8], True), Lyr(a.dWq[8:], rank == 1)]
        loss = torch.tensor([float(rank + 1)])
        sh.sync_block_gradients([a], loss)
        loss_sum = torch.tensor([float(rank + 1)])
        sh.sync_block_gradients([], loss_sum, average_loss=False)
        q.put((rank, a.dWq.float().tolist(), float(loss), float(loss_sum), [l._dw_accum[0] for l in a.layers], sh.dp_world()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradient_sync_world2_gloo():
    """sync_block_gradients: one SUM all-reduce of the arena-wide dWq buffer (bf16; fp32 staging under gloo), idle layers
    contribute zeros, mean losses are averaged and sum losses summed -- identical results on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [3.0 * i for i in range(8)] + [2.0 * i for i in range(8, 12)]       # layer 1: only rank 1's (x2) values
    for rank, g, loss, loss_sum, ran, (r, w) in res:
        assert g == expect and loss == 1.5 and loss_sum == 3.0 and ran == [True, True] and (r, w) == (rank, 2)
    assert sh.dp_world() == (0, 1)


def _front_door_worker(rank, world, port, tmp):
    """model_tuner.tune_blocks_sharded + ShardWriter(tag=...) + autoround.sync_tuned_blocks over gloo on CPU tensors: what the front
    door's sharded mode is made of (the HIP engine and packers stood in for: they need a GPU)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json

        from auto_round_amd import autoround as ar
        from auto_round_amd import model_tuner as mt
        from auto_round_amd.shard_writer import ShardWriter

        n_blocks, N, H = 5, 8, 16
        torch.manual_seed(0)
        blocks = [torch.nn.Sequential(torch.nn.Linear(H, H), torch.nn.Tanh()) for _ in range(n_blocks)]
        for b in blocks:
            b[0].bits = 4
        samples = [torch.randn(1, 4, H, generator=torch.Generator().manual_seed(7 + i)) for i in range(N)]      # un-stacked, as a cache hands them over

        class Q(_FakeQuantizer):
            device = torch.device("cpu")

            def quantize_block(self, block, xin, others, yout, q_inputs, ctx, input_ids=None, index_schedule=None):
                best = super().quantize_block(block, xin, others, yout, q_inputs, ctx, input_ids=input_ids, index_schedule=index_schedule)
                with torch.no_grad():                       # "tuning": something only the owner knows afterwards
                    block[0].weight.mul_(0.5)
                    block[0].scale = torch.full((H, 1), float(self.last_stats["sched_sum"]))
                    block[0].zp = 8
                return best

        class _Packed(torch.nn.Module):
            def __init__(self, lin):
                super().__init__()
                self.qweight, self.scales, self.qzeros = lin.weight.detach().clone(), lin.scale.clone(), torch.zeros(1)

        monkey = mt.pack_block
        mt.pack_block = lambda block, backend=None: {"0": _Packed(block[0])}
        try:
            w = ShardWriter(tmp, tag=f"rank{rank}")
            names = [f"model.layers.{k}" for k in range(n_blocks)]
            recs = mt.tune_blocks_sharded(blocks, samples if rank == 0 else [torch.zeros_like(s) for s in samples], {}, Q(), seed=42,
                                          input_ids="ids", block_names=names, shard_writer=w)
        finally:
            mt.pack_block = monkey
        assert sorted(recs) == sh.assign_blocks(n_blocks, world)[rank]
        assert all(r["name"] == names[k] and "packed" in r and r["best_params"] for k, r in recs.items())
        part = w.finish()
        parts = [None] * world
        dist.all_gather_object(parts, part)
        got = ar.sync_tuned_blocks(blocks, world, torch.device("cpu"))
        assert got == n_blocks - len(recs)                  # one tuned layer per block received from its owner
        # every rank now holds every block's tuned weight and scale
        torch.manual_seed(0)
        fresh = [torch.nn.Sequential(torch.nn.Linear(H, H), torch.nn.Tanh()) for _ in range(n_blocks)]
        for k in range(n_blocks):
            assert torch.equal(blocks[k][0].weight, fresh[k][0].weight * 0.5), (rank, k)
            assert blocks[k][0].zp == 8 and blocks[k][0].scale.shape == (H, 1)
        if rank == 0:
            index = json.load(open(ShardWriter.write_index(tmp, parts)))
            wm = index["weight_map"]
            assert len(wm) == 3 * n_blocks
            for k in range(n_blocks):
                assert wm[f"model.layers.{k}.0.qweight"].startswith(f"model-rank{k % world}-")
            open(os.path.join(tmp, "ok2"), "w").write("1")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_sharded_front_door_pieces(tmp_path):
    port = _free_port()
    mp.spawn(_front_door_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok2").exists()
