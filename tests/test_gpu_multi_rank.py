"""The front door as one process per GPU: `AutoRound(..., device_map="0,1,...", enable_quanted_input=False)` under a
launcher shards the blocks over the ranks (auto_round_amd/sharding.py), every rank packs and writes the blocks it tuned into its own
shard files, rank 0 merges the index -- and the checkpoint must equal the single-process run's tensor for tensor.  The GPU box has ONE
device, so both ranks run on cuda:0: RCCL ("nccl") is tried first and, where it refuses two ranks on one device, the same code runs
over gloo with device tensors (which backend ran is recorded).  With quantised-input chaining (the reference's default) blocks are
sequential and every block is tuned data-parallel instead; that run is held to the sequential one at trajectory level (another
summation order of the weight gradients)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(n, out_dir, device_map, quanted, backend, scheme="W4A16"):
    env = dict(os.environ, AR_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    env.pop("RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multi_rank_front_door.py"), out_dir, device_map,
           "1" if quanted else "0", scheme]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


def _two_ranks(out_dir, quanted, scheme="W4A16"):
    """2 ranks on cuda:0 -- RCCL first; gloo where RCCL refuses duplicate devices"""
    tried = []
    for backend in ("nccl", "gloo"):
        r = _launch(2, out_dir, "0,0", quanted, backend, scheme)
        tried.append((backend, r.returncode, (r.stderr or "")[-1500:]))
        if r.returncode == 0:
            return backend, tried
        for f in os.listdir(out_dir) if os.path.isdir(out_dir) else []:
            os.remove(os.path.join(out_dir, f))
    raise AssertionError(f"2-rank front-door run failed on every backend: {tried}")


def _tensors(folder):
    from safetensors import safe_open

    index = json.load(open(os.path.join(folder, "model.safetensors.index.json")))
    out = {}
    for fname in sorted(set(index["weight_map"].values())):
        with safe_open(os.path.join(folder, fname), "pt") as f:
            for k in f.keys():
                assert index["weight_map"][k] == fname
                out[k] = f.get_tensor(k)
    assert sorted(out) == sorted(index["weight_map"])
    return out, index


@pytest.mark.parametrize("scheme", ["W4A16", "MXFP4"])
def test_block_sharded_front_door_equals_the_single_process_run(tmp_path, scheme, record_property):
    seq, shd = str(tmp_path / "seq"), str(tmp_path / "sharded")
    r = _launch(1, seq, "0", False, "nccl", scheme)                      # the same script as ONE process: the sequential run
    assert r.returncode == 0, r.stderr[-2000:]
    backend, tried = _two_ranks(shd, False, scheme)
    record_property("backend", backend)
    why = [ln.strip()[:200] for b, rc, err in tried if rc for ln in err.splitlines() if ("NCCL" in ln or "RCCL" in ln or "uplicate" in ln)][-2:]
    print(f"\n[multi-rank] block-sharded front door ran over {backend} (tried: {[(b, rc) for b, rc, _ in tried]}; refused with: {why})")
    record_property("nccl_refusal", "; ".join(why))
    run = json.load(open(os.path.join(shd, "run.json")))
    assert [m["rank"] for m in run] == [0, 1] and all(m["world"] == 2 and m["sharded"] and not m["data_parallel"] for m in run)
    assert run[0]["owned_blocks"] == [0, 2] and run[1]["owned_blocks"] == [1, 3] and run[0]["backend"] == backend
    assert run[0]["tuned_weights_checksum"] == run[1]["tuned_weights_checksum"]          # every rank ends with the whole tuned model
    one = json.load(open(os.path.join(seq, "run.json")))[0]
    assert one["tuned_weights_checksum"] == run[0]["tuned_weights_checksum"] and one["owned_blocks"] == [0, 1, 2, 3]
    t_seq, _ = _tensors(seq)
    t_shd, index = _tensors(shd)
    files = set(index["weight_map"].values())
    assert any("rank0" in f for f in files) and any("rank1" in f for f in files), files     # each rank wrote its own shard
    assert index["weight_map"]["model.layers.1.mlp.down_proj." + ("qweight" if scheme == "W4A16" else "weight_packed")].startswith("model-rank1-")
    assert sorted(t_seq) == sorted(t_shd)
    for k, a in t_seq.items():
        b = t_shd[k]
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.view(torch.uint8) if a.dtype != torch.bool else a,
                                                                         b.view(torch.uint8) if b.dtype != torch.bool else b), k
    assert json.load(open(os.path.join(seq, "config.json"))) == json.load(open(os.path.join(shd, "config.json")))
    # losses per block are the sequential run's, bit for bit
    losses = {**run[0]["losses"], **run[1]["losses"]}
    assert losses == one["losses"], (losses, one["losses"])


def test_quantised_input_chaining_runs_data_parallel_under_a_launcher(tmp_path, record_property):
    seq, dp = str(tmp_path / "seq"), str(tmp_path / "dp")
    r = _launch(1, seq, "0", True, "nccl")
    assert r.returncode == 0, r.stderr[-2000:]
    backend, tried = _two_ranks(dp, True)
    record_property("backend", backend)
    run = json.load(open(os.path.join(dp, "run.json")))
    assert all(m["data_parallel"] and not m["sharded"] and m["owned_blocks"] == [0, 1, 2, 3] for m in run)
    assert run[0]["tuned_weights_checksum"] == run[1]["tuned_weights_checksum"]          # all ranks took identical sign steps
    one = json.load(open(os.path.join(seq, "run.json")))[0]
    for k, v in one["losses"].items():      # same function, another summation order of the gradients: trajectory level
        assert abs(run[0]["losses"][k] - v) <= 0.25 * abs(v), (k, run[0]["losses"][k], v)      # (tiny model, 12 iterations, losses ~1e-7)
    t_seq, _ = _tensors(seq)
    t_dp, index = _tensors(dp)
    assert sorted(t_seq) == sorted(t_dp) and all(t_seq[k].shape == t_dp[k].shape and t_seq[k].dtype == t_dp[k].dtype for k in t_seq)
    assert not any("rank" in f for f in set(index["weight_map"].values()))              # rank 0 alone wrote the checkpoint
