"""bench.py on the GPU box: the N=1 line carries the contract's objects, and the N>1 path -- exercised with two ranks sharing the
one GPU over gloo (AR_BENCH_ONE_DEVICE_DEBUG=1) -- runs the REAL sharded pipeline (broadcast, pipelined fp-chain relay, per-rank
tuning, gather) and prints one whole-job line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--workload", "opt-125m", "--iters", "4", "--nsamples", "16", "--seqlen", "64", "--batch-size", "4", "--no-cpu-baseline"]


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError(text[-2000:])


def test_single_gpu_line_has_the_contract_fields_and_live_roofline():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-extras"] + SMALL, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and "workload" in d["config"]
    rf = d["roofline"]
    # the bench measures what ships (round 6): fuse_next_forward on -- K1 runs once per block, iteration i+1's Wq comes out of the fused
    # backward kernel, whose launches are the per-iteration ones
    assert d["config"]["fuse_next_forward"] is True
    assert rf["bound"] == "hbm" and rf["launches"] == 2 and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert d["roofline_bwd_sgd"]["launches"] == 2 * 4 and "next forward" in d["roofline_bwd_sgd"]["kernel"]
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-extras", "--no-fuse-next-forward", "--no-cpu-baseline"] + SMALL,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["config"]["fuse_next_forward"] is False and d["roofline"]["launches"] == 2 * 4          # the A/B form: K1 every iteration


def test_two_ranks_run_the_sharded_pipeline():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, AR_BENCH_ONE_DEVICE_DEBUG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"] + SMALL
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["value"] > 0
    assert "tune_sharded over 4 blocks" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 / 1000.0)) < 1e-6 * d["value"] + 1e-9       # whole-job blocks / max-over-ranks time
    # the record says who took part and what each rank did
    mg = d["multi_gpu"]
    assert mg["world_size"] == 2 and [r["rank"] for r in mg["ranks"]] == [0, 1] and all(r["device"] for r in mg["ranks"])
    assert mg["blocks_tuned_per_rank"] == [2, 2] and mg["calibration_broadcast"]["GBps"] > 0
    assert "roofline" in d and d["roofline"]["launches"] > 0          # rank 0's own K1 dispatches


TINY_LLAMA = ["--workload", "llama-tiny", "--iters", "4", "--nsamples", "16", "--seqlen", "128", "--batch-size", "4", "--no-cpu-baseline"]


def test_default_path_is_exact_rounding_under_the_calibration_mask_and_says_so():
    """bench.py's defaults: the bit-identical path (`exact_rounding`, proven against the module code before the timed region) under the
    reference's calibration mask; the line names the path, the mask and the proven plan under `config` (a key the driver keeps)."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-extras"] + TINY_LLAMA, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    c = _last_json(r.stdout)["config"]
    assert c["path"] == "exact" and c["exact_rounding"] is True and c["fused_block"] is False
    assert c["attention_mask"].startswith("calibration") and c["sdpa_backend"] == "auto"
    plan = c["exact_plan"]
    assert plan["norm1"] and plan["norm2"] and plan["rope"] and plan["swiglu"], plan


def test_two_ranks_shard_blocks_on_the_exact_path():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, AR_BENCH_ONE_DEVICE_DEBUG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"] + TINY_LLAMA
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["path"] == "exact" and d["config"]["exact_rounding"] is True
    assert d["multi_gpu"]["blocks_tuned_per_rank"] == [2, 2]
