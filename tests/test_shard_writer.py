"""ShardWriter: streamed safetensors shards + index (CPU only)."""
import json
import os

import pytest
import torch

from auto_round_amd.shard_writer import ShardWriter, packed_state


class _Packed(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (16, 64), dtype=torch.int32, generator=g)
        self.qzeros = torch.full((1, 8), 0x77777777, dtype=torch.int32)
        self.scales = torch.rand(1, 64, generator=g).half()
        self.bias = None


def test_shards_and_index_roundtrip(tmp_path):
    from safetensors import safe_open

    w = ShardWriter(str(tmp_path), max_shard_bytes=6000, metadata={"quant_method": "auto-round"})
    blocks = []
    for b in range(3):
        packed = {"self_attn.q_proj": _Packed(10 * b), "mlp.down_proj.orig_layer": _Packed(10 * b + 1)}
        blocks.append(packed)
        w.write_block(f"model.layers.{b}", packed)
    idx_path = w.close()
    idx = json.load(open(idx_path))
    names = set(idx["weight_map"])
    assert "model.layers.1.self_attn.q_proj.qweight" in names and "model.layers.2.mlp.down_proj.scales" in names
    assert len(set(idx["weight_map"].values())) > 1          # several shards
    assert all(f.endswith(".safetensors") and "-of-" in f for f in idx["weight_map"].values())
    for b, packed in enumerate(blocks):
        st = packed_state(f"model.layers.{b}.self_attn.q_proj", packed["self_attn.q_proj"])
        for name, t in st.items():
            with safe_open(os.path.join(tmp_path, idx["weight_map"][name]), framework="pt") as f:
                assert torch.equal(f.get_tensor(name), t)
    assert idx["metadata"]["total_size"] == sum(t.numel() * t.element_size() for p in blocks for m in p.values()
                                               for t in (m.qweight, m.qzeros, m.scales))


def test_several_writers_stream_into_one_checkpoint_and_one_index(tmp_path):
    """Block-sharded runs (round 6): every rank writes the blocks it tuned under its own tag, rank 0 merges the weight maps."""
    import json

    from safetensors import safe_open

    parts = []
    for r in range(2):
        w = ShardWriter(str(tmp_path), max_shard_bytes=3000, tag=f"rank{r}")
        w.write({f"model.layers.{r}.w{i}": torch.full((256,), float(10 * r + i)) for i in range(5)})       # 1 KB each: 3 shards
        parts.append(w.finish())
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".safetensors"))
    assert files == ["model-rank0-00001-of-00003.safetensors", "model-rank0-00002-of-00003.safetensors", "model-rank0-00003-of-00003.safetensors",
                     "model-rank1-00001-of-00003.safetensors", "model-rank1-00002-of-00003.safetensors", "model-rank1-00003-of-00003.safetensors"]
    path = ShardWriter.write_index(str(tmp_path), parts)
    index = json.load(open(path))
    assert index["metadata"]["total_size"] == 10 * 1024 and len(index["weight_map"]) == 10
    for name, fname in index["weight_map"].items():
        with safe_open(os.path.join(tmp_path, fname), "pt") as f:
            assert float(f.get_tensor(name)[0]) == 10 * int(name.split(".")[2]) + int(name[-1])
    with pytest.raises(KeyError):                       # the same tensor from two writers is a bug, not a silent overwrite
        ShardWriter.write_index(str(tmp_path), [parts[0], parts[0]])


def test_front_door_device_map_and_launcher_environment(monkeypatch):
    """`device_map` values of the reference's front door and how a rank picks its device (auto_round_amd/autoround.py)."""
    from auto_round_amd import autoround as a

    d = lambda i: torch.device("cuda", i)  # noqa: E731
    assert a.parse_device_map("0,1,2") == [d(0), d(1), d(2)] and a.parse_device_map([0, "1"]) == [d(0), d(1)]
    assert a.parse_device_map(3) == [d(3)] and a.parse_device_map("cuda:2") == [d(2)] and a.parse_device_map("auto") == [d(0)]
    assert a.parse_device_map("cpu") == [torch.device("cpu")]                  # refused later, loudly, by the constructor
    assert a.pick_rank_device([d(0), d(1), d(2)], 1, 3, 8) == d(1)             # "0,1,2": rank r takes the r-th entry
    assert a.pick_rank_device([d(0)], 3, 8, 8) == d(3)                         # default map under a launcher: cuda:LOCAL_RANK
    assert a.pick_rank_device([d(0)], 1, 2, 1) == d(0)                         # one visible device: the ranks share it
    assert a.pick_rank_device([d(0), d(0)], 1, 2, 1) == d(0)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert a.dist_env() == (0, 1, 0)
    monkeypatch.setenv("RANK", "5"), monkeypatch.setenv("WORLD_SIZE", "8"), monkeypatch.setenv("LOCAL_RANK", "5")
    assert a.dist_env() == (5, 8, 5)
