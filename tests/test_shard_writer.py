"""ShardWriter: streamed safetensors shards + index (CPU only)."""
import json
import os

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
