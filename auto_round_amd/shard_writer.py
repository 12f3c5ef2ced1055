"""Streaming of finished (packed) blocks to safetensors shards, so that a 70B / Mixtral run never holds more than one
tuned block in host memory.  Counterpart of the reference's ShardWriter (auto_round/compressors/shard_writer.py:37) for
the tensors this path produces; tensor names follow the reference's "auto_round" checkpoint layout
(`<layer>.qweight / .qzeros / .scales [/ .bias]` for INT, `<layer>.weight_packed / .weight_scale /
.weight_global_scale` for MXFP4/NVFP4; export_to_autoround/export.py:242-330)."""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

_INT_KEYS = ("qweight", "qzeros", "scales", "g_idx", "bias")
_FP_KEYS = ("weight_packed", "weight_scale", "weight_global_scale", "input_global_scale", "bias")


def packed_state(prefix: str, module: torch.nn.Module) -> Dict[str, torch.Tensor]:
    """Flatten one packed QuantLinear into `{name: cpu tensor}` with the reference's buffer names."""
    out = {}
    for k in _INT_KEYS + _FP_KEYS:
        t = getattr(module, k, None)
        if isinstance(t, torch.Tensor) and f"{prefix}.{k}" not in out:
            out[f"{prefix}.{k}"] = t.detach().to("cpu").contiguous()
    return out


class ShardWriter:
    """reference: ShardWriter (auto_round/compressors/shard_writer.py:37): size-bounded safetensors shards written while the
    run is still tuning later blocks, `model.safetensors.index.json` on close."""

    def __init__(self, out_dir: str, max_shard_bytes: int = 5 * 1024 ** 3, metadata: Optional[dict] = None, tag: Optional[str] = None):
        """tag: a per-writer infix of the shard file names ("rank3" -> model-rank3-00001-of-00002.safetensors) for runs in which
        several processes stream into ONE checkpoint directory (block sharding: every rank writes the blocks it tuned, rank 0
        merges the weight maps into the one index -- `finish()` / `write_index()`)."""
        self.out_dir = out_dir
        self.tag = f"{tag}-" if tag else ""
        self.max_shard_bytes = int(max_shard_bytes)
        self.metadata = {"format": "pt", **(metadata or {})}
        self._pending: Dict[str, torch.Tensor] = {}
        self._pending_bytes = 0
        self._shards = []
        self._weight_map: Dict[str, str] = {}
        self._total = 0
        os.makedirs(out_dir, exist_ok=True)

    def write(self, tensors: Dict[str, torch.Tensor]) -> None:
        for name, t in tensors.items():
            if name in self._weight_map or name in self._pending:
                raise KeyError(f"tensor {name} written twice")
            nbytes = t.numel() * t.element_size()
            if self._pending and self._pending_bytes + nbytes > self.max_shard_bytes:
                self._flush()
            self._pending[name] = t
            self._pending_bytes += nbytes

    def write_block(self, block_prefix: str, packed: Dict[str, torch.nn.Module]) -> None:
        """`packed` = export.pack_block(block): {layer name within the block: packed module}."""
        for lname, mod in packed.items():
            lname = lname[:-len(".orig_layer")] if lname.endswith(".orig_layer") else lname   # act-quant wrappers
            self.write(packed_state(f"{block_prefix}.{lname}" if block_prefix else lname, mod))

    def _flush(self) -> None:
        if not self._pending:
            return
        from safetensors.torch import save_file

        fname = f"model-{self.tag}{len(self._shards) + 1:05d}.safetensors"
        save_file(self._pending, os.path.join(self.out_dir, fname), metadata={k: str(v) for k, v in self.metadata.items()})
        for name in self._pending:
            self._weight_map[name] = fname
        self._total += self._pending_bytes
        self._shards.append(fname)
        self._pending, self._pending_bytes = {}, 0

    def finish(self):
        """Flush and rename this writer's shards to their `-of-` form WITHOUT writing an index -> ({tensor: file}, total bytes):
        what a rank of a block-sharded run hands to rank 0."""
        self._flush()
        n = len(self._shards)
        renamed = {}
        for i, old in enumerate(self._shards):
            new = "model.safetensors" if (n == 1 and not self.tag) else f"model-{self.tag}{i + 1:05d}-of-{n:05d}.safetensors"
            os.replace(os.path.join(self.out_dir, old), os.path.join(self.out_dir, new))
            renamed[old] = new
        self._shards = []
        return {k: renamed[v] for k, v in self._weight_map.items()}, self._total

    @staticmethod
    def write_index(out_dir: str, parts) -> str:
        """`parts`: [(weight_map, total_bytes), ...] of every writer that streamed into `out_dir` -> the one index file."""
        weight_map, total = {}, 0
        for wm, t in parts:
            dup = set(weight_map) & set(wm)
            if dup:
                raise KeyError(f"tensors written by more than one writer: {sorted(dup)[:4]}")
            weight_map.update(wm)
            total += int(t)
        path = os.path.join(out_dir, "model.safetensors.index.json")
        with open(path, "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)
        return path

    def close(self) -> str:
        """Flush, rename shards to `-of-` form and write `model.safetensors.index.json`.  Returns the index path."""
        return self.write_index(self.out_dir, [self.finish()])
