"""Seeded random inputs shared by the live oracle-vs-reference test (CPU, build container) and the HIP-vs-oracle range test
(GPU): per-group magnitudes log-uniform over nine decades plus the structural corner cases."""
import numpy as np
import torch


def group_values(rng: np.random.Generator, G: int, gs: int, dtype: torch.dtype) -> torch.Tensor:
    mag = 10.0 ** rng.uniform(-7, 2, size=(G, 1))
    w = rng.standard_normal((G, gs)) * mag
    kinds = rng.integers(0, 10, size=G)
    w[kinds == 0] = 0.0                                           # all-zero groups
    w[kinds == 1] = np.abs(w[kinds == 1])                         # one-signed groups
    w[kinds == 2] = -np.abs(w[kinds == 2])
    tie = kinds == 3
    w[tie, 0] = -np.abs(w[tie]).max(axis=1)                       # |min| == |max|
    w[tie, 1] = np.abs(w[tie]).max(axis=1)
    return torch.from_numpy(w.astype(np.float32)).to(dtype)
