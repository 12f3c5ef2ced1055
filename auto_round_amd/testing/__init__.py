"""Synthetic model blocks for tests and benchmarks (no checkpoints exist offline)."""
