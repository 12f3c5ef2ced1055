"""auto_round_amd -- MI355X (gfx950 / CDNA4) implementation of AutoRound's block-wise sign-gradient tuning hot path.

Hand-written HIP kernels behind a C ABI (include/ar_mi355x.h), with a Python host layer that mirrors the
reference's operator interface (WrapperLinear / wrapper_block / SignSGD / SignRoundQuantizer.quantize_block (+ SignRoundV2Quantizer, the algorithm extension) /
QuantLinear.pack).  There is no CPU fallback: importing is cheap, but every op requires the built HIP library.
"""
__version__ = "0.1.0"
