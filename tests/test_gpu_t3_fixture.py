"""End-to-end parity against the REAL reference at the BASELINE configuration, checkable without the reference tree (VERDICT r02
"next round" item 1): `tests/golden/t3_opt125m_w4g128_ref_on_mi355x.npz` holds what the reference's own front door
`auto_round.AutoRound(...).quantize()` (torch-eager SignRound quantizer on cuda:0 of an MI355X, `tests/t3_baseline_shapes.py`)
produced for ONE decoder block of OPT-125M's dimensions, W4 group_size=128 sym, 200 iterations, 128 x 2048 calibration tokens,
batch 8, seed 42 -- packed by the reference's own `QuantLinear.pack` -- plus its per-iteration loss trace and checksums of the
block's inputs and targets.  Here the same seeded block is re-tuned with this package (auto_round_amd.testing.t3_fixture) and the
packed `qweight / qzeros / scales` are compared word for word.

The thresholds are the measured builder-side results (profiles/r03_t3_baseline_shapes.json) with a small margin; the module path is
the bit-parity claim, the fused path (what bench.py measures) is held to the reference's own standard for its compiled path:
same loss level, a large majority of identical codes after 200 chaotic sign-SGD iterations."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fixture_meta():
    from auto_round_amd.testing import t3_fixture as fx

    assert os.path.exists(fx.FIXTURE), f"{fx.FIXTURE} missing: it is committed with the repository (tests/t3_baseline_shapes.py --fixture)"
    return fx.load_fixture()


def test_fixture_is_the_baseline_configuration(fixture_meta):
    m = fixture_meta["meta"]
    assert (m["arch"], m["scheme"], m["iters"], m["nsamples"], m["seqlen"], m["batch_size"], m["seed"]) == ("opt125m", "W4A16", 200, 128, 2048, 8, 42)
    assert len(fixture_meta["loss_trace"]) == 200 and m["device"] and "x_sha" in m and "y_sha" in m
    assert sorted(fixture_meta["layers"]) == ["fc1", "fc2", "self_attn.k_proj", "self_attn.out_proj", "self_attn.q_proj", "self_attn.v_proj"]
    q = fixture_meta["layers"]["fc1"]["qweight"]
    assert q.shape == (768 // 8, 3072) and fixture_meta["layers"]["fc1"]["scales"].shape == (768 // 128, 3072)


def test_module_path_reproduces_the_reference_on_gpu_fixture(fixture_meta):
    """Module path (`fused_block=False`): same torch block code as the reference around this package's kernels.  At THIS shape the
    library kernels under the block (head-size-64 attention backward, stream-K GEMMs) are not run-to-run deterministic -- the
    reference does not reproduce its own result either (`ref_vs_ref` in profiles/r03_t3_baseline_shapes.json), and two runs of this
    package part after ~65 iterations -- so the comparison is statistical here; the bit-exact one is the Llama-3-8B digest below."""
    from auto_round_amd.testing import t3_fixture as fx

    r = fx.check_against_fixture(fused=False)
    assert not r["fused_block"] and r["inputs_identical"], r          # same block inputs as the reference's quantizer saw
    assert abs(r["init_loss"] - r["init_loss_ref"]) <= 1e-5 * r["init_loss_ref"], r
    assert r["identical_codes"] >= MODULE_MIN_IDENTICAL_CODES, r
    assert 1 / BEST_LOSS_BAND <= r["best_loss_ratio"] <= BEST_LOSS_BAND, r


def test_fused_path_stays_on_the_reference_trajectory_level(fixture_meta):
    """Fused block path + MFMA weight-gradient GEMM + captured hipGraph iterations (forced: the automatic mode keeps a block of this
    size host-driven)."""
    from auto_round_amd.testing import t3_fixture as fx

    r = fx.check_against_fixture(fused=True, graph=True)
    assert r["fused_block"] and r["hip_graph"] and r["inputs_identical"], r
    assert abs(r["init_loss"] - r["init_loss_ref"]) <= 2e-3 * r["init_loss_ref"], r
    assert r["identical_codes"] >= FUSED_MIN_IDENTICAL_CODES, r
    assert 1 / BEST_LOSS_BAND <= r["best_loss_ratio"] <= BEST_LOSS_BAND, r


# measured on the builder's MI355X (profiles/r03_t3_baseline_shapes.json): module path 0.87-0.93 identical codes, fused path 0.86,
# best loss within 0.2 % of the reference's in every run
MODULE_MIN_IDENTICAL_CODES = 0.80
FUSED_MIN_IDENTICAL_CODES = 0.78
BEST_LOSS_BAND = 1.01


def test_llama8b_block_at_the_full_recipe_is_bit_identical_to_the_reference_digest():
    """BASELINE configs[1]'s block dimensions at the full recipe (W4G128 sym, 200 iterations, 128 x 2048, batch 8): on the module path
    every packed `qweight / qzeros / scales` tensor of the seven layers (218 M weights) hashes to what the REAL reference produced
    on an MI355X -- "quantized integer weights and packed buffers bit-exactly on the same seed / inputs" (north-star).  The fused
    kernels' group sums follow torch's own reduction order (csrc/ar_int.hip sum8_torch / lanes_sum_torch), which is what makes the
    scale gradients -- and with them 200 sign-SGD iterations -- reproduce bit for bit."""
    import torch

    from auto_round_amd.testing import t3_fixture as fx

    assert os.path.exists(fx.DIGEST), fx.DIGEST
    r = fx.check_against_digest()
    assert not r["fused_block"] and r["inputs_identical"] and r["targets_identical"], r
    same_stack = r["torch"] == torch.__version__ and r["device"] == torch.cuda.get_device_name(0)
    if same_stack:      # the library GEMM / attention kernels are the same binaries: nothing may differ
        assert r["bit_identical"], r
        assert r["first_divergence_iter"] is None and abs(r["best_loss_ratio"] - 1.0) < 1e-5, r
    else:               # another torch / GPU: another summation order inside the library kernels -> trajectory level
        assert r["full_layer_identical_codes"] > 0.7 and abs(r["best_loss_ratio"] - 1.0) < 0.02, r


def test_llama8b_block_on_the_fused_path_stays_on_the_reference_trajectory_level():
    """The configuration bench.py's headline `value` is measured on (fused block path + MFMA weight-gradient GEMM), at the full recipe,
    against the same reference-made digest: same loss level, a large majority of identical codes in the layer the digest holds in
    full (measured builder-side: 84 % identical codes, best loss x 0.9995 -- profiles/r03_t3_baseline_shapes.json).  The calibration
    flow's attention_mask is among the block's inputs here, so the attention runs through torch SDPA as in the module path."""
    from auto_round_amd.testing import t3_fixture as fx

    r = fx.check_against_digest(fused=True)
    assert r["fused_block"] and r["inputs_identical"] and r["targets_identical"], r
    assert abs(r["init_loss"] - r["init_loss_ref"]) <= 2e-3 * r["init_loss_ref"], r
    assert r["full_layer_identical_codes"] >= 0.6, r
    assert abs(r["best_loss_ratio"] - 1.0) < 0.02, r
