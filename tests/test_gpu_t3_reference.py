"""T3: the real reference on the MI355X next to the product.  Runs only where the reference tree has
been staged (tools/stage_reference.sh -> oracle/_ref, builder-side gpurun); skipped in the driver's round-end run."""
import pytest

from ref_tree import reference_root

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(reference_root() is None, reason="reference tree not staged (tools/stage_reference.sh)")]


CHAOTIC = ("llama_w2g32_asym", "llama_mxfp4")


@pytest.mark.parametrize("name", ["llama_w4g32", "llama_w2g32_asym", "llama_mxfp4", "llama_w4a8", "opt_w4g32", "mixtral_w4g32",
                                  "llama_w2g32_alg_ext", "llama_w4g32_fp_chain"])
def test_reference_front_door_with_hip_engine_matches_reference_on_the_same_gpu(name):
    from t3_compare import run_case

    r = run_case(name, iters=20)
    assert r["same_layer_set"] and r["engine_calls"] == 2 and len(r["init_loss_ref"]) == 2, r
    # identical fake-quant weights at iteration 0 -> the same prediction -> the same loss (reduction order only) in the FIRST block
    first_rel = abs(r["init_loss_ref"][0] - r["init_loss_hip"][0]) / r["init_loss_ref"][0]
    assert first_rel < 1e-4, r
    chaotic = name in CHAOTIC
    # the symmetric INT schemes, W4A8, the algorithm extension, OPT and Mixtral come out bit-identical on this GPU; the asymmetric
    # W2 scheme (three-term fp16 scale gradient: exact zeros that one ulp turns into a full sign step) and MXFP4 with 4-bit
    # activations (power-of-two scales: a flipped exponent changes a whole group) are chaotic across engines, as they are
    # between the reference's own CPU and GPU runs
    assert r["frac_identical_weights"] >= (0.70 if chaotic else 0.97), r
    if "frac_identical_int_codes" in r:
        assert r["frac_identical_int_codes"] >= (0.93 if chaotic else 0.97), r
    if not chaotic:
        assert r["init_loss_max_rel_diff"] < 2e-3, r
        assert r["frac_identical_scale_zp_where_codes_agree"] is None or r["frac_identical_scale_zp_where_codes_agree"] >= 0.97, r


def test_two_rank_block_sharding_reproduces_the_reference_fp_chain_run():
    """(b): `enable_quanted_input=False` under 2 ranks (sharing the one GPU over gloo): calibration broadcast, pipelined fp-chain
    relay and per-block schedule replay give the weights the reference's sequential run gives on the same GPU."""
    from t3_compare import run_sharded_case

    r = run_sharded_case(iters=20)
    assert r["blocks"] == [0, 1] and r["owners"] == {"0": 0, "1": 1}, r
    assert r["frac_identical_weights"] >= 0.97, r
