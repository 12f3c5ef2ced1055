"""The driver's BENCH record keeps only `config`, `roofline` and `cpu_baseline` of bench.py's line: everything a
reader needs -- the bit-identical path's own rate, the other paths, the north-star's OPT-125M configuration with its K1 / K2
fractions, the live parity verdicts, the quoted real-reference CPU figure -- must be repeated under those three keys.  CPU only:
`nest_for_the_driver` is plain dict plumbing."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _line(path="exact"):
    rf = lambda f: {"achieved": f * 8000, "peak": 8000.0, "frac": f, "avg_launch_ms": 0.3, "algorithmic_bytes_per_launch": 1, "kernel": "k", "traffic": 1, "launches": 2}  # noqa: E731
    var = {n: {"value": v, "ms_per_step": 1000 / v, "ms_per_iter": 5 / v} for n, v in
           (("module_path_calibration_mask", 0.16), ("fused_path_calibration_mask", 0.2), ("fused_path_no_mask", 0.27), ("exact_path_calibration_mask", 0.185))}
    var.pop({"exact": "exact_path_calibration_mask", "fused": "fused_path_calibration_mask", "module": "module_path_calibration_mask"}[path])
    return {"value": 0.185, "ms_per_step": 5400.0, "ms_per_iter": 27.0, "config": {"workload": "w"}, "roofline": rf(0.7), "roofline_bwd_sgd": rf(0.72),
            "cpu_baseline": {"value": 3e-4, "unit": "blocks/s", "cores": 128, "kind": "port", "sample": "s"},
            "variants": var,
            "parity": {"llama8b_module_path_bit_identical": True, "llama8b_exact_path_bit_identical": True, "module_path_identical_codes": 0.9,
                       "fused_path_identical_codes": 0.86, "best_loss_ratio": 0.999, "llama8b_exact_path": {"tensors_identical": 21}},
            "opt125m": {"value": 3.1, "ms_per_step": 320.0, "ms_per_iter": 1.6, "hip_graph": False, "fused_block": True, "roofline": rf(0.58),
                        "roofline_bwd_sgd": rf(0.42), "speedup_vs_cpu_reference_quoted": 530.0,
                        "cpu_baseline": {"value": 3e-3, "unit": "blocks/s", "cores": 128, "kind": "port", "sec_per_iter_at_batch": 1.5},
                        "cpu_reference_quoted": {"value": 0.0058, "kind": "reference", "cores": 8}}}


def test_the_three_kept_keys_carry_the_whole_claim():
    b = _bench()
    out = _line("exact")
    b.nest_for_the_driver(out, "exact", "calibration")
    c, rf, cb = out["config"], out["roofline"], out["cpu_baseline"]
    bi = c["bit_identical_path"]
    assert bi["blocks_per_s"] == out["value"] and bi["ms_per_step"] == out["ms_per_step"] and bi["digest_bit_identical"] is True
    assert bi["digest_tensors_identical"] == 21 and bi["module_path"]["blocks_per_s"] == 0.16
    tl = c["trajectory_level_paths"]
    assert tl["fused_path_calibration_mask"]["blocks_per_s"] == 0.2 and tl["fused_path_no_mask"]["blocks_per_s"] == 0.27
    assert c["parity"]["llama8b_exact_path_bit_identical"] is True and c["parity"]["fused_path_identical_codes"] == 0.86
    assert c["opt125m"]["blocks_per_s"] == 3.1 and c["opt125m"]["ms_per_iter"] == 1.6
    assert rf["opt125m"]["k1"]["frac"] == 0.58 and rf["opt125m"]["k2"]["frac"] == 0.42 and rf["opt125m"]["ms_per_iter"] == 1.6
    assert cb["opt125m"]["port"]["value"] == 3e-3 and cb["opt125m"]["reference_quoted"]["kind"] == "reference"


def test_the_headline_path_is_reported_under_its_own_name_whichever_it_is():
    b = _bench()
    out = _line("fused")
    b.nest_for_the_driver(out, "fused", "calibration")
    c = out["config"]
    assert c["trajectory_level_paths"]["fused_path_calibration_mask"]["blocks_per_s"] == out["value"]       # the headline itself
    assert c["bit_identical_path"]["blocks_per_s"] == 0.185                                                 # from the variants
    out = _line("module")
    b.nest_for_the_driver(out, "module", "calibration")
    assert out["config"]["bit_identical_path"]["module_path"]["blocks_per_s"] == out["value"]


def test_calibration_mask_is_the_reference_flows_zero_one_bias():
    import torch

    b = _bench()
    m = b.calibration_mask(8, "cpu")
    assert m.shape == (1, 1, 8, 8) and m.dtype == torch.bfloat16
    keep = torch.tril(torch.ones(8, 8, dtype=torch.bool))
    keep[:, -1] = False
    assert torch.equal(m[0, 0].bool(), keep) and set(m.unique().tolist()) == {0.0, 1.0}


def test_flat_scalar_keys_survive_a_record_that_keeps_scalars_only():
    """the driver's `parsed` drops nested objects.  Keep only the scalars of the three kept keys, as it does, and
    the claim must still be readable: which mask, bit identity and the digest count, every path's rate, OPT-125M's rate and roofline
    fractions, K2's fraction, the quoted reference figures, the same-GPU reference time."""
    b = _bench()
    out = _line("exact")
    out["config"].update(attention_mask="calibration: the [1, 1, S, S] 0/1 additive mask ...", exact_plan={"norm1": True, "dw_gu": -1, "dw_q": 0, "rope": True})
    out["cpu_reference_quoted"] = {"value": 0.00038, "cores": 8, "sec_per_iter": 13.1}
    b.nest_for_the_driver(out, "exact", "calibration")
    scalars = lambda d: {k: v for k, v in d.items() if isinstance(v, (int, float, bool, str)) or v is None}  # noqa: E731
    c, rf, cb = scalars(out["config"]), scalars(out["roofline"]), scalars(out["cpu_baseline"])
    assert c["attention_mask"] == "calibration" and c["bit_identical"] is True and c["digest_tensors_identical"] == 21
    assert c["exact_blocks_per_s"] == out["value"] and c["module_path_blocks_per_s"] == 0.16
    assert c["fused_mask_blocks_per_s"] == 0.2 and c["fused_nomask_blocks_per_s"] == 0.27
    assert c["opt125m_blocks_per_s"] == 3.1 and c["opt125m_ms_per_iter"] == 1.6
    assert c["exact_plan_flat"] == "dw_gu=-1,norm1,rope"
    assert c["reference_same_gpu_s_per_block"] > 10 and c["speedup_vs_reference_same_gpu"] > 1        # (quoted from profiles/r03_t3_baseline_shapes.json)
    assert rf["bwd_sgd_frac"] == 0.72 and rf["opt125m_k1_frac"] == 0.58 and rf["opt125m_k2_frac"] == 0.42
    assert cb["reference_quoted_value"] == 0.00038 and cb["reference_quoted_cores"] == 8 and cb["opt125m_reference_quoted_value"] == 0.0058
    # a trajectory-level headline says so
    out = _line("fused")
    b.nest_for_the_driver(out, "fused", "none")
    assert out["config"]["bit_identical"] is False and out["config"]["fused_nomask_blocks_per_s"] == out["value"] and out["config"]["attention_mask"] == "none"


def test_traffic_is_refused_when_the_kernel_sources_changed_since_the_pmc_pass(tmp_path, monkeypatch):
    """`roofline.traffic` comes from this tree's committed PMC pass; the file carries the sha256 of the kernels' sources and is refused
    (null) when they changed since -- no more constants from another round."""
    import json
    import sys

    b = _bench()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic_merge as pm

    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    assert b.read_traffic("k_int_fwd") is None                                   # no file
    good = {"sources_sha256": pm.sources_sha256(), "k_int_fwd": 123.0, "algorithmic": {"k_int_fwd": 100}}
    (prof / "r05_pmc_traffic.json").write_text(json.dumps(good))
    assert b.read_traffic("k_int_fwd") == 123.0 and b.read_traffic("k_int_fwd", 100) == 123.0
    assert b.read_traffic("k_int_fwd", 101) is None                              # another block size than the one profiled
    (prof / "r06_pmc_traffic.json").write_text(json.dumps(dict(good, sources_sha256="0" * 64)))
    assert b.read_traffic("k_int_fwd") is None                                   # the newest file describes other sources: refused
