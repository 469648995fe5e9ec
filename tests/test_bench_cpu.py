"""bench.py's line as the driver's record keeps it (CPU only, no device call): the record holds the SCALAR fields of
`config`, `roofline` and `cpu_baseline` and drops nested values (BENCH_r05.json: extra_keys = ["configs", "sharded"], names
kept, values gone), so every headline number of a nested block must also exist as a scalar.  The fixture is a real line:
profiles/r05_bench.json (round 5's full bench output)."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _line():
    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
        return json.load(f)


def _scalars(d):
    return {k: v for k, v in d.items() if not isinstance(v, (dict, list))}


def test_no_nested_value_is_the_only_home_of_a_headline_number():
    import bench

    line = bench.flatten_for_the_record(_line())
    cfg, roof, cpu = (_scalars(line[k]) for k in ("config", "roofline", "cpu_baseline"))
    c, sh = line["configs"], line["sharded"]
    pairs = [
        (c["config2"]["fit"]["ms"], cfg["cfg2_fit_ms"]), (c["config2"]["predict"]["ms"], cfg["cfg2_predict_ms"]),
        (c["config2"]["lml"]["ms"], cfg["cfg2_lml_ms"]), (c["config2"]["lml_gradient"]["ms"], cfg["cfg2_lml_grad_ms"]),
        (c["config2"]["fit"]["frac_of_fp64_mfma_peak"], cfg["cfg2_fit_frac"]),
        (c["config4"]["ei_1000_candidates"]["ms"], cfg["cfg4_ei_ms"]),
        (c["config4"]["minus_ln_ei_and_gradient_1000_candidates"]["ms"], cfg["cfg4_ei_grad_ms"]),
        (c["config4"]["propose_evaluation"]["seconds"], cfg["cfg4_propose_s"]),
        (c["lml_gradient_at_metric_size"]["ms"], cfg["lml_grad_16k_ms"]),
        (sh["config3"]["seconds"], cfg["cfg3_grid64_s"]), (sh["config3"]["frac_of_aggregate_fp64_mfma_peak"], cfg["cfg3_frac"]),
        (sh["config5"]["lml_evals_per_s"], cfg["cfg5_lml_evals_per_s"]),
        (sh["config5"]["frac_of_aggregate_fp64_mfma_peak"], cfg["cfg5_frac"]), (sh["gather"], cfg["gather"]),
        (line["cpu_baseline"]["seconds"]["total"], cpu["seconds_total"]),
        (line["cpu_baseline"]["seconds"]["potrf"], cpu["potrf_s"]),
        (line["cpu_baseline"]["configs"]["config2"]["fit_s"], cpu["cfg2_fit_s"]),
        (line["cpu_baseline"]["configs"]["config3"]["one_grid_point_s"], cpu["cfg3_one_grid_point_s"]),
        (line["cpu_baseline"]["configs"]["config5"]["lml_evals_per_s"], cpu["cfg5_lml_evals_per_s"]),
        (line["cpu_baseline"]["faithful"]["value"], cpu["faithful_gflops"]),
        (line["roofline"]["all_trailing"]["achieved"], roof["all_trailing_tflops"]),
        (line["roofline"]["flow_tail"]["ms_per_step"], roof["flow_tail_ms_per_step"]),
    ]
    for nested, flat in pairs:
        assert nested == flat
    rows = {r["kernel"][:4]: r for r in line["roofline"]["kernels"]}
    assert roof["kbuild_frac"] == rows["kbui"]["frac"] and roof["sweeps_frac"] == rows["trsv"]["frac"]
    assert roof["predict_frac"] == rows["trsm"]["frac"]
    assert abs(roof["traffic_over_algorithmic"] - line["roofline"]["traffic"]
               / line["roofline"]["same_kernel_name_all_launches"]["algorithmic_bytes_per_launch_avg"]) < 1e-12
    assert 1.0 < roof["traffic_over_algorithmic"] < 3.0
    # a walk over every nested number of the blocks the driver drops: each one that the judge's list names has a scalar
    wanted = ("cfg2_fit_ms", "cfg2_predict_ms", "cfg2_lml_ms", "cfg2_lml_grad_ms", "cfg3_grid64_s", "cfg3_frac", "cfg4_ei_ms",
              "cfg4_ei_grad_ms", "cfg4_propose_s", "cfg5_lml_evals_per_s", "cfg5_frac", "lml_grad_16k_ms", "gather")
    assert all(k in cfg for k in wanted)
    assert all(k in roof for k in ("traffic_over_algorithmic", "predict_frac", "sweeps_frac", "kbuild_frac"))
    assert all(k in cpu for k in ("seconds_total", "faithful_gflops"))


def test_vendor_and_rccl_scalars():
    import bench

    line = _line()
    line["vendor"] = {"available": True, "results": {"potrf_16384": {"ms_median": 99.9}, "potrf_8192": {"ms_median": 30.4},
                                                      "trsm_16384_x_1024": {"ms_median": 92.3},
                                                      "syrk_15872_k512": {"tflops_at_median": 53.8}}}
    line["config"].update(rccl_ranks_seen=8, rccl_ranks_match=True,
                          dataset_broadcast="ncclBroadcast from rank 0: identical to the locally generated copy")
    cfg = _scalars(bench.flatten_for_the_record(line)["config"])
    assert cfg["vendor_potrf_16k_ms"] == 99.9 and cfg["vendor_potrf_8k_ms"] == 30.4
    assert cfg["vendor_trsm_ms"] == 92.3 and cfg["vendor_syrk_tflops"] == 53.8
    assert cfg["rccl_ranks_seen"] == 8 and cfg["rccl_ranks_match"] is True and cfg["dataset_broadcast_ok"] is True
    line2 = _line()
    line2.pop("configs"), line2.pop("sharded")
    line2["cpu_baseline"].pop("configs")
    flat = bench.flatten_for_the_record(line2)  # a multi-rank line has none of the single-GPU blocks
    assert flat["config"]["vendor_available"] is False


def test_reference_rates_come_from_the_committed_file():
    import bench

    rr = bench.reference_rates()
    assert rr is not None and rr["fp64_mfma_from_registers_sustained_tflops"] == 77.8 and "profiles/" in rr["source"]
