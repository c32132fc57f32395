"""Host-side bookkeeping of bench.py (no GPU): which roof a dominant kernel is priced against, how the committed PMC traffic is
attached to the `roofline` object, and that a stale profile is not."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stage(ms, gb, gflop, kernel, launches=1.0):
    b = _bench()
    return {"ms_per_step": ms, "launches_per_step": launches, "necessary_gb_per_step": gb, "gb_per_s": gb / (ms * 1e-3),
            "frac_of_hbm_peak": gb / (ms * 1e-3) / b.PEAK_HBM_GBS, "gflop_per_step": gflop, "tflop_per_s": gflop / ms,
            "frac_of_f32_peak": gflop / ms / b.PEAK_F32_TFLOPS, "kernel": kernel, "computes": "x"}


def test_hbm_bound_kernel_is_priced_against_hbm():
    b = _bench()
    r = b.roofline_of({"coarse_premix": _stage(0.34, 1.975, 1.97, "coarse_premix_kernel"), "coarse_fwd": _stage(0.017, 0.006, 0.03, "coarse_fwd_kernel")})
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == b.PEAK_HBM_GBS
    assert abs(r["achieved"] - 1.975e9 / 0.34e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel"] == "coarse_premix_kernel" and r["traffic"] is None


def test_matrix_core_kernel_is_priced_against_the_f32_mfma_peak_when_that_fraction_is_larger():
    b = _bench()
    r = b.roofline_of({"coarse_mac": _stage(1.6, 5.3, 126.7, "coarse_mfma16_kernel"), "coarse_fwd": _stage(1.0, 3.1, 13.0, "coarse_fwd_kernel")})
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == b.PEAK_F32_TFLOPS
    assert abs(r["frac"] - 126.7 / 1.6 / b.PEAK_F32_TFLOPS) < 1e-12 and 0.4 < r["hbm_frac"] < 0.45
    # the same kernel moving far more bytes is an HBM story
    r2 = b.roofline_of({"coarse_mac": _stage(1.6, 9.0, 126.7, "coarse_mfma16_kernel")})
    assert r2["bound"] == "hbm"


def test_committed_pmc_traffic_is_attached_only_to_the_matching_kernel_and_bytes(tmp_path):
    b = _bench()
    prof = tmp_path / "p.json"
    prof.write_text(json.dumps({"command": "python3 bench.py", "kernels": {
        "coarse_premix_kernel": {"necessary_gb_per_launch (stage, planner)": 1.975296, "pmc_total_x2_gb": 1.9745},
        "coarse_fwd_kernel": {"necessary_gb_per_launch (stage, planner)": 0.006, "pmc_total_x2_gb": 0.0061}}}))
    roof = {"kernel": "coarse_premix_kernel", "necessary_bytes_per_launch": 1.975296e9, "traffic": None}
    got = b.attach_pmc_traffic(dict(roof), str(prof))
    assert abs(got["traffic"] - 1.9745e9) < 1 and "FETCH_SIZE" in got["traffic_source"]
    stale = b.attach_pmc_traffic({"kernel": "coarse_premix_kernel", "necessary_bytes_per_launch": 3.9e9, "traffic": None}, str(prof))
    assert stale["traffic"] is None                      # another workload: the committed figure does not apply
    other = b.attach_pmc_traffic({"kernel": "coarse_mac_kernel<2,8,4,8>", "necessary_bytes_per_launch": 1.975296e9, "traffic": None}, str(prof))
    assert other["traffic"] is None
    assert b.attach_pmc_traffic(dict(roof), str(tmp_path / "missing.json"))["traffic"] is None
    assert b.attach_pmc_traffic(None, str(prof)) is None


def test_attached_traffic_says_that_it_was_not_measured_in_this_run(tmp_path):
    b = _bench()
    prof = tmp_path / "p.json"
    prof.write_text(json.dumps({"command": "python3 bench.py", "kernels": {
        "coarse_premix_kernel": {"necessary_gb_per_launch (stage, planner)": 1.975296, "pmc_total_x2_gb": 1.9745}}}))
    got = b.attach_pmc_traffic({"kernel": "coarse_premix_kernel", "necessary_bytes_per_launch": 1.975296e9, "traffic": None}, str(prof))
    assert got["traffic_measured_in_this_run"] is False


def test_serial_bound_of_the_latency_bound_configurations():
    b = _bench()
    r = b.serial_bound(480000, 2.0, "x")
    assert abs(r["bound_ms_per_step"] - 480000 * 3 * 3.46e-6) < 1e-9 and abs(r["frac"] - r["bound_ms_per_step"] / 2.0) < 1e-12
    assert r["chain_ops_per_sample"] == 3 and "r01_micro_dependent_valu_latency" in r["bound"]
