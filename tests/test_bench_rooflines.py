"""CPU tests of the bench line's per-entry rooflines (bench.py::mode_roofline) and of the committed VALU-ceiling file they are
computed from (profiles/tools/valu_ceiling.py -> profiles/r03_valu_ceilings.json)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_ceiling_file_covers_every_kernel_the_line_prices(bench):
    k = bench.load_valu_ceilings()
    for key in ("gps_l1_3tap_f32", "galileo_5tap_f32", "gps_l1_3tap_i16", "closed_loop_3tap_512", "closed_loop_3tap_1024", "closed_loop_5tap_1024",
            "closed_loop_5tap_1024_pilot"):
        e = k[key]
        # the interior loop keeps two IQ loads in flight and walks 2 samples per lane and load; ~26-41 vector instructions per sample
        assert e["iq_loads_per_iteration"] == 2 and e["samples_per_iteration_per_wave"] == 256
        assert 20.0 < e["valu_instructions_per_sample"] < 45.0
        assert e["ceiling_msamples_s"] == pytest.approx(1024 * 256 / e["cycles_per_iteration"] * 2.4e9 / 1e6)
    assert k["galileo_5tap_f32"]["ceiling_msamples_s"] < k["gps_l1_3tap_f32"]["ceiling_msamples_s"]


def test_mode_roofline_picks_the_binding_resource(bench):
    ceil = {"k": {"ceiling_msamples_s": 1.0e6}}  # 1 T samples/s
    # 8 bytes per sample from HBM: the HBM ceiling is 1 T samples/s too; 4 ms for 1 G samples = 0.25 of either
    r = bench.mode_roofline(ceil, [(1e9, 8.0, "k", 1.0)], 4.0)
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(0.25) and r["hbm_frac"] == pytest.approx(0.25) and r["valu_frac"] == pytest.approx(0.25)
    assert r["achieved"] == pytest.approx(0.25e6) and r["ceiling"] == pytest.approx(1e6)
    # input served from the caches: the VALU ceiling binds
    r = bench.mode_roofline(ceil, [(1e9, 0.25, "k", 1.0)], 2.0)
    assert r["bound"] == "valu" and r["frac"] == pytest.approx(0.5) and r["hbm_frac"] == pytest.approx(1e9 * 0.25 / 8e12 / 2e-3)
    # half the chip in use + a serial term per launch
    r = bench.mode_roofline(ceil, [(1e9, 0.0, "k", 0.5)], 4.0, serial_us_per_unit=1000.0)
    assert r["bound"].startswith("valu") and r["frac"] == pytest.approx((2e-3 + 1e-3) / 4e-3) and r["serial_us"] == 1000.0
    # two launches back to back: ceiling times add
    r = bench.mode_roofline(ceil, [(1e9, 8.0, "k", 1.0), (1e9, 0.0, "k", 1.0)], 4.0)
    assert r["frac"] == pytest.approx(0.5)
    # a kernel the file does not know: the VALU side is reported as unknown, the HBM side still counts
    r = bench.mode_roofline(ceil, [(1e9, 8.0, "missing", 1.0)], 4.0)
    assert r["valu_frac"] is None and r["hbm_frac"] == pytest.approx(0.25)


def test_committed_bench_line_is_self_consistent():
    j = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_line.json")).read())
    r = j["roofline"]
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9)
    assert len(r["segments_ms"]) == 24 and r["cold_start_frac"] <= r["frac"] * 1.02 and abs(r["steady_state_frac"] - r["frac"]) < 0.03
    assert j["preroll"]["launches"] > 0 and j["per_gpu"][0]["rank"] == 0 and j["slowest_rank"] == 0
    for k in ("shared_stream", "int16_input", "galileo_e1_5tap", "hybrid_gps_galileo_beidou", "closed_loop", "closed_loop_galileo_e1", "closed_loop_cfg5_share"):
        ro = j[k]["roofline"]
        assert 0.0 < ro["frac"] < 1.0 and ro["bound"] and ro["ceiling_source"], k
    a = j["acquisition"]["roofline"]
    assert 0.0 < a["frac"] < 1.0 and a["traffic"] < 1.1 * a["algorithmic_bytes_per_search"]
