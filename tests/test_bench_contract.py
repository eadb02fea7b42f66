"""The bench line committed under profiles/ carries every field the bench contract names, and its
numbers hang together (rays, time and rate; roofline fraction; kernel time against the rocprofv3 summary)."""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r03_d_bench_line.json")
LINE_PROFILED = os.path.join(ROOT, "profiles", "r03_d_bench_line_under_rocprof.json")      # printed while rocprofv3 made STATS
STATS = os.path.join(ROOT, "profiles", "r03_d_bench_kernel_stats.csv")


def load_line(path=LINE):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_bench_line_has_the_contract_fields():
    d = load_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["config"]["workload"] == "crytek_gi128_1080p"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


def test_bench_line_numbers_hang_together():
    d = load_line()
    c, r = d["config"], d["roofline"]
    assert d["value"] == pytest.approx(c["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6, rel=1e-6)
    assert c["reference_rays_per_frame"] == pytest.approx(c["rays_per_frame"] + c["shadow_rays_not_traversed_per_frame"])
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert 2.0 <= r["box_tests_per_node_visit"] <= c["bvh_node_width"]
    # both roofs are reported; the one with the larger share names the bound and fills achieved / peak / unit / frac
    valu, hbm = r["valu"], r["hbm"]
    assert valu["unit"] == "TFLOP/s" and valu["frac"] == pytest.approx(valu["achieved"] / valu["peak"], rel=1e-9)
    assert hbm is not None, "the committed line was measured with the traffic figure of its own kernel sources"
    assert hbm["unit"] == "GB/s" and hbm["frac"] == pytest.approx(hbm["achieved"] / hbm["peak"], rel=1e-9)
    first = hbm if hbm["frac"] > valu["frac"] else valu
    assert r["bound"] == ("hbm" if first is hbm else "valu")
    assert (r["achieved"], r["peak"], r["unit"], r["frac"]) == (first["achieved"], first["peak"], first["unit"], first["frac"])
    # the HBM share is the measured fabric-side traffic of a launch over its duration, against the peak
    assert r["hbm_frac"] == pytest.approx(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["hbm_peak_gbs"], rel=1e-6)
    assert hbm["frac"] == pytest.approx(r["hbm_frac"], rel=1e-6)
    assert 0 < r["traffic"] < r["algorithmic_bytes_per_launch"]          # the tree is served from the caches
    kinds = [k["kernel"].split()[0] for k in d["roofline_kernels"]]
    assert kinds == ["k_trace<1>", "k_shade_bounce"]
    for k in d["roofline_kernels"]:
        assert k["frac"] == pytest.approx(k["achieved"] / k["peak"], rel=1e-9) and k["bound"] in ("valu", "hbm")


def test_kernel_time_from_events_agrees_with_the_rocprofv3_summary():
    """The closest-hit kernel's launch time from HIP events and from rocprofv3, both of the same run of the same command."""
    if not (os.path.exists(STATS) and os.path.exists(LINE_PROFILED)):
        pytest.skip("the rocprofv3 summary does not travel to the GPU box")
    r = load_line(LINE_PROFILED)["roofline"]
    with open(STATS) as f:
        rows = {row["kernel"]: row for row in csv.DictReader(f)}
    k2 = next(v for k, v in rows.items() if k.startswith("k_trace<0, false, false>"))      # the timed instance (the counting one runs once more)
    assert float(k2["average_us"]) / 1e3 == pytest.approx(r["avg_launch_ms"], rel=0.02)
    # without the profiler attached the launches are a little shorter
    assert load_line()["roofline"]["avg_launch_ms"] == pytest.approx(r["avg_launch_ms"], rel=0.06)
