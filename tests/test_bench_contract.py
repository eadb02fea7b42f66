"""The bench line committed under profiles/ carries every field the bench contract names, and its
numbers hang together (rays, time and rate; roofline fraction; kernel time against the rocprofv3 summary)."""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r02_b_bench_line.json")
STATS = os.path.join(ROOT, "profiles", "r02_b_bench_kernel_stats.csv")


def load_line():
    with open(LINE) as f:
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
    # the HBM fraction is the measured fabric-side traffic of a launch over its duration, against the peak
    assert r["hbm_frac"] == pytest.approx(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["hbm_peak_gbs"], rel=1e-6)
    assert 0 < r["traffic"] < r["algorithmic_bytes_per_launch"]          # the tree is served from the caches
    # the closest-hit kernel's launch time from HIP events agrees with the rocprofv3 summary of the same command
    if not os.path.exists(STATS):
        pytest.skip("the rocprofv3 summary does not travel to the GPU box")
    with open(STATS) as f:
        rows = {row["kernel"]: row for row in csv.DictReader(f)}
    k2 = next(v for k, v in rows.items() if k.startswith("k_trace<0, false, false>"))      # the timed instance (the counting one runs once more)
    assert float(k2["average_us"]) / 1e3 == pytest.approx(r["avg_launch_ms"], rel=0.02)
