"""The measurement pipeline itself (no GPU): the committed rocprofv3 PMC collections re-summarise to the traffic figure
bench.py reports, the bench lines under profiles/ are internally consistent, and every profile DESIGN.md cites exists."""
import gzip
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def test_pmc_collections_resummarise_to_the_committed_traffic(tmp_path):
    out = tmp_path / "t.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"),
                           os.path.join(PROF, "r01_pmc_fetch_counter_collection.csv.gz"),
                           os.path.join(PROF, "r01_pmc_write_counter_collection.csv.gz"), "--json", str(out), "--key", "8b_q8_0"],
                          stdout=subprocess.DEVNULL)
    got = json.load(open(out))["8b_q8_0"]["ntk::gemv_quant_kernel"]
    want = json.load(open(os.path.join(PROF, "pmc_traffic.json")))["8b_q8_0"]["ntk::gemv_quant_kernel"]
    assert got["launches"] == want["launches"] and got["launches"] % 129 == 0          # whole decode tokens: 129 GEMV launches each
    assert abs(got["fetch_bytes_per_launch"] - want["fetch_bytes_per_launch"]) < 1.0
    assert got["fetch_bytes_per_launch"] == 2 * got["fetch_bytes_per_launch_raw"]       # the gfx950 x2 correction, applied once


@pytest.mark.parametrize("name", ["r01_bench_8b_q8_0.json", "r01_bench_8b_q4_k_m.json", "r01_bench_70b_q4_k_m.json", "r01_bench_70b_q6_k.json"])
def test_bench_lines_are_internally_consistent(name):
    d = json.load(open(os.path.join(PROF, name)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "tokens/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert abs(d["value"] * d["ms_per_step"] / 1e3 / d["n_gpus"] - 1.0) < 1e-3          # tokens/s x s/token = replicas
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 5e-3
    b_tok = d["config"]["algorithmic_bytes_per_token"]
    assert abs(d["hbm_fraction_of_8TBs_end_to_end"] - b_tok * d["value"] / 8e12) < 1e-3
    if r["traffic"] is not None:                                                         # PMC bytes within 3 % of the algorithmic bytes
        assert abs(r["traffic"] / r["bytes_per_launch"] - 1.0) < 0.03
    if name == "r01_bench_8b_q8_0.json":
        assert d["vs_baseline"] is not None and "cpu_baseline" in d and d["cpu_baseline"]["kind"] == "port"
        assert b_tok >= 7974764544                                                       # weights + norms, before KV
    else:
        assert d["vs_baseline"] is None


def test_trace_summary_agrees_with_the_bench_line():
    txt = open(os.path.join(PROF, "r01_rocprofv3_kernel_trace_8b_q8_0_graph.txt")).read()
    m = re.search(r"pooled: ([\d.]+) launches/token, avg ([\d.]+) us", txt)
    assert m and float(m.group(1)) == 129.0
    trace_us = float(m.group(2))
    bench_us = json.load(open(os.path.join(PROF, "r01_bench_8b_q8_0.json")))["roofline"]["avg_launch_us"]
    assert abs(bench_us / trace_us - 1.0) < 0.08, (bench_us, trace_us)                   # live HIP-event figure vs rocprofv3 kernel trace


def test_every_profile_cited_in_design_exists():
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    cited = set(re.findall(r"`(?:profiles/)?(r01_[A-Za-z0-9_.{},]+?\.(?:txt|json))`", design))
    assert cited
    for c in cited:
        names = [c]
        m = re.match(r"(.*)\{(.*)\}(.*)", c)
        if m: names = [m.group(1) + x + m.group(3) for x in m.group(2).split(",")]
        for n in names:
            assert os.path.exists(os.path.join(PROF, n)), n
