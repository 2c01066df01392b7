"""bench.py contract: algorithmic-byte formulas (CPU) and the JSON line it prints (GPU)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_algorithmic_bytes_match_design_doc():
    import bench

    n_b, e_b = 60000, 4 * 86350
    assert bench.algorithmic_bytes("spmm_max_fwd_f256", n_b, e_b, 1) == 432_111_204
    assert bench.algorithmic_bytes("spmm_max_fwd_f256", n_b, e_b, 4) == 432_111_204 + 3 * 256 * n_b
    assert bench.algorithmic_bytes("project_rows", n_b, e_b, 1) == 18 * 240 ** 3         # 248.8 MB
    assert bench.algorithmic_bytes("gat_fwd", n_b, e_b, 1) == 1_680_639_204
    assert bench.host_cores() >= 1
    assert set(bench.CONFIGS) == {"c2", "c3", "c5"}


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_prints_one_json_line_with_the_contract_fields(hip_lib):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--cpu-steps", "1"], capture_output=True, text=True, timeout=550, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "graphs/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["launches_timed"] == 3 * 7
    m = d["roofline_mfma"]
    assert m["bound"] == "mfma" and m["unit"] == "TFLOP/s" and m["peak"] == 157.3 and 0 < m["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
