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
    # compulsory bytes (what must cross HBM): K1 139.9 MB, K2 141.3 MB at C2 (VERDICT r01 recomputation)
    assert bench.compulsory_bytes("spmm_max_fwd_f256", n_b, e_b, 1) == 2 * 61_440_000 + 15_360_000 + 4 * (e_b + n_b + 1)
    assert bench.compulsory_bytes("spmm_max_bwd_f256", n_b, e_b, 1) == 2 * 61_440_000 + 15_360_000 + 4 * (2 * e_b + n_b + 1)
    for kernel in ("spmm_max_fwd_f256", "spmm_max_bwd_f256", "gat_fwd"):
        assert bench.compulsory_bytes(kernel, n_b, e_b, 1) < bench.algorithmic_bytes(kernel, n_b, e_b, 1)
    assert bench.host_cores() >= 1
    assert set(bench.CONFIGS) == {"c2", "c3", "c4", "c5", "real"}
    assert bench.CONFIGS["c4"]["graphs_per_gpu"] == 8          # BASELINE.json: global batch 64 over 8 GPUs
    real = bench.CONFIGS["real"]                               # /root/reference/utils/hyperparam_helpers.py:36-39, model/gnn_model.py:12
    assert (real["in_feats"], real["layer_sizes"], real["graphs_per_gpu"]) == (20, [256] * 4, 6)
    for kernel in ("gat_bwd_edge", "gat_bwd_src"):
        assert bench.compulsory_bytes(kernel, n_b, e_b, 1) < bench.algorithmic_bytes(kernel, n_b, e_b, 1)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_prints_one_json_line_with_the_contract_fields(hip_lib):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--blocks", "2", "--cpu-steps", "1"], capture_output=True, text=True, timeout=550, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "roofline_hbm", "blocks", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "graphs/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]                                        # the dominant kernel: K11
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] is None
    # 8 layers: fc_pool of layer 0, seven chained launches (fc_self + fc_neigh of layer L, fc_pool of L+1), the last pair
    assert set(r["by_kind"]) == {"fwd", "igrad", "wgrad"} and r["by_kind"]["fwd"]["launches_per_step"] == 9
    assert r["by_kind"]["igrad"]["launches_per_step"] == 9
    names = [h["kernel"] for h in d["roofline_hbm"]]
    assert names == ["spmm_max_fwd_f256", "spmm_max_bwd_f256"]
    for h in d["roofline_hbm"]:
        assert h["bound"] == "hbm" and h["peak"] == 8000.0 and h["unit"] == "GB/s" and 0 < h["frac"] < 1
        assert abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-3 and h["launches_timed"] == 3 * 7
        assert (h["traffic"] is None) == (h["traffic_source"] is None)
    b = d["blocks"]
    assert b["n"] == 2 and b["min"] <= b["median"] <= b["max"] and abs(b["median"] - d["value"]) < 1e-2
    assert d["ranks"] == 1 and d["backend"] is None
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_gpus_2_starts_its_own_ranks(hip_lib):
    """`python bench.py --gpus 2` outside torchrun spawns the ranks itself (child process, before any
    GPU call).  On the one-GPU test box both ranks share the card and gloo carries the all-reduce."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(GTS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--blocks", "2", "--graphs-per-gpu", "1"],
                         capture_output=True, text=True, timeout=550, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo"
    assert d["config"]["global_batch"] == 2 and d["config"]["parallelism"] == "dp2"
    assert d["all_reduce"]["payload_bytes"] == 4 * (1_252_888 + 2) and d["all_reduce"]["launches_timed"] == 2
    assert "cpu_baseline" not in d


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("config, kernels", [("c3", ["gat_fwd", "gat_bwd_edge", "gat_bwd_src"]),
                                             ("c4", ["spmm_max_fwd_f256", "spmm_max_bwd_f256"]),
                                             ("real", ["spmm_max_fwd_f256", "spmm_max_bwd_f256"])])
def test_other_configs_keep_the_contract(hip_lib, config, kernels):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--config", config, "--steps", "2",
                          "--warmup", "1", "--blocks", "1", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=550, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert [h["kernel"] for h in d["roofline_hbm"]] == kernels
    for h in d["roofline_hbm"]:
        assert 0 < h["frac"] < 1 and (h["traffic"] is None) == (h["traffic_source"] is None)
        assert (h["rocprof_avg_launch_us"] is None) == (h["rocprof_source"] is None)
    if config == "c3":          # hidden layers only: four launches per step, the 1-head classifier is not averaged in
        assert all(h["launches_timed"] == 2 * 4 for h in d["roofline_hbm"])
    if config == "c4":
        assert d["config"]["global_batch"] == 8 and d["config"]["nodes_per_batch"] == 120000
        assert d["config"]["workload"].startswith("C4")
    if config == "real":
        assert d["config"]["global_batch"] == 6 and d["config"]["nodes_per_batch"] == 6 * 18 ** 3
        assert d["config"]["resident_batches"]["value"] > 0 and d["config"]["host_enqueue_ms_per_step"] > 0
