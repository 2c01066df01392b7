"""Copy the judged artefacts of one `tools/gpu_validate.sh <tag>` session from gpurun_out/<tag>/ into profiles/
(bench lines, rocprof kernel stats + markdown summaries, PMC traffic, kernel-only means read by bench.py).
Usage: python tools/refresh_profiles.py <tag> [round prefix, default r02]"""
import csv
import glob
import io
import json
import os
import shutil
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_rocprof  # noqa: E402

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
src, root = f"gpurun_out/{tag}", "profiles"
# from round 4 on a round's files live in profiles/<round>/ (the JSON files bench.py reads stay in profiles/)
dst = os.path.join(root, rnd) if rnd >= "r04" else root
os.makedirs(dst, exist_ok=True)


def last_json_line(path):
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    return lines[-1]


for name in ("c2", "c3", "c5", "c2_b8", "c2_b32", "real"):
    with open(f"{dst}/{rnd}_bench_{name}.json", "w") as fh:
        fh.write(last_json_line(f"{src}/bench_{name}.json") + "\n")
for name in ("pmc_traffic.json", "pmc_traffic_b8.json", "pmc_traffic_b32.json", "pmc_traffic_c3.json"):
    shutil.copy(f"{src}/{name}", f"{root}/{name}")
shutil.copy(f"{src}/aux_kernels.jsonl", f"{dst}/{rnd}_aux_kernels.jsonl")
shutil.copy(f"{src}/epoch_throughput.jsonl", f"{dst}/{rnd}_epoch_throughput.jsonl")


def stats(prof_dir, out_stem, steps, title):
    path = glob.glob(f"{src}/{prof_dir}/**/*kernel_stats.csv", recursive=True)[0]
    out_csv = f"{dst}/{out_stem}_kernel_stats.csv"
    shutil.copy(path, out_csv)
    buf = io.StringIO()
    with redirect_stdout(buf):
        summarize_rocprof.main(out_csv, steps, title)
    open(f"{dst}/{out_stem}_summary.md", "w").write(buf.getvalue())
    return {r["Name"]: r for r in csv.DictReader(open(out_csv))}


def mean_us(rows, needle):
    hit = [r for n, r in rows.items() if all(s in n for s in needle)]
    calls = sum(int(r["Calls"]) for r in hit)
    return round(sum(float(r["TotalDurationNs"]) for r in hit) / calls / 1e3, 2), calls


c2 = stats("prof", f"{rnd}_bench", 65, f"Round {rnd[1:]}: bench.py --steps 20 --warmup 5 --blocks 2 (C2; 5 warm-up + 2 timed blocks + 1 "
           "instrumented block = 65 steps)")
c3 = stats("prof_c3", f"{rnd}_bench_c3", 12, f"Round {rnd[1:]}: bench.py --config c3 --steps 5 --warmup 2 --blocks 1 (GAT 4x4x256; 2 + 5 + 5 = 12 steps)")
b8 = stats("prof_b8", f"{rnd}_bench_c2_b8", 12, f"Round {rnd[1:]}: bench.py --config c4 --steps 5 --warmup 2 --blocks 1 (8 graphs per GPU; 2 + 5 + 5 = 12 steps)")
b32 = stats("prof_b32", f"{rnd}_bench_c2_b32", 7, f"Round {rnd[1:]}: bench.py --graphs-per-gpu 32 --steps 3 --warmup 1 --blocks 1 (1 + 3 + 3 = 7 steps)")
real = stats("prof_real", f"{rnd}_bench_real", 105, f"Round {rnd[1:]}: bench.py --config real --steps 20 --warmup 5 --blocks 1 (5 + 20 + 20 fresh-batch steps + 3 x 20 resident)")
stats("prof_aux", f"{rnd}_aux", 1, f"Round {rnd[1:]}: tools/measure_aux_kernels.py (streaming kernels around the network)")
K1 = (("spmm_cluster_stream_kernel<false, 1",), ("spmm_max_fwd_kernel<4, 64, 1>",))     # clustered form first, plain form otherwise
K2 = (("spmm_cluster_stream_kernel<true, 1",), ("spmm_max_bwd_kernel<4, 64, 1>",))


def first_mean(rows, needles):
    """(mean us, calls, which form ran) of the first kernel name pattern that occurs."""
    for needle in needles:
        if any(all(s_ in n for s_ in needle) for n in rows):
            us, calls = mean_us(rows, needle)
            return us, calls, "clustered" if "cluster" in needle[0] else "plain"
    return None, 0, None


panel, panel_calls = mean_us(c2, ("gemm_panel_direct_kernel<3, 4, 1",))
avg = {"source": f"{dst}/{rnd}_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 5 --blocks 2 "
                 "--no-cpu-baseline`, kernel-only durations)",
       "gemm_panel_direct_avg_us": panel, "gemm_panel_direct_calls": panel_calls,
       "gemm_wgrad_256x256_avg_us": mean_us(c2, ("wgrad_stream_kernel",) if any("wgrad_stream_kernel" in n for n in c2)
                                             else ("gemm_kernel<256, 256, 4, 4, false, false, true",))[0]}
for tag, rows, label in (("", c2, "C2"), ("b8_", b8, "--config c4 (8 graphs per GPU)"), ("b32_", b32, "--graphs-per-gpu 32"),
                         ("real_", real, "--config real (6 graphs of 5 832 nodes, in_feats 20)")):
    for key, needles in (("spmm_max_fwd_f256", K1), ("spmm_max_bwd_f256", K2)):
        us, calls, form = first_mean(rows, needles)
        avg[f"{tag}{key}_avg_us"], avg[f"{tag}{key}_form"] = us, form
    if tag:
        stem = "real" if tag == "real_" else f"c2_{tag[:-1]}"
        avg[f"{tag}source"] = f"{dst}/{rnd}_bench_{stem}_kernel_stats.csv ({label})"
avg["c3_source"] = f"{dst}/{rnd}_bench_c3_kernel_stats.csv (--config c3)"
# hidden-layer launches only (D = 256); the clustered form of a call = its weight pass + the streaming kernel
for key, plain, clustered in (("gat_fwd", "gat_fwd_kernel<4, 64>", ("gat_cluster_stream_kernel<0,", "gat_weights_one_chunk_kernel<false>")),
                              ("gat_bwd_edge", "gat_bwd_edge_kernel<4, 64>", ("gat_cluster_stream_kernel<2,", "gat_edge_finish_kernel")),
                              ("gat_bwd_src", "gat_bwd_src_kernel<4, 64>", ("gat_cluster_stream_kernel<1,", "gat_weights_one_chunk_kernel<true>"))):
    parts = [(needle, mean_us(c3, (needle,))[0]) for needle in clustered if any(needle in n for n in c3)]
    if parts:
        avg[f"{key}_avg_us"], avg[f"{key}_form"] = round(sum(us for _, us in parts), 2), "clustered"
        avg[f"{key}_kernels_us"] = {needle: us for needle, us in parts}
    else:
        avg[f"{key}_avg_us"] = mean_us(c3, (plain,))[0] if any(plain in n for n in c3) else None
        avg[f"{key}_form"] = "plain"
json.dump(avg, open(f"{root}/rocprof_kernel_avg.json", "w"), indent=1)
print(json.dumps(avg, indent=1))
