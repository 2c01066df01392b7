"""Plain vs clustered K1 / K2 on graphs that are not the benchmark lattice: k-nearest-neighbour graphs of random points
("SLIC-like": irregular degrees, with and without self-loops) at several mean degrees — where does the schedule stop paying?
Usage: python tools/diag/cluster_other_graphs.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnn-tumor-seg_amd"))
import torch  # noqa: E402

import gts  # noqa: E402
from gts import ops, schedule, synth  # noqa: E402

schedule.MIN_ROWS_FORWARD = 0


def timeit(fn, n_sets, reps=5):
    for i in range(n_sets):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for i in range(n_sets):
            fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (reps * n_sets)


for k, loops in ((0, False), (6, False), (8, True), (12, False), (16, True)):
    parts = [synth.geometric_graph(n=15000, k=k, seed=10 + i, self_loops=loops) if k else synth.lattice_graph() for i in range(8)]
    g = gts.batch(parts).to("cuda")
    n, e = g.n, g.number_of_edges()
    n_sets = 3
    xs = [torch.randn(n, 256, device="cuda").relu_() for _ in range(n_sets)]
    gouts = [torch.randn(n, 256, device="cuda") for _ in range(n_sets)]
    s_in, s_out = g.cluster_schedule("in"), g.cluster_schedule("out")
    args = [ops.spmm_max_fwd(g, x, relu_input=True)[1] for x in xs]
    line = f"k={k:2d} loops={int(loops)} deg={e / n:5.1f} max={g.max_in_degree:3d} "
    if s_in is None or s_out is None:
        print(line + "no worthwhile schedule", flush=True)
        continue
    line += (f"rows/cluster {n / s_in.n_clusters:4.1f}/{n / s_out.n_clusters:4.1f} staged/edge "
             f"{s_in.staged_rows / e:.2f}/{s_out.staged_rows / e:.2f} ")
    res = {}
    lib = gts._lib.load()
    for label, enabled, waves in (("plain", False, 0), ("8w", True, 8), ("12w", True, 12), ("16w", True, 16)):
        schedule.ENABLED = enabled
        lib.gts_set_option(12, waves)
        res[label] = (timeit(lambda i: ops.spmm_max_fwd(g, xs[i], relu_input=True), n_sets),
                      timeit(lambda i: ops.spmm_max_bwd(g, gouts[i], args[i]), n_sets))
    lib.gts_set_option(12, 0)
    schedule.ENABLED = True
    print(line + "| K1 " + " ".join(f"{k} {v[0]:6.1f}" for k, v in res.items()) + " | K2 "
          + " ".join(f"{k} {v[1]:6.1f}" for k, v in res.items()), flush=True)
