"""K1 / K2 at F = 256 on B lattice graphs: plain kernels vs the cluster row schedule, timed the way the training
step sees them — every call works on a DIFFERENT set of tensors out of a rotation whose footprint exceeds the
256 MiB Infinity Cache, so inputs come from HBM as they do between the GEMMs of a step.
Usage: python tools/tune_cluster.py [graphs per batch ...]      (GTS_CLUSTER_LIMITS="r,s,e;r,s,e" to try limits)"""
import sys

sys.path.insert(0, "gnn-tumor-seg_amd")
import torch  # noqa: E402

import gts  # noqa: E402
from gts import _lib, ops, schedule, synth  # noqa: E402

lib = _lib.load()


def timeit(fn, n_sets, reps=6):
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


for b in [int(a) for a in sys.argv[1:]] or [4, 8, 32]:
    g = gts.batch([synth.lattice_graph() for _ in range(b)]).to("cuda")
    n, e_b = g.n, g.number_of_edges()
    per_set = n * 256 * 4 * 3
    n_sets = max(2, int(700e6 // per_set) + 1)
    xs = [torch.randn(n, 256, device="cuda").relu_() for _ in range(n_sets)]
    gouts = [torch.randn(n, 256, device="cuda") for _ in range(n_sets)]
    args = [ops.spmm_max_fwd(g, x, relu_input=True)[1] for x in xs]
    need_f = 4 * 256 * n * 2 + 256 * n + 4 * (e_b + n + 1)
    need_b = 4 * 256 * n * 2 + 256 * n + 4 * (2 * e_b + n + 1)
    s_in, s_out = g.cluster_schedule("in"), g.cluster_schedule("out")
    print(f"B={b} N={n} sets={n_sets} limits={schedule.limits('in')};{schedule.limits('out')} "
          f"clusters={s_in.n_clusters}/{s_out.n_clusters} staged/row={s_in.staged_rows / n:.2f}/{s_out.staged_rows / n:.2f} "
          f"slot={s_in.lds_bytes(0)}/{s_out.lds_bytes(1)} B", flush=True)
    # (label, enabled, {option: value}): 9 = kernel form, 10 = ring slots, 11 = workgroups per CU, 12 = consumer waves
    variants = [("plain", False, {}), ("stream auto", True, {}),
                ("stream depth 1", True, {10: 1}), ("stream depth 2", True, {10: 2}), ("stream depth 3", True, {10: 3}),
                ("stream depth 2, 12 waves", True, {10: 2, 12: 12}), ("stream depth 2, 8 waves", True, {10: 2, 12: 8})]
    for label, enabled, options in variants:
        schedule.ENABLED = enabled
        for k, v in options.items():
            lib.gts_set_option(k, v)
        try:
            tf = timeit(lambda i: ops.spmm_max_fwd(g, xs[i], relu_input=True), n_sets)
            tb = timeit(lambda i: ops.spmm_max_bwd(g, gouts[i], args[i]), n_sets)
            print(f"  {label:22s}: K1 {tf:7.1f} us = {need_f / tf / 1e6:5.2f} TB/s ({need_f / tf / 8e6:.3f} of peak)   "
                  f"K2 {tb:7.1f} us = {need_b / tb / 1e6:5.2f} TB/s ({need_b / tb / 8e6:.3f})", flush=True)
        except Exception as exc:      # noqa: BLE001 - a geometry that does not fit the LDS
            print(f"  {label:22s}: {exc}", flush=True)
        for k in options:
            lib.gts_set_option(k, 0)
    schedule.ENABLED = True
    del xs, gouts, args
    torch.cuda.empty_cache()
